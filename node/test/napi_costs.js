'use strict'
// What the addon's calls cost on this host (GPU box): the pieces the recording context's per-frame overhead is made of.
// usage: node napi_costs.js [width=1920] [height=1080]; prints one JSON line (microseconds per call)
const { Rig } = require('../device.js')
async function main() {
	const w = parseInt(process.argv[2] || '1920')
	const h = parseInt(process.argv[3] || '1080')
	const rig = await Rig.open({ deviceIndex: 0, deferred: false })
	const native = rig.ctx._native
	const us = async (n, fn) => { const t0 = process.hrtime.bigint(); for (let i = 0; i < n; ++i) await fn(i); return +(Number(process.hrtime.bigint() - t0) / 1e3 / n).toFixed(3) }
	const out = { bench: 'napi_costs', width: w, height: h }
	for (let i = 0; i < 50; ++i) (await rig.image(w, h)).release()
	out.image_create_release = await us(2000, async () => { (await rig.image(w, h)).release() })
	out.image_create_release_native_only = await us(2000, () => { const c = native.createBuffer(rig.ctx._ctx, w * h * 16, 2, 1, w, h, 'x'); native.bufRelease(c.handle) })
	const b = await rig.image(w, h)
	out.bufRefCount = await us(200000, () => native.bufRefCount(b._handle))
	out.addRef_release_pair = await us(200000, () => { native.bufAddRef(b._handle); native.bufRelease(b._handle) })
	const read = await rig.unpack('v210', w, h, '709', '2020')
	const src = await rig.planes('v210', w, h)
	const job = read(src, b)
	const names = Object.keys(job.params)
	const values = names.map((k) => (Buffer.isBuffer(job.params[k]) ? job.params[k]._handle : job.params[k]))
	out.checkProgram_read = await us(50000, () => native.runProgram(rig.ctx._ctx, job.program._handle, names, values, 1, false, true))
	const fused = await rig.fused(4, w, h, '709', '2020')
	const srcs = []
	for (let l = 0; l < 4; ++l) srcs.push((await rig.planes('v210', w, h))[0])
	const o = (await rig.planes('v210', w, h, 'writeonly'))[0]
	const fj = fused(srcs, o)
	const fnames = Object.keys(fj.params)
	const fvalues = fnames.map((k) => (Buffer.isBuffer(fj.params[k]) ? fj.params[k]._handle : fj.params[k]))
	out.checkProgram_fused4 = await us(50000, () => native.runProgram(rig.ctx._ctx, fj.program._handle, fnames, fvalues, 1, false, true))
	await rig.ctx.drain()
	out.launch_fused4_async = await us(300, () => native.runProgram(rig.ctx._ctx, fj.program._handle, fnames, fvalues, 1, false))
	await rig.ctx.drain()
	out.queueWaitQueue = await us(20000, () => native.queueWaitQueue(rig.ctx._ctx, 2, 1))
	out.empty_await = await us(200000, async () => undefined)
	console.log(JSON.stringify(out))
	;[b, ...src, ...srcs, o].forEach((x) => x.release())
	rig.close()
}
main().catch((e) => { process.stderr.write(String(e && e.stack || e) + '\n'); process.exit(1) })
