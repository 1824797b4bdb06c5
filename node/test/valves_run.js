'use strict'
// GPU end-to-end run of the re-hosted valve graph (node/valves) on the real clContext:
//   node valves_run.js <workdir>    (driven by tests/test_node_boundary.py)
// job.json: { width, height, frames, pip: MixerParams, dissolveAt, dissolveLen, cutAt } and RGBA f32 files
// A_<i>.bin, B0_<i>.bin, B1_<i>.bin; writes out_<f>.bin + result.json.
const fs = require('fs')
const path = require('path')
const { clContext } = require('../index.js')
const { ClProcessJobs } = require('../clJobQueue.js')
const { redio, isValue, end, Mixer, Transitioner, Combiner, CombineLayer } = require('../valves')

async function main() {
	const dir = process.argv[2]
	const job = JSON.parse(fs.readFileSync(path.join(dir, 'job.json')))
	const ctx = new clContext({ platformIndex: 0, deviceIndex: 0, overlapping: true })
	await ctx.initialise()
	const jobs = new ClProcessJobs(ctx).getJobs()
	const fmt = { width: job.width, height: job.height }
	const bytes = fmt.width * fmt.height * 16

	const source = (name, n, ts0) => {
		let i = 0
		return redio(async () => {
			if (i >= n) return end
			const b = await ctx.createBuffer(bytes, 'readwrite', 'coarse', fmt, `${name} ${i}`)
			await b.hostAccess('writeonly', ctx.queue.load, fs.readFileSync(path.join(dir, `${name}_${i}.bin`)))
			await ctx.waitFinish(ctx.queue.load)
			b.timestamp = ts0 + i++
			return b
		})
	}
	const mkLayer = async (id, pipes, params) => {
		const mixers = []
		for (const [k, p] of pipes.entries()) {
			const m = new Mixer(ctx, fmt, jobs)
			if (params && params[k]) m.setMixParams(params[k])
			await m.init(`${id} src${k}`, p)
			mixers.push(m)
		}
		const t = new Transitioner(ctx, id, fmt, jobs)
		await t.initialise()
		return { mixers, t }
	}
	const A = await mkLayer('L1', [source('A', job.frames, 100)])
	const B = await mkLayer('L2', [source('B0', job.frames, 200), source('B1', job.frames, 300)], [job.pip, null])
	const C = await mkLayer('L3', [])
	A.t.update('cut', 0, [A.mixers[0].getMixVideo()])
	B.t.update('cut', 0, [B.mixers[0].getMixVideo()])
	C.t.update('cut', 0, [])
	const comb = new Combiner(ctx, 'chan1', fmt, jobs)
	await comb.initialise()
	comb.updateLayers([A, B, C].map((l) => new CombineLayer(l.t.getVideoPipe())))

	const out = comb.getVideoPipe()
	const stamps = []
	for (let f = 0; f < job.frames; ++f) {
		if (f === job.dissolveAt) B.t.update('dissolve', job.dissolveLen, [B.mixers[0].getMixVideo(), B.mixers[1].getMixVideo()])
		if (f === job.cutAt) B.t.update('cut', 0, [B.mixers[1].getMixVideo()])
		const frame = await out.next()
		if (!isValue(frame)) break
		await frame.hostAccess('readonly', ctx.queue.unload)
		fs.writeFileSync(path.join(dir, `out_${f}.bin`), frame)
		stamps.push(frame.timestamp)
		frame.release()
	}
	;[A, B, C].forEach((l) => { l.mixers.forEach((m) => m.release()); l.t.release() })
	comb.release()
	fs.writeFileSync(path.join(dir, 'result.json'), JSON.stringify({ stamps, buffers: ctx.logBuffers() }))
}

main().catch((e) => { console.error(e && e.stack || e); process.exit(1) })
