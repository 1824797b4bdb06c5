'use strict'
// GPU end-to-end run of the node boundary (tests/test_node_boundary.py drives it and checks every output
// against the oracle): the N-API addon + index.js + this repo's own front end (device.js, jobs.js,
// staging.js).  usage: node gpu_run.js <dir with job.json and input files>; writes result.json + *.bin.
const fs = require('fs')
const path = require('path')
const crypto = require('crypto')
const { Rig } = require('../device.js')
const { StagedChannel } = require('../staging.js')

const dir = process.argv[2]
const job = JSON.parse(fs.readFileSync(path.join(dir, 'job.json')))
const load = (f) => fs.readFileSync(path.join(dir, f))
const save = (f, b) => fs.writeFileSync(path.join(dir, f), b)
const sha = (bufs) => { const h = crypto.createHash('sha256'); bufs.forEach((b) => h.update(b)); return h.digest('hex') }

async function main() {
	const rig = await Rig.open({ deviceIndex: 0 })
	const ctx = rig.ctx
	const result = { platform: ctx.getPlatformInfo() }

	// ---- 1. one channel: n v210 layers -> read -> (layer 1 placed as a PiP) -> combine -> write v210 -----------
	{
		const c = job.channel
		const n = c.layers.length
		const read = await rig.unpack('v210', c.width, c.height, c.readSpec, c.writeSpec)
		const write = await rig.pack('v210', c.width, c.height, c.writeSpec, false)
		const xf = await rig.transform(c.width, c.height)
		const combine = await rig.combine(n, c.width, c.height)
		const rgba = []
		for (let l = 0; l < n; ++l) {
			const src = await rig.planes('v210', c.width, c.height, 'readonly', `L${l} src`)
			await rig.upload(src[0], load(c.layers[l]))
			const img = await rig.image(c.width, c.height, `L${l}`)
			rig.post({ source: `L${l}`, timestamp: 0 }, read(src, img), () => src[0].release())
			rgba.push(img)
		}
		await rig.sync(ctx.queue.load)
		const placed = await rig.image(c.width, c.height, 'L1 placed')
		rig.post({ source: 'L1', timestamp: 0 }, xf(rgba[1], placed, await xf.matrix(c.pip)), () => rgba[1].release())
		const layers = rgba.slice()
		layers[1] = placed
		const combined = await rig.image(c.width, c.height, 'combined')
		const out = await rig.planes('v210', c.width, c.height, 'writeonly', 'out')
		rig.post({ source: 'chan', timestamp: 0 }, combine(layers, combined), () => layers.forEach((b) => b.release()))
		rig.post({ source: 'chan', timestamp: 0 }, write(combined, out, 0), () => combined.release())
		// flushes requested together are served in order and drained once (jobs.js)
		const ids = [0, 1, 2, 3].slice(0, n).map((l) => ({ source: `L${l}`, timestamp: 0 }))
		await Promise.all(ids.concat([{ source: 'chan', timestamp: 0 }]).map((id) => rig.board.flush(id)))
		await rig.download(out[0])
		save('channel_out.bin', out[0])
		out[0].release()
		result.boardStats = Object.assign({}, rig.board.stats)
	}

	// ---- 2. the reference scripts' known answer: ramp -> read -> write gives the ramp back ("Compare returned 0")
	{
		const w = 1920
		const h = 1080
		const read = await rig.unpack('v210', w, h, '709', '709')
		const write = await rig.pack('v210', w, h, '709', false)
		const src = await rig.planes('v210', w, h)
		const dst = await rig.planes('v210', w, h, 'writeonly')
		const img = await rig.image(w, h)
		const ramp = load(job.ramp)
		await rig.upload(src[0], ramp)
		await rig.sync(ctx.queue.load)
		await rig.run(read(src, img))
		await rig.run(write(img, dst, 0))
		await rig.sync()
		await rig.download(dst[0])
		result.rampCompare = Buffer.compare(ramp, dst[0])
		;[src[0], dst[0], img].forEach((b) => b.release())
	}

	// ---- 3. yadif, send_field over a window of three frames (yadif.ts:88-145): two outputs per frame that has
	//         both neighbours, parity = tff ^ !second, timestamps cur and cur + 1 --------------------------------
	{
		const y = job.yadif
		const deint = await rig.yadif(y.width, y.height)
		const frames = []
		for (let f = 0; f < y.frames.length; ++f) {
			const img = await rig.image(y.width, y.height, `field ${f}`)
			await rig.upload(img, load(y.frames[f]))
			img.timestamp = 2 * f
			frames.push(img)
		}
		await rig.sync(ctx.queue.load)
		const stamps = []
		let k = 0
		for (let cur = 1; cur + 1 < frames.length; ++cur) {
			for (const second of [false, true]) {
				const out = await rig.image(y.width, y.height, 'deint')
				out.timestamp = frames[cur].timestamp + (second ? 1 : 0)
				const parity = (y.tff ? 1 : 0) ^ (second ? 0 : 1)
				await rig.run(deint(frames[cur - 1], frames[cur], frames[cur + 1], out, { parity, tff: y.tff, skipSpatial: false }))
				await rig.sync()
				await rig.download(out)
				save(`yadif_out${k++}.bin`, out)
				stamps.push(out.timestamp)
				out.release()
			}
		}
		result.yadifTimestamps = stamps
		frames.forEach((b) => b.release())
	}

	// ---- 4. the other wire formats: pattern -> read -> write (both fields) -> the same bytes back ------------------
	result.formats = {}
	for (const f of job.formats) {
		const read = await rig.unpack(f.fmt, f.width, f.height, f.spec, f.spec)
		const write = await rig.pack(f.fmt, f.width, f.height, f.spec, true)
		const src = await rig.planes(f.fmt, f.width, f.height)
		const dst = await rig.planes(f.fmt, f.width, f.height, 'writeonly')
		const img = await rig.image(f.width, f.height)
		const pattern = load(f.file)
		let off = 0
		const parts = []
		for (const p of src) { parts.push(pattern.slice(off, off + p.length)); off += p.length }
		for (let i = 0; i < src.length; ++i) await rig.upload(src[i], parts[i])
		await rig.sync(ctx.queue.load)
		await rig.run(read(src, img))
		await rig.run(write(img, dst, 1))
		await rig.run(write(img, dst, 3))
		await rig.sync()
		await rig.download(img)
		for (const d of dst) await rig.download(d)
		result.formats[f.fmt] = { rgbaSha256: sha([img]), backSha256: sha(dst), compare: Buffer.compare(Buffer.concat(parts), Buffer.concat(dst)) }
		;[...src, ...dst, img].forEach((b) => b.release())
	}

	// ---- 5. staged ring (queue.load / process / unload overlapped on the device) around the fused channel program -----
	{
		const s = job.staged
		const fused = await rig.fused(s.layers, s.width, s.height, s.readSpec, s.writeSpec)
		const bytes = require('../index.js').planeBytes('v210', s.width, s.height)[0]
		const chan = new StagedChannel(ctx, new Array(s.layers).fill(bytes), bytes, (sources, output) => rig.run(fused(sources, output)), 3, 'staged')
		await chan.init()
		const order = []
		const consume = async (frameNo, out) => { order.push(frameNo); save(`staged_out${frameNo}.bin`, Buffer.from(out)) }
		for (let f = 0; f < s.frames; ++f)
			await chan.submit(async (frameNo, sources) => { sources.forEach((b, l) => load(`staged_f${frameNo}_l${l}.bin`).copy(b)) }, consume)
		await chan.drain(consume)
		await chan.close()
		result.stagedOrder = order
		// the same program through the JobBoard: its callback fires after the batch has run
		const src = []
		for (let l = 0; l < s.layers; ++l) {
			const b = (await rig.planes('v210', s.width, s.height))[0]
			await rig.upload(b, load(`staged_f0_l${l}.bin`))
			src.push(b)
		}
		await rig.sync(ctx.queue.load)
		const out = (await rig.planes('v210', s.width, s.height, 'writeonly'))[0]
		let fired = false
		rig.post({ source: 'fused', timestamp: 9 }, fused(src, out), () => { fired = true })
		await rig.board.flush({ source: 'fused', timestamp: 9 })
		result.fusedCallbackFired = fired
		await rig.download(out)
		save('fused_queue_out.bin', out)
		;[...src, out].forEach((b) => b.release())
	}

	// ---- 6. errors surface as rejected promises / thrown Errors with the library's message -----------------------------
	{
		const errors = {}
		const grab = async (label, fn) => { try { await fn(); errors[label] = null } catch (e) { errors[label] = e instanceof Error ? e.message : String(e) } }
		await grab('unknownKey', () => rig.board.flush({ source: 'nobody', timestamp: 5 }))
		await grab('unknownKernel', () => ctx.createProgram('phaneron:x', { name: 'sharpen', globalWorkItems: 1 }))
		await grab('missingArgument', async () => {
			const p = await ctx.createProgram('phaneron:v210', { name: 'read', globalWorkItems: 40, workItemsPerGroup: 40 })
			await ctx.runProgram(p, { width: 1920 }, ctx.queue.process)
		})
		await grab('combineOne', () => rig.combine(1, 64, 64))
		await grab('unknownOption', async () => ctx.setOption('stream_everything', 1))
		await grab('knownOption', async () => { ctx.setOption('stream_images', 1); ctx.setOption('stream_images', 0) })
		result.errors = errors
	}

	// ---- 7. ROUTE hand-off on the library's path: a frame produced on the process queue travels through RCCL (to the
	//         own rank: one GPU here) on the communication stream and is consumed by a kernel on the process queue,
	//         ordered on the device only --------------------------------------------------------------------------
	{
		const { clContext } = require('../index.js')
		const w = 1920
		const h = 64
		const link = ctx.openRoute(clContext.routeUniqueId(), 0, 1)
		const read = await rig.unpack('v210', w, h, '709', '709')
		const write = await rig.pack('v210', w, h, '709', false)
		const src = await rig.planes('v210', w, h)
		await rig.upload(src[0], load(job.ramp).slice(0, src[0].length))
		await rig.sync(ctx.queue.load)
		const made = await rig.image(w, h, 'route source frame')
		const got = await rig.image(w, h, 'routed frame')
		const out = await rig.planes('v210', w, h, 'writeonly')
		await rig.run(read(src, made))                 // the source channel's output, on the process queue
		link.afterQueue()
		link.group(() => { link.send(made, 0); link.recv(got, 0) })
		link.queueAfter()
		await rig.run(write(got, out, 0))              // the sink consumes the routed frame
		await rig.sync()
		await rig.download(out[0])
		result.routeLoopbackCompare = Buffer.compare(load(job.ramp).slice(0, out[0].length), out[0])
		;[src[0], made, got, out[0]].forEach((b) => b.release())
	}

	// ---- 8. the de-interlacing fusions under their program names: 'yadif_pair' on the RGBA fields of step 3, and
	//         'v210_yadif_pair_<n>' (ToRGBA of the v210 window + both fields) on two layers ----------------------------
	{
		const y = job.yadif
		const pair = await rig.yadifPair(y.width, y.height)
		const f = []
		for (let i = 0; i < 3; ++i) {
			const img = await rig.image(y.width, y.height, `pair field ${i}`)
			await rig.upload(img, load(y.frames[i]))
			f.push(img)
		}
		await rig.sync(ctx.queue.load)
		const out = [await rig.image(y.width, y.height), await rig.image(y.width, y.height)]
		await rig.run(pair(f[0], f[1], f[2], out, { tff: y.tff, skipSpatial: false }))
		await rig.sync()
		for (let p = 0; p < 2; ++p) { await rig.download(out[p]); save(`yadif_pair_p${p}.bin`, out[p]) }
		;[...f, ...out].forEach((b) => b.release())

		const d = job.deint
		const reader = await rig.deinterlaceReader(d.layers.length, d.width, d.height, d.readSpec, d.writeSpec)
		const windows = []
		for (let l = 0; l < d.layers.length; ++l) {
			const win = []
			for (const file of d.layers[l]) {
				const b = (await rig.planes('v210', d.width, d.height))[0]
				await rig.upload(b, load(file))
				win.push(b)
			}
			windows.push({ prev: win[0], cur: win[1], next: win[2], out: [await rig.image(d.width, d.height), await rig.image(d.width, d.height)] })
		}
		await rig.sync(ctx.queue.load)
		await rig.run(reader(windows, { tff: d.tff, skipSpatial: false }))
		await rig.sync()
		for (let l = 0; l < windows.length; ++l)
			for (let p = 0; p < 2; ++p) { await rig.download(windows[l].out[p]); save(`deint_l${l}_p${p}.bin`, windows[l].out[p]) }
		windows.forEach((wn) => [wn.prev, wn.cur, wn.next, ...wn.out].forEach((b) => b.release()))

		// 'v210_read_batch_<n>': the three frames of layer 0's window unpacked in one launch
		const batch = await rig.unpackBatch(3, d.width, d.height, d.readSpec, d.writeSpec)
		const srcs = []
		const imgs = []
		for (const file of d.layers[0]) {
			const b = (await rig.planes('v210', d.width, d.height))[0]
			await rig.upload(b, load(file))
			srcs.push(b)
			imgs.push(await rig.image(d.width, d.height))
		}
		await rig.sync(ctx.queue.load)
		await rig.run(batch(srcs, imgs))
		await rig.sync()
		for (let i = 0; i < 3; ++i) { await rig.download(imgs[i]); save(`batch_read_${i}.bin`, imgs[i]) }
		;[...srcs, ...imgs].forEach((b) => b.release())
	}

	// ---- 9. the channel of step 1 again with its tail as ONE launch: 'compose_write_v210_<n>' = transform (layer 1) ->
	//         combine -> write; and the same with a wipe transition on the top layer inside the kernel --------------------
	{
		const c = job.channel
		const n = c.layers.length
		const read = await rig.unpack('v210', c.width, c.height, c.readSpec, c.writeSpec)
		const xf = await rig.transform(c.width, c.height)
		const compose = await rig.compose(n, c.width, c.height, c.writeSpec)
		const rgba = []
		for (let l = 0; l < n; ++l) {
			const src = await rig.planes('v210', c.width, c.height)
			await rig.upload(src[0], load(c.layers[l]))
			await rig.sync(ctx.queue.load)
			const img = await rig.image(c.width, c.height)
			await rig.run(read(src, img))
			src[0].release()
			rgba.push(img)
		}
		const pip = await xf.matrix(c.pip)
		const layers = rgba.map((image, l) => (l === 1 ? { image, matrix: pip } : { image }))
		const out = await rig.planes('v210', c.width, c.height, 'writeonly')
		await rig.run(compose(layers, out[0], 0))
		await rig.sync()
		await rig.download(out[0])
		save('compose_out.bin', out[0])
		// wipe: the top layer gives way to layer 0's picture under a mask made of layer 2's picture
		layers[n - 1] = { image: rgba[n - 1], wipe: { incoming: rgba[0], mask: rgba[2] } }
		await rig.run(compose(layers, out[0], 0))
		await rig.sync()
		await rig.download(out[0])
		save('compose_wipe_out.bin', out[0])
		;[...rgba, out[0]].forEach((b) => b.release())
	}

	// ---- 10. the same channel as ONE launch from its v210 sources: 'chan_compose_v210_<n>' = read -> transform (layer 1) ->
	//          combine -> write, and with the wipe (incoming = layer 0's source, mask = layer 2's): the outputs of step 9 again ----
	if (job.channel.width % 192 === 0) {
		const c = job.channel
		const n = c.layers.length
		const xf = await rig.transform(c.width, c.height)
		const chan = await rig.channelCompose(n, c.width, c.height, c.readSpec, c.writeSpec)
		const srcs = []
		for (let l = 0; l < n; ++l) {
			const src = await rig.planes('v210', c.width, c.height)
			await rig.upload(src[0], load(c.layers[l]))
			srcs.push(src[0])
		}
		await rig.sync(ctx.queue.load)
		const pip = await xf.matrix(c.pip)
		const layers = srcs.map((source, l) => (l === 1 ? { source, matrix: pip } : { source }))
		const out = await rig.planes('v210', c.width, c.height, 'writeonly')
		await rig.run(chan(layers, out[0], 0))
		await rig.sync()
		await rig.download(out[0])
		save('chan_out.bin', out[0])
		layers[n - 1] = { source: srcs[n - 1], transition: { type: 'wipe', incoming: { source: srcs[0] }, mask: { source: srcs[2] } } }
		await rig.run(chan(layers, out[0], 0))
		await rig.sync()
		await rig.download(out[0])
		save('chan_wipe_out.bin', out[0])
		;[...srcs, out[0]].forEach((b) => b.release())
	}

	rig.close()
	result.liveAfter = ctx.logBuffers ? rig.ctx.bufferStats().liveBuffers : -1
	save('result.json', JSON.stringify(result))
}

main().catch((e) => { process.stderr.write(String(e && e.stack || e) + '\n'); process.exit(1) })
