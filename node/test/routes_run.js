'use strict'
// GPU check: which kernels the recording context's frames reach the device as.  The reference posts a frame operator by operator
// (read / yadif / transform / transition / combine_N / write: clJobQueue.ts:126); node/defer.js folds the chain and the library
// chooses among its routes (DESIGN.md section 5.1).  Representative job streams are recorded here under a DRY trace
// (ph_trace_begin(1): everything is chosen and checked, nothing is enqueued) and the kernel names pinned - so that a change in
// either layer that silently moves a shape to another kernel fails here.  usage: node routes_run.js; prints { checks, problems, routes }
const { Rig } = require('../device.js')

async function main() {
	const problems = []
	const routes = {}
	let checks = 0
	const W = 1920, H = 1080
	const rig = await Rig.open({ deviceIndex: 0, deferred: true })
	const read = await rig.unpack('v210', W, H, '709', '709')
	const read420 = await rig.unpack('yuv420p', 1280, 720, '709', '709')
	const write = await rig.pack('v210', W, H, '709', false)
	const transform = await rig.transform(W, H)
	const combine = await rig.combine(4, W, H)
	const yadif = await rig.yadif(W, H)
	const fill = await transform.matrix({})
	const PIP = [{}, { scaleX: 0.5, scaleY: 0.5, offsetX: -0.25, offsetY: -0.25 }, { scaleX: 0.5, scaleY: 0.5, offsetX: 0.25, offsetY: -0.25 }, { scaleX: 0.5, scaleY: 0.5, offsetX: 0.25, offsetY: 0.25 }]
	const mats = []
	for (const p of PIP) mats.push(await transform.matrix(p))
	const src = []
	for (let l = 0; l < 4; ++l) src.push(await rig.planes('v210', W, H))
	const clip = await rig.planes('yuv420p', 1280, 720)
	await rig.sync(rig.ctx.queue.load)
	const keep = []
	const pin = async (what, want, post) => {
		await rig.ctx.drain()
		rig.ctx.traceBegin(true)
		let got
		try {
			const outs = await post()
			for (const o of outs) rig.ctx.realise(o)
		} finally { got = rig.ctx.traceEnd() }
		++checks
		routes[what] = got
		if (got !== want) problems.push({ what, got, want })
	}
	const out = async () => { const o = (await rig.planes('v210', W, H, 'writeonly'))[0]; keep.push(o); return o }
	const image = async (w = W, h = H) => { const im = await rig.image(w, h); keep.push(im); return im }

	await pin('the headline: read x4 -> combine_4 -> write', 'fused_v210_combine_lds', async () => {
		const ims = []
		for (let l = 0; l < 4; ++l) { const im = await image(); await rig.run(read(src[l], im)); ims.push(im) }
		const cm = await image(), o = await out()
		await rig.run(combine(ims, cm))
		await rig.run(write(cm, [o], 0))
		return [o]
	})
	const config2 = async () => {
		const placed = []
		for (let l = 0; l < 4; ++l) {
			const im = await image(), pl = await image()
			await rig.run(read(src[l], im))
			await rig.run(transform(im, pl, mats[l]))
			placed.push(pl)
		}
		const cm = await image(), o = await out()
		await rig.run(combine(placed, cm))
		await rig.run(write(cm, [o], 0))
		return o
	}
	await pin('config 2\'s shape: read -> transform x4 -> combine_4 -> write', 'chan_compose_v210<0,0>', async () => [await config2()])
	await pin('a 720p yuv420p file filling a 1080p channel: read -> transform -> write', 'clip_up_write_v210<rgb>', async () => {
		const im = await image(1280, 720), pl = await image(), o = await out()
		await rig.run(read420(clip, im))
		await rig.run(transform(im, pl, fill))
		await rig.run(write(pl, [o], 0))
		return [o]
	})
	await pin('a 1080i source on a 1080p channel, both fields posted: reader pair launch, both frames in one compositor launch', 'v210_yadif_pair+compose_up_write_v210', async () => {
		const u = []
		for (let i = 0; i < 3; ++i) { const im = await image(); await rig.run(read(src[i], im)); u.push(im) }
		const outs = []
		for (const second of [0, 1]) {
			const y = await image(), pl = await image(), o = await out()
			await rig.run(yadif(u[0], u[1], u[2], y, { parity: second ? 1 : 0, tff: 1, skipSpatial: 0 }))
			await rig.run(transform(y, pl, fill))
			await rig.run(write(pl, [o], 0))
			outs.push(o)
		}
		return outs
	})
	// config 3's own placements on a 1080p channel: four 1080i sources, the first full-frame, three picture-in-picture at half size (they SHRINK:
	// not the 2 x 2-block compositor's) - the windows' reader for all four layers, then both fields' frames in ONE launch of the batch kernel, the fields its f32 image sources
	await pin('four 1080i sources, three of them picture-in-picture, both fields posted', 'v210_yadif_pair+chan_compose_batch<0>x2', async () => {
		const placedBy = [[], []]
		for (let l = 0; l < 4; ++l) {
			const u = []
			for (let i = 0; i < 3; ++i) { const im = await image(); await rig.run(read(src[(l + i) % 4], im)); u.push(im) }
			for (const second of [0, 1]) {
				const y = await image(), pl = await image()
				await rig.run(yadif(u[0], u[1], u[2], y, { parity: second ? 1 : 0, tff: 1, skipSpatial: 0 }))
				await rig.run(transform(y, pl, mats[l]))
				placedBy[second].push(pl)
			}
		}
		const outs = []
		for (const second of [0, 1]) {
			const cm = await image(), o = await out()
			await rig.run(combine(placedBy[second], cm))
			await rig.run(write(cm, [o], 0))
			outs.push(o)
		}
		return outs
	})
	// several channels in one tick reach the device in one call (the frames are asked for together at the end of the tick)
	{
		await rig.ctx.drain()
		rig.ctx.traceBegin(true)
		let got
		try {
			const outs = []
			for (let c = 0; c < 4; ++c) outs.push(await config2())
			rig.ctx.realise(outs[0]) // (asking for one takes the others of its shape along: one runPrograms call)
		} finally { got = rig.ctx.traceEnd() }
		++checks
		routes['four channels of config 2\'s shape in one tick'] = got
		if (got !== 'chan_compose_batch<0>x4') problems.push({ what: 'four channels in one tick', got, want: 'chan_compose_batch<0>x4' })
	}
	const stats = rig.ctx.deferredStats()
	if (stats.fallbacks) problems.push({ what: 'fallbacks', got: stats.fallbacks, lastFallback: stats.lastFallback })
	console.log(JSON.stringify({ checks, problems, routes }))
	keep.forEach((b) => { try { b.release() } catch (e) { /* released by its job */ } })
	rig.close()
}
main().catch((e) => { process.stderr.write(String(e && e.stack || e) + '\n'); process.exit(1) })
