'use strict'
// Throughput through node: frames resident on the device, 4 layers -> composite -> v210, three ways
//   (a) one kernel per operator (read x N, combine_N, write), every producer / combiner / consumer flushing its own
//       key as the reference's graph does - the JobBoard drains once per turn;
//   (b) the same with coalescing off: one waitFinish per key (the reference dispatcher's behaviour);
//   (c) the fused channel program, one launch per frame;
//   (d) the job stream of (a) posted to a RECORDING context (node/defer.js) the way the reference's valves post it - a
//       fresh destination image per job, released in the job's callback - and the packed frame asked for by a consumer
//       on the device (ctx.realise): the recording folds each frame's six jobs into the one fused launch of (c).
//   (e) 'channels': PH_NODE_BENCH_CHANNELS channels (default 4) of one format in one recording context, each posting a frame per tick
//       the way (d) does - four placed layers (a full-frame one and three quarter-size insets: the channel kernel's shape), combine_4,
//       write - the frames of a tick reach the device in one launch (ph_chan_compose_batch); us_per_frame is per CHANNEL frame.
//       PHANERON_EARLY_LAUNCH=1: frames are launched at the end of the tick that posted them instead of when the consumer asks.
// usage: node bench_node.js [frames=200] [width=3840] [height=2160] [layers=4]; prints one JSON line per mode.
const { Rig } = require('../device.js')

async function main() {
	const frames = parseInt(process.argv[2] || '200')
	const w = parseInt(process.argv[3] || '3840')
	const h = parseInt(process.argv[4] || '2160')
	const n = parseInt(process.argv[5] || '4')
	const modes = process.env.PH_NODE_BENCH_MODES ? process.env.PH_NODE_BENCH_MODES.split(',') : ['coalesced', 'per-key', 'fused', 'deferred']
	for (const mode of modes) {
		if (mode === 'channels') { await channels(frames, w, h); continue }
		const rig = await Rig.open({ deviceIndex: 0, coalesce: mode !== 'per-key', spinWaitMicros: 200, deferred: mode === 'deferred' })
		const read = await rig.unpack('v210', w, h, '709', '2020')
		const write = await rig.pack('v210', w, h, '2020', false)
		const combine = n > 1 ? await rig.combine(n, w, h) : null
		const fused = await rig.fused(n, w, h, '709', '2020')
		const src = []
		for (let l = 0; l < n; ++l) {
			const p = await rig.planes('v210', w, h)
			await p[0].hostAccess('writeonly', rig.ctx.queue.load) // map, fill, unmap ('none' uploads what was mapped for writing)
			for (let i = 0; i < p[0].length; i += 4) p[0].writeUInt32LE(((0x200 + (i * 2654435761 >>> 22)) & 0x3ff) * 0x00100401 & 0x3fffffff, i)
			await p[0].hostAccess('none', rig.ctx.queue.load)
			src.push(p)
		}
		await rig.sync(rig.ctx.queue.load)
		const rgba = []
		for (let l = 0; l < n; ++l) rgba.push(await rig.image(w, h))
		const comb = await rig.image(w, h)
		const out = await rig.planes('v210', w, h, 'writeonly')
		const ring = [out, await rig.planes('v210', w, h, 'writeonly'), await rig.planes('v210', w, h, 'writeonly')]
		const slotDone = []
		const one = async (f) => {
			if (mode === 'fused') { await rig.run(fused(src.map((p) => p[0]), out[0])); return rig.sync() }
			if (mode === 'deferred') {
				const ids = []
				const fresh = []
				for (let l = 0; l < n; ++l) {
					const im = await rig.image(w, h)
					const id = { source: `L${l}`, timestamp: f }
					rig.post(id, read(src[l], im))
					fresh.push(im)
					ids.push(id)
				}
				const c = { source: 'combine', timestamp: f }
				const cm = combine ? await rig.image(w, h) : fresh[0]
				if (combine) rig.post(c, combine(fresh, cm), () => fresh.forEach((b) => b.release()))
				// three output frames in flight: the host prepares frame f + 1 while the device makes frame f.  A consumer waits for the
				// slot it is about to reuse (the frame made three frames ago), not for the whole queue
				const slot = f % ring.length
				const o = ring[slot]
				if (slotDone[slot] && !slotDone[slot].done()) await slotDone[slot].wait()
				rig.post(c, write(cm, o, 0), () => cm.release())
				ids.push(c)
				await Promise.all(ids.map((id) => rig.board.flush(id)))
				rig.ctx.realise(o[0])
				slotDone[slot] = rig.ctx.recordEvent(rig.ctx.queue.process)
				return undefined
			}
			const ids = []
			for (let l = 0; l < n; ++l) { const id = { source: `L${l}`, timestamp: f }; rig.post(id, read(src[l], rgba[l])); ids.push(id) }
			const c = { source: 'combine', timestamp: f }
			if (combine) rig.post(c, combine(rgba, comb))
			rig.post(c, write(combine ? comb : rgba[0], out, 0))
			ids.push(c)
			await Promise.all(ids.map((id) => rig.board.flush(id)))
		}
		for (let f = 0; f < 10; ++f) await one(f)
		const t0 = process.hrtime.bigint()
		for (let f = 0; f < frames; ++f) await one(10 + f)
		await rig.ctx.drain()
		const sec = Number(process.hrtime.bigint() - t0) / 1e9
		console.log(JSON.stringify({ bench: 'node', mode, width: w, height: h, layers: n, frames, frames_per_sec: +(frames / sec).toFixed(1), us_per_frame: +(1e6 * sec / frames).toFixed(1), drains: rig.board.stats.drains, deferred: rig.ctx.deferredStats() || undefined }))
		;[...src.flat(), ...rgba, comb, ...ring.flat()].forEach((b) => b.release())
		rig.close()
	}
}
// (e): C channels, each a config-2-like frame per tick through the recording context
async function channels(frames, w, h) {
	const C = parseInt(process.env.PH_NODE_BENCH_CHANNELS || '4')
	const n = 4
	const plain = process.env.PH_NODE_BENCH_PLAIN === '1' // layers that are plain reads (no Mixer in the chain): the headline shape per channel
	// file playback: ONE layer per channel, a decoder's yuv420p clip of PH_NODE_BENCH_FILE = "<w>x<h>" under the Mixer's default fill (ffmpegProducer.ts:395-442)
	const file = process.env.PH_NODE_BENCH_FILE ? process.env.PH_NODE_BENCH_FILE.split('x').map((v) => parseInt(v)) : null
	const rig = await Rig.open({ deviceIndex: 0, spinWaitMicros: 200, deferred: true })
	const read = file ? await rig.unpack('yuv420p', file[0], file[1], '709', '709') : await rig.unpack('v210', w, h, '709', '709')
	const write = await rig.pack('v210', w, h, '709', false)
	const combine = await rig.combine(n, w, h)
	const transform = await rig.transform(w, h)
	const PIP = [{}, { scaleX: 0.5, scaleY: 0.5, offsetX: -0.25, offsetY: -0.25 }, { scaleX: 0.5, scaleY: 0.5, offsetX: 0.25, offsetY: -0.25 },
		{ scaleX: 0.5, scaleY: 0.5, offsetX: 0.25, offsetY: 0.25 }]
	const mats = []
	for (const p of PIP) mats.push(await transform.matrix(p))
	const src = []
	for (let c = 0; c < C; ++c) {
		const layers = []
		for (let l = 0; l < (file ? 1 : n); ++l) {
			const p = file ? await rig.planes('yuv420p', file[0], file[1]) : await rig.planes('v210', w, h)
			if (file) {
				for (const plane of p) {
					await plane.hostAccess('writeonly', rig.ctx.queue.load)
					for (let i = 0; i + 4 <= plane.length; i += 4) plane.writeUInt32LE(((i + 977 * c) * 2654435761) >>> 0, i)
					await plane.hostAccess('none', rig.ctx.queue.load)
				}
				layers.push(p)
				continue
			}
			await p[0].hostAccess('writeonly', rig.ctx.queue.load)
			for (let i = 0; i < p[0].length; i += 4) p[0].writeUInt32LE(((0x200 + ((i + 977 * c + 13 * l) * 2654435761 >>> 22)) & 0x3ff) * 0x00100401 & 0x3fffffff, i)
			await p[0].hostAccess('none', rig.ctx.queue.load)
			layers.push(p)
		}
		src.push(layers)
	}
	await rig.sync(rig.ctx.queue.load)
	const ring = []
	for (let c = 0; c < C; ++c) ring.push([await rig.planes('v210', w, h, 'writeonly'), await rig.planes('v210', w, h, 'writeonly'), await rig.planes('v210', w, h, 'writeonly')])
	const slotDone = []
	// PH_NODE_BENCH_INTERLACED=1: the reference's own channel kind - 1080i sources (src/index.ts:45-71), each layer a Yadif window (yadif.ts:88-145,
	// send_field: two fields per source frame), shown at its own size (the Mixer's default fill), combined and written: a TICK is a source frame =
	// two output frames per channel.  The layers' v210 sources go round (the same four frames per layer; every tick one new ToRGBA per layer).
	const interlaced = process.env.PH_NODE_BENCH_INTERLACED === '1'
	const yadif = interlaced ? await rig.yadif(w, h) : null
	const window = [] // per channel and layer: the three images of the Yadif window (prev, cur, next)
	let ring2 = null
	if (interlaced) {
		ring2 = []
		for (let c = 0; c < C; ++c) ring2.push([await rig.planes('v210', w, h, 'writeonly'), await rig.planes('v210', w, h, 'writeonly'), await rig.planes('v210', w, h, 'writeonly')])
		for (let c = 0; c < C; ++c) {
			window.push([])
			for (let l = 0; l < n; ++l) {
				const three = []
				for (let i = 0; i < 3; ++i) { const im = await rig.image(w, h); rig.post({ source: `chan${c}`, timestamp: -1 }, read(src[(c + i) % C][l], im)); three.push(im) }
				window[c].push(three)
			}
		}
		for (let c = 0; c < C; ++c) await rig.board.flush({ source: `chan${c}`, timestamp: -1 })
	}
	let waited = 0, waits = 0 // the host held up by the device: the ring's slot (three ticks back) not yet written
	const oneInterlaced = async (f) => {
		const slot = f % 3
		if (slotDone[slot] && !slotDone[slot].done()) { const w0 = process.hrtime.bigint(); await slotDone[slot].wait(); waited += Number(process.hrtime.bigint() - w0); waits++ }
		const ids = []
		for (let c = 0; c < C; ++c) {
			const id = { source: `chan${c}`, timestamp: f }
			const gone = []
			for (let l = 0; l < n; ++l) { // the window moves on: one new frame per layer
				const im = await rig.image(w, h)
				rig.post(id, read(src[(c + f) % C][l], im))
				gone.push(window[c][l].shift())
				window[c][l].push(im)
			}
			for (const second of [0, 1]) { // yadif.ts:104 parity = tff ^ !isSecond, tff = 1
				const placed = []
				for (let l = 0; l < n; ++l) {
					const u = window[c][l]
					const y = await rig.image(w, h)
					rig.post(id, yadif(u[0], u[1], u[2], y, { parity: second ? 1 : 0, tff: 1, skipSpatial: 0 }))
					const pl = await rig.image(w, h)
					rig.post(id, transform(y, pl, mats[0]), () => y.release())
					placed.push(pl)
				}
				const cm = await rig.image(w, h)
				rig.post(id, combine(placed, cm), () => placed.forEach((b) => b.release()))
				rig.post(id, write(cm, (second ? ring2 : ring)[c][slot], 0), () => cm.release())
			}
			ids.push(id)
			gone.forEach((b) => b.release())
		}
		await Promise.all(ids.map((id) => rig.board.flush(id)))
		for (let c = 0; c < C; ++c) { rig.ctx.realise(ring[c][slot][0]); rig.ctx.realise(ring2[c][slot][0]) }
		slotDone[slot] = rig.ctx.recordEvent(rig.ctx.queue.process)
	}
	const one = interlaced ? oneInterlaced : async (f) => {
		const slot = f % 3
		if (slotDone[slot] && !slotDone[slot].done()) await slotDone[slot].wait()
		const ids = []
		for (let c = 0; c < C; ++c) {
			const id = { source: `chan${c}`, timestamp: f }
			const fresh = []
			const placed = []
			if (file) { // read -> transform (default fill) -> write: what a producer, the Mixer and a consumer post for one clip
				const im = await rig.image(file[0], file[1])
				rig.post(id, read(src[c][0], im))
				const pl = await rig.image(w, h)
				rig.post(id, transform(im, pl, mats[0]), () => im.release())
				rig.post(id, write(pl, ring[c][slot], 0), () => pl.release())
				ids.push(id)
				continue
			}
			for (let l = 0; l < n; ++l) {
				const im = await rig.image(w, h)
				rig.post(id, read(src[c][l], im))
				if (plain) { placed.push(im); continue }
				const pl = await rig.image(w, h)
				rig.post(id, transform(im, pl, mats[l]), () => im.release())
				fresh.push(im)
				placed.push(pl)
			}
			const cm = await rig.image(w, h)
			rig.post(id, combine(placed, cm), () => placed.forEach((b) => b.release()))
			rig.post(id, write(cm, ring[c][slot], 0), () => cm.release())
			ids.push(id)
		}
		await Promise.all(ids.map((id) => rig.board.flush(id)))
		for (let c = 0; c < C; ++c) rig.ctx.realise(ring[c][slot][0])
		slotDone[slot] = rig.ctx.recordEvent(rig.ctx.queue.process)
	}
	for (let f = 0; f < 10; ++f) await one(f)
	// PH_NODE_BENCH_DRY=1: the library chooses its kernels but enqueues nothing (ph_trace_begin(dry)): what is left is the HOST's time per tick
	const dry = process.env.PH_NODE_BENCH_DRY === '1'
	if (dry) { await rig.ctx.drain(); rig.ctx._native.traceBegin(true) }
	const t0 = process.hrtime.bigint()
	for (let f = 0; f < frames; ++f) { await one(10 + f); if (dry && f % 50 === 49) { rig.ctx._native.traceEnd(); rig.ctx._native.traceBegin(true) } }
	await rig.ctx.drain()
	const sec = Number(process.hrtime.bigint() - t0) / 1e9
	if (dry) rig.ctx._native.traceEnd()
	// PH_NODE_BENCH_TRACE=1: the launches one more tick makes, by name (ph_trace_begin: live)
	if (process.env.PH_NODE_BENCH_TRACE === '1' && !dry) { rig.ctx._native.traceBegin(false); await one(10 + frames); await rig.ctx.drain(); console.log(JSON.stringify({ tick_launches: rig.ctx._native.traceEnd() })) }
	console.log(JSON.stringify({ bench: 'node', mode: 'channels', host_only_dry_run: dry || undefined, shape: interlaced ? `${n} x 1080i sources -> yadif (send_field) -> own size -> combine_${n}: two output frames per tick` : file ? `file playback: one ${file[0]}x${file[1]} yuv420p clip under the default fill` : plain ? 'plain reads' : 'config 2', channels: C,
		slot_wait_us_per_tick: interlaced ? +(waited / 1e3 / frames).toFixed(1) : undefined, slot_waits: interlaced ? waits : undefined,
		us_per_field: interlaced ? +(1e6 * sec / frames / C / 2).toFixed(1) : undefined, pack_fields: interlaced ? process.env.PHANERON_PACK_FIELDS !== '0' : undefined, early_launch: rig.ctx.earlyLaunch, width: w, height: h, layers: n, frames, channel_frames_per_sec: +(C * frames / sec).toFixed(1),
		us_per_frame: +(1e6 * sec / frames / C).toFixed(1), us_per_tick: +(1e6 * sec / frames).toFixed(1), deferred: rig.ctx.deferredStats(),
		buffers: rig.ctx.bufferStats() }))
	;[...src.flat(2), ...ring.flat(2), ...(ring2 ? ring2.flat(2) : []), ...window.flat(2)].forEach((b) => b.release())
	rig.close()
}
main().catch((e) => { process.stderr.write(String(e && e.stack || e) + '\n'); process.exit(1) })
