'use strict'
// Random TICKS of a multi-channel playout through the recording context and through the plain one: per tick 1 - 6 channels post a frame
// each, every channel a random one of the shapes a playout server shows - plain reads, a clip under the Mixer's default fill (v210 or a
// decoder's planar frame), a clip SMALLER than the channel filling it, picture-in-picture over a full-frame clip, a graphic with alpha over
// a clip; whole frames and fields - the reference's deployment (src/index.ts:45-71: channels of one format in one context).  The frames of a tick reach the device
// as ONE runPrograms call: frames for the batch kernel, for the headline kernel's batch form, for the read + 2 x 2-block compositor route
// (alone or grouped by shape) and frames that run in their turn, in every order a seed produces.  Everything a consumer sees must be the
// bytes of the launch-as-posted context.
// usage: node channels_fuzz.js [first seed=1] [seeds=20] [ticks=12] [width=384] [height=54]; prints { seeds, ticks, problems, deferred stats }
const { Rig } = require('../device.js')

const first = parseInt(process.argv[2] || '1')
const seeds = parseInt(process.argv[3] || '20')
const TICKS = parseInt(process.argv[4] || '12')
const W = parseInt(process.argv[5] || '384')
const H = parseInt(process.argv[6] || '54')
const problems = []
const routes = process.env.PH_FUZZ_ROUTES === '1' ? new Map() : null // PH_FUZZ_ROUTES=1: count the recording context's launches by kernel
function rng(seed) { let s = (seed * 2654435761) >>> 0; return () => (s = (Math.imul(s ^ (s >>> 15), 0x2c1b3c6d) + 0x9e3779b9) >>> 0) }
// PH_FUZZ_NO_DEINT=1: progressive shapes only - every one of those folds into a fused launch (deferred.plain stays 0); a 1080i channel whose
// two fields disagree on skipSpatial, or whose placement shrinks or turns the picture, runs its Yadif jobs as recorded
const SHAPES = process.env.PH_FUZZ_NO_DEINT === '1' ? ['plain', 'fill', 'fill', 'small', 'small', 'pip', 'graphic'] : ['plain', 'fill', 'fill', 'small', 'small', 'pip', 'graphic', 'deint', 'deint']
const CLIPS = ['v210', 'v210', 'yuv420p', 'yuv422p10', 'nv12']

async function play(seed, deferred) {
	const rig = await Rig.open({ deviceIndex: 0, deferred, spinWaitMicros: 100 })
	const r = rng(seed)
	const pick = (list) => list[r() % list.length]
	const sw = W / 2
	const sh = H / 2 + ((H / 2) & 1)
	const S = { read: {}, transform: await rig.transform(W, H), write: await rig.pack('v210', W, H, '709', false), writeField: await rig.pack('v210', W, H, '709', true), combine: {} }
	for (const n of [2, 3]) S.combine[n] = await rig.combine(n, W, H)
	S.yadif = await rig.yadif(W, H)
	for (const f of ['v210', 'yuv420p', 'yuv422p10', 'nv12', 'bgra8']) {
		S.read[`${f}|full`] = await rig.unpack(f, W, H, '709', '709')
		S.read[`${f}|small`] = await rig.unpack(f, sw, sh, '709', '709')
	}
	const fillM = await S.transform.matrix({})
	const smallFill = await S.transform.matrix({ scaleX: 0.95, scaleY: 0.95 })
	const insets = [await S.transform.matrix({ scaleX: 0.5, scaleY: 0.5, offsetX: 0.25, offsetY: -0.25 }), await S.transform.matrix({ scaleX: 0.4, scaleY: 0.4, offsetX: -0.2, offsetY: 0.2, rotate: 0.05 })]
	const source = async (fmt, size) => { // a frame of pseudo-random codes on the device
		const [w, h] = size === 'full' ? [W, H] : [sw, sh]
		const planes = await rig.planes(fmt, w, h)
		for (const p of planes) {
			const b = Buffer.alloc(p.length)
			for (let i = 0; i + 4 <= b.length; i += 4) b.writeUInt32LE(fmt === 'v210' ? (((4 + (r() >>> 8) % 1016) | ((4 + (r() >>> 9) % 1016) << 10) | ((4 + (r() >>> 10) % 1016) << 20)) >>> 0) : fmt === 'yuv422p10' ? ((r() % 1024) | ((r() % 1024) << 16)) >>> 0 : r(), i)
			await rig.upload(p, b)
		}
		return planes
	}
	const seen = []
	if (deferred && routes) rig.ctx.traceBegin(false) // (ph_trace_begin: which kernels the recording context's launches were)
	for (let t = 0; t < TICKS; ++t) {
		const C = 1 + r() % 6
		const ids = []
		const outs = []
		for (let c = 0; c < C; ++c) {
			const id = { source: `chan${c}`, timestamp: t }
			const shape = pick(SHAPES)
			const layers = [] // images to combine
			const add = async (fmt, size, matrix) => {
				const planes = await source(fmt, size)
				const im = await rig.image(size === 'full' ? W : sw, size === 'full' ? H : sh)
				rig.post(id, S.read[`${fmt}|${size}`](planes, im), () => planes.forEach((p) => p.release()))
				if (!matrix) { layers.push(im); return }
				const pl = await rig.image(W, H)
				rig.post(id, S.transform(im, pl, matrix), () => im.release())
				layers.push(pl)
			}
			if (shape === 'deint') {
				// the reference's own channel kind (src/index.ts:45-71): 1 - 3 interlaced v210 sources, each a Yadif window of three frames giving
				// both fields (send_field: yadif.ts:100-145), every field placed and combined and written - TWO output frames for the tick
				const n = 1 + r() % 3
				const fields = [[], []]
				for (let l = 0; l < n; ++l) {
					const win = []
					for (let i = 0; i < 3; ++i) {
						const planes = await source('v210', 'full')
						const im = await rig.image(W, H)
						rig.post(id, S.read['v210|full'](planes, im), () => planes.forEach((p) => p.release()))
						win.push(im)
					}
					const m = l === 0 || r() % 2 ? fillM : pick([smallFill, ...insets])
					for (const parity of [0, 1]) {
						const y = await rig.image(W, H)
						rig.post(id, S.yadif(win[0], win[1], win[2], y, { parity, tff: 1, skipSpatial: r() % 4 === 0 ? 1 : 0 }), parity ? () => win.forEach((b) => b.release()) : undefined)
						const pl = await rig.image(W, H)
						rig.post(id, S.transform(y, pl, m), () => y.release())
						fields[parity].push(pl)
					}
				}
				for (const parity of [0, 1]) {
					let frame = fields[parity][0]
					if (n > 1) {
						frame = await rig.image(W, H)
						const these = fields[parity]
						rig.post(id, S.combine[n](these, frame), () => these.forEach((b) => b.release()))
					}
					const out = (await rig.planes('v210', W, H, 'writeonly'))[0]
					const last = frame
					rig.post(id, S.write(last, [out], 0), () => last.release())
					outs.push(out)
				}
				ids.push(id)
				continue
			}
			if (shape === 'plain') { const n = 1 + r() % 3; for (let l = 0; l < n; ++l) await add('v210', 'full', null) }
			else if (shape === 'fill') await add(pick(CLIPS), 'full', fillM)
			else if (shape === 'small') { await add(pick(CLIPS), 'small', fillM); if (r() % 3 === 0) await add(pick(CLIPS), 'small', smallFill) }
			else if (shape === 'pip') { await add(pick(CLIPS), 'full', fillM); await add(pick(CLIPS), r() % 2 ? 'full' : 'small', pick(insets)) }
			else { await add(pick(CLIPS), 'full', fillM); await add('bgra8', 'full', r() % 2 ? fillM : pick(insets)) }
			let frame = layers[0]
			if (layers.length > 1) {
				frame = await rig.image(W, H)
				const these = layers.slice()
				rig.post(id, S.combine[these.length](these, frame), () => these.forEach((b) => b.release()))
			}
			const out = (await rig.planes('v210', W, H, 'writeonly'))[0]
			const last = frame
			// every third frame or so is a FIELD of its channel's frame (the reference's channels are 1080i5000: macadamConsumer.ts:231); the other
			// field's lines are what the frame held before - zeros here
			const interlace = pick([0, 0, 1, 3])
			if (interlace) await rig.upload(out, Buffer.alloc(out.length))
			rig.post(id, (interlace ? S.writeField : S.write)(last, [out], interlace), () => last.release())
			ids.push(id)
			outs.push(out)
		}
		await rig.sync(rig.ctx.queue.load) // (uploads are followed by waitFinish(load) before their frames are used: ffmpegProducer.ts:514-515)
		await Promise.all(ids.map((id) => rig.board.flush(id)))
		// consumers ask in a random order; now and then one asks a tick later
		const order = outs.map((o, i) => i).sort(() => (r() % 3) - 1)
		for (const i of order) { await rig.sync(); await rig.download(outs[i]); seen.push(Buffer.from(outs[i])); outs[i].release() }
	}
	await rig.ctx.drain()
	if (deferred && routes) for (const k of rig.ctx.traceEnd().split('+')) if (k) routes.set(k.replace(/x\d+$/, 'xN'), (routes.get(k.replace(/x\d+$/, 'xN')) || 0) + 1)
	const st = rig.ctx.flushDeferred ? rig.ctx.deferredStats() : null
	rig.close()
	if (rig.ctx.flushDeferred) rig.ctx.flushDeferred()
	rig.ctx.trim()
	const live = rig.ctx.bufferStats().liveBuffers
	return { seen, st, live }
}

async function main() {
	const total = { fused: 0, launched: 0, batched: 0, fallbacks: 0, plain: 0 }
	for (let seed = first; seed < first + seeds; ++seed) {
		const a = await play(seed, false)
		const b = await play(seed, true)
		if (a.seen.length !== b.seen.length) problems.push({ seed, what: `frames seen: plain ${a.seen.length}, deferred ${b.seen.length}` })
		for (let i = 0; i < Math.min(a.seen.length, b.seen.length); ++i)
			if (Buffer.compare(a.seen[i], b.seen[i]) !== 0) { problems.push({ seed, what: `frame ${i} differs` }); break }
		if (a.live || b.live) problems.push({ seed, what: `buffers alive afterwards: plain ${a.live}, deferred ${b.live}` })
		if (b.st && b.st.pending) problems.push({ seed, what: `${b.st.pending} jobs still recorded` })
		if (b.st) for (const k of Object.keys(total)) total[k] += b.st[k] || 0
		if (b.st && b.st.fallbacks) problems.push({ seed, what: `fallback: ${b.st.lastFallback}` })
	}
	process.stdout.write(JSON.stringify({ first, seeds, ticks: TICKS, width: W, height: H, problems: problems.slice(0, 8), deferred: total, routes: routes ? Object.fromEntries([...routes].sort((a, b) => b[1] - a[1])) : undefined }) + '\n')
}
main().catch((e) => { process.stderr.write(String(e && e.stack || e) + '\n'); process.exit(1) })
