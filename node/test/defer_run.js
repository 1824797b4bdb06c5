'use strict'
// The recording context (node/defer.js) against the plain one, on the GPU: the same operator-by-operator job streams -
// shaped like the reference's valves post them (fresh destination per frame, released in the job callback, the
// consumer mapping the packed frame) - go to `new clContext({deferred: false})` and `({deferred: true})`; every
// frame a consumer would see must be the same bytes, and the counters must show that the deferred side made its
// frames with ONE launch each where the chain has a fused form.
// usage: node defer_run.js [width=384] [height=108]; prints one JSON object { scenarios: [...], problems: [...] }
const { Rig } = require('../device.js')

const W = parseInt(process.argv[2] || '384')
const H = parseInt(process.argv[3] || '108')
const problems = []
const scenarios = []

function lcg(seed) { let s = seed >>> 0; return () => (s = (Math.imul(s, 1664525) + 1013904223) >>> 0) }
function v210Frame(bytes, seed, legal = true) {
	const r = lcg(seed)
	const b = Buffer.alloc(bytes)
	const code = () => (legal ? 64 + (r() >>> 8) % 877 : (r() >>> 8) % 1024)
	for (let i = 0; i + 4 <= bytes; i += 4) b.writeUInt32LE((code() | (code() << 10) | (code() << 20)) >>> 0, i)
	return b
}
function rgbaFrame(w, h, seed, lo = 0, hi = 1) {
	const r = lcg(seed)
	const f = new Float32Array(w * h * 4)
	for (let i = 0; i < f.length; ++i) f[i] = lo + (hi - lo) * ((r() >>> 8) / 16777216)
	return Buffer.from(f.buffer)
}
const PIP = [{}, { scaleX: 0.5, scaleY: 0.5, offsetX: -0.25, offsetY: -0.25 }, { scaleX: 0.5, scaleY: 0.5, offsetX: 0.25, offsetY: -0.25 },
	{ scaleX: 0.5, scaleY: 0.5, offsetX: 0.25, offsetY: 0.25 }]

// one side (plain or deferred): the stages a channel needs and the helpers to drive them the way the valves do
async function side(deferred) {
	const rig = await Rig.open({ deviceIndex: 0, deferred, spinWaitMicros: 100 })
	const s = { rig, deferred }
	s.read = await rig.unpack('v210', W, H, '709', '709')
	s.readHalf = await rig.unpack('v210', W / 2, H / 2, '709', '709')
	s.readAs = {}
	for (const fmt of ['yuv422p10', 'yuv422p8', 'yuv420p', 'nv12', 'rgba8', 'bgra8']) s.readAs[fmt] = await rig.unpack(fmt, W, H, '709', '709')
	s.write = await rig.pack('v210', W, H, '709', false)
	s.writeField = await rig.pack('v210', W, H, '709', true)
	s.writeAs = {}
	for (const fmt of ['rgba8', 'bgra8', 'yuv422p8', 'yuv422p10', 'yuv420p', 'nv12']) s.writeAs[fmt] = await rig.pack(fmt, W, H, '709', false)
	s.combine = {}
	for (const n of [2, 3, 4]) s.combine[n] = await rig.combine(n, W, H)
	s.transform = await rig.transform(W, H)
	s.dissolve = await rig.two('transition_dissolve', W, H)
	s.mixer = await rig.two('mixer', W, H)
	s.wipe = await rig.two('transition_wipe', W, H)
	s.yadif = await rig.yadif(W, H)
	s.yadifHalf = await rig.yadif(W / 2, H / 2)
	s.frame = 0
	// a v210 source on the device
	s.source = async (bytes, w = W, h = H) => {
		const p = await rig.planes('v210', w, h)
		await rig.upload(p[0], bytes)
		await rig.sync(rig.ctx.queue.load) // (an upload is followed by waitFinish(load) before the frame is used on another queue: ffmpegProducer.ts:514-515 -
		// hostAccess('writeonly', load, bytes) only ENQUEUES the copy; a full-size frame read at once on the process queue loses that race)
		return p[0]
	}
	// a planar frame on the device: every 10-bit code in 16-bit samples, or every byte
	s.sourcePlanar = async (fmt, seed) => {
		const planes = await rig.planes(fmt, W, H)
		const r = lcg(seed)
		for (const p of planes) {
			const b = Buffer.alloc(p.length)
			if (fmt === 'yuv422p10') for (let i = 0; i + 2 <= b.length; i += 2) b.writeUInt16LE((r() >>> 8) % 1024, i)
			else for (let i = 0; i < b.length; ++i) b[i] = (r() >>> 8) & 255
			await rig.upload(p, b)
		}
		await rig.sync(rig.ctx.queue.load)
		return planes
	}
	s.image = async (bytes) => {
		const im = await rig.image(W, H)
		await rig.upload(im, bytes)
		await rig.sync(rig.ctx.queue.load)
		return im
	}
	// post a job as a valve does: destination released by the poster once the job's callback has fired
	s.id = (name) => ({ source: name, timestamp: s.frame })
	s.flush = (id) => rig.board.flush(id)
	s.consume = async (out) => { await rig.sync(); await rig.download(out); return Buffer.from(out) } // the consumer's saveFrame, after its jobs' waitFinish (clJobQueue.ts:131)
	return s
}

// a scenario: fn(side) -> array of Buffers a consumer saw; run on both sides and compared
async function scenario(name, fn, expect) {
	const got = []
	const stats = []
	// PH_DEFER_SELF=plain|deferred: the same side twice (is a side's own output repeatable?); PH_DEFER_ONLY=<text>: scenarios whose name has it
	if (process.env.PH_DEFER_ONLY && !name.includes(process.env.PH_DEFER_ONLY)) return
	const sides = process.env.PH_DEFER_SELF === 'plain' ? [false, false] : process.env.PH_DEFER_SELF === 'deferred' ? [true, true] : [false, true]
	for (const deferred of sides) {
		const s = await side(deferred)
		try {
			got.push(await fn(s))
			const st = s.rig.ctx.flushDeferred ? s.rig.ctx.deferredStats() : null
			stats.push(st)
			s.rig.close()
			const left = s.rig.ctx.flushDeferred()
			const live = s.rig.ctx.bufferStats()
			if (deferred && left && left.pending) problems.push({ scenario: name, what: `${left.pending} recorded jobs are still pending after everything was released` })
			if (live.liveBuffers !== 0) problems.push({ scenario: name, what: `${live.liveBuffers} buffers still alive on the ${deferred ? 'deferred' : 'plain'} side` })
		} catch (e) {
			problems.push({ scenario: name, what: `${deferred ? 'deferred' : 'plain'} side: ${e && e.stack || e}` })
			got.push([])
			stats.push(null)
		}
	}
	const [plain, lazy] = got
	if (process.env.PHANERON_DEFER_DUMP) {
		const fs = require('fs')
		plain.forEach((b, i) => fs.writeFileSync(`${process.env.PHANERON_DEFER_DUMP}/s${scenarios.length}_plain_${i}.bin`, b))
		lazy.forEach((b, i) => fs.writeFileSync(`${process.env.PHANERON_DEFER_DUMP}/s${scenarios.length}_deferred_${i}.bin`, b))
	}
	if (plain.length !== lazy.length || !plain.length) problems.push({ scenario: name, what: `frames seen: plain ${plain.length}, deferred ${lazy.length}` })
	for (let i = 0; i < Math.min(plain.length, lazy.length); ++i)
		if (Buffer.compare(plain[i], lazy[i]) !== 0) {
			let at = 0
			while (at < plain[i].length && plain[i][at] === lazy[i][at]) ++at
			let last = plain[i].length - 1, n = 0
			while (last > at && plain[i][last] === lazy[i][last]) --last
			for (let k = at; k <= last; ++k) if (plain[i][k] !== lazy[i][k]) ++n
			problems.push({ scenario: name, what: `frame ${i} differs from byte ${at} of ${plain[i].length}`, last, differing: n })
		}
	const st = stats[1]
	if (st && expect) for (const k of Object.keys(expect)) if (st[k] !== expect[k]) problems.push({ scenario: name, what: `deferred counter ${k} = ${st[k]}, expected ${expect[k]}`, stats: st })
	scenarios.push({ name, frames: plain.length, deferred: st })
}

const v210Bytes = (w, h) => (Math.ceil(w / 48) * 128) * h

async function main() {
	const full = v210Bytes(W, H)
	const half = v210Bytes(W / 2, H / 2)

	// the headline's chain, as the producers / combiner / consumer post it: read x4 | combine_4 + write, three frames in flight
	await scenario('read x4 -> combine_4 -> write, three frames in flight', async (s) => {
		const seen = []
		const outs = []
		for (let f = 0; f < 3; ++f) {
			s.frame = f
			const srcs = []
			for (let l = 0; l < 4; ++l) srcs.push(await s.source(v210Frame(full, 100 + 10 * f + l, l !== 2)))
			const rgba = []
			const ids = []
			for (let l = 0; l < 4; ++l) {
				const im = await s.rig.image(W, H)
				const id = s.id(`L${l}`)
				s.rig.post(id, s.read([srcs[l]], im), () => srcs[l].release()) // io.ts: the source goes once its frame is unpacked
				rgba.push(im)
				ids.push(id)
			}
			const comb = await s.rig.image(W, H)
			const out = (await s.rig.planes('v210', W, H, 'writeonly'))[0]
			const c = s.id('combine')
			s.rig.post(c, s.combine[4](rgba, comb), () => rgba.forEach((b) => b.release()))
			s.rig.post(c, s.write(comb, [out], 0), () => comb.release())
			ids.push(c)
			await Promise.all(ids.map((id) => s.flush(id)))
			outs.push(out)
		}
		for (const out of outs) { seen.push(await s.consume(out)); out.release() }
		return seen
	}, { fused: 3, plain: 0, launched: 1, batched: 3 }) // asked for together, the three frames go as one call (ph_fused_v210_combine_batch)

	// BASELINE config 2's shape: placed layers, a wipe against a placed second source with an image mask
	await scenario('PiP transforms + wipe on the top layer -> combine_4 -> write', async (s) => {
		s.frame = 1
		const srcs = []
		for (let l = 0; l < 5; ++l) srcs.push(await s.source(v210Frame(full, 200 + l)))
		const mask = await s.image(rgbaFrame(W, H, 77))
		const unpacked = []
		for (let l = 0; l < 5; ++l) {
			const im = await s.rig.image(W, H)
			s.rig.post(s.id('read'), s.read([srcs[l]], im), () => srcs[l].release())
			unpacked.push(im)
		}
		const placed = []
		for (let l = 0; l < 5; ++l) {
			const im = await s.rig.image(W, H)
			s.rig.post(s.id('mix'), s.transform(unpacked[l], im, await s.transform.matrix(PIP[Math.min(l, 3)])), () => unpacked[l].release())
			placed.push(im)
		}
		const wiped = await s.rig.image(W, H)
		s.rig.post(s.id('mix'), s.wipe(placed[3], placed[4], mask, wiped), () => { placed[3].release(); placed[4].release(); mask.release() })
		const comb = await s.rig.image(W, H)
		const layers = [placed[0], placed[1], placed[2], wiped]
		s.rig.post(s.id('mix'), s.combine[4](layers, comb), () => layers.slice(0, 3).forEach((b) => b.release()))
		s.rig.post(s.id('mix'), s.combine[4](layers, comb), () => wiped.release()) // (posted twice on purpose: the second supersedes the first)
		const out = (await s.rig.planes('v210', W, H, 'writeonly'))[0]
		s.rig.post(s.id('mix'), s.write(comb, [out], 0), () => comb.release())
		await s.flush(s.id('read'))
		await s.flush(s.id('mix'))
		const seen = [await s.consume(out)]
		out.release()
		return seen
	}, { fused: 1, plain: 0, launched: 1 })

	// the reference's deployment: several channels of one format in one context (src/index.ts:45-71), their frames posted in the same
	// tick - the recording hands them to the device as ONE launch (runPrograms -> ph_chan_compose_batch); a fifth channel with a
	// dissolve in progress and a sixth that is a plain 1:1 layer go along; so does a channel whose background is a planar (yuv420p) clip
	await scenario('six channels of one format posted in one tick', async (s) => {
		s.frame = 9
		const outs = []
		const flushes = []
		for (let c = 0; c < 6; ++c) {
			const n = c === 5 ? 1 : 2
			const srcs = []
			for (let l = 0; l < n + (c === 4 ? 1 : 0); ++l) srcs.push(await s.source(v210Frame(full, 900 + 10 * c + l, l !== 1)))
			const id = s.id(`chan${c}`)
			const unpacked = []
			for (let l = 0; l < srcs.length; ++l) {
				const im = await s.rig.image(W, H)
				if (c === 3 && l === 0) { // channel 3's background is a file decoder's frame: the library runs that channel's frame in its turn inside the same call
					const planes = await s.sourcePlanar('yuv420p', 950)
					s.rig.post(id, s.readAs.yuv420p(planes, im), () => { planes.forEach((p) => p.release()); srcs[l].release() })
				} else s.rig.post(id, s.read([srcs[l]], im), () => srcs[l].release())
				unpacked.push(im)
			}
			const placed = []
			for (let l = 0; l < srcs.length; ++l) {
				const im = await s.rig.image(W, H)
				const where = c === 5 ? {} : l === 0 ? {} : Object.assign({}, PIP[1 + (c + l) % 3], { rotate: 0.02 * c })
				s.rig.post(id, s.transform(unpacked[l], im, await s.transform.matrix(where)), () => unpacked[l].release())
				placed.push(im)
			}
			let layers = placed
			if (c === 4) { // the top layer in mid-dissolve against a third source
				const mixed = await s.rig.image(W, H)
				s.rig.post(id, s.dissolve(placed[1], placed[2], 0.375, mixed), () => { placed[1].release(); placed[2].release() })
				layers = [placed[0], mixed]
			}
			let frame = layers[0]
			if (n > 1) {
				frame = await s.rig.image(W, H)
				const these = layers
				s.rig.post(id, s.combine[2](these, frame), () => these.forEach((b) => b.release()))
			}
			const out = (await s.rig.planes('v210', W, H, 'writeonly'))[0]
			const last = frame
			s.rig.post(id, s.write(last, [out], 0), () => last.release())
			outs.push(out)
			flushes.push(id)
		}
		await Promise.all(flushes.map((id) => s.flush(id)))
		const seen = []
		for (const out of outs) { seen.push(await s.consume(out)); out.release() }
		return seen
	}, { fused: 6, plain: 0, launched: 1, batched: 6 })

	// channels whose layers are plain reads of the output's size (the headline shape): their frames of one tick are ONE call too -
	// runPrograms -> ph_fused_v210_combine_batch for the four with three layers each, the two-layer channel in its own launch inside the same call
	await scenario('five channels of plain reads posted in one tick', async (s) => {
		s.frame = 10
		const outs = []
		const flushes = []
		for (let c = 0; c < 5; ++c) {
			const n = c === 4 ? 2 : 3
			const id = s.id(`plain${c}`)
			const unpacked = []
			for (let l = 0; l < n; ++l) {
				const src = await s.source(v210Frame(full, 1200 + 10 * c + l, l !== 2))
				const im = await s.rig.image(W, H)
				s.rig.post(id, s.read([src], im), () => src.release())
				unpacked.push(im)
			}
			const frame = await s.rig.image(W, H)
			s.rig.post(id, s.combine[n](unpacked, frame), () => unpacked.forEach((b) => b.release()))
			const out = (await s.rig.planes('v210', W, H, 'writeonly'))[0]
			s.rig.post(id, s.write(frame, [out], 0), () => frame.release())
			outs.push(out)
			flushes.push(id)
		}
		await Promise.all(flushes.map((id) => s.flush(id)))
		const seen = []
		for (const out of outs) { seen.push(await s.consume(out)); out.release() }
		return seen
	}, { fused: 5, plain: 0, launched: 1, batched: 5 })

	// a dissolve against a half-size, rotated incoming source, over a plain read
	await scenario('dissolve layer over a plain read', async (s) => {
		s.frame = 2
		const a = await s.source(v210Frame(full, 300))
		const b = await s.source(v210Frame(half, 301), W / 2, H / 2)
		const ua = await s.rig.image(W, H)
		const ub = await s.rig.image(W / 2, H / 2)
		await s.rig.run(s.read([a], ua))
		await s.rig.run(s.readHalf([b], ub))
		const pa = await s.rig.image(W, H)
		const pb = await s.rig.image(W, H)
		await s.rig.run(s.transform(ua, pa, await s.transform.matrix({})))
		await s.rig.run(s.transform(ub, pb, await s.transform.matrix({ scaleX: 0.8, scaleY: 0.8, rotate: 0.05 })))
		const d = await s.rig.image(W, H)
		await s.rig.run(s.dissolve(pa, pb, 0.75, d))
		const comb = await s.rig.image(W, H)
		await s.rig.run(s.combine[2]([ua, d], comb))
		const out = (await s.rig.planes('v210', W, H, 'writeonly'))[0]
		await s.rig.run(s.write(comb, [out], 0))
		await s.rig.sync()
		const seen = [await s.consume(out)]
		// intermediates somebody asks for after all: made then, from their recipes
		seen.push(await s.consume(d), await s.consume(ub))
		;[a, b, ua, ub, pa, pb, d, comb, out].forEach((x) => x.release())
		return seen
	}, { fused: 1 })

	// `mixer` (mix.ts: the scalar mix of two images) is the dissolve's arithmetic under another name: its chain folds the same way
	await scenario('a mixer between two placed sources, under a plain layer', async (s) => {
		s.frame = 3
		const srcs = []
		for (let l = 0; l < 3; ++l) srcs.push(await s.source(v210Frame(full, 320 + l, l !== 1)))
		const u = []
		for (let l = 0; l < 3; ++l) { const im = await s.rig.image(W, H); s.rig.post(s.id('mix'), s.read([srcs[l]], im), () => srcs[l].release()); u.push(im) }
		const p = []
		for (let l = 0; l < 2; ++l) { const im = await s.rig.image(W, H); s.rig.post(s.id('mix'), s.transform(u[l], im, await s.transform.matrix(PIP[l])), () => u[l].release()); p.push(im) }
		const mixed = await s.rig.image(W, H)
		s.rig.post(s.id('mix'), s.mixer(p[0], p[1], 0.3125, mixed), () => p.forEach((b) => b.release()))
		const comb = await s.rig.image(W, H)
		s.rig.post(s.id('mix'), s.combine[2]([mixed, u[2]], comb), () => { mixed.release(); u[2].release() })
		const out = (await s.rig.planes('v210', W, H, 'writeonly'))[0]
		s.rig.post(s.id('mix'), s.write(comb, [out], 0), () => comb.release())
		await s.flush(s.id('mix'))
		const seen = [await s.consume(out)]
		out.release()
		return seen
	}, { fused: 1, plain: 0, launched: 1 })

	// a finished image (a routed frame, real alpha) as a layer over a plain read and under a placed one
	await scenario('a finished image as a layer', async (s) => {
		s.frame = 2
		const a = await s.source(v210Frame(full, 310))
		const b = await s.source(v210Frame(full, 311))
		const routed = await s.image(rgbaFrame(W, H, 302, -0.05, 1.05))
		const ua = await s.rig.image(W, H)
		const ub = await s.rig.image(W, H)
		await s.rig.run(s.read([a], ua))
		await s.rig.run(s.read([b], ub))
		const pb = await s.rig.image(W, H)
		await s.rig.run(s.transform(ub, pb, await s.transform.matrix(PIP[2])))
		const comb = await s.rig.image(W, H)
		await s.rig.run(s.combine[3]([ua, routed, pb], comb))
		const out = (await s.rig.planes('v210', W, H, 'writeonly'))[0]
		await s.rig.run(s.write(comb, [out], 0))
		const seen = [await s.consume(out)]
		;[a, b, routed, ua, ub, pb, comb, out].forEach((x) => x.release())
		return seen
	}, { fused: 1, plain: 0 })

	// one layer: the combiner passes it through (combiner.ts:213-217) - read -> write and read -> transform -> write
	await scenario('single-layer channels', async (s) => {
		s.frame = 2
		const a = await s.source(v210Frame(full, 320, false))
		const ua = await s.rig.image(W, H)
		await s.rig.run(s.read([a], ua))
		const pa = await s.rig.image(W, H)
		await s.rig.run(s.transform(ua, pa, await s.transform.matrix({ scaleX: 1.5, scaleY: 1.5, rotate: -0.1 })))
		const out1 = (await s.rig.planes('v210', W, H, 'writeonly'))[0]
		const out2 = (await s.rig.planes('v210', W, H, 'writeonly'))[0]
		await s.rig.run(s.write(ua, [out1], 0))
		await s.rig.run(s.write(pa, [out2], 0))
		const seen = [await s.consume(out1), await s.consume(out2)]
		;[a, ua, pa, out1, out2].forEach((x) => x.release())
		return seen
	}, { fused: 2, plain: 0 })

	// operands that change under a recording: a source frame and a placement matrix overwritten before the result is asked for
	await scenario('source and matrix overwritten while recorded', async (s) => {
		s.frame = 3
		const a = await s.source(v210Frame(full, 400))
		const b = await s.source(v210Frame(full, 401))
		const ua = await s.rig.image(W, H)
		const ub = await s.rig.image(W, H)
		await s.rig.run(s.read([a], ua))
		await s.rig.run(s.read([b], ub))
		const m = await s.rig.ctx.createBuffer(48, 'readonly', 'none', undefined, 'matrix')
		const set = async (placement) => {
			const f = new Float32Array(12)
			f.set(require('../index.js').colour.transformMatrix(W, H, placement))
			await m.hostAccess('writeonly', s.rig.ctx.queue.load, Buffer.from(f.buffer))
		}
		await set({ scaleX: 0.5, scaleY: 0.5 })
		const pb = await s.rig.image(W, H)
		await s.rig.run(s.transform(ub, pb, m))
		const comb = await s.rig.image(W, H)
		await s.rig.run(s.combine[2]([ua, pb], comb))
		const out = (await s.rig.planes('v210', W, H, 'writeonly'))[0]
		await s.rig.run(s.write(comb, [out], 0))
		await s.rig.sync()
		await s.rig.upload(a, v210Frame(full, 402)) // the next frame into the same source buffer
		await set({ scaleX: 0.25, scaleY: 0.25, rotate: 0.1 }) // the next placement into the same matrix buffer
		const seen = [await s.consume(out)] // must show frame 400 / 401 at half size
		;[a, b, ua, ub, m, pb, comb, out].forEach((x) => x.release())
		return seen
	})

	// interlaced output: two field writes into one frame from two different composites; a de-interlaced layer
	await scenario('field writes into one frame; a yadif layer', async (s) => {
		s.frame = 4
		const win = []
		for (let i = 0; i < 3; ++i) win.push(await s.source(v210Frame(full, 500 + i)))
		const bg = await s.source(v210Frame(full, 510))
		const u = []
		for (let i = 0; i < 3; ++i) { const im = await s.rig.image(W, H); await s.rig.run(s.read([win[i]], im)); u.push(im) }
		const ubg = await s.rig.image(W, H)
		await s.rig.run(s.read([bg], ubg))
		const out = (await s.rig.planes('v210', W, H, 'writeonly'))[0]
		await s.rig.upload(out, Buffer.alloc(full, 0x2a))
		await s.rig.sync(s.rig.ctx.queue.load)
		for (const field of [1, 3]) {
			const y = await s.rig.image(W, H)
			await s.rig.run(s.yadif(u[0], u[1], u[2], y, { parity: field === 1 ? 0 : 1, tff: 1, skipSpatial: 0 }))
			const py = await s.rig.image(W, H)
			await s.rig.run(s.transform(y, py, await s.transform.matrix({ scaleX: 0.6, scaleY: 0.6, offsetX: field === 1 ? 0.1 : -0.1 })))
			const comb = await s.rig.image(W, H)
			await s.rig.run(s.combine[2]([ubg, py], comb))
			await s.rig.run(s.writeField(comb, [out], field))
			;[y, py, comb].forEach((x) => x.release())
		}
		const seen = [await s.consume(out)]
		;[...win, bg, ...u, ubg, out].forEach((x) => x.release())
		return seen
	}, { fused: 3, plain: 0, launched: 3 }) // the yadif pair in one launch (both fields are recorded before the first is asked for), one compositor launch per field

	// BASELINE config 3's shape: interlaced half-size sources, both fields of every layer de-interlaced (the Yadif valve posts
	// the pair), each enlarged to the channel's size by the default fill, combine, one packed frame per field
	await scenario('de-interlaced layers enlarged 2x, both fields', async (s) => {
		s.frame = 5
		const L = 2
		const u = []
		const srcs = []
		for (let l = 0; l < L; ++l) {
			u.push([])
			for (let i = 0; i < 3; ++i) {
				const src = await s.source(v210Frame(half, 600 + 10 * l + i), W / 2, H / 2)
				const im = await s.rig.image(W / 2, H / 2)
				await s.rig.run(s.readHalf([src], im))
				srcs.push(src)
				u[l].push(im)
			}
		}
		const fields = [[], []]
		for (let l = 0; l < L; ++l)
			for (const second of [0, 1]) { // yadif.ts:104 parity = tff ^ !isSecond, tff = 1
				const y = await s.rig.image(W / 2, H / 2)
				await s.rig.run(s.yadifHalf(u[l][0], u[l][1], u[l][2], y, { parity: second ? 1 : 0, tff: 1, skipSpatial: 0 }))
				fields[second].push(y)
			}
		const seen = []
		const fill = await s.transform.matrix({})
		for (const second of [0, 1]) {
			const placed = []
			for (let l = 0; l < L; ++l) {
				const im = await s.rig.image(W, H)
				await s.rig.run(s.transform(fields[second][l], im, fill))
				fields[second][l].release()
				placed.push(im)
			}
			const comb = await s.rig.image(W, H)
			await s.rig.run(s.combine[2](placed, comb))
			const out = (await s.rig.planes('v210', W, H, 'writeonly'))[0]
			await s.rig.run(s.write(comb, [out], 0))
			;[...placed, comb].forEach((x) => x.release())
			seen.push(await s.consume(out))
			out.release()
		}
		;[...srcs, ...u.flat()].forEach((x) => x.release())
		return seen
	}, { fused: 3, plain: 0, launched: 3, fallbacks: 0 })

	// the same with BOTH fields' chains posted before the first frame is asked for (a consumer that takes frames in pairs, or one
	// that runs a frame behind): the de-interlacing reader's pair launch, then both fields' compositors in ONE launch
	await scenario('de-interlaced layers enlarged 2x, both fields posted before either is consumed', async (s) => {
		s.frame = 5
		const L = 3
		const u = []
		const srcs = []
		for (let l = 0; l < L; ++l) {
			u.push([])
			for (let i = 0; i < 3; ++i) {
				const src = await s.source(v210Frame(half, 650 + 10 * l + i), W / 2, H / 2)
				const im = await s.rig.image(W / 2, H / 2)
				await s.rig.run(s.readHalf([src], im))
				srcs.push(src)
				u[l].push(im)
			}
		}
		const fill = await s.transform.matrix({})
		const outs = []
		for (const second of [0, 1]) {
			const placed = []
			for (let l = 0; l < L; ++l) {
				const y = await s.rig.image(W / 2, H / 2)
				await s.rig.run(s.yadifHalf(u[l][0], u[l][1], u[l][2], y, { parity: second ? 1 : 0, tff: 1, skipSpatial: 0 }))
				const im = await s.rig.image(W, H)
				await s.rig.run(s.transform(y, im, fill))
				y.release()
				placed.push(im)
			}
			const comb = await s.rig.image(W, H)
			await s.rig.run(s.combine[3](placed, comb))
			const out = (await s.rig.planes('v210', W, H, 'writeonly'))[0]
			await s.rig.run(s.write(comb, [out], 0))
			;[...placed, comb].forEach((x) => x.release())
			outs.push(out)
		}
		const seen = []
		for (const out of outs) { seen.push(await s.consume(out)); out.release() }
		;[...srcs, ...u.flat()].forEach((x) => x.release())
		return seen
	}, { fused: 2, plain: 0, launched: 2, fallbacks: 0 })

	// the reference's deployment: several channels of 1080i sources posting both fields' frames in one tick (src/index.ts:45-71) - each channel's
	// windows go to the de-interlacing reader and, at once, its two frames to the compositor (node/defer.js _runMany; PHANERON_FIELD_BATCH=1: the
	// channels' windows and frames in shared launches) - whichever order, every consumer's frame is the launch-as-posted context's
	await scenario('three channels of two 1080i sources each, both fields, posted in one tick', async (s) => {
		s.frame = 6
		const fill = await s.transform.matrix({})
		const outs = []
		const held = []
		for (let c = 0; c < 3; ++c) {
			const u = []
			for (let l = 0; l < 2; ++l) {
				u.push([])
				for (let i = 0; i < 3; ++i) {
					const src = await s.source(v210Frame(full, 900 + 40 * c + 10 * l + i), W, H)
					const im = await s.rig.image(W, H)
					await s.rig.run(s.read([src], im))
					held.push(src, im)
					u[l].push(im)
				}
			}
			for (const second of [0, 1]) {
				const placed = []
				for (let l = 0; l < 2; ++l) {
					const y = await s.rig.image(W, H)
					await s.rig.run(s.yadif(u[l][0], u[l][1], u[l][2], y, { parity: second ? 1 : 0, tff: 1, skipSpatial: 0 }))
					const im = await s.rig.image(W, H)
					await s.rig.run(s.transform(y, im, fill))
					y.release()
					placed.push(im)
				}
				const comb = await s.rig.image(W, H)
				await s.rig.run(s.combine[2](placed, comb))
				const out = (await s.rig.planes('v210', W, H, 'writeonly'))[0]
				await s.rig.run(s.write(comb, [out], 0))
				;[...placed, comb].forEach((x) => x.release())
				outs.push(out)
			}
		}
		const seen = []
		for (const i of [3, 0, 5, 1, 2, 4]) { seen.push(await s.consume(outs[i])); outs[i].release() } // (consumers in an order of their own)
		held.forEach((x) => x.release())
		return seen
	}, { plain: 0, fallbacks: 0 })

	// the same shape, and then the APPLICATION looks at a field itself (a preview, a ROUTE of the layer): the pair launch wrote the fields packed
	// (12 bytes per pixel, only the compositor was going to read them) - whoever else asks gets the RGBA image the buffer is declared as
	await scenario('de-interlaced layers, both fields composited, then a field image read by the application', async (s) => {
		s.frame = 5
		const u = []
		const srcs = []
		for (let i = 0; i < 3; ++i) {
			const src = await s.source(v210Frame(half, 680 + i), W / 2, H / 2)
			const im = await s.rig.image(W / 2, H / 2)
			await s.rig.run(s.readHalf([src], im))
			srcs.push(src)
			u.push(im)
		}
		const fill = await s.transform.matrix({})
		const outs = []
		const fields = []
		for (const second of [0, 1]) {
			const y = await s.rig.image(W / 2, H / 2)
			await s.rig.run(s.yadifHalf(u[0], u[1], u[2], y, { parity: second ? 1 : 0, tff: 1, skipSpatial: 0 }))
			const im = await s.rig.image(W, H)
			await s.rig.run(s.transform(y, im, fill))
			fields.push(y) // (kept: the application will read them)
			const out = (await s.rig.planes('v210', W, H, 'writeonly'))[0]
			await s.rig.run(s.write(im, [out], 0))
			im.release()
			outs.push(out)
		}
		const seen = []
		for (const out of outs) { seen.push(await s.consume(out)); out.release() }
		for (const y of fields) { seen.push(await s.consume(y)); y.release() }
		;[...srcs, ...u].forEach((x) => x.release())
		return seen
	}, { fused: 2, plain: 0, launched: 4, fallbacks: 0 }) // the pair launch, both compositors in one launch, two fields unpacked

	// finished images only, one of them rotated: the tap-sharing compositor declines, the channel kernel takes them
	await scenario('finished images, placed and rotated', async (s) => {
		s.frame = 6
		const warm = await s.source(v210Frame(full, 700)) // (a Loader recipe has been seen on this context)
		const uw = await s.rig.image(W, H)
		await s.rig.run(s.read([warm], uw))
		const a = await s.image(rgbaFrame(W, H, 701, 0, 1))
		const b = await s.image(rgbaFrame(W, H, 702, -0.1, 1.1))
		const pa = await s.rig.image(W, H)
		const pb = await s.rig.image(W, H)
		await s.rig.run(s.transform(a, pa, await s.transform.matrix({ scaleX: 1.5, scaleY: 1.5 })))
		await s.rig.run(s.transform(b, pb, await s.transform.matrix({ scaleX: 0.7, scaleY: 0.7, rotate: 0.125 })))
		const comb = await s.rig.image(W, H)
		await s.rig.run(s.combine[2]([pa, pb], comb))
		const out = (await s.rig.planes('v210', W, H, 'writeonly'))[0]
		await s.rig.run(s.write(comb, [out], 0))
		const seen = [await s.consume(out)]
		;[warm, uw, a, b, pa, pb, comb, out].forEach((x) => x.release())
		return seen
	}, { fused: 1, plain: 0, fallbacks: 0 }) // (a rotated placement is none of the 2 x 2-block compositor's: decided from the matrix, not by a refused launch)

	// file sources: planar frames as decoders hand them over - 10-bit 4:2:2 (the v210 Loader recipe) and the 8-bit formats (a Loader
	// matrix of their own) - beside a v210 one, placed, combined, packed
	await scenario('planar sources of every format', async (s) => {
		s.frame = 7
		const seen = []
		for (const [fa, fb] of [['yuv422p10', 'yuv420p'], ['nv12', 'yuv422p8']]) {
			const a = await s.sourcePlanar(fa, 800)
			const b = await s.sourcePlanar(fb, 801)
			const c = await s.source(v210Frame(full, 802))
			const ua = await s.rig.image(W, H)
			const ub = await s.rig.image(W, H)
			const uc = await s.rig.image(W, H)
			s.rig.post(s.id('A'), s.readAs[fa](a, ua), () => a.forEach((p) => p.release()))
			s.rig.post(s.id('B'), s.readAs[fb](b, ub), () => b.forEach((p) => p.release()))
			s.rig.post(s.id('C'), s.read([c], uc), () => c.release())
			const pb = await s.rig.image(W, H)
			const pc = await s.rig.image(W, H)
			s.rig.post(s.id('B'), s.transform(ub, pb, await s.transform.matrix(PIP[1])), () => ub.release())
			s.rig.post(s.id('C'), s.transform(uc, pc, await s.transform.matrix({ scaleX: 0.6, scaleY: 0.6, rotate: -0.08, offsetX: 0.15 })), () => uc.release())
			const comb = await s.rig.image(W, H)
			s.rig.post(s.id('mix'), s.combine[3]([ua, pb, pc], comb), () => [ua, pb, pc].forEach((x) => x.release()))
			const out = (await s.rig.planes('v210', W, H, 'writeonly'))[0]
			s.rig.post(s.id('mix'), s.write(comb, [out], 0), () => comb.release())
			for (const k of ['A', 'B', 'C', 'mix']) await s.flush(s.id(k))
			seen.push(await s.consume(out))
			out.release()
			s.frame++
		}
		return seen
	}, { fused: 2, plain: 0, launched: 2, fallbacks: 0 })

	// graphics: packed 8-bit RGB frames with real alpha (a logo, a lower third) over a v210 frame, one placed small
	await scenario('packed RGB graphics with alpha over v210', async (s) => {
		s.frame = 9
		const bg = await s.source(v210Frame(full, 900))
		const ga = await s.sourcePlanar('bgra8', 901)
		const gb = await s.sourcePlanar('rgba8', 902)
		const ubg = await s.rig.image(W, H)
		const ua = await s.rig.image(W, H)
		const ub = await s.rig.image(W, H)
		await s.rig.run(s.read([bg], ubg))
		await s.rig.run(s.readAs.bgra8(ga, ua))
		await s.rig.run(s.readAs.rgba8(gb, ub))
		const pb = await s.rig.image(W, H)
		await s.rig.run(s.transform(ub, pb, await s.transform.matrix({ scaleX: 0.25, scaleY: 0.25, offsetX: 0.35, offsetY: -0.35 })))
		const comb = await s.rig.image(W, H)
		await s.rig.run(s.combine[3]([ubg, ua, pb], comb))
		const out = (await s.rig.planes('v210', W, H, 'writeonly'))[0]
		await s.rig.run(s.write(comb, [out], 0))
		const seen = [await s.consume(out)]
		;[bg, ...ga, ...gb, ubg, ua, ub, pb, comb, out].forEach((x) => x.release())
		return seen
	}, { fused: 1, plain: 0, launched: 1, fallbacks: 0 })

	// the reference's other consumers: the screen maps an rgba8 frame (screenConsumer.ts:131), an encoder yuv422p8 planes (ffmpegConsumer.ts:144)
	await scenario('frames for the screen and for an encoder', async (s) => {
		s.frame = 10
		const seen = []
		for (const fmt of ['rgba8', 'yuv422p8', 'bgra8', 'yuv422p10', 'yuv420p', 'nv12']) { // (the last two: 4:2:0 frames, chroma from the upper line of a pair)
			const a = await s.source(v210Frame(full, 1000))
			const b = await s.sourcePlanar('yuv420p', 1001)
			const ua = await s.rig.image(W, H)
			const ub = await s.rig.image(W, H)
			s.rig.post(s.id('A'), s.read([a], ua), () => a.release())
			s.rig.post(s.id('B'), s.readAs.yuv420p(b, ub), () => b.forEach((p) => p.release()))
			const pb = await s.rig.image(W, H)
			s.rig.post(s.id('B'), s.transform(ub, pb, await s.transform.matrix(PIP[2])), () => ub.release())
			const comb = await s.rig.image(W, H)
			s.rig.post(s.id('mix'), s.combine[2]([ua, pb], comb), () => [ua, pb].forEach((x) => x.release()))
			const out = await s.rig.planes(fmt, W, H, 'writeonly')
			s.rig.post(s.id('mix'), s.writeAs[fmt](comb, out, 0), () => comb.release())
			for (const k of ['A', 'B', 'mix']) await s.flush(s.id(k))
			for (const p of out) { seen.push(await s.consume(p)); p.release() }
			s.frame++
		}
		return seen
	}, { fused: 6, plain: 0, launched: 6, fallbacks: 0 })

	// interlaced file sources: windows of planar frames (4:2:2: yuv422p10, yuv422p8; 4:2:0: yuv420p, nv12 - interlaced H.264), both fields de-interlaced, placed, packed
	await scenario('interlaced file sources: planar windows de-interlaced', async (s) => {
		s.frame = 20
		const seen = []
		for (const fmt of ['yuv422p10', 'yuv422p8', 'yuv420p', 'nv12']) {
			const win = []
			const u = []
			for (let i = 0; i < 3; ++i) {
				const planes = await s.sourcePlanar(fmt, 1100 + i)
				const im = await s.rig.image(W, H)
				await s.rig.run(s.readAs[fmt](planes, im))
				win.push(planes)
				u.push(im)
			}
			const bg = await s.source(v210Frame(full, 1110))
			const ubg = await s.rig.image(W, H)
			await s.rig.run(s.read([bg], ubg))
			const fields = []
			for (const parity of [0, 1]) {
				const y = await s.rig.image(W, H)
				await s.rig.run(s.yadif(u[0], u[1], u[2], y, { parity, tff: 1, skipSpatial: 0 }))
				fields.push(y)
			}
			for (const y of fields) {
				const py = await s.rig.image(W, H)
				await s.rig.run(s.transform(y, py, await s.transform.matrix({ scaleX: 0.6, scaleY: 0.6, offsetX: 0.1 })))
				const comb = await s.rig.image(W, H)
				await s.rig.run(s.combine[2]([ubg, py], comb))
				const out = (await s.rig.planes('v210', W, H, 'writeonly'))[0]
				await s.rig.run(s.write(comb, [out], 0))
				;[y, py, comb].forEach((x) => x.release())
				seen.push(await s.consume(out))
				out.release()
			}
			;[...win.flat(), ...u, bg, ubg].forEach((x) => x.release())
		}
		return seen
	}, { fused: 12, plain: 0, launched: 12, fallbacks: 0 })

	// a packed frame made on the device and unpacked again (a channel whose consumer's frame feeds another channel's producer):
	// write -> read -> write.  The second frame's fused launch reads `mid` itself, so the job that makes `mid` must run first
	await scenario('a packed frame made on the device and read back: write -> read -> write', async (s) => {
		s.frame = 12
		const a = await s.source(v210Frame(full, 1200))
		const b = await s.source(v210Frame(full, 1201))
		const ua = await s.rig.image(W, H)
		const ub = await s.rig.image(W, H)
		await s.rig.run(s.read([a], ua))
		await s.rig.run(s.read([b], ub))
		const comb = await s.rig.image(W, H)
		await s.rig.run(s.combine[2]([ua, ub], comb))
		const mid = (await s.rig.planes('v210', W, H, 'readwrite'))[0]
		await s.rig.run(s.write(comb, [mid], 0))
		const um = await s.rig.image(W, H)
		await s.rig.run(s.read([mid], um))
		const pm = await s.rig.image(W, H)
		await s.rig.run(s.transform(um, pm, await s.transform.matrix(PIP[1])))
		const comb2 = await s.rig.image(W, H)
		await s.rig.run(s.combine[2]([ua, pm], comb2))
		const out = (await s.rig.planes('v210', W, H, 'writeonly'))[0]
		await s.rig.run(s.write(comb2, [out], 0))
		const seen = [await s.consume(out), await s.consume(mid)]
		;[a, b, ua, ub, comb, mid, um, pm, comb2, out].forEach((x) => x.release())
		return seen
	}, { fused: 2, plain: 0, launched: 2, fallbacks: 0 })

	// 720p50, the reference's third format (src/config.ts:43-54): 1280 % 48 = 32 - every line ends in a tail quad and cleared slots
	if (W === 384) {
		const w7 = 1280
		const h7 = 24
		await scenario('a 1280-wide channel: tails in every read and in the write', async (s) => {
			s.frame = 13
			const read7 = await s.rig.unpack('v210', w7, h7, '709', '709')
			const write7 = await s.rig.pack('v210', w7, h7, '709', false)
			const combine7 = await s.rig.combine(3, w7, h7)
			const transform7 = await s.rig.transform(w7, h7)
			const srcs = []
			for (let l = 0; l < 3; ++l) srcs.push(await s.source(v210Frame(v210Bytes(w7, h7), 1300 + l, l !== 1), w7, h7))
			const placed = []
			for (let l = 0; l < 3; ++l) {
				const u = await s.rig.image(w7, h7)
				await s.rig.run(read7([srcs[l]], u))
				const p = await s.rig.image(w7, h7)
				await s.rig.run(transform7(u, p, await transform7.matrix(PIP[l])))
				u.release()
				placed.push(p)
			}
			const comb = await s.rig.image(w7, h7)
			await s.rig.run(combine7(placed, comb))
			const out = (await s.rig.planes('v210', w7, h7, 'writeonly'))[0]
			await s.rig.run(write7(comb, [out], 0))
			const seen = [await s.consume(out)]
			// and the headline's shape at that width: plain reads, no placement
			const plain = []
			for (let l = 0; l < 3; ++l) { const u = await s.rig.image(w7, h7); await s.rig.run(read7([srcs[l]], u)); plain.push(u) }
			const comb2 = await s.rig.image(w7, h7)
			await s.rig.run(combine7(plain, comb2))
			const out2 = (await s.rig.planes('v210', w7, h7, 'writeonly'))[0]
			await s.rig.run(write7(comb2, [out2], 0))
			seen.push(await s.consume(out2))
			;[...srcs, ...placed, comb, out, ...plain, comb2, out2].forEach((x) => x.release())
			return seen
		}, { fused: 2, plain: 0, launched: 2, fallbacks: 0 })
	}

	process.stdout.write(JSON.stringify({ width: W, height: H, scenarios, problems }) + '\n')
}
main().catch((e) => { process.stderr.write(String(e && e.stack || e) + '\n'); process.exit(1) })
