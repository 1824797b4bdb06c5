'use strict'
// Random job streams through the recording context (node/defer.js) and through the plain one: the same seeded sequence of
// reads, transforms, transitions, combines, de-interlaces, frame and field writes, source and matrix overwrites, releases
// and host reads goes to both; everything a host read sees must be the same bytes, and when the owners have let go of
// everything nothing may be left alive or pending.  The streams are NOT shaped like a channel: images are reused as
// operands of several jobs, released early or late, asked for in the middle of a chain, overwritten while recorded -
// whatever order the recording has to keep, a random stream will sooner or later need it.
// usage: node defer_fuzz.js [first seed=1] [streams=20] [steps=60]; prints one JSON object { streams, problems, fusedLaunches }
const { Rig } = require('../device.js')
const { colour } = require('../index.js')

// frame size: small by default (many streams per second); PHANERON_FUZZ_SIZE=1920x270 makes the kernels long enough for an ordering
// mistake between queues, or a block recycled too early, to show
const [W, H] = (process.env.PHANERON_FUZZ_SIZE || '192x12').split('x').map((v) => parseInt(v)) // (widths: multiples of 8; one that is not a multiple of 48 - 176 - puts a tail quad into every v210 line)
const first = parseInt(process.argv[2] || '1')
const streams = parseInt(process.argv[3] || '20')
const steps = parseInt(process.argv[4] || '60')

function rng(seed) { let s = (seed * 2654435761) >>> 0; return () => (s = (Math.imul(s ^ (s >>> 15), 0x2c1b3c6d) + 0x9e3779b9) >>> 0) }
const v210Bytes = Math.ceil(W / 48) * 128 * H
function v210Frame(seed) {
	const r = rng(seed)
	const b = Buffer.alloc(v210Bytes)
	const code = () => 4 + (r() >>> 8) % 1016
	for (let i = 0; i < v210Bytes; i += 4) b.writeUInt32LE((code() | (code() << 10) | (code() << 20)) >>> 0, i)
	return b
}
const PLACEMENTS = [{}, { scaleX: 0.5, scaleY: 0.5, offsetX: 0.25, offsetY: -0.25 }, { scaleX: 2, scaleY: 2 }, { scaleX: 0.7, scaleY: 0.9, rotate: 0.11 }, { flipH: true }]

async function play(seed, deferred) {
	const rig = await Rig.open({ deviceIndex: 0, deferred, spinWaitMicros: 100 })
	const S = {
		read: await rig.unpack('v210', W, H, '709', '2020'), readAs: {}, write: await rig.pack('v210', W, H, '2020', false), writeField: await rig.pack('v210', W, H, '2020', true),
		transform: await rig.transform(W, H), dissolve: await rig.two('transition_dissolve', W, H), wipe: await rig.two('transition_wipe', W, H), yadif: await rig.yadif(W, H),
		combine: { 2: await rig.combine(2, W, H), 3: await rig.combine(3, W, H), 4: await rig.combine(4, W, H) }
	}
	const FORMATS = ['v210', 'v210', 'yuv422p10', 'yuv422p8', 'yuv420p', 'nv12', 'rgba8', 'bgra8'] // sources come in every pack format (v210 twice as often)
	for (const f of FORMATS.slice(2)) S.readAs[f] = await rig.unpack(f, W, H, '709', '2020')
	const OUT_FORMATS = ['v210', 'v210', 'v210', 'rgba8', 'bgra8', 'yuv422p8', 'yuv422p10', 'yuv420p', 'nv12'] // packed frames in every Writer's format (v210 most often)
	S.writeAs = { v210: [S.write, S.writeField] }
	for (const f of OUT_FORMATS.slice(3)) S.writeAs[f] = [await rig.pack(f, W, H, '2020', false), await rig.pack(f, W, H, '2020', true)]
	const r = rng(seed)
	const pick = (list) => list[r() % list.length]
	const seen = []
	const sources = [] // v210 on the device
	const images = [] // defined RGBA images the "application" still holds
	const outs = [] // packed frames: { buf, fields } (fields: which field writes it has had; 0 = whole frame)
	const mats = []
	for (const p of PLACEMENTS) mats.push(await S.transform.matrix(p))
	// a matrix buffer of its own that the stream overwrites now and then
	const live = await rig.ctx.createBuffer(48, 'readonly', 'none', undefined, 'live matrix')
	// (an upload is followed by waitFinish(load) before anything uses the buffer, as ffmpegProducer.ts:514-515 does)
	const up = async (buf, bytes) => { await rig.upload(buf, bytes); await rig.sync(rig.ctx.queue.load) }
	const setLive = async (i) => { const f = new Float32Array(12); f.set(colour.transformMatrix(W, H, PLACEMENTS[i % PLACEMENTS.length])); await up(live, Buffer.from(f.buffer)) }
	await setLive(1)
	const fill = (fmt, bytes, seed) => {
		if (fmt === 'v210') return v210Frame(seed)
		const q = rng(seed)
		const b = Buffer.alloc(bytes)
		if (fmt === 'yuv422p10') for (let i = 0; i + 2 <= bytes; i += 2) b.writeUInt16LE((q() >>> 8) % 1024, i)
		else for (let i = 0; i < bytes; ++i) b[i] = (q() >>> 8) & 255
		return b
	}
	const refill = async (src) => { for (const p of src.planes) await up(p, fill(src.fmt, p.length, r())) }
	const newSource = async () => { const fmt = pick(FORMATS); const src = { fmt, planes: await rig.planes(fmt, W, H) }; await refill(src); sources.push(src); return src }
	for (let i = 0; i < 4; ++i) await newSource()
	const newImage = async () => rig.image(W, H)
	const log = []
	// (a consumer maps a frame after its jobs' waitFinish: clJobQueue.ts:131, macadamConsumer.ts:233-254)
	const consume = async (b) => { await rig.sync(); await rig.download(b); seen.push(Buffer.from(b)); log.push(`#${seen.length - 1}`) }
	for (let step = 0; step < steps; ++step) {
		const op = r() % 16
		if (process.env.PHANERON_DEFER_DEBUG && deferred) process.stderr.write(`step ${step} op ${op} after: ${log[log.length - 1]}\n`)
		if (op < 3 || images.length < 2) { // read
			const im = await newImage()
			const src = pick(sources)
			await rig.run(src.fmt === 'v210' ? S.read(src.planes, im) : S.readAs[src.fmt](src.planes, im))
			images.push(im); log.push('read')
		} else if (op < 5) { // transform, through a constant matrix or the live one
			const im = await newImage()
			await rig.run(S.transform(pick(images), im, r() % 3 === 0 ? live : pick(mats)))
			images.push(im); log.push('transform')
		} else if (op === 5) { // dissolve
			const im = await newImage()
			await rig.run(S.dissolve(pick(images), pick(images), (r() % 5) / 4, im))
			images.push(im); log.push('dissolve')
		} else if (op === 6 && images.length >= 3) { // wipe
			const im = await newImage()
			await rig.run(S.wipe(pick(images), pick(images), pick(images), im))
			images.push(im); log.push('wipe')
		} else if (op < 9) { // combine
			const n = 2 + r() % 3
			const ls = []
			for (let i = 0; i < n; ++i) ls.push(pick(images))
			const im = await newImage()
			await rig.run(S.combine[n](ls, im))
			images.push(im); log.push(`combine_${n}`)
		} else if (op === 9 && images.length >= 3) { // both fields of a window, or one
			const win = [pick(images), pick(images), pick(images)]
			for (const parity of r() % 2 ? [0, 1] : [r() % 2]) {
				const im = await newImage()
				await rig.run(S.yadif(win[0], win[1], win[2], im, { parity, tff: 1, skipSpatial: 0 }))
				images.push(im)
			}
			log.push('yadif')
		} else if (op < 12) { // a packed frame, or a field into an existing one
			if (outs.length && r() % 3 === 0) {
				const o = pick(outs)
				const field = r() % 2 ? 1 : 3
				await rig.run(S.writeAs[o.fmt][1](pick(images), o.planes, field))
				log.push(`write field ${field}`)
			} else {
				const fmt = pick(OUT_FORMATS)
				const planes = await rig.planes(fmt, W, H, 'writeonly')
				for (const p of planes) await up(p, Buffer.alloc(p.length, 0x15))
				await rig.run(S.writeAs[fmt][0](pick(images), planes, 0))
				outs.push({ fmt, planes }); log.push(`write ${fmt}`)
			}
		} else if (op === 12) { // the next frame into an old source, or a new placement into the live matrix
			if (r() % 2) await refill(pick(sources)); else await setLive(r())
			log.push('overwrite')
		} else if (op === 13 && images.length > 2) { // the owner lets an image go (jobs recorded on it may still need it)
			const i = r() % images.length
			images[i].release()
			images.splice(i, 1); log.push('release')
		} else if (op === 14 && outs.length) { // a consumer maps a packed frame
			for (const p of pick(outs).planes) await consume(p)
			log.push('consume out')
		} else if (op === 15) { // somebody maps an image in the middle of everything
			await consume(pick(images)); log.push('consume image')
		}
		if (r() % 7 === 0) await rig.sync()
	}
	for (const o of outs) for (const p of o.planes) await consume(p)
	;[...images, ...sources.map((x) => x.planes).flat(), ...outs.map((o) => o.planes).flat(), live].forEach((b) => b.release())
	rig.close()
	const stats = rig.ctx.deferredStats()
	const left = rig.ctx.flushDeferred()
	const liveBuffers = rig.ctx.bufferStats().liveBuffers
	return { seen, stats, pending: left ? left.pending : 0, liveBuffers, log }
}

async function main() {
	const problems = []
	let fusedLaunches = 0
	let launchesSaved = 0
	let fallbacks = 0
	let lastFallback = null
	for (let seed = first; seed < first + streams; ++seed) {
		let a, b
		try {
			a = await play(seed, false)
			b = await play(seed, true)
		} catch (e) {
			problems.push({ seed, what: String(e && e.stack || e) })
			continue
		}
		if (a.seen.length !== b.seen.length) problems.push({ seed, what: `host reads: ${a.seen.length} plain, ${b.seen.length} deferred` })
		for (let i = 0; i < Math.min(a.seen.length, b.seen.length); ++i)
			if (Buffer.compare(a.seen[i], b.seen[i]) !== 0) { problems.push({ seed, what: `host read ${i} of ${a.seen.length} differs (${a.seen[i].length} bytes)`, log: b.log.join(' ') }); break }
		if (b.pending) problems.push({ seed, what: `${b.pending} jobs still recorded after every owner let go` })
		if (a.liveBuffers || b.liveBuffers) problems.push({ seed, what: `buffers left alive: plain ${a.liveBuffers}, deferred ${b.liveBuffers}` })
		fusedLaunches += b.stats.fused
		fallbacks += b.stats.fallbacks
		if (b.stats.lastFallback) lastFallback = b.stats.lastFallback
		launchesSaved += b.stats.recorded - b.stats.launched
	}
	process.stdout.write(JSON.stringify({ first, streams, steps, problems, fusedLaunches, launchesSaved, fallbacks, lastFallback }) + '\n')
}
main().catch((e) => { process.stderr.write(String(e && e.stack || e) + '\n'); process.exit(1) })
