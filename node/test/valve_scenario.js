'use strict'
// The reference's OWN video valves, driven frame by frame: Mixer.mixVidValve (producer/mixer.ts:189-236),
// Transitioner.transitionVidValve (transitioner.ts:123-201), Combiner.vidEndValve / combineVidValve (combiner.ts:184-267)
// and Black (blackSilence.ts:96-158), type-erased into <root> by oracle/refbuild/ts_erase.py (build container only), wired
// to the recording clContext (mock_context.js), to stand-ins for `redioactive` and `beamcoder`, and to the reference's own
// ImageProcess / Transform / Transition / Combine / ClJobs (ts_strip.py).  This script plays the part of layer.ts and of the
// pipes: it tells a Transitioner which sources it has (`update`, as Layer.update does) and hands every valve the frames the
// zip in front of it would have collected.  The trace - every createBuffer / hostAccess / createProgram / runProgram /
// addRef / release the valves caused, in order - is printed as JSON (tests/golden/valve_trace.json) and replayed on the real
// addon by replay.js.
// Schedule (11 output frames of a 3-layer channel):
//   layer 1: clip A full frame; frames 7..10: wipe to clip C through mask clip M (the mask is a Mixer output like any source)
//   layer 2: clip B0 picture-in-picture; frames 2..5: dissolve into clip B1 (full frame); from frame 6: B1
//   layer 3: nothing loaded (the Transitioner passes its black frame through)
// With --v210 the ends of the path are the reference's too: every source frame is a v210 frame that goes through ToRGBA
// (io.ts:61-118: createSources / loadFrame / createDest / processFrame, as ffmpegProducer.ts:501-538 calls them, one Loader
// per producer) and every output frame through FromRGBA (io.ts:141-174: createDests / processFrame / saveFrame, as
// macadamConsumer.ts:221-256 calls them) - the whole per-frame video path of a channel, v210 in, v210 out
// (tests/golden/channel_trace.json).
// usage: node valve_scenario.js <root> [--v210]
const path = require('path')
const { EventEmitter } = require('events')
const { makeMock } = require('./mock_context')

const root = path.resolve(process.argv[2])
const req = (m) => require(path.join(root, m))
const { ClProcessJobs } = req('clJobQueue.js')
const { Mixer } = req('producer/mixer.js')
const { Transitioner } = req('transitioner.js')
const { Combiner, CombineLayer } = req('combiner.js')
const redio = require('./redioactive_mock.js')
const V210 = process.argv.includes('--v210')
const { ToRGBA, FromRGBA } = V210 ? req('process/io.js') : {}
const v210 = V210 ? req('process/v210.js') : null

const W = 192
const H = 64
const FRAMES = 11
const PIP = { anchor: { x: 0.25, y: 0.75 }, rotation: 30, fill: { xOffset: 0.25, yOffset: -0.125, xScale: 0.5, yScale: 0.5 }, volume: 1 }

// a clip's k-th frame: f32 RGBA noise in [0, 1) from a 32-bit hash of (element index, seed) - tests/test_node_boundary.py restates it
function clipFrame(seed, k) {
	const n = W * H * 4
	const f = new Float32Array(n)
	const s = (Math.imul(seed, 0x9E3779B1) ^ Math.imul(k + 1, 0x85EBCA6B)) >>> 0
	for (let i = 0; i < n; ++i) {
		let h = Math.imul((i + 1) ^ s, 0x9E3779B1) >>> 0
		h = (h ^ (h >>> 15)) >>> 0
		h = Math.imul(h, 0x85EBCA6B) >>> 0
		h = (h ^ (h >>> 13)) >>> 0
		f[i] = (h >>> 8) / 16777216
	}
	return Buffer.from(f.buffer)
}
const CLIPS = { A: 11, B0: 22, B1: 33, C: 44, M: 55 }
// the same clip as v210: word i of the frame packs three legal-range codes taken from the hash of elements 3i .. 3i + 2
function clipFrameV210(seed, k) {
	const words = (Math.ceil(W / 48) * 128 * H) / 4
	const b = Buffer.alloc(words * 4)
	const s = (Math.imul(seed, 0x9E3779B1) ^ Math.imul(k + 1, 0x85EBCA6B)) >>> 0
	const code = (i) => {
		let h = Math.imul((i + 1) ^ s, 0x9E3779B1) >>> 0
		h = (h ^ (h >>> 15)) >>> 0
		h = Math.imul(h, 0x85EBCA6B) >>> 0
		h = (h ^ (h >>> 13)) >>> 0
		return 64 + (h >>> 8) % 877
	}
	for (let i = 0; i < words; ++i) b.writeUInt32LE((code(3 * i) | (code(3 * i + 1) << 10) | (code(3 * i + 2) << 20)) >>> 0, 4 * i)
	return b
}

async function main() {
	const ctx = makeMock({ recordMappedWrites: true })
	const trace = ctx.trace
	const note = (what, extra) => trace.push(Object.assign({ op: 'note', what }, extra || {}))
	const jobs = new ClProcessJobs(ctx).getJobs()
	const fmt = { name: 'test', fields: 1, width: W, height: H, squareWidth: W, squareHeight: H, timescale: 50, duration: 1, audioSampleRate: 48000, audioChannels: 2 }
	const dims = { width: W, height: H }

	// a producer's output: an f32 RGBA frame on the device, stamped like ffmpegProducer does; then its Mixer
	const counters = {}
	const loaders = {}
	async function sourceFrame(clip) {
		const k = counters[clip] = (counters[clip] === undefined ? 0 : counters[clip] + 1)
		if (V210) { // ffmpegProducer.ts:501-538 vidLoader + vidProcess
			const sourceID = `P${CLIPS[clip]} ${clip}`
			if (!loaders[clip]) {
				loaders[clip] = new ToRGBA(ctx, '709', '709', new v210.Reader(W, H), jobs)
				await loaders[clip].init()
			}
			const convert = loaders[clip]
			const ts = 1000 * CLIPS[clip] + k
			const clSources = await convert.createSources(`${sourceID} ${ts}`)
			clSources.forEach((s) => { s.timestamp = ts })
			await convert.loadFrame(clipFrameV210(CLIPS[clip], k), clSources, ctx.queue.load)
			await ctx.waitFinish(ctx.queue.load)
			const clDest = await convert.createDest(dims, `${sourceID} ${clSources[0].timestamp}`)
			clDest.timestamp = clSources[0].timestamp
			convert.processFrame(sourceID, clSources, clDest)
			note('source frame', { clip, k, buf: clDest._mockId })
			return clDest
		}
		const buf = await ctx.createBuffer(W * H * 16, 'readwrite', 'coarse', dims, `P ${clip}`)
		await buf.hostAccess('writeonly', ctx.queue.load, clipFrame(CLIPS[clip], k))
		buf.timestamp = 1000 * CLIPS[clip] + k
		note('source frame', { clip, k, buf: buf._mockId })
		return buf
	}
	const mixers = {}
	async function mixerOf(clip, params) {
		const m = new Mixer(ctx, fmt, jobs)
		const src = redio(() => redio.nil)
		await m.init(`P${CLIPS[clip]} ${clip}`, src, src, fmt)
		if (params) m.setMixParams(params)
		mixers[clip] = m
		return m
	}
	for (const [clip, params] of [['A', null], ['B0', PIP], ['B1', null], ['C', null], ['M', null]]) await mixerOf(clip, params)
	const mixed = async (clip) => {
		const f = await sourceFrame(clip)
		if (!V210) await ctx.waitFinish(ctx.queue.load)
		return mixers[clip].getVideoPipe().fn(f)
	}
	let fromRGBA = null
	if (V210) {
		fromRGBA = new FromRGBA(ctx, '709', new v210.Writer(W, H, false), jobs)
		await fromRGBA.init()
	}
	const vid = (clip) => mixers[clip].getVideoPipe()

	// three layers' Transitioners and the channel's Combiner
	const layers = []
	for (let l = 1; l <= 3; ++l) {
		const endEvent = new EventEmitter()
		const t = new Transitioner(ctx, `chan1 L${l}`, fmt, jobs, endEvent, () => {})
		await t.initialise()
		layers.push({ t, endEvent, getEndEvent: () => endEvent })
	}
	const combiner = new Combiner(ctx, 'chan1', fmt, jobs)
	await combiner.initialise()
	combiner.updateLayers(layers.map((L) => new CombineLayer(L, L.t.getAudioPipe(), L.t.getVideoPipe())))
	const black = (pipe) => pipe.root().generator()
	const combineValve = combiner.videoPipe.fn
	const endValve = combiner.videoPipe.up.fn

	layers[0].t.update('cut', 0, [], [vid('A')])
	layers[1].t.update('cut', 0, [], [vid('B0')])
	layers[2].t.update('cut', 0, [], [])
	for (let f = 0; f < FRAMES; ++f) {
		note('frame', { f })
		// layer.ts `update()` at the frames where the play-out changes
		if (f === 2) layers[1].t.update('dissolve', 4, [], [vid('B0'), vid('B1')])
		if (f === 6) layers[1].t.update('cut', 0, [], [vid('B1')])
		if (f === 7) layers[0].t.update('wipe', 4, [], [vid('A'), vid('C'), vid('M')])
		const in1 = f < 7 ? [await mixed('A')] : [await mixed('A'), await mixed('C'), await mixed('M')]
		const in2 = f < 2 ? [await mixed('B0')] : f < 6 ? [await mixed('B0'), await mixed('B1')] : [await mixed('B1')]
		const outs = []
		for (const [L, ins] of [[layers[0], in1], [layers[1], in2], [layers[2], []]]) {
			const pipe = L.t.getVideoPipe()
			outs.push(await pipe.fn([await black(pipe)].concat(ins)))
		}
		const zipped = await endValve([await black(combiner.videoPipe)].concat(outs))
		const out = await combineValve(zipped)
		note('output', { f, buf: out._mockId, timestamp: out.timestamp })
		if (V210) { // macadamConsumer.ts:221-256 vidProcess + vidSaver (progressive format); the spout lets the frame go
			const clDests = await fromRGBA.createDests(`chan1 ${out.timestamp}`)
			clDests.forEach((d) => { d.timestamp = out.timestamp })
			const ts = out.timestamp
			fromRGBA.processFrame('chan1', out, clDests, 0)
			await jobs.runQueue({ source: 'chan1', timestamp: ts })
			await fromRGBA.saveFrame(clDests[0], ctx.queue.unload)
			await ctx.waitFinish(ctx.queue.unload)
			clDests.forEach((d) => d.release())
			continue
		}
		await out.hostAccess('readonly', ctx.queue.unload)  // what a consumer's saveFrame does after FromRGBA; here the f32 frame itself
		out.release()
	}
	trace.push({ op: 'liveBuffers', ids: Array.from(ctx.live.keys()).sort((a, b) => a - b), refs: Array.from(ctx.live.keys()).sort((a, b) => a - b).map((id) => ctx.live.get(id)._refs) })
	process.stdout.write(JSON.stringify(trace))
}

main().catch((e) => { process.stderr.write(String(e && e.stack || e) + '\n'); process.exit(1) })
