'use strict'
// Host-logic scenario for the re-hosted valve graph (node/valves) on the recording mock context:
// three layers (default-fill source; PiP source that dissolves to a second source; an empty layer),
// Mixer -> Transitioner -> Combiner, 7 frames.  Prints what the graph asked the device to do and the
// buffers left alive.  usage: node valves_scenario.js
const path = require('path')
const { makeMock } = require('./mock_context')
const root = path.join(__dirname, '..')
const { ClProcessJobs } = require(path.join(root, 'clJobQueue.js'))
const { redio, isValue, end, Mixer, Transitioner, Combiner, CombineLayer } = require(path.join(root, 'valves'))

async function main() {
	const ctx = makeMock()
	const jobs = new ClProcessJobs(ctx).getJobs()
	const fmt = { width: 64, height: 36 }
	const NF = 7
	const made = []
	const source = (name, n, ts0) => {
		let i = 0
		return redio(async () => {
			if (i >= n) return end
			const b = await ctx.createBuffer(fmt.width * fmt.height * 16, 'readwrite', 'coarse', fmt, `${name} ${i}`)
			b.timestamp = ts0 + i++
			made.push(b._mockId)
			return b
		})
	}
	const layerEvents = []
	const mkLayer = async (id, pipes, params) => {
		const mixers = []
		for (const [k, p] of pipes.entries()) {
			const m = new Mixer(ctx, fmt, jobs)
			if (params && params[k]) m.setMixParams(params[k])
			await m.init(`${id} src${k}`, p)
			mixers.push(m)
		}
		const t = new Transitioner(ctx, id, fmt, jobs, (ts) => layerEvents.push({ layer: id, ts }))
		await t.initialise()
		return { mixers, t }
	}
	const pip = { anchor: { x: 0.25, y: 0.75 }, rotation: 90, fill: { xOffset: 0.25, yOffset: -0.125, xScale: 0.5, yScale: 0.5 }, volume: 1 }
	const A = await mkLayer('L1', [source('A', NF, 100)])
	const B = await mkLayer('L2', [source('B0', NF, 200), source('B1', 5, 300)], [pip, null])
	const C = await mkLayer('L3', [])
	A.t.update('cut', 0, [A.mixers[0].getMixVideo()])
	B.t.update('cut', 0, [B.mixers[0].getMixVideo()])
	C.t.update('cut', 0, [])

	const comb = new Combiner(ctx, 'chan1', fmt, jobs)
	await comb.initialise()
	comb.updateLayers([A, B, C].map((l) => new CombineLayer(l.t.getVideoPipe())))

	const outputs = []
	const out = comb.getVideoPipe()
	for (let f = 0; f < NF; ++f) {
		if (f === 2) B.t.update('dissolve', 4, [B.mixers[0].getMixVideo(), B.mixers[1].getMixVideo()])
		if (f === 6) B.t.update('cut', 0, [B.mixers[1].getMixVideo()])
		const frame = await out.next()
		if (!isValue(frame)) { outputs.push({ frame: f, ended: true }); break }
		outputs.push({ frame: f, buf: frame._mockId, ts: frame.timestamp, refs: frame._refs })
		frame.release()
	}
	const kernels = ctx.trace.filter((e) => e.op === 'runProgram').map((e) => {
		const o = { name: e.name, ts: e.params.output ? e.params.output.ts : null }
		for (const k of ['mix', 'offsetX', 'offsetY']) if (k in e.params) o[k] = e.params[k]
		if (e.data && e.data.transformMatrix) o.matrix = e.data.transformMatrix
		o.inputs = Object.keys(e.params).filter((k) => /^(input\d?|l\dIn|maskIn)$/.test(k)).map((k) => e.params[k].buf)
		return o
	})
	const liveOwners = ctx.trace.filter((e) => e.op === 'createBuffer' && ctx.live.has(e.buf)).map((e) => ({ buf: e.buf, owner: e.owner, refs: ctx.live.get(e.buf)._refs }))
	const leakedFrames = made.filter((id) => ctx.live.has(id))
	const second = await endAndWipe()
	process.stdout.write(JSON.stringify({ outputs, kernels, layerEvents, liveOwners, leakedFrames, sourceFrames: made, second }))
}

// Second graph: a source that ENDS after three frames (its layer falls back to the transitioner's black
// frame, transitioner.ts:186-192), a wipe transition with a mask source (:168-170), and a combiner with
// one route fork (every output carries one reference per fork, combiner.ts:255).
async function endAndWipe() {
	const ctx = makeMock()
	const jobs = new ClProcessJobs(ctx).getJobs()
	const fmt = { width: 32, height: 18 }
	const made = []
	const source = (name, n, ts0) => {
		let i = 0
		return redio(async () => {
			if (i >= n) return end
			const b = await ctx.createBuffer(fmt.width * fmt.height * 16, 'readwrite', 'coarse', fmt, `${name} ${i}`)
			b.timestamp = ts0 + i++
			made.push(b._mockId)
			return b
		})
	}
	const mix = async (id, pipe) => {
		const m = new Mixer(ctx, fmt, jobs)
		await m.init(id, pipe)
		return m.getMixVideo()
	}
	const short = new Transitioner(ctx, 'S', fmt, jobs)
	await short.initialise()
	short.update('cut', 0, [await mix('S src', source('S', 3, 10))])
	const wiper = new Transitioner(ctx, 'W', fmt, jobs)
	await wiper.initialise()
	wiper.update('wipe', 6, [await mix('W a', source('Wa', 6, 20)), await mix('W b', source('Wb', 6, 30)), await mix('W mask', source('Wm', 6, 40))])
	const comb = new Combiner(ctx, 'chan2', fmt, jobs)
	await comb.initialise()
	comb.updateLayers([new CombineLayer(short.getVideoPipe()), new CombineLayer(wiper.getVideoPipe())])
	const tap = comb.getSourcePipes() // one fork = the only reader here
	const outputs = []
	for (let f = 0; f < 6; ++f) {
		const frame = await tap.video.next()
		if (!isValue(frame)) { outputs.push({ frame: f, ended: true }); break }
		outputs.push({ frame: f, ts: frame.timestamp, refs: frame._refs })
		frame.release()
	}
	tap.release()
	const kernels = ctx.trace.filter((e) => e.op === 'runProgram').map((e) => ({ name: e.name, inputs: Object.keys(e.params).filter((k) => /^(input\d?|l\dIn|maskIn)$/.test(k)).sort() }))
	const blackId = ctx.trace.find((e) => e.op === 'createBuffer' && e.owner === 'black-S transition').buf
	const combineFirstInputs = ctx.trace.filter((e) => e.op === 'runProgram' && e.name === 'combine_2').map((e) => e.params.l0In.buf)
	return { outputs, kernels, blackId, combineFirstInputs, leakedFrames: made.filter((id) => ctx.live.has(id)), forksAfterRelease: comb.numForks }
}

main().catch((e) => { process.stderr.write(String(e && e.stack || e) + '\n'); process.exit(1) })
