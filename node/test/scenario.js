'use strict'
// The host-logic scenario: drives the REFERENCE's operator layer + dispatcher (src/process/*, src/clJobQueue.ts,
// type-stripped into <root> by oracle/refbuild/ts_strip.py - build container only) against the recording mock
// and prints the trace as JSON.  usage: node scenario.js <root> [--resolve]
//   --resolve: every createProgram also records what the addon's device-free resolver makes of the kernel text
//              the reference passed (packer.ts:98, imageProcess.ts:69)
const path = require('path')
const { makeMock } = require('./mock_context')

const root = path.resolve(process.argv[2] || path.join(__dirname, '..'))
const req = (m) => require(path.join(root, m))
const { ClProcessJobs } = req('clJobQueue.js')
const { ToRGBA, FromRGBA } = req('process/io.js')
const v210 = req('process/v210.js')
const { Interlace } = req('process/packer.js')
const ImageProcess = req('process/imageProcess.js').default
const Yadif = req('process/yadif.js').default
const Transform = req('process/transform.js').default
const Resize = req('process/resize.js').default
const Combine = req('process/combine.js').default
const Transition = req('process/transition.js').default
const Mix = req('process/mix.js').default
const Wipe = req('process/wipe.js').default

const note = (trace, what) => trace.push({ op: 'note', what })
async function expectThrow(trace, label, fn) {
	try {
		await fn()
		trace.push({ op: 'threw', label, threw: false })
	} catch (e) {
		trace.push({ op: 'threw', label, threw: true, isError: e instanceof Error, message: e instanceof Error ? e.message : String(e) })
	}
}

async function main() {
	const resolve = process.argv.includes('--resolve') ? require('../index.js').resolveProgram : undefined
	const ctx = makeMock({ resolve })
	const trace = ctx.trace
	const processJobs = new ClProcessJobs(ctx)
	const jobs = processJobs.getJobs()
	const W = 1920
	const H = 1080
	const dims = { width: W, height: H }

	note(trace, 'v210 round trip (the reference test scripts: src/process/test/*.ts)')
	const toRGBA = new ToRGBA(ctx, '709', '709', new v210.Reader(W, H), jobs)
	await toRGBA.init()
	const fromRGBA = new FromRGBA(ctx, '709', new v210.Writer(W, H, false), jobs)
	await fromRGBA.init()
	const srcs = await toRGBA.createSources('src0')
	const rgba = await toRGBA.createDest(dims, 'src0')
	const dsts = await fromRGBA.createDests('out0')
	const frame = Buffer.alloc(toRGBA.getTotalBytes())
	v210.fillBuf(frame, W, H)
	await toRGBA.loadFrame(frame, srcs, ctx.queue.load)
	await ctx.waitFinish(ctx.queue.load)
	toRGBA.processFrame('yuvRead', srcs, rgba)
	await jobs.runQueue({ source: 'yuvRead', timestamp: 0 })
	rgba.addRef() // FromRGBA releases its source when the job has run (io.ts:152-164); this scenario uses the frame again below
	fromRGBA.processFrame('yuvWrite', rgba, dsts, Interlace.Progressive)
	await jobs.runQueue({ source: 'yuvWrite', timestamp: 0 })
	await fromRGBA.saveFrame(dsts, ctx.queue.unload)
	await expectThrow(trace, 'loadFrame plane mismatch', () => toRGBA.loadFrame([frame, frame], srcs, 0))
	await expectThrow(trace, 'runQueue unknown key', () => jobs.runQueue({ source: 'nobody', timestamp: 5 }))

	note(trace, 'interlaced writer, both fields (macadamConsumer.ts:224-244)')
	const fromI = new FromRGBA(ctx, '2020', new v210.Writer(W, H, true), jobs)
	await fromI.init()
	const idsts = await fromI.createDests('outI')
	const rgbaI = await ctx.createBuffer(W * H * 16, 'readwrite', 'coarse', dims, 'combined')
	rgbaI.timestamp = 7
	rgbaI.addRef() // one reference per field job, each dropped when its job has run ...
	rgbaI.addRef() // ... and the creator's stays: the frame is reused below
	fromI.processFrame('chan1 decklink', rgbaI, idsts, Interlace.TopField)
	fromI.processFrame('chan1 decklink', rgbaI, idsts, Interlace.BottomField)
	await jobs.runQueue({ source: 'chan1 decklink', timestamp: 7 })

	note(trace, 'yadif send_field over 4 interlaced frames (yadif.ts:115-145)')
	const yadif = new Yadif(ctx, jobs, W, H, { mode: 'send_field', tff: true }, true)
	await yadif.init()
	const toI = new ToRGBA(ctx, '709', '2020', new v210.Reader(W, H), jobs)
	await toI.init()
	const yadifOut = []
	for (let f = 0; f < 4; ++f) {
		const s = await toI.createSources('P1 L1')
		s[0].timestamp = 2 * f
		const d = await toI.createDest(dims, 'P1 L1')
		d.timestamp = 2 * f
		toI.processFrame('P1 L1', s, d)
		const outs = []
		await yadif.processFrame(d, outs, 'P1 L1')
		outs.forEach((o) => yadifOut.push({ buf: o._mockId, ts: o.timestamp }))
	}
	trace.push({ op: 'yadifOutputs', outs: yadifOut })
	yadif.release()
	const yadifP = new Yadif(ctx, jobs, W, H, { mode: 'send_frame_nospatial', tff: false }, false)
	await yadifP.init()
	const pass = []
	await yadifP.processFrame(rgbaI, pass, 'P2')
	trace.push({ op: 'yadifProgressive', same: pass[0] === rgbaI })

	note(trace, 'transform: default, PiP, repeat (no re-upload), rotate (mixer.ts:209-223)')
	const xf = new ImageProcess(ctx, new Transform(ctx, 3840, 2160), jobs)
	await xf.init()
	const xfOut = await ctx.createBuffer(3840 * 2160 * 16, 'readwrite', 'coarse', { width: 3840, height: 2160 }, 'mixer')
	const sets = [
		{ flipH: false, flipV: false, anchorX: 0, anchorY: 0, scaleX: 1, scaleY: 1, rotate: -0, offsetX: -0, offsetY: -0 },
		{ flipH: false, flipV: false, anchorX: -0.25, anchorY: 0.25, scaleX: 0.5, scaleY: 0.5, rotate: -45 / 360.0, offsetX: -0.25, offsetY: 0.125 },
		{ flipH: false, flipV: false, anchorX: -0.25, anchorY: 0.25, scaleX: 0.5, scaleY: 0.5, rotate: -45 / 360.0, offsetX: -0.25, offsetY: 0.125 },
		{ flipH: true, flipV: true, anchorX: 0, anchorY: 0, scaleX: 2, scaleY: 0.75, rotate: 0.3, offsetX: 0.1, offsetY: -0.6 }
	]
	let ts = 100
	for (const p of sets) {
		rgba.addRef()
		await xf.run(Object.assign({ input: rgba, output: xfOut }, p), { source: 'P3 L2', timestamp: ts }, () => rgba.release())
		await jobs.runQueue({ source: 'P3 L2', timestamp: ts++ })
	}
	xf.finish()

	note(trace, 'resize incl. parameter validation (resize.ts:104-132)')
	const rs = new ImageProcess(ctx, new Resize(ctx, W, H), jobs)
	await rs.init()
	for (const p of [{ flipH: false, flipV: false, scale: 1.0, offsetX: 0, offsetY: 0 }, { flipH: true, flipV: false, scale: 0.5, offsetX: 0.25, offsetY: -1.0 }, { flipH: true, flipV: true }]) {
		await rs.run(Object.assign({ input: rgba, output: rgbaI }, p), { source: 'rs', timestamp: ts }, () => {})
		await jobs.runQueue({ source: 'rs', timestamp: ts++ })
	}
	await expectThrow(trace, 'resize scale <= 0', () => rs.run({ input: rgba, output: rgbaI, flipH: true, flipV: true, scale: -1 }, { source: 'rs', timestamp: ts }, () => {}))
	await expectThrow(trace, 'resize offsetX range', () => rs.run({ input: rgba, output: rgbaI, flipH: true, flipV: true, scale: 1, offsetX: 1.5 }, { source: 'rs', timestamp: ts }, () => {}))

	note(trace, 'combine / transition / mix / wipe (combiner.ts:219-254, transitioner.ts:165-176)')
	const layers = []
	for (let i = 0; i < 4; ++i) layers.push(await ctx.createBuffer(W * H * 16, 'readwrite', 'coarse', dims, `layer${i}`))
	for (const n of [1, 2, 4]) {
		const comb = new ImageProcess(ctx, new Combine(n, W, H), jobs)
		await comb.init()
		if (n === 1) {
			await expectThrow(trace, 'combine with one input', () => comb.run({ inputs: layers.slice(0, 1), output: rgbaI }, { source: 'chan combine', timestamp: ts }, () => {}))
			continue
		}
		await comb.run({ inputs: layers.slice(0, n), output: rgbaI }, { source: 'chan combine', timestamp: ts }, () => {})
		await jobs.runQueue({ source: 'chan combine', timestamp: ts++ })
	}
	const dis = new ImageProcess(ctx, new Transition('dissolve', W, H), jobs)
	await dis.init()
	for (const cur of [0, 7, 24]) {
		await dis.run({ inputs: layers.slice(0, 2), output: rgbaI, mix: 1.0 - cur / 24 }, { source: 'L1 transition', timestamp: ts }, () => {})
		await jobs.runQueue({ source: 'L1 transition', timestamp: ts++ })
	}
	const wp = new ImageProcess(ctx, new Transition('wipe', W, H), jobs)
	await wp.init()
	await wp.run({ inputs: layers.slice(0, 2), output: rgbaI, mask: layers[2] }, { source: 'L1 transition', timestamp: ts }, () => {})
	await jobs.runQueue({ source: 'L1 transition', timestamp: ts++ })
	await expectThrow(trace, 'wipe without mask', () => wp.run({ inputs: layers.slice(0, 2), output: rgbaI }, { source: 'L1 transition', timestamp: ts }, () => {}))
	await expectThrow(trace, 'transition with 3 inputs', () => dis.run({ inputs: layers.slice(0, 3), output: rgbaI, mix: 0.5 }, { source: 'L1 transition', timestamp: ts }, () => {}))
	await expectThrow(trace, 'transition bad type', async () => new Transition('fade', W, H))
	const mix = new ImageProcess(ctx, new Mix(W, H), jobs)
	await mix.init()
	await mix.run({ input0: layers[0], input1: layers[1], mix: 0.62, output: rgbaI }, { source: 'sw', timestamp: ts }, () => {})
	const wipe = new ImageProcess(ctx, new Wipe(W, H), jobs)
	await wipe.init()
	await wipe.run({ input0: layers[0], input1: layers[1], wipe: 0.37, output: rgbaI }, { source: 'sw', timestamp: ts }, () => {})
	await jobs.runQueue({ source: 'sw', timestamp: ts++ }) // one batch, two kernels, one waitFinish

	note(trace, 'the other pack formats: load -> read -> write (both fields) -> save, plane-count errors')
	const fmtColours = { yuv422p10: ['709', '709'], yuv422p8: ['601-625', '709'], yuv420p: ['709', '2020'], nv12: ['709', '709'], rgba8: ['sRGB', '709'], bgra8: ['sRGB', 'sRGB'] }
	for (const fmt of Object.keys(fmtColours)) {
		const mod = req(`process/${fmt}.js`)
		const [inSpec, outSpec] = fmtColours[fmt]
		const rd = new mod.Reader(W, H)
		const to = new ToRGBA(ctx, inSpec, outSpec, rd, jobs)
		await to.init()
		const wrI = new mod.Writer(W, H, true)
		const from = new FromRGBA(ctx, outSpec, wrI, jobs)
		await from.init()
		trace.push({ op: 'formatGeometry', fmt, name: rd.getName(), numBytes: rd.getNumBytes(), total: rd.getTotalBytes(), rgba: rd.getNumBytesRGBA(), isRGB: rd.getIsRGB(), readWipg: rd.getWorkItemsPerGroup(), readGwi: rd.getGlobalWorkItems(), writeWipg: wrI.getWorkItemsPerGroup(), writeGwi: wrI.getGlobalWorkItems(), progGwi: new mod.Writer(W, H, false).getGlobalWorkItems() })
		const fs = await to.createSources(`${fmt} src`)
		const fd = await to.createDest(dims, `${fmt} src`)
		const fo = await from.createDests(`${fmt} out`)
		const whole = Buffer.alloc(to.getTotalBytes())
		mod.fillBuf(whole, W, H)
		const planes = []
		let off = 0
		for (const n of to.getNumBytes()) { planes.push(whole.slice(off, off + n)); off += n }
		await to.loadFrame(planes.length === 1 ? whole : planes, fs, ctx.queue.load)
		await ctx.waitFinish(ctx.queue.load)
		to.processFrame(`${fmt} rd`, fs, fd)
		await jobs.runQueue({ source: `${fmt} rd`, timestamp: 0 })
		fd.addRef() // one reference per field job (macadamConsumer.ts:224-244 does the same for its two fields)
		from.processFrame(`${fmt} wr`, fd, fo, Interlace.TopField)
		from.processFrame(`${fmt} wr`, fd, fo, Interlace.BottomField)
		await jobs.runQueue({ source: `${fmt} wr`, timestamp: 0 })
		await from.saveFrame(fo, ctx.queue.unload)
		await expectThrow(trace, `${fmt} reader plane count`, async () => rd.getKernelParams({ sources: fs.concat(fs), dest: fd }))
		await expectThrow(trace, `${fmt} writer plane count`, async () => wrI.getKernelParams({ source: fd, dests: [] }))
	}

	note(trace, 'dispatcher: FIFO across keys, late arrivals, clearQueue (clJobQueue.ts:53-141)')
	const order = []
	const mk = (src, t, tag) => mix.run({ input0: layers[0], input1: layers[1], mix: t / 10, output: rgbaI }, { source: src, timestamp: t }, () => order.push(tag))
	await mk('A', 1, 'A1a'); await mk('A', 1, 'A1b'); await mk('B', 2, 'B2'); await mk('A', 3, 'A3'); await mk('C gone', 4, 'C4')
	const p1 = jobs.runQueue({ source: 'B', timestamp: 2 })
	const p2 = jobs.runQueue({ source: 'A', timestamp: 1 })
	const p3 = p1.then(() => jobs.runQueue({ source: 'A', timestamp: 3 })) // requested while draining
	await Promise.all([p1, p2, p3])
	jobs.clearQueue('C gone')
	const left = jobs.get({ source: 'C gone', timestamp: 4 })
	trace.push({ op: 'callbackOrder', order, pendingAfterClear: left ? left.length : 0 })

	trace.push({ op: 'liveBuffers', ids: Array.from(ctx.live.keys()).sort((a, b) => a - b), refs: Array.from(ctx.live.values()).map((b) => b._refs) })
	process.stdout.write(JSON.stringify(trace))
}

main().catch((e) => { process.stderr.write(String(e && e.stack || e) + '\n'); process.exit(1) })
