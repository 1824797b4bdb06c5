'use strict'
// GPU end-to-end check of the node boundary: the JS operator layer + dispatcher (node/) on the
// real clContext (N-API addon -> libphaneron_hip.so).  Driven by tests/test_node_boundary.py:
//   node gpu_run.js <workdir>
// reads <workdir>/job.json + input .bin files, writes output .bin files and result.json.
const fs = require('fs')
const path = require('path')
const crypto = require('crypto')
const { clContext } = require('../index.js')
const { ClProcessJobs } = require('../clJobQueue.js')
const { ToRGBA, FromRGBA } = require('../process/io.js')
const v210 = require('../process/v210.js')
const { Interlace } = require('../process/packer.js')
const ImageProcess = require('../process/imageProcess.js').default
const Combine = require('../process/combine.js').default
const Yadif = require('../process/yadif.js').default
const Transform = require('../process/transform.js').default

async function main() {
	const dir = process.argv[2]
	const job = JSON.parse(fs.readFileSync(path.join(dir, 'job.json')))
	const result = {}
	const ctx = new clContext({ platformIndex: 0, deviceIndex: 0, overlapping: true })
	await ctx.initialise()
	result.platform = ctx.getPlatformInfo()
	const jobs = new ClProcessJobs(ctx).getJobs()

	// 1. the reference's implied known-answer test: ramp -> read -> write == input
	{
		const W = 1920, H = 1080
		const toRGBA = new ToRGBA(ctx, '709', '709', new v210.Reader(W, H), jobs)
		await toRGBA.init()
		const fromRGBA = new FromRGBA(ctx, '709', new v210.Writer(W, H, false), jobs)
		await fromRGBA.init()
		const srcs = await toRGBA.createSources('kat')
		const rgba = await toRGBA.createDest({ width: W, height: H }, 'kat')
		const dsts = await fromRGBA.createDests('kat')
		const yuvSrc = Buffer.allocUnsafe(toRGBA.getTotalBytes())
		v210.fillBuf(yuvSrc, W, H)
		await toRGBA.loadFrame(yuvSrc, srcs, ctx.queue.load)
		await ctx.waitFinish(ctx.queue.load)
		toRGBA.processFrame('yuvRead', srcs, rgba)
		await jobs.runQueue({ source: 'yuvRead', timestamp: 0 })
		fromRGBA.processFrame('yuvWrite', rgba, dsts, Interlace.Progressive)
		await jobs.runQueue({ source: 'yuvWrite', timestamp: 0 })
		await fromRGBA.saveFrame(dsts, ctx.queue.unload)
		result.rampCompare = yuvSrc.compare(Buffer.from(dsts[0]))
		dsts[0].release()
	}

	// 2. N-layer channel: read xN -> transform(layer 1 only, PiP) -> combine_N -> write, like
	//    producer -> Mixer -> Combiner -> consumer (SURVEY 3.3)
	{
		const { width: W, height: H, layers, readSpec, writeSpec } = job.channel
		const dims = { width: W, height: H }
		const toRGBA = new ToRGBA(ctx, readSpec, writeSpec, new v210.Reader(W, H), jobs)
		await toRGBA.init()
		const fromRGBA = new FromRGBA(ctx, writeSpec, new v210.Writer(W, H, false), jobs)
		await fromRGBA.init()
		const xf = new ImageProcess(ctx, new Transform(ctx, W, H), jobs)
		await xf.init()
		const comb = new ImageProcess(ctx, new Combine(layers.length, W, H), jobs)
		await comb.init()
		const frames = []
		for (let l = 0; l < layers.length; ++l) {
			const srcs = await toRGBA.createSources(`L${l}`)
			await toRGBA.loadFrame(fs.readFileSync(path.join(dir, layers[l])), srcs, ctx.queue.load)
			await ctx.waitFinish(ctx.queue.load)
			let rgba = await toRGBA.createDest(dims, `L${l}`)
			toRGBA.processFrame(`P${l}`, srcs, rgba)
			if (l === 1) {
				const dst = await ctx.createBuffer(W * H * 16, 'readwrite', 'coarse', dims, 'mixer')
				const src = rgba
				await xf.run(Object.assign({ input: src, output: dst }, job.channel.pip), { source: `P${l}`, timestamp: 0 }, () => src.release())
				rgba = dst
			}
			await jobs.runQueue({ source: `P${l}`, timestamp: 0 })
			frames.push(rgba)
		}
		const out = await ctx.createBuffer(W * H * 16, 'readwrite', 'coarse', dims, 'chan')
		await comb.run({ inputs: frames, output: out }, { source: 'chan combine', timestamp: 0 }, () => frames.forEach((f) => f.release()))
		await jobs.runQueue({ source: 'chan combine', timestamp: 0 })
		const dsts = await fromRGBA.createDests('chan')
		fromRGBA.processFrame('chan out', out, dsts, Interlace.Progressive)
		await jobs.runQueue({ source: 'chan out', timestamp: 0 })
		await fromRGBA.saveFrame(dsts, ctx.queue.unload)
		fs.writeFileSync(path.join(dir, 'channel_out.bin'), dsts[0])
		dsts[0].release()
	}

	// 3. yadif send_field through the Yadif wrapper over already converted RGBA fields
	{
		const { width: W, height: H, frames: files, tff } = job.yadif
		const dims = { width: W, height: H }
		const yadif = new Yadif(ctx, jobs, W, H, { mode: 'send_field', tff }, true)
		await yadif.init()
		const outs = []
		for (let f = 0; f < files.length; ++f) {
			const b = await ctx.createBuffer(W * H * 16, 'readwrite', 'coarse', dims, `field${f}`)
			await b.hostAccess('writeonly', ctx.queue.load, fs.readFileSync(path.join(dir, files[f])))
			await ctx.waitFinish(ctx.queue.load)
			b.timestamp = 2 * f
			// the producer queues its loader job under this key; give runQueue something to run
			jobs.add({ source: 'P9', timestamp: b.timestamp }, 'noop', null, {}, () => {})
			jobs.get({ source: 'P9', timestamp: b.timestamp }).length = 0
			await yadif.processFrame(b, outs, 'P9')
		}
		result.yadifTimestamps = outs.map((o) => o.timestamp)
		for (let i = 0; i < outs.length; ++i) {
			await outs[i].hostAccess('readonly', ctx.queue.unload)
			fs.writeFileSync(path.join(dir, `yadif_out${i}.bin`), outs[i])
		}
		yadif.release()
	}

	// 4. the reference's other round-trip scripts (src/process/test/<fmt>Test.ts): pattern -> read -> write
	//    (progressive) -> compare with the input; RGBA and output hashes are checked against the
	//    reference kernels' own results (tests/golden/kat.json)
	result.formats = {}
	for (const { fmt, width: W, height: H, spec } of job.formats || []) {
		const mod = require(`../process/${fmt}.js`)
		const toRGBA = new ToRGBA(ctx, spec, spec, new mod.Reader(W, H), jobs)
		await toRGBA.init()
		const fromRGBA = new FromRGBA(ctx, spec, new mod.Writer(W, H, false), jobs)
		await fromRGBA.init()
		const srcs = await toRGBA.createSources(fmt)
		const rgba = await toRGBA.createDest({ width: W, height: H }, fmt)
		const dsts = await fromRGBA.createDests(fmt)
		const whole = Buffer.alloc(toRGBA.getTotalBytes())
		mod.fillBuf(whole, W, H)
		const planes = []
		let off = 0
		for (const n of toRGBA.getNumBytes()) { planes.push(whole.slice(off, off + n)); off += n }
		await toRGBA.loadFrame(planes, srcs, ctx.queue.load)
		await ctx.waitFinish(ctx.queue.load)
		rgba.addRef() // keep it for the read-back below; the writer's callback releases one reference
		toRGBA.processFrame(`${fmt} rd`, srcs, rgba)
		await jobs.runQueue({ source: `${fmt} rd`, timestamp: 0 })
		fromRGBA.processFrame(`${fmt} wr`, rgba, dsts, Interlace.Progressive)
		await jobs.runQueue({ source: `${fmt} wr`, timestamp: 0 })
		await fromRGBA.saveFrame(dsts, ctx.queue.unload)
		await rgba.hostAccess('readonly', ctx.queue.unload)
		const back = Buffer.concat(dsts.map((d) => Buffer.from(d)))
		result.formats[fmt] = {
			compare: whole.compare(back),
			rgbaSha256: crypto.createHash('sha256').update(rgba).digest('hex'),
			backSha256: crypto.createHash('sha256').update(back).digest('hex')
		}
		rgba.release()
		dsts.forEach((d) => d.release())
	}

	// 5. staged ring (node/staging.js) around the fused channel program: frames with different content
	//    through 3 slots, outputs written per frame; also the same frames through the dispatcher
	if (job.staged) {
		const { width: W, height: H, layers: N, frames: F } = job.staged
		const { StagedChannel } = require('../staging.js')
		const { FusedV210Channel } = require('../process/fusedChannel.js')
		const fused = new FusedV210Channel(ctx, job.staged.readSpec, job.staged.writeSpec, N, W, H, jobs)
		await fused.init()
		const vb = fused.getNumBytes()
		const chan = new StagedChannel(ctx, Array(N).fill(vb), vb, (sources, output) => fused.launch(sources, output), 3, 'staged')
		await chan.init()
		const order = []
		const consume = async (f, out) => { order.push(f); fs.writeFileSync(path.join(dir, `staged_out${f}.bin`), out) }
		const fill = async (f, sources) => {
			for (let l = 0; l < N; ++l) fs.readFileSync(path.join(dir, `staged_f${f}_l${l}.bin`)).copy(sources[l])
		}
		for (let f = 0; f < F; ++f) await chan.submit(fill, consume)
		await chan.drain(consume)
		result.stagedOrder = order
		await chan.close()
		// the same program through the job queue (one frame)
		const srcs = []
		for (let l = 0; l < N; ++l) {
			const s = await fused.createSource(`L${l}`)
			await s.hostAccess('writeonly', ctx.queue.load, fs.readFileSync(path.join(dir, `staged_f0_l${l}.bin`)))
			srcs.push(s)
		}
		await ctx.waitFinish(ctx.queue.load)
		const dst = await fused.createDest('q')
		let fired = false
		fused.processFrame({ source: 'fused chan', timestamp: 0 }, srcs, dst, () => { fired = true })
		await jobs.runQueue({ source: 'fused chan', timestamp: 0 })
		await dst.hostAccess('readonly', ctx.queue.unload)
		fs.writeFileSync(path.join(dir, 'fused_queue_out.bin'), dst)
		result.fusedCallbackFired = fired
		srcs.forEach((s) => s.release())
		dst.release()
	}

	result.buffers = ctx.logBuffers()
	fs.writeFileSync(path.join(dir, 'result.json'), JSON.stringify(result))
}

main().catch((e) => { console.error(e && e.stack || e); process.exit(1) })
