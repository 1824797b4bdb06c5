'use strict'
// Throughput of the path THROUGH the node layer (JS operators + dispatcher + N-API), frames resident
// on the device: (a) the reference-shaped batch per frame - ToRGBA x N, Combine, FromRGBA, each its own
// job key and runQueue like producer / combiner / consumer do; (b) the same as one job key (one
// waitFinish per frame); (c) the fused channel program.  usage: node bench_node.js [frames] [w h layers]
const { clContext } = require('../index.js')
const { ClProcessJobs } = require('../clJobQueue.js')
const { ToRGBA, FromRGBA } = require('../process/io.js')
const v210 = require('../process/v210.js')
const { Interlace } = require('../process/packer.js')
const ImageProcess = require('../process/imageProcess.js').default
const Combine = require('../process/combine.js').default
const { FusedV210Channel } = require('../process/fusedChannel.js')

async function main() {
	const frames = parseInt(process.argv[2] || '200', 10)
	const W = parseInt(process.argv[3] || '3840', 10), H = parseInt(process.argv[4] || '2160', 10), N = parseInt(process.argv[5] || '4', 10)
	const spin = parseInt(process.env.PH_SPIN_WAIT_US || '0', 10)
	const ctx = new clContext({ platformIndex: 0, deviceIndex: 0, overlapping: true, spinWaitMicros: spin })
	await ctx.initialise()
	const jobs = new ClProcessJobs(ctx).getJobs()
	const dims = { width: W, height: H }
	const toRGBA = new ToRGBA(ctx, '709', '2020', new v210.Reader(W, H), jobs)
	await toRGBA.init()
	const fromRGBA = new FromRGBA(ctx, '2020', new v210.Writer(W, H, false), jobs)
	await fromRGBA.init()
	const comb = new ImageProcess(ctx, new Combine(N, W, H), jobs)
	await comb.init()
	const fused = new FusedV210Channel(ctx, '709', '2020', N, W, H, jobs)
	await fused.init()

	const frame = Buffer.alloc(toRGBA.getTotalBytes())
	v210.fillBuf(frame, W, H)
	const srcs = []
	for (let l = 0; l < N; ++l) {
		const s = (await toRGBA.createSources(`L${l}`))[0]
		await s.hostAccess('writeonly', ctx.queue.load, frame)
		srcs.push(s)
	}
	await ctx.waitFinish(ctx.queue.load)
	const rgba = []
	for (let l = 0; l < N; ++l) rgba.push(await toRGBA.createDest(dims, `L${l}`))
	const mixed = await ctx.createBuffer(W * H * 16, 'readwrite', 'coarse', dims, 'chan')
	const dsts = await fromRGBA.createDests('chan')
	const keep = (b) => b.addRef() // the operators release their inputs per job; the bench reuses them

	const shaped = async (f, separateKeys) => {
		for (let l = 0; l < N; ++l) {
			srcs[l].timestamp = f
			keep(srcs[l])
			toRGBA.processFrame(separateKeys ? `P${l}` : 'chan', [srcs[l]], rgba[l])
			if (separateKeys) await jobs.runQueue({ source: `P${l}`, timestamp: f })
		}
		const cid = { source: separateKeys ? 'chan combine' : 'chan', timestamp: f }
		await comb.run({ inputs: rgba, output: mixed }, cid, () => {})
		if (separateKeys) await jobs.runQueue(cid)
		mixed.timestamp = f
		keep(mixed)
		fromRGBA.processFrame(separateKeys ? 'chan out' : 'chan', mixed, dsts, Interlace.Progressive)
		await jobs.runQueue({ source: separateKeys ? 'chan out' : 'chan', timestamp: f })
	}
	const one = async (f) => {
		fused.processFrame({ source: 'fused', timestamp: f }, srcs, dsts[0])
		await jobs.runQueue({ source: 'fused', timestamp: f })
	}
	const time = async (name, fn) => {
		for (let f = 0; f < 10; ++f) await fn(f)
		const t0 = process.hrtime.bigint()
		for (let f = 0; f < frames; ++f) await fn(1000 + f)
		const el = Number(process.hrtime.bigint() - t0) / 1e9
		console.log(JSON.stringify({ path: name, spinWaitMicros: spin, frames, size: `${W}x${H}`, layers: N, frames_per_sec: +(frames / el).toFixed(1), ms_per_frame: +(1000 * el / frames).toFixed(3) }))
	}
	await time(`node: reference-shaped, ${N + 2} job keys per frame (read x${N}, combine, write)`, (f) => shaped(f, true))
	await time(`node: reference-shaped, one job key per frame (${N + 2} kernels, one waitFinish)`, (f) => shaped(f, false))
	await time('node: fused channel program (1 kernel per frame)', one)
}

main().catch((e) => { console.error(e && e.stack || e); process.exit(1) })
