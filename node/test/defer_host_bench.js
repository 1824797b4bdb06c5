'use strict'
// The host side of the recording context WITHOUT a device: bench_node.js's 'deferred' job stream (the valves' posting pattern - a
// fresh destination image per job, released in the job's callback, five flushes per frame, the packed frame asked for on the
// device) against a stand-in for the addon that does nothing but count.  What is left is this repository's JavaScript per frame:
// JobBoard, clContext, defer.js.  Prints one JSON line: us_per_frame and the addon calls per frame by name.
// usage: node defer_host_bench.js [frames=20000] [width=3840] [height=2160] [layers=4] [--plain]
const real = require('../phaneron_napi.node')
const { Rig } = require('../device.js')

function standIn(calls) {
	let next = 1
	const count = (k) => { calls[k] = (calls[k] || 0) + 1 }
	const refs = new Map()
	const small = Buffer.alloc(1 << 20) // every big buffer's "mirror": nobody reads pixels here
	const fake = Object.assign({}, real, {
		createContext: () => ({ ctx: true }),
		contextInfo: () => ({ vendor: 'stand-in', device: 'none' }),
		createBuffer: (_ctx, bytes) => { count('createBuffer'); const handle = { id: next++ }; refs.set(handle, 1); return { buffer: bytes > small.length ? Buffer.from(small.buffer, 0, 64) : Buffer.alloc(bytes), handle } },
		bufAddRef: (h) => { count('bufAddRef'); refs.set(h, refs.get(h) + 1) },
		bufRelease: (h) => { count('bufRelease'); const n = refs.get(h) - 1; if (n) refs.set(h, n); else refs.delete(h) },
		bufRefCount: (h) => { count('bufRefCount'); return refs.get(h) || 0 },
		hostAccess: async () => { count('hostAccess') },
		downloadAsync: () => { count('downloadAsync') },
		waitFinish: async () => { count('waitFinish') },
		waitFinishSpin: () => { count('waitFinishSpin'); return true },
		createProgram: (_ctx, _src, name) => { count('createProgram'); return { name } },
		runProgram: (_ctx, _prog, _names, _values, _queue, _timed, checkOnly) => { count(checkOnly ? 'checkProgram' : 'runProgram'); return { dataToKernel: 0, kernelExec: 0, totalTime: 0 } },
		queueWaitQueue: () => { count('queueWaitQueue') },
		runPrograms: () => { count('runPrograms') },
		runProgramsProgress: () => 0,
		bufferStats: () => ({ liveBuffers: refs.size, liveBytes: 0, pooledBytes: 0 })
	})
	return fake
}

async function main() {
	const args = process.argv.slice(2).filter((a) => !a.startsWith('--'))
	const frames = parseInt(args[0] || '20000')
	const w = parseInt(args[1] || '3840')
	const h = parseInt(args[2] || '2160')
	const n = parseInt(args[3] || '4')
	const deferred = !process.argv.includes('--plain')
	const calls = {}
	const rig = await Rig.open({ deviceIndex: 0, spinWaitMicros: 200, deferred, addon: standIn(calls) })
	const read = await rig.unpack('v210', w, h, '709', '2020')
	const write = await rig.pack('v210', w, h, '2020', false)
	const combine = n > 1 ? await rig.combine(n, w, h) : null
	const src = []
	for (let l = 0; l < n; ++l) src.push(await rig.planes('v210', w, h))
	const ring = [await rig.planes('v210', w, h, 'writeonly'), await rig.planes('v210', w, h, 'writeonly'), await rig.planes('v210', w, h, 'writeonly')]
	const one = async (f) => {
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
		const o = ring[f % ring.length]
		rig.post(c, write(cm, o, 0), () => cm.release())
		ids.push(c)
		await Promise.all(ids.map((id) => rig.board.flush(id)))
		rig.ctx.realise(o[0])
		return f % ring.length === ring.length - 1 ? rig.ctx.drain() : undefined
	}
	// --interlaced: bench_node.js's PH_NODE_BENCH_INTERLACED tick (one channel): per source frame n new ToRGBA, then twice (send_field)
	// n x (yadif -> transform) -> combine_n -> write; frames counts TICKS, us_per_frame is per tick (two output frames)
	let tick = one
	if (process.argv.includes('--interlaced')) {
		const yadif = await rig.yadif(w, h)
		const transform = await rig.transform(w, h)
		const mat = await transform.matrix({})
		const ring2 = [await rig.planes('v210', w, h, 'writeonly'), await rig.planes('v210', w, h, 'writeonly'), await rig.planes('v210', w, h, 'writeonly')]
		const window = []
		for (let l = 0; l < n; ++l) {
			const three = []
			for (let i = 0; i < 3; ++i) { const im = await rig.image(w, h); rig.post({ source: 'chan0', timestamp: -1 }, read(src[l], im)); three.push(im) }
			window.push(three)
		}
		await rig.board.flush({ source: 'chan0', timestamp: -1 })
		tick = async (f) => {
			const slot = f % 3
			const id = { source: 'chan0', timestamp: f }
			const gone = []
			for (let l = 0; l < n; ++l) {
				const im = await rig.image(w, h)
				rig.post(id, read(src[l], im))
				gone.push(window[l].shift())
				window[l].push(im)
			}
			for (const second of [0, 1]) {
				const placed = []
				for (let l = 0; l < n; ++l) {
					const u = window[l]
					const y = await rig.image(w, h)
					rig.post(id, yadif(u[0], u[1], u[2], y, { parity: second ? 1 : 0, tff: 1, skipSpatial: 0 }))
					const pl = await rig.image(w, h)
					rig.post(id, transform(y, pl, mat), () => y.release())
					placed.push(pl)
				}
				const cm = await rig.image(w, h)
				rig.post(id, combine(placed, cm), () => placed.forEach((b) => b.release()))
				rig.post(id, write(cm, (second ? ring2 : ring)[slot], 0), () => cm.release())
			}
			gone.forEach((b) => b.release())
			await rig.board.flush(id)
			rig.ctx.realise(ring[slot][0])
			rig.ctx.realise(ring2[slot][0])
			return slot === 2 ? rig.ctx.drain() : undefined
		}
	}
	for (let f = 0; f < 2000; ++f) await tick(f)
	for (const k of Object.keys(calls)) calls[k] = 0
	const t0 = process.hrtime.bigint()
	for (let f = 0; f < frames; ++f) await tick(2000 + f)
	const sec = Number(process.hrtime.bigint() - t0) / 1e9
	const per = {}
	for (const k of Object.keys(calls)) per[k] = +(calls[k] / frames).toFixed(2)
	console.log(JSON.stringify({ bench: 'defer_host', interlaced: process.argv.includes('--interlaced') || undefined, deferred, width: w, height: h, layers: n, frames, us_per_frame: +(1e6 * sec / frames).toFixed(2), addon_calls_per_frame: per, stats: rig.ctx.deferredStats() || undefined }))
}
main().catch((e) => { process.stderr.write(String(e && e.stack || e) + '\n'); process.exit(1) })
