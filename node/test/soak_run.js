'use strict'
// The recording context (the default clContext) over a long stream, with a fault in the middle, and with `profile` (VERDICT r4 item 4):
//   soak     FRAMES frames of the valves' posting pattern (a fresh destination per job, released in its callback, three output frames
//            in flight), the format changing every FRAMES / 4 frames (1080 -> 720 -> 2160 -> 1080): the library's buffer counters and
//            pinned bytes must come back to where they were, nothing pinned in steady state, no fused launch refused;
//   channels four channels posting a frame per tick (placed layers / plain reads alternately) for FRAMES / 5 ticks, the format changing half
//            way: one runPrograms call per tick, every frame through a batch launch, the same counters flat;
//   fault    launches made to fail (context option fail_launches) while frame K's consumer maps its frame: that hostAccess rejects,
//            every job callback has fired, the frames after it are the launch-as-posted context's bytes, nothing leaks;
//   timings  `profile: true`: the fused launch's device time is shared out over the frame's jobs (every row non-zero, the rows sum to the launch).
// usage: node soak_run.js [frames=100000] (PH_SOAK_ONLY=soak,channels,fault,timings picks phases); prints one JSON object { soak, channels, fault, timings, problems }
const { Rig } = require('../device.js')

const problems = []
const FRAMES = parseInt(process.argv[2] || '100000')
const v210Bytes = (w, h) => Math.ceil(w / 48) * 128 * h
function fill(buf, seed) {
	let s = seed >>> 0
	for (let i = 0; i + 4 <= buf.length; i += 4) { s = (Math.imul(s, 1664525) + 1013904223) >>> 0; buf.writeUInt32LE(((64 + (s >>> 8) % 877) * 0x00100401) & 0x3fffffff, i) }
}

// one channel of `n` 1:1 layers at w x h: sources resident, stages made once
async function channel(rig, w, h, n) {
	const c = { w, h, n, read: await rig.unpack('v210', w, h, '709', '709'), write: await rig.pack('v210', w, h, '709', false), combine: await rig.combine(n, w, h), src: [], ring: [] }
	for (let l = 0; l < n; ++l) {
		const p = await rig.planes('v210', w, h)
		await p[0].hostAccess('writeonly', rig.ctx.queue.load) // map, fill, unmap: 'none' uploads what was mapped for writing (io.ts:89-94)
		fill(p[0], 7 * w + l)
		await p[0].hostAccess('none', rig.ctx.queue.load)
		c.src.push(p)
	}
	for (let i = 0; i < 3; ++i) c.ring.push(await rig.planes('v210', w, h, 'writeonly'))
	await rig.sync(rig.ctx.queue.load)
	c.close = () => [...c.src.flat(), ...c.ring.flat()].forEach((b) => b.release())
	return c
}
// a frame as the valves post it; returns the output plane of ring slot f % 3 (not yet asked for)
async function post(rig, c, f, fired) {
	const ids = []
	const fresh = []
	for (let l = 0; l < c.n; ++l) {
		const im = await rig.image(c.w, c.h)
		const id = { source: `L${l}`, timestamp: f }
		rig.post(id, c.read(c.src[l], im), () => { if (fired) fired.n++ })
		fresh.push(im)
		ids.push(id)
	}
	const k = { source: 'combine', timestamp: f }
	const cm = await rig.image(c.w, c.h)
	rig.post(k, c.combine(fresh, cm), () => { fresh.forEach((b) => b.release()); if (fired) fired.n++ })
	const o = c.ring[f % 3]
	rig.post(k, c.write(cm, o, 0), () => { cm.release(); if (fired) fired.n++ })
	ids.push(k)
	await Promise.all(ids.map((id) => rig.board.flush(id)))
	return o[0]
}

async function soak() {
	const rig = await Rig.open({ deviceIndex: 0, spinWaitMicros: 200 })
	const formats = [[1920, 1080], [1280, 720], [3840, 2160], [1920, 1080]]
	const per = Math.max(8, Math.floor(FRAMES / formats.length))
	const marks = []
	const t0 = process.hrtime.bigint()
	for (const [w, h] of formats) {
		const c = await channel(rig, w, h, 4)
		const done = []
		let warm = null
		for (let f = 0; f < per; ++f) {
			const slot = f % 3
			if (done[slot] && !done[slot].done()) await done[slot].wait()
			rig.ctx.realise(await post(rig, c, f))
			done[slot] = rig.ctx.recordEvent(rig.ctx.queue.process)
			if (f === Math.min(64, per - 1)) warm = rig.ctx.bufferStats() // the working set of this format is in place
		}
		await rig.ctx.drain()
		const end = rig.ctx.bufferStats()
		if (end.pins !== warm.pins) problems.push({ soak: `${w}x${h}`, what: `${end.pins - warm.pins} blocks pinned after the format's first frames (the pools do not cover the working set)` })
		if (end.liveBuffers !== warm.liveBuffers) problems.push({ soak: `${w}x${h}`, what: `live buffers ${warm.liveBuffers} -> ${end.liveBuffers} over ${per} frames` })
		if (end.pinnedInUse + end.pinnedPooled !== warm.pinnedInUse + warm.pinnedPooled) problems.push({ soak: `${w}x${h}`, what: `pinned bytes ${warm.pinnedInUse + warm.pinnedPooled} -> ${end.pinnedInUse + end.pinnedPooled}` })
		c.close()
		marks.push({ format: `${w}x${h}`, frames: per, live: end.liveBuffers, parked: end.parkedBuffers, pinned_mb: Math.round((end.pinnedInUse + end.pinnedPooled) / 1048576), pins: end.pins })
	}
	const sec = Number(process.hrtime.bigint() - t0) / 1e9
	const st = rig.ctx.deferredStats()
	rig.close()
	rig.ctx.trim()
	const left = rig.ctx.bufferStats()
	if (st.fallbacks) problems.push({ soak: 'all', what: `${st.fallbacks} fused launches refused: ${st.lastFallback}` })
	if (st.pending) problems.push({ soak: 'all', what: `${st.pending} jobs still recorded at the end` })
	if (st.fused !== formats.length * per) problems.push({ soak: 'all', what: `${st.fused} fused launches for ${formats.length * per} frames` })
	if (left.liveBuffers) problems.push({ soak: 'all', what: `${left.liveBuffers} buffers alive after everything was released` })
	return { frames: formats.length * per, seconds: +sec.toFixed(2), us_per_frame: +(1e6 * sec / (formats.length * per)).toFixed(1), formats: marks, deferred: st }
}

// the reference's deployment over a long stream: C channels in one context (src/index.ts:45-71), every channel posting a frame per tick -
// two placed layers each (the channel kernel), two plain reads (the headline kernel), or - the last channel - two clips of half the
// channel's size filling the frame (read + 2 x 2-block compositor inside the call) - so that a tick is ONE
// runPrograms call; the format changes half way (1080 -> 720); counters flat, every frame through a batch, nothing refused
async function soakChannels(ticks) {
	const C = 4
	const rig = await Rig.open({ deviceIndex: 0, spinWaitMicros: 200 })
	const marks = []
	const per = Math.max(8, Math.floor(ticks / 2))
	const t0 = process.hrtime.bigint()
	for (const [w, h] of [[1920, 1080], [1280, 720]]) {
		const chans = []
		for (let c = 0; c < C; ++c) chans.push(await channel(rig, w, h, 2))
		// the last channel shows clips of HALF the channel's size, filling the frame: the library makes such frames by read + 2 x 2-block compositor
		const small = await channel(rig, w / 2, h / 2, 2)
		chans[C - 1].src.flat().forEach((b) => b.release())
		chans[C - 1].src = small.src
		chans[C - 1].read = small.read
		const transform = await rig.transform(w, h)
		const mats = [await transform.matrix({}), await transform.matrix({ scaleX: 0.5, scaleY: 0.5, offsetX: 0.25, offsetY: -0.25 })]
		const fills = [await transform.matrix({}), await transform.matrix({ scaleX: 0.9, scaleY: 0.9 })]
		const done = []
		let warm = null
		for (let f = 0; f < per; ++f) {
			const slot = f % 3
			if (done[slot] && !done[slot].done()) await done[slot].wait()
			const ids = []
			for (let c = 0; c < C; ++c) {
				const ch = chans[c]
				const id = { source: `chan${c}`, timestamp: f }
				const layers = []
				for (let l = 0; l < 2; ++l) {
					const enlarged = c === C - 1
					const im = enlarged ? await rig.image(w / 2, h / 2) : await rig.image(w, h)
					rig.post(id, ch.read(ch.src[l], im))
					if (c % 2 && !enlarged) { layers.push(im); continue }
					const pl = await rig.image(w, h)
					rig.post(id, transform(im, pl, enlarged ? fills[l] : mats[l]), () => im.release())
					layers.push(pl)
				}
				const cm = await rig.image(w, h)
				rig.post(id, ch.combine(layers, cm), () => layers.forEach((b) => b.release()))
				rig.post(id, ch.write(cm, ch.ring[slot], 0), () => cm.release())
				ids.push(id)
			}
			await Promise.all(ids.map((id) => rig.board.flush(id)))
			for (const ch of chans) rig.ctx.realise(ch.ring[slot][0])
			done[slot] = rig.ctx.recordEvent(rig.ctx.queue.process)
			if (f === Math.min(64, per - 1)) warm = rig.ctx.bufferStats()
		}
		await rig.ctx.drain()
		const end = rig.ctx.bufferStats()
		if (end.pins !== warm.pins) problems.push({ channels: `${w}x${h}`, what: `${end.pins - warm.pins} blocks pinned after the format's first ticks` })
		if (end.liveBuffers !== warm.liveBuffers) problems.push({ channels: `${w}x${h}`, what: `live buffers ${warm.liveBuffers} -> ${end.liveBuffers} over ${per} ticks` })
		if (end.pinnedInUse + end.pinnedPooled !== warm.pinnedInUse + warm.pinnedPooled) problems.push({ channels: `${w}x${h}`, what: `pinned bytes ${warm.pinnedInUse + warm.pinnedPooled} -> ${end.pinnedInUse + end.pinnedPooled}` })
		chans[C - 1].src = []
		chans.forEach((ch) => ch.close())
		small.close()
		marks.push({ format: `${w}x${h}`, ticks: per, live: end.liveBuffers, parked: end.parkedBuffers, pinned_mb: Math.round((end.pinnedInUse + end.pinnedPooled) / 1048576), pins: end.pins })
	}
	const sec = Number(process.hrtime.bigint() - t0) / 1e9
	const st = rig.ctx.deferredStats()
	rig.close()
	rig.ctx.trim()
	const left = rig.ctx.bufferStats()
	if (st.fallbacks) problems.push({ channels: 'all', what: `${st.fallbacks} fused launches refused: ${st.lastFallback}` })
	if (st.pending) problems.push({ channels: 'all', what: `${st.pending} jobs still recorded at the end` })
	if (st.fused !== 2 * per * C || st.batched !== st.fused || st.launched !== 2 * per) problems.push({ channels: 'all', what: `fused ${st.fused}, batched ${st.batched}, launched ${st.launched} for ${2 * per} ticks of ${C} channels` })
	if (left.liveBuffers) problems.push({ channels: 'all', what: `${left.liveBuffers} buffers alive after everything was released` })
	return { channels: C, ticks: 2 * per, seconds: +sec.toFixed(2), us_per_channel_frame: +(1e6 * sec / (2 * per * C)).toFixed(1), formats: marks, deferred: st }
}

async function fault() {
	const W = 384
	const H = 108
	const N = 8
	const K = 3
	const seen = {}
	const out = {}
	for (const deferred of [false, true]) {
		const rig = await Rig.open({ deviceIndex: 0, deferred, spinWaitMicros: 100 })
		const c = await channel(rig, W, H, 2)
		const frames = []
		const fired = { n: 0 }
		let rejected = null
		for (let f = 0; f < N; ++f) {
			// every frame another picture: the top source is rewritten (its pending readers - none by now - would run first)
			await c.src[1][0].hostAccess('writeonly', rig.ctx.queue.load)
			fill(c.src[1][0], 1000 + f)
			await c.src[1][0].hostAccess('none', rig.ctx.queue.load)
			await rig.sync(rig.ctx.queue.load)
			const broken = deferred && f === K
			let o = null
			try {
				if (broken) {
					// the frame is posted and flushed like any other (its jobs are only recorded); the launches fail when the consumer asks
					o = await post(rig, c, f, fired)
					rig.ctx.setOption('fail_launches', 1)
				} else o = await post(rig, c, f, fired)
				await rig.sync()
				await rig.download(o)
				frames.push(Buffer.from(o))
			} catch (e) {
				rejected = String(e && e.message || e)
				frames.push(null)
			} finally { if (broken) rig.ctx.setOption('fail_launches', 0) }
		}
		await rig.ctx.drain()
		const st = rig.ctx.deferredStats()
		c.close()
		rig.close()
		rig.ctx.trim()
		const left = rig.ctx.bufferStats()
		seen[deferred] = frames
		out[deferred ? 'deferred' : 'plain'] = { callbacks: fired.n, rejected, live_after: left.liveBuffers, stats: st }
		if (fired.n !== N * 4) problems.push({ fault: deferred, what: `${fired.n} job callbacks fired, ${N * 4} jobs posted` })
		if (left.liveBuffers) problems.push({ fault: deferred, what: `${left.liveBuffers} buffers alive after everything was released` })
		if (deferred && st.pending) problems.push({ fault: true, what: `${st.pending} jobs still recorded at the end` })
		if (deferred && !/injected/.test(rejected || '')) problems.push({ fault: true, what: `frame ${K}'s consumer was not told: ${rejected}` })
	}
	if (seen[false][0] && seen[false][1] && Buffer.compare(seen[false][0], seen[false][1]) === 0) problems.push({ fault: false, what: 'frames 0 and 1 are the same picture: the rewritten source did not reach the device' })
	for (let f = 0; f < N; ++f) {
		if (f === K) { if (seen[true][f] !== null) problems.push({ fault: true, what: `frame ${K} was delivered although its launches failed` }); continue }
		if (!seen[true][f] || Buffer.compare(seen[true][f], seen[false][f]) !== 0) {
			let at = 0
			let count = 0
			if (seen[true][f]) for (let i = 0; i < seen[true][f].length; ++i) if (seen[true][f][i] !== seen[false][f][i]) { if (!count) at = i; ++count }
			problems.push({ fault: true, what: `frame ${f} differs from the launch-as-posted context's`, first_byte: at, bytes: count })
		}
	}
	return out
}

async function timings() {
	const rig = await Rig.open({ deviceIndex: 0, profile: true })
	const c = await channel(rig, 1920, 1080, 4)
	const got = []
	for (let f = 0; f < 4; ++f) {
		const imgs = []
		const t = []
		for (let l = 0; l < 4; ++l) { const im = await rig.image(c.w, c.h); t.push(await rig.run(c.read(c.src[l], im))); imgs.push(im) }
		const cm = await rig.image(c.w, c.h)
		t.push(await rig.run(c.combine(imgs, cm)))
		const w = await rig.run(c.write(cm, c.ring[f % 3], 0))
		;[...imgs, cm].forEach((b) => b.release())
		// the frame reached the device as ONE launch when its write was posted: that launch's device time is shared out over the frame's six
		// jobs (reads 3 parts each, combine 1, write 3), filled into the RunTimings objects the folded jobs were handed when they were recorded
		const sum = t.reduce((a, x) => a + x.kernelExec, 0) + w.kernelExec
		got.push({ folded: t.map((x) => x.kernelExec), write: w, sum })
		if (t.some((x) => !(x.kernelExec > 0))) problems.push({ timings: f, what: `a job folded into the frame's launch has no share of its time: ${JSON.stringify(t)}` })
		if (!(t[0].kernelExec === t[1].kernelExec && t[0].kernelExec >= 2 * t[4].kernelExec)) problems.push({ timings: f, what: `shares: reads alike, a read about three combines: ${JSON.stringify(t)}` })
		if (!(sum > 5 && sum < 5000 && w.kernelExec > 0 && w.totalTime >= w.kernelExec)) problems.push({ timings: f, what: `the frame's rows must sum to the launch's device time: ${sum}, write ${JSON.stringify(w)}` })
	}
	const st = rig.ctx.deferredStats()
	if (st.fused !== 4 || st.plain !== 0) problems.push({ timings: 'all', what: `fused ${st.fused}, plain ${st.plain}` })
	c.close()
	rig.close()
	return got
}

async function main() {
	const only = process.env.PH_SOAK_ONLY ? process.env.PH_SOAK_ONLY.split(',') : null
	const want = (what) => !only || only.includes(what)
	const result = { soak: want('soak') ? await soak() : null, channels: want('channels') ? await soakChannels(Math.max(16, Math.floor(FRAMES / 5))) : null, fault: want('fault') ? await fault() : null,
		timings: want('timings') ? await timings() : null, problems }
	process.stdout.write(JSON.stringify(result) + '\n')
}
main().catch((e) => { process.stderr.write(String(e && e.stack || e) + '\n'); process.exit(1) })
