'use strict'
// GPU check of buffer parking (node/index.js): a released frame is taken over whole by the next createBuffer of its shape - and the
// hazards the library's own pools guard against still hold (ADVICE r5): a download in flight at release, a mirror the previous owner
// filled but never handed back, a ROUTE transfer still using the block.  usage: node park_run.js; prints one JSON line {checks, problems}
const { clContext } = require('../index.js')
async function main() {
	const problems = []
	let checks = 0
	const expect = (what, got, want) => { ++checks; if (JSON.stringify(got) !== JSON.stringify(want)) problems.push({ what, got, want }) }
	const ctx = new clContext({ deviceIndex: 0 })
	await ctx.initialise()
	const native = ctx._native
	let reuses = 0
	const bufReuse = native.bufReuse
	native.bufReuse = (h) => { ++reuses; return bufReuse(h) }
	const bytes = 5529600 * 4 // four 1080p v210 frames' worth: a copy long enough to still be in flight at the release
	const fill = (v) => Buffer.alloc(bytes, v)
	// (a) release right after downloadAsync, before its waitFinish; the next owner fills its mirror at once
	for (let round = 0; round < 8; ++round) {
		const a = await ctx.createBuffer(bytes, 'readwrite', 'coarse', undefined, 'first owner')
		await a.hostAccess('writeonly', ctx.queue.load, fill(0x11 + round))
		await ctx.drain(ctx.queue.load)
		a.downloadAsync(ctx.queue.unload)
		a.release()
		const b = await ctx.createBuffer(bytes, 'readwrite', 'coarse', undefined, 'second owner')
		expect(`round ${round}: the parked buffer is taken over`, b === a, true)
		await b.hostAccess('writeonly', ctx.queue.load, fill(0xA0 + round))
		await ctx.drain(ctx.queue.load)
		await ctx.drain(ctx.queue.unload)
		await b.hostAccess('readonly', ctx.queue.unload)
		let bad = 0
		for (let i = 0; i < bytes; i += 4099) if (b[i] !== 0xA0 + round) ++bad
		expect(`round ${round}: the second owner reads back what it wrote`, bad, 0)
		b.release()
	}
	expect('every takeover of a buffer whose mirror had been in use was settled by the library (each round takes over twice, the first round once)', reuses, 15)
	// (b) a mirror mapped for writing and never handed back: the next owner's first job must not upload it
	const c = await ctx.createBuffer(bytes, 'readwrite', 'coarse', undefined, 'maps and leaves')
	await c.hostAccess('writeonly', ctx.queue.load, fill(0x33))
	await ctx.drain(ctx.queue.load)
	await c.hostAccess('writeonly', ctx.queue.load) // mapped: the library now expects host data ...
	c.fill(0x77)
	c.release() // ... which never comes
	const d = await ctx.createBuffer(bytes, 'readwrite', 'coarse', undefined, 'next')
	expect('the mapped buffer is taken over', d === c, true)
	await d.hostAccess('readonly', ctx.queue.unload) // device contents: still the 0x33 upload, not the abandoned 0x77
	let stale = 0
	for (let i = 0; i < bytes; i += 4099) if (d[i] !== 0x33) ++stale
	expect('the abandoned host data did not reach the device', stale, 0)
	const before = reuses
	const e = await ctx.createBuffer(bytes, 'readwrite', 'coarse', undefined, 'x')
	e.release()
	const f = await ctx.createBuffer(bytes, 'readwrite', 'coarse', undefined, 'y')
	expect('a buffer that was only created and released is taken over without a call', [f === e, reuses - before], [true, 0])
	f.release()
	d.release()
	ctx.trim()
	// (c) strictHandles: a fresh object per takeover; what the previous owner still holds is refused as a released nodencl buffer is
	const strict = new clContext({ deviceIndex: 0, strictHandles: true })
	await strict.initialise()
	const g = await strict.createBuffer(bytes, 'readwrite', 'coarse', undefined, 'first owner')
	await g.hostAccess('writeonly', strict.queue.load, fill(0x44))
	await strict.drain(strict.queue.load)
	g.release()
	const h = await strict.createBuffer(bytes, 'readwrite', 'coarse', undefined, 'second owner')
	let msgs = []
	try { g.release() } catch (e) { msgs.push(e.message) }
	try { await g.hostAccess('readonly', strict.queue.unload) } catch (e) { msgs.push(/released/.test(e.message)) }
	expect('strictHandles: another object over the parked block; the stale one is refused', [h !== g, h.refCount(), msgs], [true, 1, ['release on a released buffer', true]])
	await h.hostAccess('readonly', strict.queue.unload)
	let kept = 0
	for (let i = 0; i < bytes; i += 4099) if (h[i] !== 0x44) ++kept
	expect('strictHandles: it IS the parked block (the first owner\'s bytes are still on the device)', kept, 0)
	h.release()
	strict.trim()
	// (d) an event wait with spinWaitMicros: polled on this thread first, handed to the pool when the poll runs out - either way it ends after the work
	for (const spinWaitMicros of [0, 20, 5000]) {
		const c2 = new clContext({ deviceIndex: 0, spinWaitMicros })
		await c2.initialise()
		const big = await c2.createBuffer(bytes, 'readwrite', 'coarse', undefined, 'event')
		await big.hostAccess('writeonly', c2.queue.load, fill(0x5A))
		const ev = c2.recordEvent(c2.queue.load)
		await ev.wait()
		expect(`spinWaitMicros ${spinWaitMicros}: the event is done when wait() resolves`, ev.done(), true)
		big.release()
		c2.trim()
	}
	console.log(JSON.stringify({ checks, problems, reuses }))
}
main().catch((e) => { process.stderr.write(String(e && e.stack || e) + '\n'); process.exit(1) })
