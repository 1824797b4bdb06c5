'use strict'
// The JobBoard contract (node/jobs.js) against the recording mock context - no GPU.  Prints a JSON report that
// tests/test_node_boundary.py asserts on.
const { makeMock } = require('./mock_context')
const { JobBoard } = require('../jobs.js')
const { placementToTransform, dissolveMix } = require('../channel.js')

async function main() {
	const report = {}
	for (const coalesce of [true, false]) {
		const ctx = makeMock()
		const board = new JobBoard(ctx, { coalesce })
		const prog = await ctx.createProgram('phaneron:mixer', { name: 'mixer' })
		const order = []
		const mk = (src, ts, tag) => board.post({ source: src, timestamp: ts }, 'mixer', prog, { mix: ts / 10 }, () => order.push(tag))
		mk('A', 1, 'A1a'); mk('A', 1, 'A1b'); mk('B', 2, 'B2'); mk('A', 3, 'A3'); mk('C gone', 4, 'C4')
		const p1 = board.flush({ source: 'B', timestamp: 2 })
		const p2 = board.flush({ source: 'A', timestamp: 1 })
		const p3 = p1.then(() => board.flush({ source: 'A', timestamp: 3 })) // requested while the board is busy
		await Promise.all([p1, p2, p3])
		let unknown = null
		try { await board.flush({ source: 'nobody', timestamp: 5 }) } catch (e) { unknown = { isError: e instanceof Error, message: e.message } }
		board.cancel('C gone')
		const left = board.peek({ source: 'C gone', timestamp: 4 })
		// what reached the device, in order, and where the drains fell
		const device = ctx.trace.filter((e) => e.op === 'runProgram' || e.op === 'waitFinish').map((e) => e.op === 'waitFinish' ? 'wait' : e.params.mix)
		report[coalesce ? 'coalesced' : 'oneByOne'] = { order, unknown, pendingAfterCancel: left ? left.length : 0, device, stats: board.stats }
	}
	// failures: a rejecting waitFinish and a failing job must settle every flush, fire every callback and leave the board serving
	{
		const ctx = makeMock()
		const board = new JobBoard(ctx)
		const prog = await ctx.createProgram('phaneron:mixer', { name: 'mixer' })
		const fired = []
		const mk = (src, ts, tag) => board.post({ source: src, timestamp: ts }, 'mixer', prog, { mix: ts / 10 }, () => fired.push(tag))
		const realWait = ctx.waitFinish, realRun = ctx.runProgram
		let waits = 0
		ctx.waitFinish = (q) => (++waits === 1 ? Promise.reject(new Error('device lost')) : realWait.call(ctx, q))
		mk('A', 1, 'A1'); mk('B', 2, 'B2')
		const outcome = await Promise.all([board.flush({ source: 'A', timestamp: 1 }), board.flush({ source: 'B', timestamp: 2 })]
			.map((p) => p.then(() => 'ok', (e) => e.message)))
		// a job that throws: its flush rejects, the job behind it is skipped but its callback fires; the other flush of the turn runs
		ctx.runProgram = (pr, params, q) => (params.mix === 0.3 ? Promise.reject(new Error('bad launch')) : realRun.call(ctx, pr, params, q))
		mk('C', 3, 'C3a'); mk('C', 3, 'C3b'); mk('D', 4, 'D4')
		const outcome2 = await Promise.all([board.flush({ source: 'C', timestamp: 3 }), board.flush({ source: 'D', timestamp: 4 })]
			.map((p) => p.then(() => 'ok', (e) => e.message)))
		ctx.runProgram = realRun
		mk('E', 5, 'E5')
		const after = await board.flush({ source: 'E', timestamp: 5 }).then(() => 'ok', (e) => e.message)
		await new Promise((r) => setImmediate(r))
		report.failures = { outcome, outcome2, after, fired, pumpIdle: board.pump === null }
	}
	report.placement = placementToTransform({ anchor: { x: 0.25, y: 0.75 }, rotation: 30, fill: { xOffset: 0.25, yOffset: -0.125, xScale: 0.5, yScale: 0.5 } })
	report.placementDefault = placementToTransform()
	report.dissolve = [0, 1, 2, 3].map((k) => dissolveMix(k, 4)).concat([dissolveMix(0, 1), dissolveMix(0, 0)])
	process.stdout.write(JSON.stringify(report))
}
main().catch((e) => { process.stderr.write(String(e && e.stack || e) + '\n'); process.exit(1) })
