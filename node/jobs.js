'use strict'
// JobBoard - a keyed kernel-job dispatcher with the contract the reference's producers, combiner and
// consumers rely on (src/clJobQueue.ts:53-141, SURVEY 8 a14):
//   (i)   jobs posted under one key run in the order they were posted;
//   (ii)  flushes are served first come first served, including ones requested while the board is busy;
//   (iii) every completion callback of a flush fires after the whole batch has finished on the device;
//   (iv)  flushing a key nobody posted to throws Error('Failed to run queue for id <key>');
//   (v)   cancel(prefix) fires the callbacks of matching pending jobs without running them (so their
//         owners drop their buffer references) and leaves the entries where they are.
// Own design, for a device whose launches are asynchronous: a batch is ENQUEUED without awaiting the single
// launches, flushes that are already waiting when the board turns to the device are enqueued together, and
// ONE waitFinish covers them all - on MI355X a waitFinish hand-off to the libuv pool costs about as much as
// a 1080p kernel runs, so draining once per turn instead of once per key is most of the host-side gain.
// A key is `${source} ts ${timestamp}`, the reference's spelling, so logs read the same.

const keyOf = (id) => `${id.source} ts ${id.timestamp}`

class JobBoard {
	constructor(clContext, options = {}) {
		this.ctx = clContext
		this.pending = new Map() // key -> [{ name, program, params, done }]
		this.waiting = [] // flushes not yet enqueued: { key, jobs, settle, fail }
		this.pump = null
		this.coalesce = options.coalesce !== false
		this.stats = { flushes: 0, drains: 0, kernels: 0 }
	}

	post(id, name, program, params, done) {
		const key = keyOf(id)
		let list = this.pending.get(key)
		if (!list) this.pending.set(key, (list = []))
		list.push({ name, program, params, done: done || (() => {}) })
	}

	peek(id) { return this.pending.get(keyOf(id)) }

	flush(id) {
		const key = keyOf(id)
		const jobs = this.pending.get(key)
		if (!jobs) return Promise.reject(new Error(`Failed to run queue for id ${key}`))
		this.pending.delete(key)
		return new Promise((settle, fail) => {
			this.waiting.push({ key, jobs, settle, fail })
			this._kick()
		})
	}

	// the turn starts on a microtask, so flushes requested in the same tick share its first drain; a flush that
	// arrives between a turn's last look at the list and its end starts the next turn
	_kick() {
		if (this.pump || !this.waiting.length) return
		// whatever a turn throws, the board keeps serving: the pump is cleared in every outcome
		this.pump = Promise.resolve()
			.then(() => this._turn())
			.catch((e) => this._failAll(e))
			.then(() => { this.pump = null; this._kick() })
	}

	// a turn died outside its own error handling: nobody may be left waiting on a flush that will never settle
	_failAll(e) {
		for (const f of this.waiting.splice(0)) {
			f.jobs.forEach((j) => { try { j.done() } catch (_) { /* the owner's callback is not the board's problem */ } })
			f.fail(e)
		}
	}

	cancel(prefix) {
		for (const [key, jobs] of this.pending) if (key.startsWith(prefix)) jobs.forEach((j) => j.done())
	}

	// A failing job ends its flush: the jobs behind it under that key are NOT run (they consume its output), the flush
	// rejects with the job's error, and every job's completion callback still fires - as cancel() does - so that the
	// owners drop their buffer references.  A failing waitFinish rejects every flush of the batch the same way.
	async _turn() {
		while (this.waiting.length) {
			const batch = this.coalesce ? this.waiting.splice(0) : [this.waiting.shift()]
			const q = this.ctx.queue.process
			for (const f of batch) {
				try {
					for (const j of f.jobs) {
						// (a recording context takes the job without a promise: clContext.recordProgram)
						if (!this.ctx.recordProgram || this.ctx.recordProgram(j.program, j.params, q) === null) await this.ctx.runProgram(j.program, j.params, q)
						this.stats.kernels++
					}
				} catch (e) { f.error = e }
			}
			let drainError = null
			try {
				await this.ctx.waitFinish(q)
				this.stats.drains++
			} catch (e) { drainError = e }
			for (const f of batch) {
				this.stats.flushes++
				for (const j of f.jobs) { try { j.done() } catch (e) { f.error = f.error || e } }
				const err = f.error || drainError
				if (err) f.fail(err); else f.settle()
			}
		}
	}
}

module.exports = { JobBoard, keyOf }
