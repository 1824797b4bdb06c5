'use strict'
// The handful of redioactive notions the video valves need (reference: the `redioactive` npm
// dependency, used by src/producer/mixer.ts, src/transitioner.ts, src/combiner.ts): the `end` and `nil`
// sentinels, isValue / isEnd / isNil, and a minimal pull pipe - enough to wire producer -> valve ->
// zipEach -> valve -> consumer in node/test and in a host application.  Own code; redioactive's
// scheduling (buffer sizes, back pressure, http transport) is out of scope (SURVEY 8: plumbing).
const end = Object.freeze({ end: true })
const nil = Object.freeze({ nil: true })
const isEnd = (t) => t === end
const isNil = (t) => t === nil
const isValue = (t) => t !== end && t !== nil

class Pipe {
	// pull: () => value | end | nil | Promise of those.  After `end` the pipe keeps answering `end`.
	constructor(pull) {
		this._pull = pull
		this._ended = false
		this._forks = null
	}

	async next() {
		if (this._forks) throw new Error('a forked pipe is read through its forks')
		if (this._ended) return end
		for (;;) {
			const v = await this._pull()
			if (isNil(v)) continue
			if (isEnd(v)) this._ended = true
			return v
		}
	}

	// fn sees values and the final `end`; returning nil drops the item
	valve(fn) {
		return new Pipe(async () => fn(await this.next()))
	}

	// [own, ...others] per step; the zip ends when THIS pipe ends (the others may have ended
	// already: their slots then carry `end`, which is what the reference valves test for)
	zipEach(others) {
		return new Pipe(async () => {
			const own = await this.next()
			if (isEnd(own)) return end
			const rest = await Promise.all(others.map((p) => p.next()))
			return [own, ...rest]
		})
	}

	// every fork sees every item; a fork that lags keeps its own queue
	fork() {
		if (!this._forks) {
			const source = new Pipe(this._pull)
			source._ended = this._ended
			this._forks = { source, queues: [] }
		}
		const shared = this._forks
		const queue = []
		shared.queues.push(queue)
		const f = new Pipe(async () => {
			if (queue.length === 0) {
				const v = await shared.source.next()
				shared.queues.forEach((q) => q.push(v))
			}
			return queue.shift()
		})
		f._unfork = () => {
			const i = shared.queues.indexOf(queue)
			if (i >= 0) shared.queues.splice(i, 1)
		}
		return f
	}

	unfork(f) {
		if (f && f._unfork) f._unfork()
	}

	// drain into fn until end
	async each(fn) {
		for (;;) {
			const v = await this.next()
			if (isEnd(v)) return
			await fn(v)
		}
	}
}

// source from a function (redioactive's `redio(() => value | end)`) or from an array
function redio(src) {
	if (Array.isArray(src)) {
		let i = 0
		return new Pipe(() => (i < src.length ? src[i++] : end))
	}
	return new Pipe(src)
}

module.exports = { end, nil, isEnd, isNil, isValue, Pipe, redio }
