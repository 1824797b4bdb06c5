'use strict'
// node/clJobQueue.js - the dispatcher of the reference (src/clJobQueue.ts:40-157) over any
// nodencl-shaped context.  Contract kept (SURVEY.md 8 a14):
//  (i) jobs under one key run in insertion order; (ii) requests run FIFO, including ones that
//  arrive while the queue drains; (iii) every callback of a request fires after the whole
//  batch has finished on the device (one waitFinish per batch); (iv) runQueue on an unknown
//  key throws; (v) clearQueue(prefix) fires callbacks without running and keeps the entries.
const { EventEmitter } = require('events')

class ClJobs {
	constructor(processJobs) {
		this.processJobs = processJobs
		this.jobs = new Map()
	}

	makeKey(id) {
		return `${id.source} ts ${id.timestamp}`
	}

	add(id, name, program, params, cb) {
		const key = this.makeKey(id)
		if (!this.jobs.has(key)) this.jobs.set(key, [])
		this.jobs.get(key).push({ name, program, params, cb })
	}

	get(id) {
		return this.jobs.get(this.makeKey(id))
	}

	delete(id) {
		this.jobs.delete(this.makeKey(id))
	}

	clear() {
		this.jobs.clear()
	}

	async runQueue(id) {
		const key = this.makeKey(id)
		const batch = this.jobs.get(key)
		if (!batch) throw new Error(`Failed to run queue for id ${key}`)
		return new Promise((resolve) => {
			this.processJobs.requestRun(key, { id: key, jobs: batch, start: process.hrtime(), done: resolve })
			this.delete(id)
		})
	}

	clearQueue(src) {
		for (const [key, batch] of this.jobs) {
			if (key.startsWith(src)) batch.forEach((j) => j.cb()) // release held references; entries stay
		}
	}
}

class ClProcessJobs {
	constructor(clContext) {
		this.clContext = clContext
		this.requests = new Map()
		this.runEvents = new EventEmitter()
		this.clJobs = new ClJobs(this)
		this.showTimings = 0
		this.runEvents.once('run', () => this.processQueue())
	}

	async processQueue() {
		// Map iteration sees entries appended while we await, which is what keeps late requests FIFO
		for (const [key, req] of this.requests) {
			const timings = new Map()
			const queued = process.hrtime(req.start)
			for (const job of req.jobs) {
				timings.set(job.name, await this.clContext.runProgram(job.program, job.params, this.clContext.queue.process))
			}
			const submitted = process.hrtime(req.start)
			await this.clContext.waitFinish(this.clContext.queue.process)
			req.jobs.forEach((j) => j.cb())
			const finished = process.hrtime(req.start)
			this.logTimings(req.id, queued, submitted, finished, timings)
			req.done()
			this.requests.delete(key)
		}
		this.runEvents.once('run', () => this.processQueue())
	}

	getJobs() {
		return this.clJobs
	}

	requestRun(id, request) {
		this.requests.set(id, request)
		this.runEvents.emit('run')
	}

	logRequests() {
		let i = 0
		this.requests.forEach((r) => console.log(`${i++}: ${r.id} ${r.jobs.map((j) => j.name)}`))
	}

	logTimings(id, queued, submitted, finished, timings) {
		if (this.showTimings <= 0) return
		const ms = (t) => t[0] * 1e3 + t[1] / 1e6
		const label = id.slice(-20)
		if (this.showTimings > 1) {
			console.log(`\n${label} | toGPU | process | total (microseconds)`)
			let total = 0
			for (const [name, t] of timings) {
				console.log(`${name.padEnd(26)}| ${String(t.dataToKernel).padStart(7)} | ${String(t.kernelExec).padStart(7)} | ${String(t.totalTime).padStart(7)}`)
				total += t.totalTime
			}
			console.log(`execute ${((ms(finished) - ms(submitted)) * 1000) >>> 0} us, kernels total ${total} us`)
		}
		console.log(`${label}: ${ms(finished).toFixed(2)}ms elapsed (${ms(queued).toFixed(2)}ms job queued, ${(ms(submitted) - ms(queued)).toFixed(2)}ms submit, ${(ms(finished) - ms(submitted)).toFixed(2)}ms execute)`)
	}
}

module.exports = { ClJobs, ClProcessJobs }
