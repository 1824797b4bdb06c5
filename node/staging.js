'use strict'
// Producer / consumer staging for one channel (SURVEY 8f-3): the queue.load / queue.unload roles of
// the reference's io.ts:79-98,166-174 with a ring of frames in flight, ordered on the device
// (clContext.queueWaitQueue / recordEvent) instead of through host-side waitFinish.  Same design as
// phaneron_amd/staging.py:
//   fill(frameNo, sourceBuffers)   write the frame's bytes into the pinned mirrors (the buffers ARE
//                                  node Buffers), e.g. straight from a decoder
//   process(sources, output)       enqueue the frame's kernels (ToRGBA.processFrame ... runQueue, or
//                                  one fused program) on queue.process
//   consume(frameNo, outputBuffer) called when the frame's bytes are back in the output mirror

class StagedChannel {
	constructor(clContext, sourceBytes, outputBytes, process, depth = 3, tag = 'chan') {
		this.ctx = clContext
		this.sourceBytes = sourceBytes
		this.outputBytes = outputBytes
		this.process = process
		this.depth = depth
		this.tag = tag
		this.slots = []
		this.submitted = 0
	}

	async init() {
		for (let k = 0; k < this.depth; ++k) {
			const sources = []
			for (let i = 0; i < this.sourceBytes.length; ++i)
				sources.push(await this.ctx.createBuffer(this.sourceBytes[i], 'readonly', 'coarse', undefined, `${this.tag} slot${k} src${i}`))
			const output = await this.ctx.createBuffer(this.outputBytes, 'writeonly', 'coarse', undefined, `${this.tag} slot${k} out`)
			this.slots.push({ sources, output, done: null, frame: -1 })
		}
	}

	async _retire(slot, consume) {
		if (slot.done) {
			await slot.done.wait()
			slot.done = null
			if (consume) await consume(slot.frame, slot.output)
		}
	}

	async submit(fill, consume) {
		const q = this.ctx.queue
		const slot = this.slots[this.submitted % this.depth]
		await this._retire(slot, consume)
		for (const b of slot.sources) await b.hostAccess('writeonly', q.load)
		await fill(this.submitted, slot.sources)
		for (const b of slot.sources) await b.hostAccess('none', q.load)
		this.ctx.queueWaitQueue(q.process, q.load)
		await this.process(slot.sources, slot.output)
		this.ctx.queueWaitQueue(q.unload, q.process)
		slot.output.downloadAsync(q.unload)
		slot.done = this.ctx.recordEvent(q.unload)
		slot.frame = this.submitted
		return this.submitted++
	}

	async drain(consume) {
		for (let f = Math.max(0, this.submitted - this.depth); f < this.submitted; ++f)
			await this._retire(this.slots[f % this.depth], consume)
	}

	async close() {
		await this.drain()
		for (const s of this.slots) {
			s.sources.forEach((b) => b.release())
			s.output.release()
		}
		this.slots = []
	}
}

module.exports = { StagedChannel }
