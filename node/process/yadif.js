'use strict'
// Yadif: three-frame window around the kernel, one (send_frame) or two (send_field) outputs per
// interlaced input frame (reference: src/process/yadif.ts).
const ImageProcess = require('./imageProcess').default
const YadifCl = require('./yadifCl').default

class Yadif {
	constructor(clContext, clJobs, width, height, config, interlaced) {
		this.clContext = clContext
		this.clJobs = clJobs
		this.width = width
		this.height = height
		this.config = config
		this.interlaced = interlaced
		const mode = config.mode
		this.sendField = interlaced && (mode === 'send_field' || mode === 'send_field_nospatial')
		this.skipSpatial = mode === 'send_frame_nospatial' || mode === 'send_field_nospatial'
		this.yadifCl = null
		this.window = []
		this.out = null
	}

	async init() {
		this.yadifCl = new ImageProcess(this.clContext, new YadifCl(this.width, this.height), this.clJobs)
		await this.yadifCl.init()
	}

	async makeOutput(isSecond, sourceID, timestamp) {
		this.out = await this.clContext.createBuffer(this.width * this.height * 4 * 4, 'readwrite', 'coarse',
			{ width: this.width, height: this.height }, `yadif ${isSecond ? '2' : '1'} ${sourceID} ${timestamp}`)
	}

	async runYadif(isSecond, sourceID) {
		if (!this.yadifCl) throw new Error('Yadif needs to be initialised')
		const srcs = this.window.slice(0) // held until the job's callback
		srcs.forEach((s) => s.addRef())
		const out = this.out
		out.timestamp = srcs[1].timestamp + (isSecond ? 1 : 0)
		await this.yadifCl.run(
			{
				prev: srcs[0],
				cur: srcs[1],
				next: srcs[2],
				parity: (this.config.tff ? 1 : 0) ^ (!isSecond ? 1 : 0),
				tff: this.config.tff,
				skipSpatial: this.skipSpatial,
				output: out
			},
			{ source: sourceID, timestamp: out.timestamp },
			() => srcs.forEach((s) => s.release())
		)
		await this.clJobs.runQueue({ source: sourceID, timestamp: out.timestamp })
	}

	async processFrame(input, outputs, sourceID) {
		if (!this.interlaced) {
			outputs.push(input)
			return
		}
		this.window.push(input)
		if (this.window.length < 3) {
			// run whatever was queued for this input so its sources are released
			await this.clJobs.runQueue({ source: sourceID, timestamp: input.timestamp })
			return
		}
		if (this.window.length > 3) {
			const old = this.window.shift()
			if (old) old.release()
		}
		await this.makeOutput(false, sourceID, this.window[1].timestamp)
		await this.runYadif(false, sourceID)
		outputs.push(this.out)
		if (this.sendField) {
			await this.makeOutput(true, sourceID, this.window[1].timestamp + 1)
			await this.runYadif(true, sourceID)
			outputs.push(this.out)
		}
	}

	release() {
		this.window.forEach((i) => i.release())
	}
}

module.exports = { default: Yadif }
