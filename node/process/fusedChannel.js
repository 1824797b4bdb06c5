'use strict'
// FusedV210Channel - NOT a reference operator: the headline chain of the reference
// (ToRGBA x N -> Combine -> FromRGBA on v210 frames; producer -> combiner.ts:219-254 -> consumer) as
// ONE program of the HIP library (ph_fused_v210_combine).  It owns a Loader and a Saver only for
// their colour parameter buffers, exactly the ones the separate operators would upload, and queues
// its job on the same ClJobs dispatcher.  Results are bit-identical to the separate operators.
const { Loader, Saver } = require('./loadSave')
const v210 = require('./v210')

class FusedV210Channel {
	constructor(clContext, colSpecRead, colSpecWrite, numLayers, width, height, clJobs) {
		if (numLayers < 1 || numLayers > 8) throw new Error(`FusedV210Channel supports 1..8 layers, got ${numLayers}`)
		this.clContext = clContext
		this.clJobs = clJobs
		this.numLayers = numLayers
		this.width = width
		this.height = height
		this.name = `fused_v210_combine_${numLayers}`
		this.loader = new Loader(clContext, colSpecRead, colSpecWrite, new v210.Reader(width, height), clJobs)
		this.saver = new Saver(clContext, colSpecWrite, new v210.Writer(width, height, false), clJobs)
		this.numBytes = this.loader.packImpl.getNumBytes()[0]
		this.program = null
	}

	async init() {
		await this.loader.init()
		await this.saver.init()
		this.program = await this.clContext.createProgram('phaneron:fused', {
			name: this.name,
			globalWorkItems: Uint32Array.from([this.width, this.height])
		})
	}

	getNumBytes() { return this.numBytes }

	async createSource(id) {
		return this.clContext.createBuffer(this.numBytes, 'readonly', 'coarse', undefined, `fused src ${id}`)
	}
	async createDest(id) {
		return this.clContext.createBuffer(this.numBytes, 'writeonly', 'coarse', undefined, `fused out ${id}`)
	}

	kernelParams(sources, dest) {
		if (sources.length !== this.numLayers)
			throw new Error(`${this.name} requires ${this.numLayers} source buffers, found ${sources.length}`)
		const kp = { output: dest }
		sources.forEach((s, i) => { kp[`l${i}In`] = s })
		kp.colMatrix = this.loader.colMatrix
		kp.gammaLut = this.loader.gammaLut
		kp.gamutMatrix = this.loader.gamutMatrix
		kp.outColMatrix = this.saver.colMatrix
		kp.outGammaLut = this.saver.gammaLut
		return kp
	}

	// queue the frame on the dispatcher under `id` ({source, timestamp}); cb fires after the batch ran
	processFrame(id, sources, dest, cb) {
		if (this.program === null) throw new Error('FusedV210Channel.processFrame failed with no program available')
		this.loader.addRefs()
		this.saver.addRefs()
		this.clJobs.add(id, this.name, this.program, this.kernelParams(sources, dest), () => {
			this.loader.releaseRefs()
			this.saver.releaseRefs()
			if (cb) cb()
		})
	}

	// launch straight on the process queue (staged rings order queues on the device, node/staging.js)
	async launch(sources, dest) {
		if (this.program === null) throw new Error('FusedV210Channel.launch failed with no program available')
		return this.clContext.runProgram(this.program, this.kernelParams(sources, dest), this.clContext.queue.process)
	}

	finish() {
		this.loader.releaseRefs()
		this.saver.releaseRefs()
	}
}

module.exports = { FusedV210Channel, default: FusedV210Channel }
