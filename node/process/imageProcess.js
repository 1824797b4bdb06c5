'use strict'
// Image operator base classes (reference: src/process/imageProcess.ts).
class ProcessImpl {
	constructor(name, width, height, kernel, programName) {
		this.name = name
		this.width = width
		this.height = height
		this.kernel = kernel
		this.programName = programName
		this.globalWorkItems = 0
	}
	getName() { return this.name }
	getNumBytesRGBA() { return this.width * this.height * 4 * 4 }
	getGlobalWorkItems() { return Uint32Array.from([this.width, this.height]) }
}

class ImageProcess {
	constructor(clContext, processImpl, clJobs) {
		this.clContext = clContext
		this.processImpl = processImpl
		this.clJobs = clJobs
		this.program = null
	}
	async init() {
		this.program = await this.clContext.createProgram(this.processImpl.kernel, {
			name: this.processImpl.programName,
			globalWorkItems: this.processImpl.getGlobalWorkItems()
		})
		return this.processImpl.init()
	}
	async run(params, id, cb) {
		if (this.program == null) throw new Error('Loader.run failed with no program available')
		const kernelParams = await this.processImpl.getKernelParams(params)
		this.clJobs.add(id, this.processImpl.getName(), this.program, kernelParams, () => {
			this.processImpl.releaseRefs()
			cb()
		})
	}
	finish() {
		this.processImpl.releaseRefs()
	}
}

module.exports = { ProcessImpl, default: ImageProcess }
