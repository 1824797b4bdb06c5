'use strict'
// Packed-format operator base classes (reference: src/process/packer.ts).
const Interlace = Object.freeze({ Progressive: 0, TopField: 1, BottomField: 3 })

class PackImpl {
	constructor(name, width, height, kernel, programName) {
		this.name = name
		this.width = width
		this.height = height
		this.interlaced = false
		this.kernel = kernel // "phaneron:<format>" tag - the HIP kernels are precompiled
		this.programName = programName
		this.numBits = 10
		this.lumaBlack = 64
		this.lumaWhite = 940
		this.chromaRange = 896
		this.isRGB = true
		this.numBytes = [0]
		this.globalWorkItems = 0
		this.workItemsPerGroup = 0
	}
	getName() { return this.name }
	getWidth() { return this.width }
	getHeight() { return this.height }
	getNumBytes() { return this.numBytes }
	getNumBytesRGBA() { return this.width * this.height * 4 * 4 }
	getIsRGB() { return this.isRGB }
	getTotalBytes() { return this.numBytes.reduce((acc, n) => acc + n, 0) }
	getGlobalWorkItems() { return this.globalWorkItems }
	getWorkItemsPerGroup() { return this.workItemsPerGroup }
	getKernelParams() { throw new Error('getKernelParams is abstract') }
}

class Packer {
	constructor(clContext, packImpl, clJobs) {
		this.clContext = clContext
		this.packImpl = packImpl
		this.clJobs = clJobs
		this.program = null
	}
	async init() {
		this.program = await this.clContext.createProgram(this.packImpl.kernel, {
			name: this.packImpl.programName,
			globalWorkItems: this.packImpl.getGlobalWorkItems(),
			workItemsPerGroup: this.packImpl.getWorkItemsPerGroup()
		})
	}
}

module.exports = { Interlace, PackImpl, default: Packer }
