'use strict'
// Loader / Saver: own the LUT and matrix device buffers of a packed-format reader / writer and
// inject them into every queued job (reference: src/process/loadSave.ts).
const Packer = require('./packer').default
const cm = require('./colourMaths')

async function upload(clContext, array, svm, owner) {
	const buf = await clContext.createBuffer(array.byteLength, 'readonly', svm, undefined, owner)
	await buf.hostAccess('writeonly')
	Buffer.from(array.buffer).copy(buf)
	return buf
}

class Loader extends Packer {
	constructor(clContext, colSpec, outColSpec, packImpl, clJobs) {
		super(clContext, packImpl, clJobs)
		this.gammaArray = cm.gamma2linearLUT(colSpec)
		this.colMatrixArray = null
		if (!packImpl.getIsRGB())
			this.colMatrixArray = cm.matrixFlatten(
				cm.ycbcr2rgbMatrix(colSpec, packImpl.numBits, packImpl.lumaBlack, packImpl.lumaWhite, packImpl.chromaRange))
		this.gamutMatrixArray = cm.matrixFlatten(cm.rgb2rgbMatrix(colSpec, outColSpec))
		this.gammaLut = null
		this.colMatrix = null
		this.gamutMatrix = null
	}

	async init() {
		await super.init()
		this.gammaLut = await upload(this.clContext, this.gammaArray, 'coarse', 'loader gammaLut')
		if (this.colMatrixArray) this.colMatrix = await upload(this.clContext, this.colMatrixArray, 'none', 'loader colMatrix')
		this.gamutMatrix = await upload(this.clContext, this.gamutMatrixArray, 'none', 'loader gamutMatrix')
	}

	addRefs() {
		for (const b of [this.gammaLut, this.colMatrix, this.gamutMatrix]) if (b) b.addRef()
	}
	releaseRefs() {
		for (const b of [this.gammaLut, this.colMatrix, this.gamutMatrix]) if (b) b.release()
	}

	run(params, id, cb) {
		if (this.program === null) throw new Error('Loader.run failed with no program available')
		this.addRefs()
		const kernelParams = this.packImpl.getKernelParams(params)
		kernelParams.gammaLut = this.gammaLut
		kernelParams.gamutMatrix = this.gamutMatrix
		if (this.colMatrix) kernelParams.colMatrix = this.colMatrix
		this.clJobs.add(id, this.packImpl.getName(), this.program, kernelParams, () => {
			this.releaseRefs()
			cb()
		})
	}
}

class Saver extends Packer {
	constructor(clContext, colSpec, packImpl, clJobs) {
		super(clContext, packImpl, clJobs)
		this.gammaArray = cm.linear2gammaLUT(colSpec)
		this.colMatrixArray = null
		if (!packImpl.getIsRGB())
			this.colMatrixArray = cm.matrixFlatten(
				cm.rgb2ycbcrMatrix(colSpec, packImpl.numBits, packImpl.lumaBlack, packImpl.lumaWhite, packImpl.chromaRange))
		this.gammaLut = null
		this.colMatrix = null
	}

	async init() {
		await super.init()
		this.gammaLut = await upload(this.clContext, this.gammaArray, 'coarse', 'saver gammaLut')
		if (this.colMatrixArray) this.colMatrix = await upload(this.clContext, this.colMatrixArray, 'none', 'saver colMatrix')
	}

	addRefs() {
		for (const b of [this.gammaLut, this.colMatrix]) if (b) b.addRef()
	}
	releaseRefs() {
		for (const b of [this.gammaLut, this.colMatrix]) if (b) b.release()
	}

	run(params, id, cb) {
		if (this.program === null) throw new Error('Saver.run failed with no program available')
		this.addRefs()
		const kernelParams = this.packImpl.getKernelParams(params)
		kernelParams.gammaLut = this.gammaLut
		if (this.colMatrix) kernelParams.colMatrix = this.colMatrix
		this.clJobs.add(id, this.packImpl.getName(), this.program, kernelParams, () => {
			this.releaseRefs()
			cb()
		})
	}
}

module.exports = { Loader, Saver }
