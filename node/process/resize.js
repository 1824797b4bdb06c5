'use strict'
// Resize: scale / offset / flip variant of the bilinear sampler (reference: src/process/resize.ts).
const { ProcessImpl } = require('./imageProcess')

class Resize extends ProcessImpl {
	constructor(clContext, width, height) {
		super('resize', width, height, 'phaneron:resize', 'resize')
		this.clContext = clContext
		this.flipH = false
		this.flipV = false
		this.flipArr = Float32Array.from([0.0, 1.0, 0.0, 1.0])
		this.flipArrBytes = this.flipArr.length * this.flipArr.BYTES_PER_ELEMENT
		this.flipVals = null
	}

	async updateFlip(flipH, flipV, clQueue) {
		if (this.flipVals === null) throw new Error('Resize.updateFlip failed with no program available')
		this.flipH = flipH
		this.flipV = flipV
		this.flipArr = Float32Array.from([flipH ? 1.0 : 0.0, flipH ? -1.0 : 1.0, flipV ? 1.0 : 0.0, flipV ? -1.0 : 1.0])
		await this.flipVals.hostAccess('writeonly', clQueue, Buffer.from(this.flipArr.buffer))
		return this.flipVals.hostAccess('none', clQueue)
	}

	async init() {
		this.flipVals = await this.clContext.createBuffer(this.flipArrBytes, 'readonly', 'coarse', undefined, 'flipVals')
		return this.updateFlip(false, false, this.clContext.queue.load)
	}

	async getKernelParams(params) {
		const { flipH, flipV, scale, offsetX, offsetY } = params
		if (!(this.flipH === flipH && this.flipV === flipV)) await this.updateFlip(flipH, flipV, this.clContext.queue.load)
		// the reference throws bare strings here (resize.ts:116-122)
		if (scale && !(scale > 0.0)) throw 'resize scale factor must be greater than zero'
		if (offsetX && !(offsetX >= -1.0 && offsetX <= 1.0)) throw 'resize offsetX must be between -1.0 and +1.0'
		if (offsetY && !(offsetY >= -1.0 && offsetY <= 1.0)) throw 'resize offsetX must be between -1.0 and +1.0'
		if (this.flipVals) this.flipVals.addRef()
		return {
			input: params.input,
			scale: params.scale || 1.0,
			offsetX: params.offsetX || 0.0,
			offsetY: params.offsetY || 0.0,
			flip: this.flipVals,
			output: params.output
		}
	}

	releaseRefs() {
		if (this.flipVals) this.flipVals.release()
	}
}

module.exports = { default: Resize }
