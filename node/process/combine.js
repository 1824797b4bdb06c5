'use strict'
// Combine: premultiplied "over" of N >= 2 layers (reference: src/process/combine.ts:70-104).
const { ProcessImpl } = require('./imageProcess')

class Combine extends ProcessImpl {
	constructor(numLayers, width, height) {
		const n = numLayers < 2 ? 2 : numLayers // combine is not used below 2 layers
		super(`combine-${numLayers}`, width, height, 'phaneron:combine', `combine_${n}`)
	}
	async init() {}
	async getKernelParams(params) {
		const kernelParams = { output: params.output }
		const inArray = params.inputs
		if (inArray.length < 2) throw new Error("Combine requires an 'inputs' array parameter with at least 2 OpenCL buffers")
		inArray.forEach((b, i) => { kernelParams[`l${i}In`] = b })
		return kernelParams
	}
	releaseRefs() {}
}

module.exports = { default: Combine }
