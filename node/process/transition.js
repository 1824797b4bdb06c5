'use strict'
// Transition: dissolve (scalar mix) or wipe (mask image) between two inputs
// (reference: src/process/transition.ts:83-116).
const { ProcessImpl } = require('./imageProcess')

class Transition extends ProcessImpl {
	constructor(type, width, height) {
		if (!['dissolve', 'wipe'].includes(type))
			throw new Error(`Transition requires a 'type' parameter that is either 'dissolve' or 'wipe' - found '${type}'`)
		super(type, width, height, 'phaneron:transition', `transition_${type}`)
	}
	async init() {}
	async getKernelParams(params) {
		const kernelParams = { output: params.output }
		const inArray = params.inputs
		if (inArray.length !== 2) throw new Error(`Transition requires an 'inputs' array parameter with 2 OpenCL buffers`)
		inArray.forEach((b, i) => { kernelParams[`input${i}`] = b })
		if (this.name === 'dissolve') kernelParams.mix = params.mix
		else if (params.mask) kernelParams.maskIn = params.mask
		else throw new Error(`Transition '${this.name}' expected a 'mask' buffer which wasn't found`)
		return kernelParams
	}
	releaseRefs() {}
}

module.exports = { default: Transition }
