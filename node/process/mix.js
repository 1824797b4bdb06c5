'use strict'
// Mix: scalar cross-fade of two images (reference: src/process/mix.ts:48-69).
const { ProcessImpl } = require('./imageProcess')

class Mix extends ProcessImpl {
	constructor(width, height) {
		super('mixer', width, height, 'phaneron:mixer', 'mixer')
	}
	async init() {}
	async getKernelParams(params) {
		return { input0: params.input0, input1: params.input1, mix: params.mix, output: params.output }
	}
	releaseRefs() {}
}

module.exports = { default: Mix }
