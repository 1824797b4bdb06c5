'use strict'
// Wipe: hard vertical edge between two images (reference: src/process/wipe.ts:49-70).
const { ProcessImpl } = require('./imageProcess')

class Wipe extends ProcessImpl {
	constructor(width, height) {
		super('wipe', width, height, 'phaneron:wipe', 'wipe')
	}
	async init() {}
	async getKernelParams(params) {
		return { input0: params.input0, input1: params.input1, wipe: params.wipe, output: params.output }
	}
	releaseRefs() {}
}

module.exports = { default: Wipe }
