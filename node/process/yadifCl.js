'use strict'
// yadif kernel binding (reference: src/process/yadifCl.ts:170-194)
const { ProcessImpl } = require('./imageProcess')

class YadifCL extends ProcessImpl {
	constructor(width, height) {
		super('yadif', width, height, 'phaneron:yadif', 'yadif')
	}
	async init() {}
	async getKernelParams(params) {
		return {
			prev: params.prev,
			cur: params.cur,
			next: params.next,
			parity: params.parity,
			tff: params.tff ? 1 : 0,
			skipSpatial: params.skipSpatial ? 1 : 0,
			output: params.output
		}
	}
	releaseRefs() {}
}

module.exports = { default: YadifCL }
