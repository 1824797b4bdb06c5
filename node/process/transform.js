'use strict'
// Transform: 3x3 affine (anchor, scale, rotate, translate, anchor^-1, project) uploaded when the
// parameters change, bilinear sampling in the kernel (reference: src/process/transform.ts).
const { ProcessImpl } = require('./imageProcess')
const { matrixFlatten, matrixMultiply } = require('./colourMaths')

const PARAM_KEYS = ['flipH', 'flipV', 'anchorX', 'anchorY', 'scaleX', 'scaleY', 'offsetX', 'offsetY', 'rotate']
const m3 = (a, b, c, d, e, f) => [Float32Array.from([a, b, c]), Float32Array.from([d, e, f]), Float32Array.from([0.0, 0.0, 1.0])]

class Transform extends ProcessImpl {
	constructor(clContext, width, height) {
		super('transform', width, height, 'phaneron:transform', 'transform')
		this.clContext = clContext
		this.transformMatrix = m3(1.0, 0.0, 0.0, 0.0, 1.0, 0.0)
		this.transformArray = matrixFlatten(this.transformMatrix)
		this.matrixBuffer = null
		this.curParams = null
	}

	async updateMatrix(clQueue) {
		if (!this.matrixBuffer) throw new Error('Transform needs to be initialised')
		this.transformArray = matrixFlatten(this.transformMatrix)
		await this.matrixBuffer.hostAccess('writeonly', clQueue, Buffer.from(this.transformArray.buffer))
		return this.matrixBuffer.hostAccess('none', clQueue)
	}

	async init() {
		this.matrixBuffer = await this.clContext.createBuffer(this.transformArray.byteLength, 'readonly', 'coarse', undefined, 'transformMatrix')
		await this.updateMatrix(this.clContext.queue.load)
		return this.clContext.waitFinish(this.clContext.queue.load)
	}

	paramsUnchanged(params) {
		return this.curParams !== null && PARAM_KEYS.every((k) => params[k] === this.curParams[k])
	}

	async getKernelParams(params) {
		if (!this.paramsUnchanged(params)) {
			const aspect = this.width / this.height
			const flipX = params.flipH || false ? -1.0 : 1.0
			const flipY = params.flipV || false ? -1.0 : 1.0
			const anchorX = params.anchorX || 0.0
			const anchorY = params.anchorY || 0.0
			const scaleX = (params.scaleX || 1.0) * flipX
			const scaleY = (params.scaleY || 1.0) * flipY
			const offsetX = params.offsetX || 0.0
			const offsetY = params.offsetY || 0.0
			const rotate = (params.rotate || 0.0) * 2 * Math.PI
			const chain = [
				m3(1.0, 0.0, anchorX, 0.0, 1.0, anchorY),
				m3(1.0 / (scaleX * aspect), 0.0, 0.0, 0.0, 1.0 / scaleY, 0.0),
				m3(Math.cos(rotate), -Math.sin(rotate), 0.0, Math.sin(rotate), Math.cos(rotate), 0.0),
				m3(1.0, 0.0, offsetX * aspect, 0.0, 1.0, offsetY),
				m3(1.0, 0.0, -anchorX * aspect, 0.0, 1.0, -anchorY),
				m3(aspect, 0.0, 0.0, 0.0, 1.0, 0.0)
			]
			this.transformMatrix = chain.reduce((acc, m) => matrixMultiply(acc, m))
			await this.updateMatrix(this.clContext.queue.load)
			await this.clContext.waitFinish(this.clContext.queue.load)
		}
		this.curParams = params
		if (this.matrixBuffer) this.matrixBuffer.addRef()
		return { input: params.input, transformMatrix: this.matrixBuffer, output: params.output }
	}

	releaseRefs() {
		if (this.matrixBuffer) this.matrixBuffer.release()
	}
}

module.exports = { default: Transform }
