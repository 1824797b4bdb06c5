'use strict'
// Black: the transparent-black RGBA frame source that heads every zip in the Transitioner and the
// Combiner (reference: src/blackSilence.ts:96-158).  One device buffer, handed out with an extra
// reference per pull while running; released and ended after release().
const { redio, end } = require('./redio')

class Black {
	constructor(clContext, consumerFormat, id) {
		this.clContext = clContext
		this.consumerFormat = consumerFormat
		this.id = id
		this.running = true
	}

	async initialise() {
		const { width, height } = this.consumerFormat
		const numBytesRGBA = width * height * 4 * 4
		let black = await this.clContext.createBuffer(numBytesRGBA, 'readwrite', 'coarse', { width, height }, `black-${this.id}`)
		await black.hostAccess('writeonly')
		black.fill(0) // r = g = b = a = 0.0f
		return redio(() => {
			if (this.running) {
				black.addRef()
				return black
			}
			if (black) {
				black.release()
				black = null
			}
			return end
		})
	}

	release() {
		this.running = false
	}
}

module.exports = { Black }
