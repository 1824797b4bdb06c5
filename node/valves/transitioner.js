'use strict'
// Transitioner video valve (reference: src/transitioner.ts:32-297, video side).  Zips a Black
// source with 0..3 source pipes (current, incoming, optional wipe mask):
//   no sources           -> the black frame
//   one source           -> that frame (addRef)
//   two / three sources  -> transition_dissolve with mix = 1 - cur/numFrames (numFrames = requested
//                           length - 1, :269) / transition_wipe with the third frame as mask; a source
//                           that has ended falls back to the other (:186-189)
// keyed `${layerID} transition` on the job queue; every input frame is released after use (:193).
const ImageProcess = require('../process/imageProcess').default
const Transition = require('../process/transition').default
const { Black } = require('./black')
const { isValue, isEnd, end } = require('./redio')

class Transitioner {
	constructor(clContext, layerID, consumerFormat, clJobs, layerUpdate) {
		this.clContext = clContext
		this.layerID = `${layerID} transition`
		this.consumerFormat = consumerFormat
		this.clJobs = clJobs
		this.layerUpdate = layerUpdate || (() => {})
		this.black = new Black(clContext, consumerFormat, this.layerID)
		this.vidType = 'cut'
		this.numFrames = 0
		this.curFrame = 0
		this.nextType = 'cut'
		this.nextNumFrames = 0
		this.vidTransition = null
		this.videoPipe = undefined
		this.vidSourcePipes = []
		this.updating = true
	}

	async initialise() {
		const blackPipe = await this.black.initialise()
		const { width, height } = this.consumerFormat

		const transitionVidValve = async (frames) => {
			let transitionResult = end
			if (isValue(frames) && isValue(frames[0])) {
				const srcFrames = frames.slice(1)
				const numSrcs = srcFrames.length
				this.layerUpdate(srcFrames.map((f) => (isValue(f) ? f.timestamp : this.updating ? 0 : -1)))

				if (numSrcs === 0) {
					transitionResult = frames[0]
					transitionResult.addRef()
				} else {
					if (this.vidType !== this.nextType && numSrcs === this.vidSourcePipes.length) {
						this.vidType = this.nextType
						this.numFrames = this.nextNumFrames
						this.curFrame = 0
						await this.makeVidTransition()
					}

					if (srcFrames.every((f) => isValue(f))) {
						this.updating = false
						if (numSrcs === 1) {
							transitionResult = srcFrames[0]
							transitionResult.addRef()
						} else {
							const timestamp = srcFrames[1].timestamp
							const transitionDest = await this.clContext.createBuffer(width * height * 4 * 4, 'readwrite', 'coarse', { width, height }, `${this.layerID} ${timestamp}`)
							transitionDest.timestamp = timestamp
							const params = { inputs: srcFrames.slice(0, 2), output: transitionDest }
							if (numSrcs === 2) params.mix = this.numFrames > 0 ? 1.0 - this.curFrame / this.numFrames : 0.0
							else params.mask = srcFrames[2]
							this.curFrame++
							if (this.vidTransition) await this.vidTransition.run(params, { source: this.layerID, timestamp }, () => {})
							await this.clJobs.runQueue({ source: this.layerID, timestamp })
							transitionResult = transitionDest
						}
					} else {
						transitionResult = numSrcs > 1 && isValue(srcFrames[1]) ? srcFrames[1] : srcFrames[0]
						if (isValue(transitionResult)) transitionResult.addRef()
					}

					if (isEnd(transitionResult)) {
						transitionResult = frames[0]
						if (isValue(transitionResult)) transitionResult.addRef()
					}
				}
				frames.forEach((f) => { if (isValue(f)) f.release() })
			} else {
				this.layerUpdate([])
			}
			return transitionResult
		}

		this.videoPipe = blackPipe.zipEach(this.vidSourcePipes).valve(transitionVidValve)
	}

	async makeVidTransition() {
		if (this.vidType === 'cut') this.vidTransition = null
		else {
			this.vidTransition = new ImageProcess(this.clContext, new Transition(this.vidType, this.consumerFormat.width, this.consumerFormat.height), this.clJobs)
			await this.vidTransition.init()
		}
	}

	// type: 'cut' | 'dissolve' | 'wipe'; numFrames: requested length of the transition in frames
	update(type, numFrames, videoSrcPipes) {
		this.nextType = type
		this.nextNumFrames = numFrames > 0 ? numFrames - 1 : 0
		this.updating = true
		this.vidSourcePipes.splice(0)
		videoSrcPipes.forEach((p) => this.vidSourcePipes.push(p))
	}

	getVideoPipe() { return this.videoPipe }

	release() {
		this.black.release()
		if (this.vidTransition) this.vidTransition.finish()
		this.vidTransition = null
	}
}

module.exports = { Transitioner }
