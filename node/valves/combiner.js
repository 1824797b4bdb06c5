'use strict'
// Combiner video valves (reference: src/combiner.ts:32-360, video side).  Zips a Black source with
// the layers' video pipes; vidEndValve drops layers that have ended (CombineLayer.checkVideo :74-87),
// combineVidValve:
//   no layers   -> the black frame        one layer -> that frame
//   N >= 2      -> combine_N into a new consumer-sized buffer, keyed `${chanID} combine`
// Output frames are renumbered with the combiner's own counter (:211) and carry one reference per
// fork (route consumers, :255).
const { EventEmitter } = require('events')
const ImageProcess = require('../process/imageProcess').default
const Combine = require('../process/combine').default
const { Black } = require('./black')
const { isValue, isEnd, end } = require('./redio')

class CombineLayer {
	constructor(videoPipe, endEvent) {
		this.videoPipe = videoPipe
		this.endEvent = endEvent || new EventEmitter()
		this.videoState = 'start'
	}
	getVideoPipe() { return this.videoPipe }
	getEndEvent() { return this.endEvent }
	checkVideo(frame) {
		if (isValue(frame)) {
			if (this.videoState === 'start') this.videoState = 'run'
			return true
		}
		if (this.videoState === 'run') {
			this.videoState = 'end'
			this.endEvent.emit('end') // no audio side here
		}
		return false
	}
}

class Combiner {
	constructor(clContext, chanID, consumerFormat, clJobs) {
		this.clContext = clContext
		this.chanID = `${chanID} combine`
		this.consumerFormat = consumerFormat
		this.clJobs = clJobs
		this.lastNumVidLayers = 0
		this.vidCombiner = undefined
		this.videoPipe = undefined
		this.combineLayers = []
		this.vidLayerPipes = []
		this.vidTimestamp = 0
		this.numForks = 0
	}

	async initialise() {
		const black = new Black(this.clContext, this.consumerFormat, this.chanID)
		const blackPipe = await black.initialise()
		const { width, height } = this.consumerFormat

		const vidEndValve = async (frames) => {
			if (!isValue(frames)) return frames
			return frames.filter((f, i) => (i > 0 ? (this.combineLayers.length > i - 1 ? this.combineLayers[i - 1].checkVideo(f) : false) : true))
		}

		const combineVidValve = async (frames) => {
			let result = end
			if (isValue(frames) && isValue(frames[0])) {
				const layerFrames = frames.slice(1)
				const numLayers = layerFrames.length
				const timestamp = this.vidTimestamp++

				const numCombineLayers = numLayers < 2 ? 0 : numLayers
				if (numCombineLayers && this.lastNumVidLayers !== numCombineLayers) {
					await this.makeVidCombiner(numCombineLayers)
					this.lastNumVidLayers = numCombineLayers
				}

				if (numLayers === 0) {
					frames[0].timestamp = timestamp
					frames[0].addRef()
					result = frames[0]
				} else if (numLayers === 1) {
					if (!isEnd(frames[1])) {
						frames[1].timestamp = timestamp
						frames[1].addRef()
					}
					result = frames[1]
				} else if (layerFrames.every((f) => isValue(f))) {
					const combineDest = await this.clContext.createBuffer(width * height * 4 * 4, 'readwrite', 'coarse', { width, height }, `${this.chanID} ${timestamp}`)
					combineDest.timestamp = timestamp
					await this.vidCombiner.run({ inputs: layerFrames, output: combineDest }, { source: this.chanID, timestamp }, () => {})
					await this.clJobs.runQueue({ source: this.chanID, timestamp })
					result = combineDest
				}

				if (isValue(result)) for (let d = 1; d < this.numForks; ++d) result.addRef()
				frames.forEach((f) => { if (isValue(f)) f.release() })
			} else if (this.vidCombiner) {
				this.clJobs.clearQueue(this.chanID)
				black.release()
				this.vidCombiner = undefined
			}
			return result
		}

		this._black = black
		this.videoPipe = blackPipe.zipEach(this.vidLayerPipes).valve(vidEndValve).valve(combineVidValve)
	}

	async makeVidCombiner(numLayers) {
		this.vidCombiner = new ImageProcess(this.clContext, new Combine(numLayers, this.consumerFormat.width, this.consumerFormat.height), this.clJobs)
		await this.vidCombiner.init()
	}

	getLayers() { return this.combineLayers }

	updateLayers(layers) {
		this.combineLayers = layers.slice(0)
		this.vidLayerPipes.splice(0)
		layers.forEach((l) => this.vidLayerPipes.push(l.getVideoPipe()))
	}

	getVideoPipe() { return this.videoPipe }

	// a route consumer's tap (RouteSource.getSourcePipes, :332-359)
	getSourcePipes() {
		if (!this.videoPipe) throw new Error('Combiner failed to find source pipes for route')
		this.numForks++
		const vidFork = this.videoPipe.fork()
		return {
			video: vidFork,
			format: this.consumerFormat,
			release: () => {
				try {
					this.videoPipe.unfork(vidFork)
					this.numForks--
				} catch (err) { /* as the reference: ignore */ }
			}
		}
	}

	release() { if (this._black) this._black.release() }
}

module.exports = { Combiner, CombineLayer }
