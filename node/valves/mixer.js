'use strict'
// Mixer video valve (reference: src/producer/mixer.ts:104-269, video side): every source frame goes
// through `transform` into a consumer-sized RGBA buffer, with the fill / anchor / rotation parameters
// mapped exactly as mixVidValve does (:209-223).  Audio (beamcoder filter graphs) is out of scope.
const ImageProcess = require('../process/imageProcess').default
const Transform = require('../process/transform').default
const { isValue, nil } = require('./redio')

const MixerDefaults = '{ "anchor": { "x": 0, "y": 0 }, "rotation": 0, "fill": { "xOffset": 0, "yOffset": 0, "xScale": 1, "yScale": 1 }, "volume": 1 }'

class Mixer {
	constructor(clContext, consumerFormat, clJobs) {
		this.clContext = clContext
		this.consumerFormat = consumerFormat
		this.clJobs = clJobs
		this.transform = new ImageProcess(clContext, new Transform(clContext, consumerFormat.width, consumerFormat.height), clJobs)
		this.mixParams = JSON.parse(MixerDefaults)
		this.running = true
		this.vidDone = false
		this.mixVideo = null
	}

	async init(sourceID, srcVideo) {
		await this.transform.init()
		const { width, height } = this.consumerFormat
		const numBytesRGBA = width * height * 4 * 4

		const mixVidValve = async (frame) => {
			if (isValue(frame)) {
				if (!this.running) {
					frame.release()
					return nil
				}
				const timestamp = frame.timestamp
				const xfDest = await this.clContext.createBuffer(numBytesRGBA, 'readwrite', 'coarse', { width, height }, `mixer ${sourceID} ${timestamp}`)
				xfDest.timestamp = timestamp
				await this.transform.run(
					{
						input: frame,
						flipH: false,
						flipV: false,
						anchorX: this.mixParams.anchor.x - 0.5,
						anchorY: this.mixParams.anchor.y - 0.5,
						scaleX: this.mixParams.fill.xScale,
						scaleY: this.mixParams.fill.yScale,
						rotate: -this.mixParams.rotation / 360.0,
						offsetX: -this.mixParams.fill.xOffset,
						offsetY: -this.mixParams.fill.yOffset,
						output: xfDest
					},
					{ source: sourceID, timestamp },
					() => frame.release()
				)
				await this.clJobs.runQueue({ source: sourceID, timestamp })
				return xfDest
			}
			this.clJobs.clearQueue(sourceID)
			if (this.transform) this.transform.finish()
			this.transform = null
			this.vidDone = true
			this.running = false // no audio side here: video done == all done
			return frame
		}

		this.mixVideo = srcVideo.valve(mixVidValve)
	}

	release() { this.running = false }
	setMixParams(mixParams) { this.mixParams = mixParams }
	getMixParams() { return this.mixParams }
	getMixVideo() { return this.mixVideo }
}

module.exports = { Mixer, MixerDefaults }
