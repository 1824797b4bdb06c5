'use strict'
// ToRGBA / FromRGBA: buffer factories, host<->device staging and job enqueue for a packed
// format (reference: src/process/io.ts).
const { Loader, Saver } = require('./loadSave')

class ToRGBA {
	constructor(clContext, colSpecRead, colSpecWrite, readImpl, clJobs) {
		this.clContext = clContext
		this.loader = new Loader(clContext, colSpecRead, colSpecWrite, readImpl, clJobs)
		this.numBytes = readImpl.getNumBytes()
		this.numBytesRGBA = readImpl.getNumBytesRGBA()
		this.totalBytes = readImpl.getTotalBytes()
	}
	async init() { await this.loader.init() }
	getNumBytes() { return this.numBytes }
	getNumBytesRGBA() { return this.numBytesRGBA }
	getTotalBytes() { return this.totalBytes }

	async createSources(srcID) {
		return Promise.all(this.numBytes.map((bytes) =>
			this.clContext.createBuffer(bytes, 'readonly', 'coarse', undefined, `ToRGBA src ${srcID}`)))
	}
	async createDest(imageDims, srcID) {
		return this.clContext.createBuffer(this.numBytesRGBA, 'readonly', 'coarse', imageDims, `ToRGBA ${srcID}`)
	}
	async loadFrame(input, sources, clQueue) {
		const inputs = Array.isArray(input) ? input : [input]
		if (sources.length !== inputs.length)
			throw new Error(`Expected buffer array of ${sources.length} sources, found ${inputs.length}`)
		for (let i = 0; i < inputs.length; ++i) {
			await sources[i].hostAccess('writeonly', clQueue ? clQueue : 0, inputs[i].slice(0, this.numBytes[i]))
			await sources[i].hostAccess('none', clQueue ? clQueue : 0)
		}
	}
	processFrame(sourceID, sources, dest) {
		return this.loader.run({ sources, dest }, { source: sourceID, timestamp: sources[0].timestamp },
			() => sources.forEach((s) => s.release()))
	}
	finish() { this.loader.releaseRefs() }
}

class FromRGBA {
	constructor(clContext, colSpecRead, writeImpl, clJobs) {
		this.clContext = clContext
		this.saver = new Saver(clContext, colSpecRead, writeImpl, clJobs)
		this.numBytes = writeImpl.getNumBytes()
		this.numBytesRGBA = writeImpl.getNumBytesRGBA()
		this.totalBytes = writeImpl.getTotalBytes()
	}
	async init() { await this.saver.init() }
	getNumBytes() { return this.numBytes }
	getNumBytesRGBA() { return this.numBytesRGBA }
	getTotalBytes() { return this.totalBytes }

	async createDests(sourceID) {
		return Promise.all(this.numBytes.map((bytes) =>
			this.clContext.createBuffer(bytes, 'writeonly', 'coarse', undefined, `FromRGBA ${sourceID}`)))
	}
	processFrame(sourceID, source, dests, interlace) {
		this.saver.run({ source, dests, interlace }, { source: sourceID, timestamp: source.timestamp }, () => source.release())
	}
	async saveFrame(output, clQueue) {
		const outputs = Array.isArray(output) ? output : [output]
		for (const o of outputs) await o.hostAccess('readonly', clQueue ? clQueue : 0)
	}
	finish() { this.saver.releaseRefs() }
}

module.exports = { ToRGBA, FromRGBA }
