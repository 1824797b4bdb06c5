'use strict'
// v210 Reader / Writer geometry and parameter mapping (reference: src/process/v210.ts:198-339).
// The OpenCL text of the reference is replaced by a tag; the kernels live in libphaneron_hip.
const { PackImpl, Interlace } = require('./packer')

const KERNEL_TAG = 'phaneron:v210'
const pixelsPerWorkItem = 48

const getPitch = (width) => width + 47 - ((width - 1) % 48)
const getPitchBytes = (width) => (getPitch(width) * 8) / 3

// reference test pattern: one luma step per 6-pixel group, 64..940 wrapping, neutral chroma
function fillBuf(buf, width, height) {
	const pitchBytes = getPitchBytes(width)
	const Cb = 512
	const Cr = 512
	let Y = 64
	buf.fill(0)
	for (let y = 0; y < height; ++y) {
		const line = y * pitchBytes
		const groups = (width - (width % 6)) / 6
		for (let g = 0; g < groups; ++g) {
			const o = line + g * 16
			buf.writeUInt32LE(((Cr << 20) | (Y << 10) | Cb) >>> 0, o)
			buf.writeUInt32LE(((Y << 20) | (Cb << 10) | Y) >>> 0, o + 4)
			buf.writeUInt32LE(((Cb << 20) | (Y << 10) | Cr) >>> 0, o + 8)
			buf.writeUInt32LE(((Y << 20) | (Cr << 10) | Y) >>> 0, o + 12)
			Y = Y === 940 ? 64 : Y + 1
		}
		const remain = width % 6
		if (remain) {
			const o = line + groups * 16
			buf.writeUInt32LE(((Cr << 20) | (Y << 10) | Cb) >>> 0, o)
			if (remain === 2) buf.writeUInt32LE(Y, o + 4)
			else if (remain === 4) {
				buf.writeUInt32LE(((Y << 20) | (Cb << 10) | Y) >>> 0, o + 4)
				buf.writeUInt32LE(((Y << 10) | Cr) >>> 0, o + 8)
			}
		}
	}
}

function dumpBuf(buf, width, numLines) {
	const pitch = getPitchBytes(width)
	for (let l = 0; l < numLines; ++l) {
		const hex = (off) => buf.readUInt32LE(l * pitch + off).toString(16)
		console.log(`Line ${l}: ${hex(0)}, ${hex(4)}, ${hex(8)}, ${hex(12)} ... ${hex(128)}, ${hex(132)}, ${hex(136)}, ${hex(140)}`)
	}
}

class Reader extends PackImpl {
	constructor(width, height) {
		super('v210', width, height, KERNEL_TAG, 'read')
		this.numBits = 10
		this.lumaBlack = 64
		this.lumaWhite = 940
		this.chromaRange = 896
		this.isRGB = false
		this.numBytes = [getPitchBytes(width) * height]
		this.workItemsPerGroup = getPitch(width) / pixelsPerWorkItem
		this.globalWorkItems = this.workItemsPerGroup * height
	}
	getKernelParams(params) {
		const srcArray = params.sources
		if (srcArray.length !== 1) throw new Error(`Reader for ${this.name} requires sources parameter with 1 OpenCL buffer`)
		return {
			input: srcArray[0],
			output: params.dest,
			width: this.width,
			colMatrix: params.colMatrix,
			gammaLut: params.gammaLut,
			gamutMatrix: params.gamutMatrix
		}
	}
}

class Writer extends PackImpl {
	constructor(width, height, interlaced) {
		super('v210', width, height, KERNEL_TAG, 'write')
		this.interlaced = interlaced
		this.numBits = 10
		this.lumaBlack = 64
		this.lumaWhite = 940
		this.chromaRange = 896
		this.isRGB = false
		this.numBytes = [getPitchBytes(width) * height]
		this.workItemsPerGroup = getPitch(width) / pixelsPerWorkItem
		this.globalWorkItems = (this.workItemsPerGroup * height) / (interlaced ? 2 : 1)
	}
	getKernelParams(params) {
		const dstArray = params.dests
		if (dstArray.length !== 1) throw new Error(`Writer for ${this.name} requires dests parameter with 1 OpenCL buffer`)
		return {
			input: params.source,
			output: dstArray[0],
			width: this.width,
			interlace: this.interlaced ? params.interlace : Interlace.Progressive,
			colMatrix: params.colMatrix,
			gammaLut: params.gammaLut
		}
	}
}

module.exports = { Reader, Writer, fillBuf, dumpBuf, getPitch, getPitchBytes }
