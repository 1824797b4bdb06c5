'use strict'
// Reader / Writer geometry, parameter mapping and test patterns of the planar / semi-planar /
// RGBA pack formats (reference: src/process/{yuv422p10,yuv422p8,yuv420p,nv12,rgba8,bgra8}.ts).
// One table-driven factory instead of six near-identical modules; the per-format files
// (yuv422p10.js ...) re-export what the reference modules export.
const { PackImpl, Interlace } = require('./packer')

const pixelsPerWorkItem = 64
const pitch8 = (width) => width + 7 - ((width - 1) % 8)

const FORMATS = {
	yuv422p10: {
		name: 'yuv422p10le', bits: 10, black: 64, white: 940, chroma: 896, bytesPerSample: 2,
		planes: (luma) => [luma, luma / 2, luma / 2], vertical: 1,
		inputs: ['inputY', 'inputU', 'inputV'], outputs: ['outputY', 'outputU', 'outputV'], quote: false
	},
	yuv422p8: {
		name: 'yuv422p8', bits: 8, black: 16, white: 235, chroma: 224, bytesPerSample: 1,
		planes: (luma) => [luma, luma / 2, luma / 2], vertical: 1,
		inputs: ['inputY', 'inputU', 'inputV'], outputs: ['outputY', 'outputU', 'outputV'], quote: true
	},
	yuv420p: {
		name: 'yuv420p', bits: 8, black: 16, white: 235, chroma: 224, bytesPerSample: 1,
		planes: (luma) => [luma, luma / 4, luma / 4], vertical: 2,
		inputs: ['inputY', 'inputU', 'inputV'], outputs: ['outputY', 'outputU', 'outputV'], quote: true
	},
	nv12: {
		name: 'nv12', bits: 8, black: 16, white: 235, chroma: 224, bytesPerSample: 1,
		planes: (luma) => [luma, luma / 2], vertical: 2,
		inputs: ['inputY', 'inputC'], outputs: ['outputY', 'outputC'], quote: true
	},
	rgba8: { name: 'rgba8', rgb: true, bits: 8, inputs: ['input'], outputs: ['output'], quote: true },
	bgra8: { name: 'bgra8', rgb: true, bits: 8, black: 16, white: 235, chroma: 224, inputs: ['input'], outputs: ['output'], quote: true }
}

function geometry(f, width, height) {
	if (f.rgb) return { numBytes: [width * 4 * height], wipg: width / pixelsPerWorkItem, lines: height }
	const pitch = pitch8(width)
	const luma = pitch * f.bytesPerSample * height
	return { numBytes: f.planes(luma), wipg: Math.ceil(pitch / pixelsPerWorkItem), lines: height / f.vertical }
}

function plural(n) {
	return `${n} OpenCL buffer${n === 1 ? '' : 's'}`
}

function makeFormat(key) {
	const f = FORMATS[key]
	const tag = `phaneron:${key}`

	function configure(impl, width, height) {
		impl.numBits = f.bits
		if (f.black !== undefined) impl.lumaBlack = f.black
		if (f.white !== undefined) impl.lumaWhite = f.white
		if (f.chroma !== undefined) impl.chromaRange = f.chroma
		if (!f.rgb) impl.isRGB = false
		const g = geometry(f, width, height)
		impl.numBytes = g.numBytes
		impl.workItemsPerGroup = g.wipg
		return g
	}

	class Reader extends PackImpl {
		constructor(width, height) {
			super(f.name, width, height, tag, 'read')
			const g = configure(this, width, height)
			this.globalWorkItems = g.wipg * g.lines
		}
		getKernelParams(params) {
			const srcArray = params.sources
			const word = f.quote ? "'sources'" : 'sources'
			if (srcArray.length !== f.inputs.length)
				throw new Error(`Reader for ${this.name} requires ${word} parameter with ${plural(f.inputs.length)}`)
			const kp = {}
			f.inputs.forEach((n, i) => { kp[n] = srcArray[i] })
			kp.output = params.dest
			kp.width = this.width
			if (!f.rgb) kp.colMatrix = params.colMatrix
			kp.gammaLut = params.gammaLut
			kp.gamutMatrix = params.gamutMatrix
			return kp
		}
	}

	class Writer extends PackImpl {
		constructor(width, height, interlaced) {
			super(f.name, width, height, tag, 'write')
			this.interlaced = interlaced
			const g = configure(this, width, height)
			// 4:2:0 writers always run one group per line pair; the others halve when interlaced
			this.globalWorkItems = f.vertical === 2 ? g.wipg * g.lines : (g.wipg * height) / (interlaced ? 2 : 1)
		}
		getKernelParams(params) {
			const dstArray = params.dests
			const word = f.quote ? "'dests'" : 'dests'
			if (dstArray.length !== f.outputs.length)
				throw new Error(`Writer for ${this.name} requires ${word} parameter with ${plural(f.outputs.length)}`)
			const kp = { input: params.source }
			f.outputs.forEach((n, i) => { kp[n] = dstArray[i] })
			kp.width = this.width
			kp.interlace = this.interlaced ? params.interlace : Interlace.Progressive
			if (!f.rgb) kp.colMatrix = params.colMatrix
			kp.gammaLut = params.gammaLut
			return kp
		}
	}

	// the reference's deterministic test patterns (fillBuf of each module)
	function fillBuf(buf, width, height) {
		if (f.rgb) {
			const px = key === 'rgba8' ? [16, 32, 64, 255] : [16, 16, 16, 255]
			buf.fill(0)
			for (let i = 0; i < width * height; ++i) for (let c = 0; c < 4; ++c) buf[4 * i + c] = px[c]
			return
		}
		const pitch = pitch8(width)
		const wide = f.bytesPerSample === 2
		const put = wide ? (v, o) => buf.writeUInt16LE(v, o) : (v, o) => buf.writeUInt8(v, o)
		const lumaPitchBytes = pitch * f.bytesPerSample
		const lumaBytes = lumaPitchBytes * height
		if (wide) {
			for (let o = 0; o < lumaBytes; o += 2) buf.writeUInt16LE(64, o)
			for (let o = lumaBytes; o < 2 * lumaBytes; o += 2) buf.writeUInt16LE(512, o)
		} else {
			buf.fill(16, 0)
			buf.fill(128, lumaBytes)
		}
		if (f.vertical === 1) {
			const chromaPitchBytes = lumaPitchBytes / 2
			const lo = wide ? 64 : 16
			const top = wide ? 938 : 234
			let Y = lo
			for (let y = 0; y < height; ++y) {
				for (let x = 0, xl = 0; x < width; x += 2, xl += 2 * f.bytesPerSample) {
					put(Y, y * lumaPitchBytes + xl)
					put(Y + 1, y * lumaPitchBytes + xl + f.bytesPerSample)
					Y = Y === top ? lo : Y + 2
				}
			}
			void chromaPitchBytes // chroma stays at the neutral fill value
			return
		}
		let Y0 = 16
		let Y1 = 234
		for (let y = 0; y < height; y += 2) {
			for (let x = 0; x < width; x += 2) {
				buf.writeUInt8(Y0, y * lumaPitchBytes + x)
				buf.writeUInt8(Y0 + 1, y * lumaPitchBytes + x + 1)
				buf.writeUInt8(Y1 + 1, (y + 1) * lumaPitchBytes + x)
				buf.writeUInt8(Y1, (y + 1) * lumaPitchBytes + x + 1)
				Y0 = Y0 === 234 ? 16 : Y0 + 2
				Y1 = Y1 === 16 ? 234 : Y1 - 2
			}
		}
	}

	const getPitchBytes = f.rgb ? (w) => w * 4 : (w) => pitch8(w) * f.bytesPerSample

	// debugging aid of the reference's test scripts: print the first 4 (RGBA) / 8 (YUV) pixels of
	// the first numLines lines in hex, "Line n: U, Y, V, Y; ..." (or "R, G, B, A; ...")
	function dumpBuf(buf, width, ...rest) {
		const hex = (v) => v.toString(16)
		if (f.rgb) {
			const [numLines] = rest
			for (let l = 0; l < numLines; ++l) {
				const px = []
				for (let p = 0; p < 4; ++p) px.push([0, 1, 2, 3].map((c) => hex(buf.readUInt8(getPitchBytes(width) * l + 4 * p + c))).join(', '))
				console.log(`Line ${l}: ${px.join('; ')}`)
			}
			return
		}
		const [height, numLines, lineEnds] = rest
		const bps = f.bytesPerSample
		const get = bps === 2 ? (o) => buf.readUInt16LE(o) : (o) => buf.readUInt8(o)
		const lumaPitch = getPitchBytes(width)
		const sizes = f.planes(lumaPitch * height)
		const semi = sizes.length === 2
		const chromaPitch = semi ? lumaPitch : lumaPitch / 2
		const end = lineEnds ? lumaPitch - 8 * bps : 0
		console.log()
		for (let l = 0; l < numLines; ++l) {
			const cl = Math.floor(l / f.vertical)
			let s = `Line ${l}:`
			for (let p = 0; p < 8; p += 2) {
				const yOff = lumaPitch * l + end + p * bps
				const cOff = chromaPitch * cl + (semi ? end + p * bps : (end + p * bps) / 2)
				const u = get(sizes[0] + cOff)
				const v = semi ? get(sizes[0] + cOff + bps) : get(sizes[0] + sizes[1] + cOff)
				s += ` ${hex(u)}, ${hex(get(yOff))}, ${hex(v)}, ${hex(get(yOff + bps))};`
			}
			console.log(s)
		}
	}

	const out = { Reader, Writer, fillBuf, dumpBuf }
	if (f.rgb) out.getPitchBytes = getPitchBytes
	return out
}

module.exports = { makeFormat, FORMATS }
