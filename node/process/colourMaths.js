'use strict'
// Host colour maths of the operator layer: gamma LUTs, YCbCr<->RGB and gamut matrices.  Same
// API and results (bit for bit, see node/test) as the reference's src/process/colourMaths.ts;
// matrices are arrays of Float32Array rows, products and sums are JS doubles, every element
// rounds to f32 when stored.
const SPECS = {
	'601-625': { kR: 0.299, kB: 0.114, prim: [0.64, 0.33, 0.29, 0.6, 0.15, 0.06], white: [0.3127, 0.329], oetf: [1.099, 0.018, 0.45, 4.5] },
	'601_525': { kR: 0.299, kB: 0.114, prim: [0.63, 0.34, 0.31, 0.595, 0.155, 0.07], white: [0.3127, 0.329], oetf: [1.099, 0.018, 0.45, 4.5] },
	'709': { kR: 0.2126, kB: 0.0722, prim: [0.64, 0.33, 0.3, 0.6, 0.15, 0.06], white: [0.3127, 0.329], oetf: [1.099, 0.018, 0.45, 4.5] },
	'2020': { kR: 0.2627, kB: 0.0593, prim: [0.708, 0.292, 0.17, 0.797, 0.131, 0.046], white: [0.3127, 0.329], oetf: [1.099, 0.018, 0.45, 4.5] },
	sRGB: { kR: 0.0, kB: 0.0, prim: [0.64, 0.33, 0.3, 0.6, 0.15, 0.06], white: [0.3127, 0.329], oetf: [1.055, 0.0031308, 1.0 / 2.4, 12.92] }
}

function spec(colSpec) {
	if (!(colSpec in SPECS)) {
		console.error(`Unrecognised colourspace ${colSpec} - defaulting to BT.709`)
		return SPECS['709']
	}
	return SPECS[colSpec]
}

const rows = (...r) => r.map((x) => Float32Array.from(x))

function matrixMultiply(a, b) {
	return a.map((arow) => {
		const out = new Float32Array(b[0].length)
		for (let j = 0; j < out.length; ++j) {
			let sum = 0.0
			for (let k = 0; k < arow.length; ++k) sum = sum + arow[k] * b[k][j]
			out[j] = sum
		}
		return out
	})
}

const scaled = (a, c) => a.map((row) => row.map((v) => v * c))

function invert3(a) {
	const pick = [[1, 2], [0, 2], [0, 1]]
	const minors = rows([0, 0, 0], [0, 0, 0], [0, 0, 0])
	for (let i = 0; i < 3; ++i)
		for (let j = 0; j < 3; ++j) {
			const [r0, r1] = pick[i]
			const [c0, c1] = pick[j]
			minors[i][j] = a[r0][c0] * a[r1][c1] - a[r0][c1] * a[r1][c0]
		}
	const adj = rows([0, 0, 0], [0, 0, 0], [0, 0, 0])
	for (let i = 0; i < 3; ++i)
		for (let j = 0; j < 3; ++j) {
			const cof = Math.fround(minors[i][j] * ((i + j) % 2 ? -1 : 1))
			adj[j][i] = cof
		}
	const det = a[0][0] * minors[0][0] - a[0][1] * minors[0][1] + a[0][2] * minors[0][2]
	return scaled(adj, 1.0 / det)
}

function rgb2xyz(colSpec) {
	const s = spec(colSpec)
	const [rx, ry, gx, gy, bx, by] = s.prim
	const [wx, wy] = s.white
	const w = rows([wx], [wy], [1.0 - wx - wy])
	const W = scaled(w, 1.0 / w[1][0])
	const xyz = rows([rx, gx, bx], [ry, gy, by], [1.0 - rx - ry, 1.0 - gx - gy, 1.0 - bx - by])
	const f = matrixMultiply(invert3(xyz), W)
	return matrixMultiply(xyz, rows([f[0][0], 0, 0], [0, f[1][0], 0], [0, 0, f[2][0]]))
}

function gamma2linearLUT(colSpec) {
	const [alpha, beta0, gamma, delta] = spec(colSpec).oetf
	const beta = beta0 * delta
	const lut = new Float32Array(65536)
	for (let i = 0; i < 65536; ++i) {
		const fi = i / 65535
		lut[i] = fi < beta ? fi / delta : Math.pow((fi + (alpha - 1)) / alpha, 1 / gamma)
	}
	return lut
}

function linear2gammaLUT(colSpec) {
	const [alpha, beta, gamma, delta] = spec(colSpec).oetf
	const lut = new Float32Array(65536)
	for (let i = 0; i < 65536; ++i) {
		const fi = i / 65535
		lut[i] = fi < beta ? fi * delta : alpha * Math.pow(fi, gamma) - (alpha - 1)
	}
	return lut
}

function ycbcr2rgbMatrix(colSpec, numBits, lumaBlack, lumaWhite, chrRange) {
	const { kR, kB } = spec(colSpec)
	const kG = 1.0 - kR - kB
	const chrNull = 128.0 << (numBits - 8)
	const lumaRange = lumaWhite - lumaBlack
	const col = rows([1.0, 0.0, 1.0 - kR], [1.0, (-(1.0 - kB) * kB) / kG, (-(1.0 - kR) * kR) / kG], [1.0, 1.0 - kB, 0.0])
	const scale = rows(
		[1.0 / lumaRange, 0.0, 0.0, -lumaBlack / lumaRange],
		[0.0, (1.0 / chrRange) * 2, 0.0, -(chrNull / chrRange) * 2],
		[0.0, 0.0, (1.0 / chrRange) * 2, -(chrNull / chrRange) * 2]
	)
	return matrixMultiply(col, scale)
}

function rgb2ycbcrMatrix(colSpec, numBits, lumaBlack, lumaWhite, chrRange) {
	const { kR, kB } = spec(colSpec)
	const kG = 1.0 - kR - kB
	const chrNull = 128.0 << (numBits - 8)
	const lumaRange = lumaWhite - lumaBlack
	const scale = rows([lumaRange, 0.0, 0.0], [0.0, chrRange / 2.0, 0.0], [0.0, 0.0, chrRange / 2.0])
	const col = rows(
		[kR, kG, kB, lumaBlack / lumaRange],
		[-kR / (1.0 - kB), -kG / (1.0 - kB), (1.0 - kB) / (1.0 - kB), (chrNull / chrRange) * 2.0],
		[(1.0 - kR) / (1.0 - kR), -kG / (1.0 - kR), -kB / (1.0 - kR), (chrNull / chrRange) * 2.0]
	)
	return matrixMultiply(scale, col)
}

function rgb2rgbMatrix(srcColSpec, dstColSpec) {
	return matrixMultiply(invert3(rgb2xyz(dstColSpec)), rgb2xyz(srcColSpec))
}

function matrixFlatten(a) {
	const out = new Float32Array(a.length * a[0].length)
	a.forEach((row, i) => out.set(row, i * row.length))
	return out
}

module.exports = { gamma2linearLUT, linear2gammaLUT, matrixMultiply, ycbcr2rgbMatrix, rgb2ycbcrMatrix, rgb2rgbMatrix, matrixFlatten }
