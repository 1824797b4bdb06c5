'use strict'
// Runs the REFERENCE's own host maths (src/process/colourMaths.ts, transform.ts, v210.ts fillBuf),
// type-stripped into oracle/_ref/work/js by oracle/refbuild/ts_strip.py, under node 12 and prints the
// results as JSON (f32 bit patterns / sha256).  Build container only; called by gen_golden.py.
const path = require('path')
const crypto = require('crypto')
const root = path.join(__dirname, '..', '..', 'oracle', '_ref', 'work', 'js')
const cm = require(path.join(root, 'process', 'colourMaths.js'))
const v210 = require(path.join(root, 'process', 'v210.js'))
const Transform = require(path.join(root, 'process', 'transform.js')).default

const hex = (f32) =>
	Array.from(new Uint32Array(Float32Array.from(f32).buffer)).map((x) => x.toString(16).padStart(8, '0'))
const sha = (buf) => crypto.createHash('sha256').update(buf).digest('hex')

const specs = ['601-625', '601_525', '709', '2020', 'sRGB', 'bogus']
const out = { ycbcr2rgb: {}, rgb2ycbcr: {}, rgb2rgb: {}, lut: {}, ramp: {}, transform: [] }

// console.error is how the reference reports an unknown colourspace; keep the dump clean
console.error = () => {}

const ranges = { '10': [10, 64, 940, 896], '8': [8, 16, 235, 224] }
for (const s of specs) {
	for (const r of Object.keys(ranges)) {
		const a = ranges[r]
		out.ycbcr2rgb[`${s}/${r}`] = hex(cm.matrixFlatten(cm.ycbcr2rgbMatrix(s, a[0], a[1], a[2], a[3])))
		out.rgb2ycbcr[`${s}/${r}`] = hex(cm.matrixFlatten(cm.rgb2ycbcrMatrix(s, a[0], a[1], a[2], a[3])))
	}
	for (const d of specs) out.rgb2rgb[`${s}->${d}`] = hex(cm.matrixFlatten(cm.rgb2rgbMatrix(s, d)))
	const g2l = cm.gamma2linearLUT(s)
	const l2g = cm.linear2gammaLUT(s)
	const sample = (lut) => {
		const idx = []
		for (let i = 0; i < 65536; i += 257) idx.push(i)
		return hex(idx.map((i) => lut[i]))
	}
	if (process.env.REF_LUT_DIR) {
		// full tables for gen_golden.py's kernel runs (written under oracle/_ref, never committed)
		require('fs').writeFileSync(path.join(process.env.REF_LUT_DIR, `g2l_${s}.bin`), Buffer.from(g2l.buffer))
		require('fs').writeFileSync(path.join(process.env.REF_LUT_DIR, `l2g_${s}.bin`), Buffer.from(l2g.buffer))
	}
	out.lut[s] = {
		g2l_sha256: sha(Buffer.from(g2l.buffer)),
		l2g_sha256: sha(Buffer.from(l2g.buffer)),
		g2l_every257: sample(g2l),
		l2g_every257: sample(l2g)
	}
}

for (const [w, h] of [[1920, 1080], [3840, 2160], [1280, 720], [96, 4], [100, 3], [98, 2]]) {
	const pitchBytes = ((w + 47 - ((w - 1) % 48)) * 8) / 3
	const buf = Buffer.alloc(pitchBytes * h, 0xa5) // fillBuf must clear it
	v210.fillBuf(buf, w, h)
	out.ramp[`${w}x${h}`] = sha(buf)
}

// the other pack formats: test pattern hashes and Reader / Writer geometry (SURVEY 8f-1)
out.ramp_fmt = {}
out.format_geometry = {}
for (const fmt of ['yuv422p10', 'yuv422p8', 'yuv420p', 'nv12', 'rgba8', 'bgra8']) {
	const m = require(path.join(root, 'process', `${fmt}.js`))
	for (const [w, h] of [[1920, 1080], [718, 480], [128, 4]]) {
		if ((fmt === 'rgba8' || fmt === 'bgra8') && w === 718) continue
		const r = new m.Reader(w, h)
		const wr = new m.Writer(w, h, true)
		const buf = Buffer.alloc(r.getTotalBytes(), 0x5a)
		m.fillBuf(buf, w, h)
		out.ramp_fmt[`${fmt}/${w}x${h}`] = sha(buf)
		out.format_geometry[`${fmt}/${w}x${h}`] = {
			numBytes: r.getNumBytes(), readWipg: r.getWorkItemsPerGroup(), readGwi: r.getGlobalWorkItems(),
			writeWipgInterlaced: wr.getWorkItemsPerGroup(), writeGwiInterlaced: wr.getGlobalWorkItems(),
			numBits: r.numBits, lumaBlack: r.lumaBlack, lumaWhite: r.lumaWhite, chromaRange: r.chromaRange, isRGB: r.getIsRGB(), name: r.getName()
		}
	}
}

// Transform.getKernelParams against a recording stand-in for the nodencl context: the
// matrix it uploads is what the device kernel receives (transform.ts:119-175).
const paramSets = [
	{ w: 1920, h: 1080, p: {} },
	{ w: 3840, h: 2160, p: { scaleX: 0.5, scaleY: 0.5, offsetX: 0.25, offsetY: -0.25 } },
	{ w: 1920, h: 1080, p: { scaleX: 0.5, scaleY: 0.5, offsetX: -0.25, offsetY: 0.25, anchorX: 0.1, anchorY: -0.2 } },
	{ w: 1920, h: 1080, p: { rotate: 0.125 } },
	{ w: 1920, h: 1080, p: { rotate: -0.25, scaleX: 0.75, scaleY: 1.25, anchorX: -0.5, anchorY: -0.5 } },
	{ w: 1280, h: 720, p: { flipH: true } },
	{ w: 1280, h: 720, p: { flipV: true, rotate: 1 / 360 } },
	{ w: 64, h: 36, p: { flipH: true, flipV: true, scaleX: 2.0, scaleY: 2.0, offsetX: 0.125, offsetY: 0.0625 } },
	{ w: 720, h: 576, p: { scaleX: 1.5, scaleY: 0.5, rotate: 0.3, offsetX: 0.3, offsetY: -0.4, anchorX: 0.25, anchorY: 0.125 } }
]
const run = async () => {
	for (const ps of paramSets) {
		let uploaded = null
		const fakeBuf = {
			hostAccess: async (dir, q, src) => { if (src) uploaded = Buffer.from(src) },
			addRef: () => {},
			release: () => {}
		}
		const fakeCtx = { queue: { load: 0, process: 1, unload: 2 }, createBuffer: async () => fakeBuf, waitFinish: async () => {} }
		const t = new Transform(fakeCtx, ps.w, ps.h)
		await t.init()
		await t.getKernelParams(Object.assign({ input: null, output: null }, ps.p))
		const f = new Float32Array(uploaded.buffer, uploaded.byteOffset, 9)
		out.transform.push({ width: ps.w, height: ps.h, params: ps.p, matrix: hex(f) })
	}
	process.stdout.write(JSON.stringify(out, null, 1))
}
run().catch((e) => { process.stderr.write(String(e.stack || e)); process.exit(1) })
