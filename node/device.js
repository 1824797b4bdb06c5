'use strict'
// Rig - a small own-design front end over the nodencl-shaped surface (index.js) for code in THIS repository
// that drives the MI355X kernels from node: the GPU tests, the node benchmark, the channel compositor
// (channel.js).  In a phaneron deployment this role is played by the reference's own src/process operators and
// src/clJobQueue.ts, unchanged, on top of index.js; nothing here is needed for that.
//
// Shape: no class per operator.  A Rig caches programs and colour-parameter buffers; every `rig.<op>(...)`
// call returns a *stage*: a function from buffers (and per-frame scalars) to a job { name, program, params }
// that can be launched at once (rig.run) or posted on the JobBoard (rig.board.post).  Parameter names are the
// kernels' OpenCL argument names, the contract of runProgram (SURVEY 8b).
const { clContext, colour, planeBytes, FORMATS } = require('./index.js')
const { JobBoard } = require('./jobs.js')

// code ranges of each wire format: numBits, lumaBlack, lumaWhite, chromaRange (v210.ts:284-290 and the
// corresponding constants of the other Readers / Writers); null = RGB formats, no YCbCr matrix
const RANGE = {
	v210: [10, 64, 940, 896], yuv422p10: [10, 64, 940, 896], yuv422p8: [8, 16, 235, 224],
	yuv420p: [8, 16, 235, 224], nv12: [8, 16, 235, 224], rgba8: null, bgra8: null
}
const PLANE_ARGS = { 1: [''], 2: ['Y', 'C'], 3: ['Y', 'U', 'V'] }
const ceilTo = (v, m) => Math.ceil(v / m) * m

// work-group geometry the library derives the frame height from (ph_api.cpp dispatch: height = global / local,
// doubled for 4:2:0 line pairs and for interlaced writers)
function packGeometry(format, width, height, writer, interlaced) {
	const wipg = format === 'v210' ? ceilTo(width, 48) / 48
		: Math.ceil((format === 'rgba8' || format === 'bgra8' ? width : ceilTo(width, 8)) / 64)
	const v420 = format === 'yuv420p' || format === 'nv12'
	const groups = v420 ? height / 2 : (writer && interlaced ? height / 2 : height)
	return { workItemsPerGroup: wipg, globalWorkItems: wipg * groups }
}

class Rig {
	static async open(options = {}) {
		const rig = new Rig()
		rig.ctx = new clContext(Object.assign({ overlapping: true }, options))
		await rig.ctx.initialise()
		rig.board = new JobBoard(rig.ctx, options)
		rig.programs = new Map()
		rig.constants = new Map()
		return rig
	}

	// ---- buffers ------------------------------------------------------------------------------------
	image(width, height, owner = 'image') {
		return this.ctx.createBuffer(width * height * 16, 'readwrite', 'coarse', { width, height }, owner)
	}
	async planes(format, width, height, dir = 'readonly', owner = format) {
		const out = []
		for (const n of planeBytes(format, width, height)) out.push(await this.ctx.createBuffer(n, dir, 'coarse', undefined, owner))
		return out
	}
	// a read-only parameter buffer holding `f32`, uploaded once and shared (key identifies the contents)
	async constant(key, f32, svm = 'none') {
		let b = this.constants.get(key)
		if (!b) {
			b = await this.ctx.createBuffer(f32.byteLength, 'readonly', svm, undefined, key)
			await b.hostAccess('writeonly', this.ctx.queue.load, Buffer.from(f32.buffer, f32.byteOffset, f32.byteLength))
			this.constants.set(key, b)
		}
		return b
	}
	async program(name, tag, geometry) {
		const key = `${name}|${tag}|${geometry.globalWorkItems}|${geometry.workItemsPerGroup || 0}`
		let p = this.programs.get(key)
		if (!p) this.programs.set(key, (p = await this.ctx.createProgram(`phaneron:${tag}`, Object.assign({ name }, geometry))))
		return p
	}

	// ---- colour parameters (what loadSave.ts:50-63,139-149 binds for a Loader / Saver) -----------------
	async colourIn(format, spec, workSpec) {
		const r = RANGE[format]
		return {
			colMatrix: r ? await this.constant(`ycbcr2rgb ${spec} ${r}`, colour.ycbcr2rgbMatrix(spec, ...r)) : undefined,
			gammaLut: await this.constant(`g2l ${spec}`, colour.gamma2linearLUT(spec), 'coarse'),
			gamutMatrix: await this.constant(`gamut ${spec}>${workSpec}`, colour.rgb2rgbMatrix(spec, workSpec))
		}
	}
	async colourOut(format, spec) {
		const r = RANGE[format]
		return {
			colMatrix: r ? await this.constant(`rgb2ycbcr ${spec} ${r}`, colour.rgb2ycbcrMatrix(spec, ...r)) : undefined,
			gammaLut: await this.constant(`l2g ${spec}`, colour.linear2gammaLUT(spec), 'coarse')
		}
	}

	// ---- stages --------------------------------------------------------------------------------------
	// wire format -> linear RGBA f32 in workSpec: stage(planes[], image)
	async unpack(format, width, height, spec, workSpec) {
		if (!FORMATS.includes(format)) throw new Error(`unknown pack format '${format}'`)
		const program = await this.program('read', format, packGeometry(format, width, height, false, false))
		const c = await this.colourIn(format, spec, workSpec)
		const args = PLANE_ARGS[planeBytes(format, width, height).length]
		return (planes, image) => {
			if (planes.length !== args.length) throw new Error(`${format} read needs ${args.length} plane(s), got ${planes.length}`)
			const params = { output: image, width, gammaLut: c.gammaLut, gamutMatrix: c.gamutMatrix }
			if (c.colMatrix) params.colMatrix = c.colMatrix
			args.forEach((a, i) => { params[`input${a}`] = planes[i] })
			return { name: 'read', program, params }
		}
	}
	// linear RGBA f32 -> wire format: stage(image, planes[], field) with field 0 progressive / 1 top / 3 bottom
	// (packer.ts:24-28); an interlaced writer handles half the lines per call
	async pack(format, width, height, spec, interlaced = false) {
		const program = await this.program('write', format, packGeometry(format, width, height, true, interlaced))
		const c = await this.colourOut(format, spec)
		const args = PLANE_ARGS[planeBytes(format, width, height).length]
		return (image, planes, field = 0) => {
			if (planes.length !== args.length) throw new Error(`${format} write needs ${args.length} plane(s), got ${planes.length}`)
			if (interlaced !== (field !== 0)) throw new Error('field 1 / 3 needs an interlaced writer, field 0 a progressive one')
			const params = { input: image, width, interlace: field, gammaLut: c.gammaLut }
			if (c.colMatrix) params.colMatrix = c.colMatrix
			args.forEach((a, i) => { params[`output${a}`] = planes[i] })
			return { name: 'write', program, params }
		}
	}
	async yadif(width, height) {
		const program = await this.program('yadif', 'yadif', { globalWorkItems: [width, height] })
		return (prev, cur, next, output, o) => ({
			name: 'yadif', program,
			params: { prev, cur, next, parity: o.parity ? 1 : 0, tff: o.tff ? 1 : 0, skipSpatial: o.skipSpatial ? 1 : 0, output }
		})
	}
	// the channel's tail as one launch: stage([{ image, matrix?, wipe?: { incoming, mask } }...], output v210, field)
	// = [transform] per layer (+ transition_wipe) -> combine_n -> FromRGBA; bit-identical to the separate stages
	async compose(n, width, height, spec) {
		const program = await this.program(`compose_write_v210_${n}`, 'compose', { globalWorkItems: [width, height] })
		const c = await this.colourOut('v210', spec)
		return (layers, output, field = 0) => {
			if (layers.length !== n) throw new Error(`compose_write_v210_${n} needs ${n} layers, got ${layers.length}`)
			const params = { output, outColMatrix: c.colMatrix, outGammaLut: c.gammaLut, interlace: field }
			layers.forEach((l, i) => {
				params[`l${i}In`] = l.image
				if (l.matrix) params[`l${i}Matrix`] = l.matrix
				if (l.wipe) { params[`l${i}WipeIn`] = l.wipe.incoming; params[`l${i}WipeMask`] = l.wipe.mask }
			})
			return { name: `compose_write_v210_${n}`, program, params }
		}
	}
	// the channel's whole frame as one launch, straight from the v210 sources (no f32 frame in between):
	// stage([{ source (v210 buffer | RGBA image), width?, height?, matrix?, transition?: { type: 'dissolve' | 'wipe', mix?, incoming: {...}, mask?: {...} } }...], output v210, field)
	// = ToRGBA -> [transform] (-> transition) per layer -> combine_n -> FromRGBA; bit-identical to the separate stages
	async channelCompose(n, width, height, readSpec, writeSpec) {
		const program = await this.program(`chan_compose_v210_${n}`, 'chan', { globalWorkItems: [width, height] })
		const ci = await this.colourIn('v210', readSpec, writeSpec)
		const co = await this.colourOut('v210', writeSpec)
		return (layers, output, field = 0) => {
			if (layers.length !== n) throw new Error(`chan_compose_v210_${n} needs ${n} layers, got ${layers.length}`)
			const params = { output, colMatrix: ci.colMatrix, gammaLut: ci.gammaLut, gamutMatrix: ci.gamutMatrix, outColMatrix: co.colMatrix, outGammaLut: co.gammaLut, interlace: field }
			const put = (prefix, s) => {
				params[`${prefix}In`] = s.source
				if (s.matrix) params[`${prefix}Matrix`] = s.matrix
				if (s.width) { params[`${prefix}Width`] = s.width; params[`${prefix}Height`] = s.height }
			}
			layers.forEach((l, i) => {
				put(`l${i}`, l)
				if (l.transition) {
					params[`l${i}Transition`] = l.transition.type === 'wipe' ? 2 : 1
					if (l.transition.type !== 'wipe') params[`l${i}Mix`] = l.transition.mix
					put(`l${i}Incoming`, l.transition.incoming)
					if (l.transition.type === 'wipe') put(`l${i}Mask`, l.transition.mask)
				}
			})
			return { name: `chan_compose_v210_${n}`, program, params }
		}
	}
	// ToRGBA of n v210 frames of one size and colour recipe in one launch: stage([planes...], [images...])
	async unpackBatch(n, width, height, spec, workSpec) {
		const c = await this.colourIn('v210', spec, workSpec)
		const program = await this.program(`v210_read_batch_${n}`, 'batch', { globalWorkItems: [width, height] })
		return (sources, images) => {
			if (sources.length !== n || images.length !== n) throw new Error(`unpackBatch: ${n} frames expected`)
			const params = { colMatrix: c.colMatrix, gammaLut: c.gammaLut, gamutMatrix: c.gamutMatrix }
			sources.forEach((src, i) => { params[`l${i}In`] = src; params[`l${i}Out`] = images[i] })
			return { name: `v210_read_batch_${n}`, program, params }
		}
	}
	// both send_field outputs of a frame in one pass: out[0] / out[1] are what yadif writes with parity 0 / 1
	async yadifPair(width, height) {
		const program = await this.program('yadif_pair', 'yadif', { globalWorkItems: [width, height] })
		return (prev, cur, next, out, o) => ({
			name: 'yadif_pair', program,
			params: { prev, cur, next, tff: o.tff ? 1 : 0, skipSpatial: o.skipSpatial ? 1 : 0, output0: out[0], output1: out[1] }
		})
	}
	// ToRGBA over the v210 windows of n layers + both de-interlaced fields of each, one kernel:
	// stage([{ prev, cur, next, out: [parity0, parity1] }, ...], { tff, skipSpatial })
	async deinterlaceReader(layers, width, height, readSpec, writeSpec) {
		const c = await this.colourIn('v210', readSpec, writeSpec)
		const program = await this.program(`v210_yadif_pair_${layers}`, 'deint', { globalWorkItems: [width, height] })
		return (windows, o) => {
			if (windows.length !== layers) throw new Error(`deinterlaceReader: ${layers} layers expected, got ${windows.length}`)
			const params = { colMatrix: c.colMatrix, gammaLut: c.gammaLut, gamutMatrix: c.gamutMatrix, tff: o.tff ? 1 : 0, skipSpatial: o.skipSpatial ? 1 : 0 }
			windows.forEach((wn, i) => {
				params[`l${i}Prev`] = wn.prev; params[`l${i}Cur`] = wn.cur; params[`l${i}Next`] = wn.next
				params[`l${i}Out0`] = wn.out[0]; params[`l${i}Out1`] = wn.out[1]
			})
			return { name: `v210_yadif_pair_${layers}`, program, params }
		}
	}
	// stage(input, output, placement): the 3x3 matrix of a placement is uploaded once per distinct placement
	async transform(outWidth, outHeight) {
		const program = await this.program('transform', 'transform', { globalWorkItems: [outWidth, outHeight] })
		const cache = new Map()
		const stage = (input, output, matrix) => ({ name: 'transform', program, params: { input, transformMatrix: matrix, output } })
		stage.matrix = async (placement) => {
			const key = JSON.stringify(placement)
			if (!cache.has(key)) {
				const m = new Float32Array(12) // the kernel reads three float4 rows (transform.ts:47-49)
				m.set(colour.transformMatrix(outWidth, outHeight, placement))
				cache.set(key, await this.constant(`xf ${outWidth}x${outHeight} ${key}`, m))
			}
			return cache.get(key)
		}
		return stage
	}
	async resize(outWidth, outHeight) {
		const program = await this.program('resize', 'resize', { globalWorkItems: [outWidth, outHeight] })
		return async (input, output, o) => {
			const scale = o.scale === undefined ? 1 : o.scale
			const offsetX = o.offsetX || 0
			const offsetY = o.offsetY || 0
			if (!(scale > 0)) throw new Error('resize: scale must be greater than 0')
			if (Math.abs(offsetX) > 1 || Math.abs(offsetY) > 1) throw new Error('resize: offsets must be between -1 and 1')
			const flip = await this.constant(`flip ${!!o.flipH} ${!!o.flipV}`,
				Float32Array.from([o.flipH ? 1 : 0, o.flipH ? -1 : 1, o.flipV ? 1 : 0, o.flipV ? -1 : 1]))
			return { name: 'resize', program, params: { input, scale, offsetX, offsetY, flip, output } }
		}
	}
	async combine(n, width, height) {
		if (n < 2) throw new Error(`combine needs at least 2 layers, got ${n}`)
		const program = await this.program(`combine_${n}`, 'combine', { globalWorkItems: [width, height] })
		return (layers, output) => {
			if (layers.length !== n) throw new Error(`combine_${n} needs ${n} layers, got ${layers.length}`)
			const params = { output }
			layers.forEach((l, i) => { params[`l${i}In`] = l })
			return { name: `combine_${n}`, program, params }
		}
	}
	async two(kind, width, height) { // 'transition_dissolve' | 'mixer' (scalar mix), 'wipe' (scalar wipe), 'transition_wipe' (mask image)
		const program = await this.program(kind, kind, { globalWorkItems: [width, height] })
		return (input0, input1, by, output) => {
			const params = { input0, input1, output }
			if (kind === 'transition_wipe') {
				if (!by) throw new Error('a wipe transition needs a mask image')
				params.maskIn = by
			} else params[kind === 'wipe' ? 'wipe' : 'mix'] = by
			return { name: kind, program, params }
		}
	}
	// extension: n v210 layers -> unpack, combine_n, pack as ONE kernel (ph_fused_v210_combine)
	async fused(n, width, height, spec, outSpec) {
		const program = await this.program(`fused_v210_combine_${n}`, 'fused', { globalWorkItems: [width, height] })
		const ci = await this.colourIn('v210', spec, outSpec)
		const co = await this.colourOut('v210', outSpec)
		return (layers, output) => {
			if (layers.length !== n) throw new Error(`fused_v210_combine_${n} needs ${n} layers, got ${layers.length}`)
			const params = { output, colMatrix: ci.colMatrix, gammaLut: ci.gammaLut, gamutMatrix: ci.gamutMatrix, outColMatrix: co.colMatrix, outGammaLut: co.gammaLut }
			layers.forEach((l, i) => { params[`l${i}In`] = l })
			return { name: `fused_v210_combine_${n}`, program, params }
		}
	}

	// ---- running ---------------------------------------------------------------------------------------
	run(job, queue) { return this.ctx.runProgram(job.program, job.params, queue === undefined ? this.ctx.queue.process : queue) }
	post(id, job, done) { this.board.post(id, job.name, job.program, job.params, done) }
	sync(queue) { return this.ctx.waitFinish(queue) }
	// a frame onto the device as a producer loads one: hostAccess('writeonly') ENQUEUES the copy on the load queue, then waitFinish(load) before
	// the frame is used on another queue (ffmpegProducer.ts:514-515) - without it a job launched at once on the process queue races the copy
	async upload(buffer, bytes) { await buffer.hostAccess('writeonly', this.ctx.queue.load, bytes); await this.sync(this.ctx.queue.load) }
	async download(buffer) { await buffer.hostAccess('readonly', this.ctx.queue.unload); return buffer }
	close() {
		for (const b of this.constants.values()) b.release()
		this.constants.clear()
	}
}

module.exports = { Rig, RANGE, packGeometry }
