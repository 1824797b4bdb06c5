'use strict'
// node/defer.js without a device: the recording's graph logic (what is launched, in which order, what is dropped, who
// holds which buffer) against a stand-in for the addon that only counts.  Prints one JSON object { checks, problems }.
const { Deferral } = require('../defer.js')
const { bufferPrototype, newPark } = require('../index.js')

const problems = []
let checks = 0
const expect = (what, got, want) => {
	++checks
	if (JSON.stringify(got) !== JSON.stringify(want)) problems.push({ what, got, want })
}

function rig(opt) { // opt.early: frames are launched at the end of the posting tick (clContext's earlyLaunch); opt.batch: the addon has runPrograms
	opt = opt || {}
	const launches = [] // [program name, queue]
	const orders = [] // [waiter, signal]
	let nextId = 1
	const refs = new Map() // handle -> count (0 = freed)
	const native = {
		bufAddRef: (h) => { if (!refs.get(h)) throw new Error(`addRef on freed buffer ${h}`); refs.set(h, refs.get(h) + 1) },
		bufRelease: (h) => { if (!refs.get(h)) throw new Error(`release on freed buffer ${h}`); refs.set(h, refs.get(h) - 1) },
		bufRefCount: (h) => refs.get(h) || 0,
		createProgram: (_ctx, _src, name) => ({ name }),
		runProgram: (_ctx, prog, names, values, queue, _timed, checkOnly) => {
			if (checkOnly) return null
			if (native.refuse && native.refuse(prog.name)) throw new Error(`${prog.name}: refused`)
			for (const v of values) if (v && typeof v === 'object' && !refs.get(v)) throw new Error(`${prog.name} launched on a freed buffer`)
			launches.push([prog.name, queue, names.join(',')])
			return { dataToKernel: 0, kernelExec: 0, totalTime: 0 }
		},
		queueWaitQueue: (_ctx, waiter, signal) => orders.push([waiter, signal])
	}
	if (opt.batch) native.runPrograms = (_ctx, progs, names, values, queue) => {
		if (native.refuse && native.refuse('batch')) throw new Error('batch: refused')
		for (const vs of values) for (const v of vs) if (v && typeof v === 'object' && !refs.get(v)) throw new Error('batch launched on a freed buffer')
		launches.push([`batch:${progs.map((p) => p.name).join('+')}`, queue, names.map((n) => n.join(',')).join(';'), values])
	}
	const ctx = { _native: native, _ctx: {}, queue: { load: 0, process: 1, unload: 2 }, earlyLaunch: !!opt.early }
	const d = new Deferral(ctx)
	const proto = bufferPrototype(native, d, newPark(false, 0)) // the real reference counting of node/index.js; nothing is parked here
	const buffer = (bytes, dims, owner) => {
		const b = Buffer.alloc(bytes)
		const h = { id: nextId++ }
		refs.set(h, 1)
		Object.setPrototypeOf(b, proto)
		Object.defineProperty(b, '_handle', { value: h })
		b._refs = 1
		b._dead = false
		b.imageDims = dims
		b.owner = owner || ''
		b.alive = () => refs.get(h) > 0
		b.appRefs = () => b._refs
		Deferral.adopt(b, true)
		return b
	}
	const W = 96
	const H = 4
	const program = (name, extra) => Object.assign({ name, globalWorkItems: [W, H], workItemsPerGroup: 0, _handle: { name } }, extra || {})
	const P = {
		read: program('read', { format: 'v210', globalWorkItems: [2 * H], workItemsPerGroup: 2 }),
		write: program('write', { format: 'v210', globalWorkItems: [2 * H], workItemsPerGroup: 2 }),
		writeField: program('write', { format: 'v210', globalWorkItems: [H], workItemsPerGroup: 2 }),
		transform: program('transform'), combine2: program('combine_2'), dissolve: program('transition_dissolve'), yadif: program('yadif'),
		other: program('resize')
	}
	const param = (tag, bytes, fill) => { const b = buffer(bytes, undefined, tag); b.fill(fill); return b }
	const loader = (fill = 1) => ({ colMatrix: param('cm', 48, fill), gammaLut: param('lut', 64, fill + 1), gamutMatrix: param('gm', 36, fill + 2) })
	const saver = { colMatrix: param('wcm', 48, 7), gammaLut: param('wlut', 64, 8) }
	const image = (owner) => buffer(W * H * 16, { width: W, height: H }, owner)
	const v210 = (owner) => buffer(256 * H, undefined, owner)
	// a placement the 2 x 2-block compositor takes (node/defer.js `enlarged`): half a source texel per output pixel, no rotation
	const enlarging = () => { const b = buffer(48, undefined, 'matrix'); b.fill(0); new Float32Array(b.buffer, b.byteOffset, 9).set([0.5, 0, 0, 0, 0.5, 0, 0, 0, 1]); return b }
	return { d, native, launches, orders, buffer, image, v210, P, loader, saver, enlarging, W, H, names: () => launches.map((l) => l[0]) }
}

// 1. read x2 -> combine_2 -> write: one fused launch, intermediates never made, everything let go when the owners release
{
	const r = rig()
	const L = r.loader()
	const src = [r.v210('s0'), r.v210('s1')]
	const img = [r.image('u0'), r.image('u1')]
	const comb = r.image('comb')
	const out = r.v210('out')
	src.forEach((s, i) => r.d.record(r.P.read, Object.assign({ input: s, output: img[i], width: r.W }, L), 1))
	r.d.record(r.P.combine2, { l0In: img[0], l1In: img[1], output: comb }, 1)
	r.d.record(r.P.write, Object.assign({ input: comb, output: out, width: r.W, interlace: 0 }, r.saver), 1)
	src.forEach((s) => s.release()) // the read jobs' callbacks
	img.forEach((s) => s.release()) // the combine job's callback
	comb.release() // the write job's callback
	expect('sources survive their owners while recorded', src.map((s) => s.alive()), [true, true])
	expect('nothing launched before the frame is asked for', r.names(), [])
	r.d.touch(out, 'readonly', 2)
	expect('one fused launch', r.names(), ['fused_v210_combine_2'])
	expect('the unload queue is ordered behind the process queue', r.orders, [[2, 1]])
	expect('recording empty', r.d.pending.size, 0)
	expect('sources and intermediates freed', [...src, ...img, comb].map((b) => b.alive()), [false, false, false, false, false])
	expect('counters', [r.d.stats.recorded, r.d.stats.launched, r.d.stats.fused, r.d.stats.plain, r.d.stats.dropped], [4, 1, 1, 0, 3])
}

// 2. Loaders with equal contents are one recipe; a Loader with other contents makes its layer a finished image first
{
	const r = rig()
	const out = r.v210('out')
	const build = (loaders) => {
		const img = loaders.map((L, i) => { const im = r.image(`u${i}`); r.d.record(r.P.read, Object.assign({ input: r.v210(`s${i}`), output: im, width: r.W }, L), 1); return im })
		const comb = r.image('comb')
		r.d.record(r.P.combine2, { l0In: img[0], l1In: img[1], output: comb }, 1)
		r.d.record(r.P.write, Object.assign({ input: comb, output: out, width: r.W, interlace: 0 }, r.saver), 1)
	}
	build([r.loader(1), r.loader(1)])
	r.d.touch(out, 'readonly', 2)
	expect('equal Loader contents: one fused launch', r.names(), ['fused_v210_combine_2'])
	build([r.loader(1), r.loader(5)])
	r.d.touch(out, 'readonly', 2)
	expect('another Loader: its layer is unpacked as recorded, the rest goes to the channel kernel', r.names().slice(1), ['read', 'chan_compose_v210_2'])
}

// 3. hazards: a source overwritten while recorded; a destination reused for the next frame; a field write over a field write
{
	const r = rig()
	const L = r.loader()
	const s = r.v210('s')
	const u = r.image('u')
	const out = r.v210('out')
	r.d.record(r.P.read, Object.assign({ input: s, output: u, width: r.W }, L), 1)
	r.d.record(r.P.write, Object.assign({ input: u, output: out, width: r.W, interlace: 0 }, r.saver), 1)
	r.d.touch(s, 'writeonly', 0) // the next frame into the same source buffer
	expect('the reader of an overwritten source runs first, as recorded', r.names(), ['read'])
	expect('the upload is ordered behind it', r.orders, [[0, 1]])
	r.d.touch(out, 'readonly', 2)
	expect('then the write has only a finished image in front of it: as recorded', r.names(), ['read', 'write'])

	const r2 = rig()
	const u2 = r2.image('u')
	const L2 = r2.loader()
	for (let f = 0; f < 3; ++f) r2.d.record(r2.P.read, Object.assign({ input: r2.v210(`s${f}`), output: u2, width: r2.W }, L2), 1)
	expect('a result overwritten unseen is dropped, not run', [r2.names(), r2.d.stats.dropped, r2.d.pending.size], [[], 2, 1])

	const r3 = rig()
	const out3 = r3.v210('out')
	const L3 = r3.loader()
	for (const field of [1, 3]) {
		const im = r3.image(`u${field}`)
		r3.d.record(r3.P.read, Object.assign({ input: r3.v210(`s${field}`), output: im, width: r3.W }, L3), 1)
		r3.d.record(r3.P.writeField, Object.assign({ input: im, output: out3, width: r3.W, interlace: field }, r3.saver), 1)
	}
	expect('the first field is made when the second is posted (it fills every other line of the same frame)', r3.names(), ['chan_compose_v210_1'])
	r3.d.touch(out3, 'readonly', 2)
	expect('then the second', r3.names(), ['chan_compose_v210_1', 'chan_compose_v210_1'])
}

// 4. an intermediate somebody holds is made when asked for; an in-place job keeps its operand's producer
{
	const r = rig()
	const L = r.loader()
	const u = r.image('u')
	const t = r.image('t')
	const m = r.buffer(48, undefined, 'matrix')
	const out = r.v210('out')
	r.d.record(r.P.read, Object.assign({ input: r.v210('s'), output: u, width: r.W }, L), 1)
	r.d.record(r.P.transform, { input: u, transformMatrix: m, output: t }, 1)
	r.d.record(r.P.write, Object.assign({ input: t, output: out, width: r.W, interlace: 0 }, r.saver), 1)
	r.d.touch(out, 'readonly', 2)
	expect('fused frame', r.names(), ['chan_compose_v210_1'])
	expect('the images in between stay recipes while their owner holds them', r.d.pending.size, 2)
	r.d.touch(t, 'readonly', 2)
	expect('asked for after all: producers first', r.names(), ['chan_compose_v210_1', 'read', 'transform'])
	r.d.record(r.P.other, { input: t, output: t }, 1)
	r.d.record(r.P.other, { input: t, output: t }, 1)
	r.d.touch(t, 'readonly', 2)
	expect('in-place jobs run in order', r.names().slice(3), ['resize', 'resize'])
}

// 5. both fields of a de-interlaced layer: one pair launch; a refused fused launch falls back to the recorded jobs
{
	const r = rig()
	const L = r.loader()
	const win = [0, 1, 2].map((i) => { const im = r.image(`w${i}`); r.d.record(r.P.read, Object.assign({ input: r.v210(`s${i}`), output: im, width: r.W }, L), 1); return im })
	const m = r.enlarging()
	const outs = []
	for (const parity of [0, 1]) {
		const y = r.image(`y${parity}`)
		r.d.record(r.P.yadif, { prev: win[0], cur: win[1], next: win[2], parity, tff: 1, skipSpatial: 0, output: y }, 1)
		const t = r.image(`t${parity}`)
		r.d.record(r.P.transform, { input: y, transformMatrix: m, output: t }, 1)
		const out = r.v210(`out${parity}`)
		r.d.record(r.P.write, Object.assign({ input: t, output: out, width: r.W, interlace: 0 }, r.saver), 1)
		outs.push(out)
	}
	r.d.touch(outs[0], 'readonly', 2)
	expect('pair launch, then the tap-sharing compositor for BOTH fields of the frame in one launch (the other field\'s chain is recorded too)',
		r.launches.map((l) => [l[0], /output2/.test(l[2]), /l0In2/.test(l[2])]), [['v210_yadif_pair_1', false, false], ['compose_up_write_v210_1', true, true]])
	expect('the fields travel packed (12 bytes per pixel) between the two launches, and the compositor is told their size',
		r.launches.map((l) => [/packedRgb/.test(l[2]), /l0Width,l0Height/.test(l[2])]), [[true, false], [true, true]])
	r.d.touch(outs[1], 'readonly', 2)
	expect('the second field has been made already', r.names().length, 2)
	expect('nothing pending but the recipes of images their owners still hold', Array.from(r.d.pending).map((nd) => nd.program.name).sort(), ['read', 'read', 'read', 'transform', 'transform'])
}
// 5a. a de-interlaced layer shown at its OWN size (1080i on a 1080 channel): under the default fill the 2 x 2-block compositor takes it (the one
// placement at scale one it serves); moved by a fraction of a pixel it is the channel kernel's - decided from the matrix, no refused attempt
for (const moved of [false, true]) {
	const r = rig()
	const L = r.loader()
	const win = [0, 1, 2].map((i) => { const im = r.image(`w${i}`); r.d.record(r.P.read, Object.assign({ input: r.v210(`s${i}`), output: im, width: r.W }, L), 1); return im })
	const m = r.enlarging()
	new Float32Array(m.buffer, m.byteOffset, 9).set([1, 0, moved ? 0.01 : 0, 0, 1, 0, 0, 0, 1])
	const y = r.image('y0')
	r.d.record(r.P.yadif, { prev: win[0], cur: win[1], next: win[2], parity: 0, tff: 1, skipSpatial: 0, output: y }, 1)
	const t = r.image('t0')
	r.d.record(r.P.transform, { input: y, transformMatrix: m, output: t }, 1)
	const out = r.v210('out0')
	r.d.record(r.P.write, Object.assign({ input: t, output: out, width: r.W, interlace: 0 }, r.saver), 1)
	r.d.touch(out, 'readonly', 2)
	expect(`own-size field${moved ? ', moved' : ' under the default fill'}: the field is made, then ${moved ? 'the channel kernel' : 'the 2 x 2-block compositor'} - no refused attempt`,
		[r.names().slice(-2), r.d.stats.fallbacks], [['yadif', moved ? 'chan_compose_v210_1' : 'compose_up_write_v210_1'], 0])
}
// 5b. the same, refused: first the two-field form, then the single one, then the jobs as recorded; the other field likewise
{
	const r = rig()
	const L = r.loader()
	const win = [0, 1, 2].map((i) => { const im = r.image(`w${i}`); r.d.record(r.P.read, Object.assign({ input: r.v210(`s${i}`), output: im, width: r.W }, L), 1); return im })
	const m = r.enlarging()
	const outs = []
	for (const parity of [0, 1]) {
		const y = r.image(`y${parity}`)
		r.d.record(r.P.yadif, { prev: win[0], cur: win[1], next: win[2], parity, tff: 1, skipSpatial: 0, output: y }, 1)
		const t = r.image(`t${parity}`)
		r.d.record(r.P.transform, { input: y, transformMatrix: m, output: t }, 1)
		const out = r.v210(`out${parity}`)
		r.d.record(r.P.write, Object.assign({ input: t, output: out, width: r.W, interlace: 0 }, r.saver), 1)
		outs.push(out)
	}
	r.native.refuse = (name) => name !== 'write' && name !== 'transform' && name !== 'v210_yadif_pair_1' && name !== 'rgb_unpack'
	r.d.touch(outs[0], 'readonly', 2)
	// (the pair launch wrote the fields packed - the frame was the 2 x 2-block compositor's to make; when somebody else takes them they are unpacked first)
	expect('refused fused launches: the field unpacked, then the jobs as recorded', r.names(), ['v210_yadif_pair_1', 'rgb_unpack', 'transform', 'write', 'rgb_unpack']) // (the last one: the other field's frame was tried along with this one)
	// (the other field's frame is planned after the first has been made as recorded: no pair form left for it - its single form and the channel kernel)
	expect('fallbacks counted: the pair form, the single form, the channel kernel - and the other field\'s frame, tried along with the one asked for: two more', r.d.stats.fallbacks, 5)
	r.d.touch(outs[1], 'readonly', 2)
	expect('the other field: as recorded too', r.names().slice(5), ['transform', 'write'])
}

// 6. a packed frame made on the device and read back (write -> read -> write): the fused launch reads `mid` itself, so the job
//    that makes `mid` runs first (ADVICE r3: it used to stay pending, and `out` was computed from an unwritten buffer)
{
	const r = rig()
	const L = r.loader()
	const s0 = r.v210('s0')
	const a = r.image('a')
	const mid = r.v210('mid')
	const b = r.image('b')
	const out = r.v210('out')
	r.d.record(r.P.read, Object.assign({ input: s0, output: a, width: r.W }, L), 1)
	r.d.record(r.P.write, Object.assign({ input: a, output: mid, width: r.W, interlace: 0 }, r.saver), 1)
	r.d.record(r.P.read, Object.assign({ input: mid, output: b, width: r.W }, L), 1)
	r.d.record(r.P.write, Object.assign({ input: b, output: out, width: r.W, interlace: 0 }, r.saver), 1)
	r.d.touch(out, 'readonly', 2)
	expect('the frame in the middle is made first, then read back', r.launches.map((l) => [l[0], /l0In/.test(l[2])]), [['fused_v210_combine_1', true], ['fused_v210_combine_1', true]])
	expect('nothing is left pending but the recipes of images their owners still hold', r.d.pending.size, 2)
	;[s0, a, mid, b, out].forEach((x) => x.release())
	expect('and those go with their owners', r.d.pending.size, 0)
	// the same through the de-interlacing reader: a window frame that is a pending packed result
	const r2 = rig()
	const L2 = r2.loader()
	const pre = r2.image('pre')
	const packed = r2.v210('packed')
	r2.d.record(r2.P.read, Object.assign({ input: r2.v210('s'), output: pre, width: r2.W }, L2), 1)
	r2.d.record(r2.P.write, Object.assign({ input: pre, output: packed, width: r2.W, interlace: 0 }, r2.saver), 1)
	const win = [r2.v210('w0'), packed, r2.v210('w2')].map((src, i) => { const im = r2.image(`u${i}`); r2.d.record(r2.P.read, Object.assign({ input: src, output: im, width: r2.W }, L2), 1); return im })
	const m = r2.enlarging()
	const outs = []
	for (const parity of [0, 1]) {
		const y = r2.image(`y${parity}`)
		r2.d.record(r2.P.yadif, { prev: win[0], cur: win[1], next: win[2], parity, tff: 1, skipSpatial: 0, output: y }, 1)
		const t = r2.image(`t${parity}`)
		r2.d.record(r2.P.transform, { input: y, transformMatrix: m, output: t }, 1)
		const o = r2.v210(`o${parity}`)
		r2.d.record(r2.P.write, Object.assign({ input: t, output: o, width: r2.W, interlace: 0 }, r2.saver), 1)
		outs.push(o)
	}
	r2.d.touch(outs[0], 'readonly', 2)
	expect('a window frame that is a pending result is made before the pair launch reads it', r2.names(), ['fused_v210_combine_1', 'v210_yadif_pair_1', 'compose_up_write_v210_1'])
}

// 7. a job that fails is not forgotten: every later consumer of its output is told, until somebody writes the buffer again;
//    a job that may fill only PART of a buffer never drops the pending producer of the rest
{
	const r = rig()
	const L = r.loader()
	const u = r.image('u')
	const t1 = r.image('t1')
	const t2 = r.image('t2')
	const m = r.buffer(48, undefined, 'matrix')
	r.d.record(r.P.read, Object.assign({ input: r.v210('s'), output: u, width: r.W }, L), 1)
	r.d.record(r.P.transform, { input: u, transformMatrix: m, output: t1 }, 1)
	r.d.record(r.P.other, { input: u, output: t2 }, 1)
	r.native.refuse = (name) => name === 'read'
	const caught = []
	for (const b of [t1, t2, u]) { try { r.d.touch(b, 'readonly', 2); caught.push(null) } catch (e) { caught.push(String(e.message)) } }
	expect('the failed read is reported to each of its consumers, and to whoever asks for its own output', caught, ['read: refused', 'read: refused', 'read: refused'])
	expect('and nothing of it was launched', r.names(), [])
	r.native.refuse = null
	r.d.record(r.P.read, Object.assign({ input: r.v210('s2'), output: u, width: r.W }, L), 1)
	r.d.touch(u, 'readonly', 2)
	expect('a new job into the buffer clears the failure', r.names(), ['read'])

	const r2 = rig()
	const L2 = r2.loader()
	const frame = r2.image('frame')
	const partial = Object.assign({}, r2.P.other, { name: 'paint_region' }) // a program this layer does not know: it may write part of its output
	r2.d.record(r2.P.read, Object.assign({ input: r2.v210('s'), output: frame, width: r2.W }, L2), 1)
	r2.d.record(partial, { input: r2.image('logo'), output: frame }, 1)
	expect('an unknown program over a pending result: the producer runs first, it is not dropped', [r2.names(), r2.d.stats.dropped], [['read'], 0])
	r2.d.record(r2.P.read, Object.assign({ input: r2.v210('s3'), output: frame, width: r2.W }, L2), 1)
	expect('a whole-frame operator over it: the superseded job is dropped unseen', [r2.names(), r2.d.stats.dropped], [['read'], 1])
}

// 8. ADVICE r4: field images re-recorded since the pair launch made them are no twins any more
{
	const r = rig()
	const L = r.loader()
	const m = r.enlarging()
	const y = [r.image('y0'), r.image('y1')]
	const frame = () => {
		const win = [0, 1, 2].map((i) => { const im = r.image(`w${i}`); r.d.record(r.P.read, Object.assign({ input: r.v210(`s${i}`), output: im, width: r.W }, L), 1); return im })
		for (const parity of [0, 1]) r.d.record(r.P.yadif, { prev: win[0], cur: win[1], next: win[2], parity, tff: 1, skipSpatial: 0, output: y[parity] }, 1)
	}
	const writes = () => [0, 1].map((parity) => {
		const t = r.image(`t${parity}`)
		r.d.record(r.P.transform, { input: y[parity], transformMatrix: m, output: t }, 1)
		const out = r.v210(`out${parity}`)
		r.d.record(r.P.write, Object.assign({ input: t, output: out, width: r.W, interlace: 0 }, r.saver), 1)
		return out
	})
	frame()
	let outs = writes()
	r.d.touch(outs[0], 'readonly', 2)
	expect('frame k: the pair launch and both fields\' frames in one launch', r.names(), ['v210_yadif_pair_1', 'compose_up_write_v210_1'])
	// frame k + 1 into the SAME field images, but y1 gets another producer (not a pairable one) while y0 keeps its contents
	r.d.record(r.P.other, { input: r.image('still'), output: y[1] }, 1)
	outs = writes()
	r.launches.length = 0
	r.d.touch(outs[0], 'readonly', 2)
	expect('frame k + 1: y1 is no finished field image any more - no launch takes it as the other field of y0 (no output2 / l0In2); the other frame, tried along with the one asked for, is made from y1 AFTER y1\'s new producer has run',
		r.launches.map((l) => [l[0], /output2/.test(l[2])]), [['resize', false], ['compose_up_write_v210_1', false], ['compose_up_write_v210_1', false]])
	r.d.touch(outs[1], 'readonly', 2)
	expect('and asking for that frame launches nothing more', r.names().length, 3)
}

// 9. ADVICE r4: a failure while the jobs a new job displaces are run leaves nothing of the new job behind
{
	const r = rig()
	const L = r.loader()
	const a = r.image('a')
	const b = r.image('b')
	const dst = r.image('dst')
	const m = r.buffer(48, undefined, 'matrix')
	r.d.record(r.P.read, Object.assign({ input: r.v210('s'), output: a, width: r.W }, L), 1)
	r.d.record(r.P.other, { input: a, output: dst }, 1) // a pending producer of dst that depends on the read ...
	r.native.refuse = (name) => name === 'read' // ... which will fail
	let msg = null
	try { r.d.record(Object.assign({}, r.P.other, { name: 'paint_region' }), { input: b, output: dst }, 1) } catch (e) { msg = e.message } // a partial writer: dst's producer runs first
	expect('the new job is refused with the displaced job\'s failure', msg, 'read: refused')
	expect('and holds nothing: b is only its owner\'s, nobody reads it, nothing pending writes dst', [b.appRefs(), b._held, b._readers.length, dst._producer], [1, 0, 0, null])
	b.release()
	expect('b goes with its owner', b.alive(), false)
}

// 10. frames are launched at the end of the tick that posted them; channels of one shape go to the device in one launch
{
	const r = rig({ early: true, batch: true })
	const m = [r.buffer(48, undefined, 'm0'), r.buffer(48, undefined, 'm1')]
	m[1].fill(3)
	const channel = (tag, fill, lines) => { // a channel of its own Loader (equal contents: fill) - two placed layers -> combine_2 -> write
		const L = r.loader(fill)
		const img = [0, 1].map((i) => { const im = r.image(`${tag}u${i}`); r.d.record(r.P.read, Object.assign({ input: r.v210(`${tag}s${i}`), output: im, width: r.W }, L), 1); return im })
		const placed = img.map((im, i) => { const t = r.image(`${tag}t${i}`); r.d.record(r.P.transform, { input: im, transformMatrix: m[i], output: t }, 1); return t })
		const comb = r.image(`${tag}comb`)
		r.d.record(r.P.combine2, { l0In: placed[0], l1In: placed[1], output: comb }, 1)
		const out = r.v210(`${tag}out`)
		r.d.record(r.P.write, Object.assign({ input: comb, output: out, width: r.W, interlace: 0 }, r.saver), 1)
		;[...img, ...placed, comb].forEach((b) => b.release())
		return out
	}
	const outs = [channel('A', 1), channel('B', 1), channel('C', 5)] // C: another colour recipe
	expect('nothing is launched while the tick records', r.names(), [])
	setImmediate(() => {
		expect('end of the tick: A and B in one launch, C (another Loader) on its own', r.names().sort(), ['batch:chan_compose_v210_2+chan_compose_v210_2', 'chan_compose_v210_2'])
		const batch = r.launches.find((l) => l[0].startsWith('batch'))
		const at = (job, name) => batch[3][job][batch[2].split(';')[job].split(',').indexOf(name)]
		expect('the batch names ONE Loader / Saver: the first job\'s buffers', ['colMatrix', 'gammaLut', 'gamutMatrix', 'outColMatrix', 'outGammaLut'].map((k) => at(0, k) === at(1, k)), [true, true, true, true, true])
		expect('every frame is made: nothing recorded is left', [r.d.pending.size, r.d.stats.fused, r.d.stats.launched], [0, 3, 2])
		outs.forEach((o) => r.d.touch(o, 'readonly', 2))
		expect('asking for the frames afterwards launches nothing', r.launches.length, 2)
		// a frame asked for BEFORE the tick ends takes the others of its shape with it
		const r2 = rig({ early: true, batch: true })
		const L = r2.loader()
		const m2 = r2.buffer(48, undefined, 'm')
		const outs2 = [0, 1, 2].map((c) => {
			const im = r2.image(`u${c}`)
			r2.d.record(r2.P.read, Object.assign({ input: r2.v210(`s${c}`), output: im, width: r2.W }, L), 1)
			const t = r2.image(`t${c}`)
			r2.d.record(r2.P.transform, { input: im, transformMatrix: m2, output: t }, 1)
			const out = r2.v210(`out${c}`)
			r2.d.record(r2.P.write, Object.assign({ input: t, output: out, width: r2.W, interlace: 0 }, r2.saver), 1)
			return out
		})
		r2.d.touch(outs2[1], 'readonly', 2)
		expect('asked for in the middle of the tick: all three channels\' frames in one launch', r2.names(), ['batch:chan_compose_v210_1+chan_compose_v210_1+chan_compose_v210_1'])
		// the batch refused: every frame on its own
		const r3 = rig({ early: true, batch: true })
		const L3 = r3.loader()
		const m3 = r3.buffer(48, undefined, 'm')
		const outs3 = [0, 1].map((c) => {
			const im = r3.image(`u${c}`)
			r3.d.record(r3.P.read, Object.assign({ input: r3.v210(`s${c}`), output: im, width: r3.W }, L3), 1)
			const t = r3.image(`t${c}`)
			r3.d.record(r3.P.transform, { input: im, transformMatrix: m3, output: t }, 1)
			const out = r3.v210(`out${c}`)
			r3.d.record(r3.P.write, Object.assign({ input: t, output: out, width: r3.W, interlace: 0 }, r3.saver), 1)
			return out
		})
		r3.native.refuse = (name) => name === 'batch'
		r3.d.touch(outs3[0], 'readonly', 2)
		expect('a refused batch: the frames one launch each', [r3.names(), r3.d.stats.fallbacks], [['chan_compose_v210_1', 'chan_compose_v210_1'], 1])
		// a batch that fails HALF WAY (a later group refused at its launch, ph_run_programs_progress = 2): the two frames already on the device
		// are done, only the third is launched on its own - nothing twice
		const r4 = rig({ early: true, batch: true })
		const L4 = r4.loader()
		const m4 = r4.buffer(48, undefined, 'm')
		const outs4 = [0, 1, 2].map((c) => {
			const im = r4.image(`u${c}`)
			r4.d.record(r4.P.read, Object.assign({ input: r4.v210(`s${c}`), output: im, width: r4.W }, L4), 1)
			const t = r4.image(`t${c}`)
			r4.d.record(r4.P.transform, { input: im, transformMatrix: m4, output: t }, 1)
			const out = r4.v210(`out${c}`)
			r4.d.record(r4.P.write, Object.assign({ input: t, output: out, width: r4.W, interlace: 0 }, r4.saver), 1)
			return out
		})
		r4.native.refuse = (name) => name === 'batch'
		r4.native.runProgramsProgress = () => 2
		r4.d.touch(outs4[0], 'readonly', 2)
		expect('a batch refused after two of its three frames: one more launch, for the third', [r4.names(), r4.d.stats.fallbacks, r4.d.stats.batched, (outs4.forEach((o) => r4.d.touch(o, 'readonly', 2)), r4.names().length)],
			[['chan_compose_v210_1'], 1, 2, 1])
		// profile contexts: the device time of the launch that makes a frame is shared out over the frame's jobs, into the RunTimings objects they
		// were handed when they were recorded (clJobQueue.ts:121-138 keeps them, :159-215 prints them after the batch)
		const r5 = rig({ early: true })
		const L5 = r5.loader()
		const m5 = r5.buffer(48, undefined, 'm')
		const im5 = r5.image('u'), t5 = r5.image('t'), out5 = r5.v210('out')
		const rows = [r5.d.record(r5.P.read, Object.assign({ input: r5.v210('s'), output: im5, width: r5.W }, L5), 1),
			r5.d.record(r5.P.transform, { input: im5, transformMatrix: m5, output: t5 }, 1),
			r5.d.record(r5.P.write, Object.assign({ input: t5, output: out5, width: r5.W, interlace: 0 }, r5.saver), 1)]
		expect('recorded jobs report zeros until their frame has run', rows.map((t) => t.kernelExec), [0, 0, 0])
		r5.d.timedBegin()
		r5.d.touch(out5, 'readonly', 2)
		r5.d.timedEnd(100, null)
		expect('one fused launch of 100 us over read (3 parts), transform (2), write (3): every row non-zero, the rows sum to the launch',
			[r5.names(), rows.map((t) => t.kernelExec)], [['chan_compose_v210_1'], [37, 25, 38]])
		// two channels of 1080i sources posting their field pairs in one tick: their Yadif windows share ONE launch of the de-interlacing reader
		// (the reference's four channels are all 1080i: src/index.ts:45-71), then each channel's two frames are one compositor launch
		// (PHANERON_FIELD_BATCH=1's form: r6, r7; the default - channel by channel, each reader followed at once by its two fields' compositor launch - r8)
		const r6 = rig({ early: true })
		r6.d.fieldBatch = true
		const L6 = r6.loader()
		const m6 = r6.enlarging()
		const outs6 = []
		for (const ch of [0, 1]) {
			const win = [0, 1, 2].map((i) => { const im = r6.image(`c${ch}w${i}`); r6.d.record(r6.P.read, Object.assign({ input: r6.v210(`c${ch}s${i}`), output: im, width: r6.W }, L6), 1); return im })
			for (const parity of [0, 1]) {
				const y = r6.image(`c${ch}y${parity}`)
				r6.d.record(r6.P.yadif, { prev: win[0], cur: win[1], next: win[2], parity, tff: 1, skipSpatial: 0, output: y }, 1)
				const t = r6.image(`c${ch}t${parity}`)
				r6.d.record(r6.P.transform, { input: y, transformMatrix: m6, output: t }, 1)
				const out = r6.v210(`c${ch}out${parity}`)
				r6.d.record(r6.P.write, Object.assign({ input: t, output: out, width: r6.W, interlace: 0 }, r6.saver), 1)
				outs6.push(out)
			}
		}
		// the same on an addon with runPrograms: the two channels' compositor jobs (each both fields of its frame) go down in ONE call -
		// the library makes the four frames in one launch (ph_compose_up_write_v210_batch)
		const r7 = rig({ early: true, batch: true })
		r7.d.fieldBatch = true
		const L7 = r7.loader()
		const m7 = r7.enlarging()
		for (const ch of [0, 1]) {
			const win = [0, 1, 2].map((i) => { const im = r7.image(`c${ch}w${i}`); r7.d.record(r7.P.read, Object.assign({ input: r7.v210(`c${ch}s${i}`), output: im, width: r7.W }, L7), 1); return im })
			for (const parity of [0, 1]) {
				const y = r7.image(`c${ch}y${parity}`)
				r7.d.record(r7.P.yadif, { prev: win[0], cur: win[1], next: win[2], parity, tff: 1, skipSpatial: 0, output: y }, 1)
				const t = r7.image(`c${ch}t${parity}`)
				r7.d.record(r7.P.transform, { input: y, transformMatrix: m7, output: t }, 1)
				r7.d.record(r7.P.write, Object.assign({ input: t, output: r7.v210(`c${ch}out${parity}`), width: r7.W, interlace: 0 }, r7.saver), 1)
			}
		}
		const r8 = rig({ early: true, batch: true })
		const L8 = r8.loader()
		const m8 = r8.enlarging()
		for (const ch of [0, 1]) {
			const win = [0, 1, 2].map((i) => { const im = r8.image(`c${ch}w${i}`); r8.d.record(r8.P.read, Object.assign({ input: r8.v210(`c${ch}s${i}`), output: im, width: r8.W }, L8), 1); return im })
			for (const parity of [0, 1]) {
				const y = r8.image(`c${ch}y${parity}`)
				r8.d.record(r8.P.yadif, { prev: win[0], cur: win[1], next: win[2], parity, tff: 1, skipSpatial: 0, output: y }, 1)
				const t = r8.image(`c${ch}t${parity}`)
				r8.d.record(r8.P.transform, { input: y, transformMatrix: m8, output: t }, 1)
				r8.d.record(r8.P.write, Object.assign({ input: t, output: r8.v210(`c${ch}out${parity}`), width: r8.W, interlace: 0 }, r8.saver), 1)
			}
		}
		setImmediate(() => {
			expect('two 1080i channels in one tick, the default: channel by channel - its windows\' reader, then at once its two fields in one compositor launch',
				[r8.names(), r8.d.stats.unpacked || 0, Array.from(r8.d.pending).filter((n) => n.program.name === 'write').length],
				[['v210_yadif_pair_1', 'compose_up_write_v210_1', 'v210_yadif_pair_1', 'compose_up_write_v210_1'], 0, 0])
			expect('two 1080i channels in one tick: one reader launch for both windows, one compositor launch per channel (both fields)',
				r6.names(), ['v210_yadif_pair_2', 'compose_up_write_v210_1', 'compose_up_write_v210_1'])
			const b7 = r7.launches.find((l) => l[0].startsWith('batch'))
			expect('with runPrograms: the channels\' compositor jobs in one call, each with both fields, the packed fields not unpacked on the way',
				[r7.names(), b7 ? /packedRgb/.test(b7[2]) && /output2/.test(b7[2]) : false, r7.d.stats.unpacked || 0, r7.d.pending.size > 0 ? Array.from(r7.d.pending).filter((n) => n.program.name === 'write').length : 0],
				[['v210_yadif_pair_2', 'batch:compose_up_write_v210_1+compose_up_write_v210_1'], true, 0, 0])
			process.stdout.write(JSON.stringify({ checks, problems }) + '\n')
		})
	})
}
