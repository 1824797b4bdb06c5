'use strict'
// defer.js - runProgram as a RECORDING, so that the reference's per-frame job stream reaches the MI355X as the fused
// kernels instead of one launch and one f32 frame per operator.
//
// The reference (and anything written against nodencl) asks for a channel's frame operator by operator, from
// different valves and at different times: ToRGBA `read` in the producer (io.ts:100-118), `transform` in the Mixer
// (producer/mixer.ts:209-223), `transition_*` in the Transitioner (transitioner.ts:165-176), `combine_N` in the
// Combiner (combiner.ts:219-254), `write` in the consumer (io.ts:150-164) - every arrow a full-size f32 RGBA frame
// through HBM (SURVEY 3.3).  The library has kernels that do the whole chain in one pass straight from the v210 words
// (ph_chan_compose_v210, ph_fused_v210_combine), but nobody who speaks nodencl calls them.  With
// `new clContext({ deferred: true })` (or PHANERON_DEFERRED=1) this layer makes the connection:
//
//   * runProgram does not launch.  It records a node - program, arguments, the buffers it reads and writes - and
//     resolves at once.  Every buffer knows the pending node that will produce it and the pending nodes that read it.
//     A node keeps a reference of its own on each of its buffers, so the owners' release() calls (the job callbacks of
//     clJobQueue.ts:133-137) cannot free what a recorded job still needs.
//   * The recording is forced only where a result becomes observable or an operand is about to change: hostAccess
//     'readonly' / downloadAsync / a route send of a buffer with a pending producer (the consumer mapping its output
//     frame, io.ts:166-174); hostAccess 'writeonly' of a buffer pending nodes still read (a new frame or a new
//     placement matrix into an old buffer: transform.ts:84-89); a new job writing into such a buffer.  waitFinish does
//     not force: nothing can tell a frame that sits in HBM from one that was never made.
//   * Forcing a v210 `write` looks at what produces its input: [combine_N of] per layer [transition_* of]
//     [transform of] (`read` of a v210 or planar YCbCr frame | an image that exists).  That shape is ONE launch of chan_compose_v210_N
//     (or fused_v210_combine_N when every layer is a plain read of the output's size) on the ORIGINAL v210 sources.
//     Anything else runs as recorded, producers first.  Both kernels are bit-identical to the chain of separate
//     operators (tests/test_chan_gpu.py, tests/test_hip_parity.py), so deferring changes no result.
//   * Intermediate images are never made unless somebody asks for them: their nodes stay pending as recipes.  When the
//     owner's last reference goes (release() finds only recorded jobs holding the buffer) and nobody pending reads it,
//     the recipe is dropped and its own operands are let go - the v210 source of a frame is recycled as soon as the
//     last thing that could still need it is gone.
//
// What changes for the caller: RunTimings of a deferred job are zeros.  A job's arguments are checked when it is recorded
// (ph_check_program: the checks of a launch without the launch), so a bad job is refused by runProgram as on a plain
// context.  Own design; nothing of this exists in the reference, whose OpenCL queue runs every job as posted.

const OUTPUT_ARG = /^(output|l\d+Out)/
const isOutputArg = (name) => { const c = name.charCodeAt(0); return (c === 111 || c === 108) && OUTPUT_ARG.test(name) } // 'o' | 'l'
const ZERO_TIMINGS = () => ({ dataToKernel: 0, kernelExec: 0, totalTime: 0 })
// Jobs that write EVERY byte of their output (a frame-sized operator, a progressive `write`): a pending producer of that buffer
// can be dropped unseen.  Anything else - a field `write`, a program this layer does not know - may fill only part of it: the
// producer runs first (ADVICE r3).
const WHOLE_OUTPUT = /^(read|write|yadif|transform|resize|combine_\d+|transition_dissolve|transition_wipe|mixer|wipe)$/
// parameter names of the fused programs, made once: `${prefix}${suffix}` per source (l0, l0Incoming, l0Mask ...)
const SOURCE_KEYS = new Map()
const sourceKeys = (prefix) => {
	let k = SOURCE_KEYS.get(prefix)
	if (!k) SOURCE_KEYS.set(prefix, (k = { In: `${prefix}In`, Packing: `${prefix}Packing`, InU: `${prefix}InU`, InV: `${prefix}InV`, ColMatrix: `${prefix}ColMatrix`,
		Matrix: `${prefix}Matrix`, Width: `${prefix}Width`, Height: `${prefix}Height`, Transition: `${prefix}Transition`, Mix: `${prefix}Mix` }))
	return k
}
const LAYER_PREFIX = [0, 1, 2, 3, 4, 5, 6, 7].map((i) => `l${i}`)
const LAYER_INCOMING = LAYER_PREFIX.map((p) => `${p}Incoming`)
const LAYER_MASK = LAYER_PREFIX.map((p) => `${p}Mask`)
const NONE = Object.freeze([])
const pushNew = (list, v) => { if (!list.includes(v)) list.push(v) }

class Deferral {
	constructor(ctx) {
		this.ctx = ctx // the clContext: _native, _ctx, queue, createProgram
		this.programs = new Map() // fused programs by `${name}|${w}|${h}`
		this.pending = new Set() // recorded nodes that have neither run nor been dropped
		this.lastReader = null // the Loader recipe of the newest v210 `read` (colMatrix / gammaLut / gamutMatrix buffers)
		this.launchedOn = new Map() // queue -> number of launches made on it
		this.orderedAt = new Map() // `${waiter}<${signal}` -> the signal queue's launch count the waiter is ordered behind
		this.stats = { recorded: 0, launched: 0, fused: 0, fusedNodes: 0, plain: 0, dropped: 0, fallbacks: 0, lastFallback: null }
		this.packFields = process.env.PHANERON_PACK_FIELDS !== '0' // de-interlaced fields as packed RGB while only the compositor reads them (_deinterlace)
		// PHANERON_FIELD_BATCH=1: the Yadif windows of ALL the channels of a tick in shared launches and their compositor frames four to a launch
		// (round 6's first form).  Measured on four 1080i channels (profiles/r06_node_interlaced.jsonl): the eight-window reader launch gains
		// 4 % on two of four, the four-frame compositor launch LOSES 33 % on two of two (113 against 2 x 42.5 us: eight 25 MB field images per
		// channel pair and tick leave the 256 MB MALL before their reader has finished) - so the default is channel by channel, each
		// channel's reader followed at once by its own two fields' compositor launch
		this.fieldBatch = process.env.PHANERON_FIELD_BATCH === '1'
		this.fieldTwin = new WeakMap() // a de-interlaced field image -> the other field of the same frame (set by the pair launch that made both)
		// Terminal wire-format `write` jobs that are recorded and not launched yet.  When somebody asks for one of them, the others whose
		// frames the same kernel can make in the same launch go with it (channels of one format in one context: src/index.ts:45-71).
		// With `earlyLaunch` they are launched at the END OF THE TICK that posted them (setImmediate) instead - for a host whose consumers
		// map their frames long after posting them.  Not the default: where the consumer asks right after its jobs' flush (the reference's
		// saveFrame, this repo's benches) the tick-end launch only adds its own bookkeeping (measured: 143 against 132 us per frame for
		// one 1080p channel, 92 against 82 for four - profiles/r05_node_bench.jsonl)
		this.terminals = []
		this.tickScheduled = false
		this.running = false // inside _runMany (a launch that forces another terminal write runs that one on its own)
		this._tick = () => {
			this.tickScheduled = false
			const list = this.terminals.filter((n) => n.state === 'pending')
			this.terminals = []
			// nobody is waiting for these frames yet: a failure stays with the buffers (_failed) for whoever asks, as everywhere
			try { this._runMany(list, null) } catch (e) { /* recorded with the buffers */ }
		}
		this.earlyLaunch = ctx.earlyLaunch === true
	}

	// ---- bookkeeping on buffers -----------------------------------------------------------------------------
	// (plain assignments: Object.defineProperty is a runtime call per field and buffer - a fifth of the recording's host time was here)
	static adopt(buf, fresh) { // fresh: a new buffer, or a parked one taken over (node/index.js createBuffer)
		if (buf._readers !== undefined && !fresh) return
		buf._digest = null
		buf._readers = NONE // pending nodes that read the buffer (a small array; NONE until there is one)
		buf._producer = null // the pending node that will write it
		buf._held = 0 // recorded nodes that hold it (ONE native reference stands for all of them: _hold)
		buf._failed = null // the error of the job that should have produced it (ADVICE r3: every later consumer sees it, not only the first)
		buf._packed = null // set: the image buffer holds PACKED f32 RGB (12 bytes per pixel) for now; the value is the queue its launch ran on (see _deinterlace)
	}
	// two parameter buffers that hold the same bytes (every producer's Loader makes its own LUT and matrices: loadSave.ts:50-99);
	// the host mirror is what hostAccess wrote, its digest is dropped when the buffer is written again (touch)
	static same(a, b) {
		if (a === b) return true
		if (!a || !b || a.length !== b.length) return false
		for (const x of [a, b])
			if (!x._digest) x._digest = require('crypto').createHash('sha1').update(x).digest('hex')
		return a._digest === b._digest
	}
	static sameRecipe(r, q) { return Deferral.same(r.colMatrix, q.colMatrix) && Deferral.same(r.gammaLut, q.gammaLut) && Deferral.same(r.gamutMatrix, q.gamutMatrix) }
	// The recorded nodes' hold on a buffer: a count beside the owners' own (buf._refs, node/index.js) - the buffer goes when both are
	// zero.  (29 addRef + 34 release calls into the addon per 4-layer frame in round 3, one pair per buffer in round 4, none now.)
	_hold(buf) { buf._held++ }
	_unhold(buf) { if (--buf._held === 0 && buf._refs === 0) buf._free() }
	_appRefs(buf) { return buf._refs }

	// ---- recording --------------------------------------------------------------------------------------------
	record(program, params, queue) {
		const ins = []
		const outs = []
		const names = Object.keys(params)
		const kept = {} // the job's own copy of its parameters (the caller may reuse its object)
		for (let k = 0; k < names.length; ++k) {
			const v = params[names[k]]
			kept[names[k]] = v
			if (typeof v !== 'object' || v === null) continue // (scalars: most of a job's parameters)
			if (!v._handle) { // a matrix (typed array) - or a plain Buffer, which is refused
				if (Buffer.isBuffer(v)) throw new Error(`runProgram: parameter '${names[k]}' is a plain Buffer, not an OpenCLBuffer`)
				continue
			}
			if (v._dead) throw new Error('runProgram: a buffer argument has already been released')
			if (v._readers === undefined) Deferral.adopt(v)
			pushNew(isOutputArg(names[k]) ? outs : ins, v)
		}
		// what runProgram would refuse - a missing argument, a buffer too small for the frame - is refused here, with the same
		// message, where the reference awaits it (clJobQueue.ts:126); nothing is enqueued (ph_check_program).  A job that looks
		// exactly like the last one this program accepted (names, buffer sizes and image dimensions, scalars, queue) is not asked again.
		if (!this._sameAsChecked(program, names, params, queue)) {
			this._launch(program, params, queue, true)
			program._checked = this._signature(names, params, queue)
		}
		const node = { program, params: kept, queue, ins, outs, state: 'pending', timings: null }
		for (let k = 0; k < ins.length; ++k) { const b = ins[k]; this._hold(b); if (b._readers === NONE) b._readers = [node]; else b._readers.push(node) }
		for (let k = 0; k < outs.length; ++k) if (!ins.includes(outs[k])) this._hold(outs[k])
		// a buffer this job overwrites: whoever still wants its present (or pending) contents goes first.  A pending producer is
		// dropped unseen only if this job is known to write the WHOLE buffer; a job that fills part of it (an interlaced `write`:
		// every other line) or a program this layer does not know runs the producer first.
		// (The job holds its buffers and is registered as their reader BEFORE this - dropping a displaced producer lets go of recipes
		// nobody reads, and this job's own operands must not be among them - so a failure here, e.g. the stored error of an unrelated
		// job that one of the displaced ones depended on, has to undo that: ADVICE r4.)
		// (asked once per program; not enumerable, so that a program object copied under another name asks again)
		if (program._whole === undefined) Object.defineProperty(program, '_whole', { value: WHOLE_OUTPUT.test(program.name), enumerable: false })
		const whole = program._whole && !params.interlace
		try {
			for (let k = 0; k < outs.length; ++k) {
				const o = outs[k]
				if (o._readers.length) for (const r of o._readers.slice()) if (r !== node) this._run(r)
				const p = o._producer
				if (p && whole && p.outs.length === 1 && !ins.includes(o)) { this.stats.dropped++; this._retire(p, 'dropped') } // its result would be overwritten unseen
				else if (p) this._run(p)
			}
		} catch (e) {
			this._retire(node, 'error') // never recorded: its holds and reader entries go; the caller's runProgram rejects
			throw e
		}
		for (let k = 0; k < outs.length; ++k) {
			const o = outs[k]
			if (o._packed != null) { if (whole) o._packed = null; else this._unpack(o) } // (a job that fills part of it needs the real image under its part)
			o._producer = node
			o._failed = null
			this._untwin(o) // (it is no longer the field image a pair launch made)
		}
		if (Deferral._isV210(program, 'read') && params.colMatrix && params.gammaLut && params.gamutMatrix)
			this.lastReader = { colMatrix: params.colMatrix, gammaLut: params.gammaLut, gamutMatrix: params.gamutMatrix }
		this.pending.add(node)
		this.stats.recorded++
		if (program.name === 'write' && program.format !== undefined) this._noteTerminal(node)
		// The job's RunTimings: zeros now (nothing has run).  The object is the job's own and is kept with it: on a `profile` context
		// the launch that finally makes the frame is timed, and its device time is shared out over the jobs it stood in for - the
		// reference keeps what runProgram returned and prints it after the batch has finished (clJobQueue.ts:121-138, 159-215)
		node.timings = ZERO_TIMINGS()
		return node.timings
	}
	// profile contexts (node/index.js _timedWrite): between begin and end every job that reaches the device - launched as recorded, or stood
	// in for by a fused launch - is noted; end(us) shares the measured device time out over them by a rough weight of each operator's
	// arithmetic, so that the rows of the reference's table are non-zero and sum to the launch's time
	timedBegin() { this.timed = [] }
	timedEnd(kernelExec, own) {
		const jobs = this.timed || []
		this.timed = null
		const WEIGHT = { read: 3, write: 3, yadif: 4, transform: 2, resize: 2 } // (everything else - combine_N, transitions, mixer, wipe: 1)
		const seen = new Set()
		const list = []
		for (const n of jobs) if (n.timings && !seen.has(n)) { seen.add(n); list.push(n) }
		let total = 0
		for (const n of list) total += WEIGHT[n.program.name] || 1
		let left = kernelExec
		list.forEach((n, i) => {
			const share = i === list.length - 1 ? left : Math.floor(kernelExec * (WEIGHT[n.program.name] || 1) / total)
			left -= share
			n.timings.kernelExec = share
			n.timings.totalTime = share
		})
		return own && seen.has(own) ? own.timings : null
	}
	_noteTerminal(node) {
		if (this.terminals.length >= 32) this.terminals = this.terminals.filter((n) => n.state === 'pending')
		this.terminals.push(node)
		if (this.earlyLaunch && !this.tickScheduled) { this.tickScheduled = true; setImmediate(this._tick) }
	}
	_untwin(buf) {
		const t = this.fieldTwin.get(buf)
		if (t) { this.fieldTwin.delete(buf); this.fieldTwin.delete(t) }
	}
	// what ph_check_program looks at, as a flat list: the names in order, per buffer its size and image dimensions, the scalars
	_signature(names, params, queue) {
		const sig = [queue, names.length]
		for (let k = 0; k < names.length; ++k) {
			const v = params[names[k]]
			sig.push(names[k])
			if (Buffer.isBuffer(v)) sig.push(v.length, v.imageDims ? v.imageDims.width : 0, v.imageDims ? v.imageDims.height : 0)
			else sig.push(v === undefined || v === null ? null : typeof v === 'boolean' ? (v ? 1 : 0) : v, -1, -1)
		}
		return sig
	}
	_sameAsChecked(program, names, params, queue) {
		const sig = program._checked
		if (!sig || sig[0] !== queue || sig[1] !== names.length) return false
		for (let k = 0, j = 2; k < names.length; ++k, j += 4) {
			const v = params[names[k]]
			if (sig[j] !== names[k]) return false
			if (Buffer.isBuffer(v)) {
				if (sig[j + 1] !== v.length || sig[j + 2] !== (v.imageDims ? v.imageDims.width : 0) || sig[j + 3] !== (v.imageDims ? v.imageDims.height : 0)) return false
			} else if (sig[j + 2] !== -1 || sig[j + 1] !== (v === undefined || v === null ? null : typeof v === 'boolean' ? (v ? 1 : 0) : v)) return false
		}
		return true
	}

	// the node has run, or will never have to: take it out of the graph and let go of its buffers
	_retire(node, state) {
		node.state = state
		this.pending.delete(node)
		const { ins, outs } = node
		for (let k = 0; k < outs.length; ++k) if (outs[k]._producer === node) outs[k]._producer = null
		for (let k = 0; k < ins.length; ++k) { const rd = ins[k]._readers; const at = rd.indexOf(node); if (at >= 0) { if (rd.length === 1) ins[k]._readers = NONE; else rd.splice(at, 1) } }
		// operands that are recipes themselves go with it if nobody else can ask for them (we still hold them here)
		for (let k = 0; k < ins.length; ++k) this._reap(ins[k])
		for (let k = 0; k < ins.length; ++k) this._unhold(ins[k])
		for (let k = 0; k < outs.length; ++k) if (!ins.includes(outs[k])) this._unhold(outs[k])
	}
	// a recipe nobody can ask for any more: none of its outputs has a pending reader or a reference outside the recording
	_reap(buf) {
		const p = buf._producer
		if (!p || p.state !== 'pending') return
		for (const o of p.outs) if (o._readers.length || this._appRefs(o) > 0) return
		this.stats.dropped++
		this._retire(p, 'dropped')
	}
	// the owner called release(): index.js tells us after the native count went down
	released(buf) {
		if (buf._held > 0) this._reap(buf)
	}

	// ---- where the recording meets the host -------------------------------------------------------------------
	// hostAccess / downloadAsync / route traffic on `queue`: 'readonly' needs the buffer's contents, 'writeonly' and
	// 'none' replace them.  Launches made here (and earlier ones the caller's waitFinish could not have covered,
	// because they had not been made yet) are ordered in front of the access ON THE DEVICE.
	touch(buf, dir, queue) {
		Deferral.adopt(buf)
		if (dir === 'readonly') { this.force(buf); if (buf._packed != null) this._unpack(buf) } // (whoever reads it on the host or sends it away gets the image it is declared as)
		else {
			buf._packed = null
			if (buf._digest) buf._digest = null
			this._untwin(buf)
			this.beforeWrite(buf)
			if (buf._producer) this.force(buf) // (a recorded result the host overwrites: run it rather than reason about partial writes)
			buf._failed = null
		}
		for (const [q, epoch] of this.launchedOn) {
			if (q === queue) continue
			const key = `${queue}<${q}`
			if (this.orderedAt.get(key) === epoch) continue
			this.ctx._native.queueWaitQueue(this.ctx._ctx, queue, q)
			this.orderedAt.set(key, epoch)
		}
	}

	// ---- forcing ------------------------------------------------------------------------------------------------
	// the buffer's contents are about to be replaced: every pending job that reads them runs now
	beforeWrite(buf) {
		if (!buf._readers || !buf._readers.length) return
		for (const r of buf._readers.slice()) this._run(r)
	}
	// make the buffer's contents real
	// (a buffer whose producer FAILED stays failed until somebody writes it again: every consumer gets the error, not only
	// the one whose request happened to run the job)
	force(buf) {
		if (buf._producer) this._run(buf._producer)
		if (buf._failed) throw buf._failed
	}
	// whatever is still recorded runs (clContext.flushDeferred; recipes whose images nobody holds were dropped already)
	forceAll() {
		for (const n of Array.from(this.pending)) this._run(n)
	}
	_run(node) {
		if (node.state !== 'pending') return
		if (!this.running && node.program.name === 'write' && this.terminals.length > 1) {
			// somebody needs this frame now: the other recorded frames the same launch can make go with it
			const others = this.terminals.filter((n) => n !== node && n.state === 'pending' && n.queue === node.queue)
			this.terminals = this.terminals.filter((n) => n.state === 'pending' && n !== node && n.queue !== node.queue)
			if (others.length) return this._runMany([node, ...others], node)
		}
		if (!this._fused(node)) this._plain(node)
	}
	// Several terminal writes at once.  `must` (or null) has to be done when this returns - fused, or as recorded; the others are
	// launched only if their chain folds (what does not fold stays recorded until somebody asks).  Frames the channel kernel (or the
	// headline kernel: plain reads) makes from sources of one recipe, at one size, go to the device as ONE launch (runPrograms ->
	// ph_chan_compose_batch / ph_fused_v210_combine_batch).
	_runMany(nodes, must) {
		this.running = true
		const plans = []
		let failure = null
		try {
			if (nodes.length > 2 && this.fieldBatch) this._deinterlaceAhead(nodes) // (two: the fields of one frame - their windows are one launch anyway)
			const covered = new Set() // writes that another plan's launch makes along with its own (the other field of a de-interlaced frame)
			for (const n of nodes) {
				if (n.state !== 'pending' || covered.has(n)) continue
				let plan = null
				try { plan = this._plan(n) } catch (e) { if (n === must) failure = e; continue }
				if (!plan) continue
				// a de-interlaced frame's two fields (planning has just launched its windows' reader): their compositor launch follows at
				// once, while the fields are still in the cache - not after the other channels' readers (fieldBatch: see the constructor)
				const twin = plan.up && !this.fieldBatch && plan.candidates[0][2]
				if (twin) { // (the other field's write is done with it - or still pending, and planned next)
					if (this._fresh(plan) && !this._commit(plan) && plan.node === must) this._plain(plan.node)
					continue
				}
				plans.push(plan)
			}
			// groups of plans one launch can take
			const groups = []
			for (const p of plans) {
				if (covered.has(p.node)) continue
				if (!p.batchable || !this.ctx._native.runPrograms) { groups.push([p]); continue }
				if (p.up) { // frames of the 2 x 2-block compositor: those of one Saver go down together, the library puts like ones into shared launches
					const twin = p.candidates[0][2]
					if (twin) covered.add(twin.node)
					let g = groups.find((v) => v[0].up && v[0].batchable && v.length < 8 && v[0].width === p.width && v[0].height === p.height && v[0].node.queue === p.node.queue &&
						Deferral.same(v[0].saver.outColMatrix, p.saver.outColMatrix) && Deferral.same(v[0].saver.outGammaLut, p.saver.outGammaLut))
					if (!g) groups.push((g = []))
					g.push(p)
					continue
				}
				let g = groups.find((v) => !v[0].up && v[0].batchable && v.length < 8 && v[0].width === p.width && v[0].height === p.height && v[0].node.queue === p.node.queue &&
					Deferral.sameRecipe(v[0].loader, p.loader) && Deferral.same(v[0].saver.outColMatrix, p.saver.outColMatrix) && Deferral.same(v[0].saver.outGammaLut, p.saver.outGammaLut))
				if (!g) groups.push((g = []))
				g.push(p)
			}
			for (const g of groups) {
				const live = g.filter((p) => this._fresh(p))
				for (const p of g) if (!live.includes(p) && p.node.state === 'pending' && p.node === must) { if (!this._fused(p.node)) this._plain(p.node) } // (planning another frame ran part of this one's chain: look again)
				if (live.length > 1 && this._batch(live)) continue
				for (const p of live) if (this._fresh(p) && !this._commit(p) && p.node === must) this._plain(p.node)  // (fresh: a batch that failed half way has made some)
			}
			if (must && must.state === 'pending' && !failure) { if (!this._fused(must)) this._plain(must) }
		} finally { this.running = false }
		if (failure) throw failure
	}
	// a plan is good as long as the write and every node it stands in for are still pending
	_fresh(plan) {
		if (plan.node.state !== 'pending') return false
		for (const u of plan.used) if (u.state !== 'pending') return false
		return true
	}
	// several channel frames in one launch: every plan's first candidate with the FIRST plan's Loader / Saver buffers (equal contents)
	_batch(plans) {
		// (frames of a group do not depend on each other - what a frame needs of another was made real when it was planned - so like
		// programs may stand together: the library puts CONSECUTIVE jobs of one kind and shape into a launch)
		plans = plans.map((p, i) => [p, i]).sort((a, b) => (a[0].candidates[0][0] < b[0].candidates[0][0] ? -1 : a[0].candidates[0][0] > b[0].candidates[0][0] ? 1 : a[1] - b[1])).map((v) => v[0])
		const first = plans[0].candidates[0][1]
		const progs = []
		const names = []
		const values = []
		for (const p of plans) {
			const [name, params] = p.candidates[0]
			progs.push(this._program(name, p.width, p.height)._handle)
			const nm = []
			const vs = []
			for (const k of Object.keys(params)) {
				let v = params[k]
				if (v === undefined || v === null) continue
				if ((k === 'colMatrix' || k === 'gammaLut' || k === 'gamutMatrix' || k === 'outColMatrix' || k === 'outGammaLut') && first[k]) v = first[k]
				// (a packed field image as a source of a batched frame: the real image first - unless the frame is the compositor's and was told)
				if (v && v._packed != null && !(params.packedRgb && /^l\d+In2?$/.test(k))) this._unpack(v)
				nm.push(k)
				vs.push(Buffer.isBuffer(v) ? v._handle : typeof v === 'boolean' ? (v ? 1 : 0) : v)
			}
			names.push(nm)
			values.push(vs)
		}
		const queue = plans[0].node.queue
		try {
			this.ctx._native.runPrograms(this.ctx._ctx, progs, names, values, queue)
		} catch (e) {
			this.stats.fallbacks++
			this.stats.lastFallback = `batch of ${plans.length}: ${e && e.message || e}`
			// a call refused at a launch has made the launches of the jobs before the failing group (ph_run_programs_progress): those
			// frames are done - the caller commits the rest one by one, not these a second time
			const made = this.ctx._native.runProgramsProgress ? this.ctx._native.runProgramsProgress() : 0
			if (made > 0) {
				this.stats.launched++
				this.stats.batched = (this.stats.batched || 0) + made
				this.launchedOn.set(queue, (this.launchedOn.get(queue) || 0) + 1)
				for (const p of plans.slice(0, made)) this._done(p, p.candidates[0][2] || null)
			}
			return false
		}
		this.stats.launched++
		this.stats.batched = (this.stats.batched || 0) + plans.length
		this.launchedOn.set(queue, (this.launchedOn.get(queue) || 0) + 1)
		for (const p of plans) this._done(p, p.candidates[0][2] || null) // (a compositor job that made both fields' frames: the twin's write is done too)
		return true
	}

	_launch(program, params, queue, checkOnly = false) {
		const names = []
		const values = []
		for (const name of Object.keys(params)) {
			const v = params[name]
			if (v === undefined || v === null) continue
			names.push(name)
			values.push(Buffer.isBuffer(v) ? v._handle : typeof v === 'boolean' ? (v ? 1 : 0) : v)
		}
		if (checkOnly) return this.ctx._native.runProgram(this.ctx._ctx, program._handle, names, values, queue, false, true)
		// packed field images (see _deinterlace) are images to everybody but the 2 x 2-block compositor that was told so
		for (let k = 0; k < names.length; ++k) {
			const v = params[names[k]]
			if (v && v._packed != null && !(params.packedRgb && /^l\d+In2?$/.test(names[k])) && program.name !== 'rgb_unpack') this._unpack(v)
		}
		this.stats.launched++
		this.launchedOn.set(queue, (this.launchedOn.get(queue) || 0) + 1)
		return this.ctx._native.runProgram(this.ctx._ctx, program._handle, names, values, queue, false)
	}
	// a fused launch that may be refused (a shape the kernel does not take): false = nothing was launched
	_try(program, params, queue) {
		if (process.env.PHANERON_DEFER_DEBUG) {
			const show = {}
			for (const k of Object.keys(params)) show[k] = Buffer.isBuffer(params[k]) ? `${params[k].owner}:${params[k].length}` : params[k]
			process.stderr.write(`deferred launch ${program.name} ${JSON.stringify(show)}\n`)
		}
		try {
			this._launch(program, params, queue)
			return true
		} catch (e) {
			this.stats.launched--
			this.launchedOn.set(queue, this.launchedOn.get(queue) - 1)
			this.stats.fallbacks++
			this.stats.lastFallback = `${program.name}: ${e && e.message || e}`
			return false
		}
	}
	// A de-interlaced field the pair launch wrote as packed f32 RGB into the application's RGBA image buffer (12 of its 16 bytes per
	// pixel used) becomes the image it is declared as: in place, on the queue that made it (ph_image_unpack_rgb).  Only when somebody
	// other than the 2 x 2-block compositor wants it - the host, a ROUTE, another kernel, a job run as recorded.
	_unpack(buf) {
		const queue = buf._packed
		if (!buf.imageDims) { buf._packed = null; return }
		this._launch(this._program('rgb_unpack', buf.imageDims.width, buf.imageDims.height), { image: buf }, queue) // (a failure leaves it marked: nobody takes packed pixels for an image)
		buf._packed = null
		this.stats.unpacked = (this.stats.unpacked || 0) + 1
	}
	_plain(node) {
		let failure = null
		try {
			for (const i of node.ins) {
				if (i._producer && i._producer !== node) this.force(i) // (an in-place job is its own operand's producer)
				else if (i._failed) throw i._failed
			}
			this._launch(node.program, node.params, node.queue)
		} catch (e) { failure = e }
		this.stats.plain++
		if (this.timed && !failure) this.timed.push(node)
		if (failure) for (const o of node.outs) o._failed = failure // whoever asks for them later is told, too
		this._retire(node, failure ? 'error' : 'done')
		if (failure) throw failure
	}

	// ---- the fused shapes -------------------------------------------------------------------------------------
	_program(name, width, height) {
		const key = `${name}|${width}|${height}`
		let p = this.programs.get(key)
		if (!p) {
			const handle = this.ctx._native.createProgram(this.ctx._ctx, 'phaneron:deferred', name, [width, height], 0)
			this.programs.set(key, (p = { name, globalWorkItems: [width, height], workItemsPerGroup: 0, _handle: handle }))
		}
		return p
	}
	static _isV210(program, which) { return program.name === which && program.format === 'v210' }
	static _frameOf(node) { // a `read` / `write` job's frame: packer.ts:58-66 geometry (a 4:2:0 work group handles a line pair: yuv420p.ts:381)
		const wipg = node.program.workItemsPerGroup
		const pairs = node.program.format === 'yuv420p' || node.program.format === 'nv12' ? 2 : 1
		const lines = wipg ? pairs * node.program.globalWorkItems[0] / wipg : 0
		return { width: node.params.width, lines }
	}

	// Layers that are de-interlaced sources: [transform of] `yadif` of three images that are pending reads of v210 frames (SDI) or of
	// planar 4:2:2 frames (yuv422p10 / yuv422p8: interlaced files).  The Yadif
	// valve posts both fields of a frame (yadif.ts:100-145, send_field: parity 1 ^ tff, then parity tff); the pair over one
	// window, for every such layer of the channel, is ONE launch of v210_yadif_pair_<k> on the v210 frames themselves
	// (unpack + filter, both fields: ph_kernels_deint.hip), bit-identical to read x 3 -> yadif x 2.  Anything that does not
	// fit (one field only, an image of the window already real, mixed sizes) is left to run as recorded.
	// packed: the fields are written as packed f32 RGB (12 bytes per pixel: a v210 source's alpha is 1) into the application's RGBA image
	// buffers - the caller has seen that the frame's every layer goes to the 2 x 2-block compositor, which reads 25 % less for it and drops
	// the alpha arithmetic (the library's best route for 1080i sources: DESIGN.md section 5).  Anybody else who asks gets them unpacked.
	_deinterlace(layerImages, packed) {
		const found = new Map() // cur image -> { windows of wire-format sources, the two yadif nodes }
		const PACKING = { v210: 0, yuv422p10: 1, yuv422p8: 2, yuv420p: 3, nv12: 4 } // SDI frames, or the planar frames of interlaced files (PH_FMT_*)
		// the wire-format frame behind an image of the window: [planes], if it is a pending ToRGBA of `fmt` with the window's Loader recipe
		const framesOf = (img, w, h, reader, fmt) => {
			const p = img && img._producer
			if (!p || p.state !== 'pending' || p.program.name !== 'read' || p.program.format !== fmt) return null
			const q = p.params
			const planes = fmt === 'v210' ? [q.input] : fmt === 'nv12' ? [q.inputY, q.inputC] : [q.inputY, q.inputU, q.inputV]
			if (planes.some((b) => !b)) return null
			// a frame that is itself the pending result of a recorded job (a packed frame made on the device and read back) is made
			// real first: the fused launch reads the planes, not the images (ADVICE r3)
			for (const b of planes) if (b._producer || b._failed) this.force(b)
			if (p.state !== 'pending') return null
			const f = Deferral._frameOf(p)
			if (f.width !== w || f.lines !== h || !img.imageDims || img.imageDims.width !== w || img.imageDims.height !== h) return null
			if (!Deferral.sameRecipe(reader, q)) return null
			return planes
		}
		for (let img of layerImages) {
			let p = img._producer
			if (p && p.state === 'pending' && p.program.name === 'transform') { img = p.params.input; p = img && img._producer }
			if (!p || p.state !== 'pending' || p.program.name !== 'yadif' || found.has(p.params.cur)) continue
			const { prev, cur, next } = p.params
			const rd = cur && cur._producer
			if (!prev || !cur || !next || !rd || !p.params.output || rd.program.name !== 'read' || PACKING[rd.program.format] === undefined) continue
			const fmt = rd.program.format
			const [w, h] = p.program.globalWorkItems
			const reader = { colMatrix: rd.params.colMatrix, gammaLut: rd.params.gammaLut, gamutMatrix: rd.params.gamutMatrix }
			if (!reader.colMatrix || !reader.gammaLut || !reader.gamutMatrix || w % (fmt === 'v210' ? 6 : 2)) continue
			const src = [framesOf(prev, w, h, reader, fmt), framesOf(cur, w, h, reader, fmt), framesOf(next, w, h, reader, fmt)]
			if (src.includes(null)) continue
			if (p.state !== 'pending' || [prev, cur, next].some((im) => !im._producer || im._producer.state !== 'pending')) continue // (forcing a plane ran one of them)
			// the other field of the same window
			let twin = null
			for (const r of cur._readers)
				if (r !== p && r.state === 'pending' && r.program.name === 'yadif' && r.params.prev === prev && r.params.cur === cur && r.params.next === next &&
					!!r.params.tff === !!p.params.tff && !!r.params.skipSpatial === !!p.params.skipSpatial && !!r.params.parity !== !!p.params.parity &&
					r.params.output && r.params.output !== p.params.output) twin = r
			if (!twin) continue
			const out = p.params.parity ? [twin.params.output, p.params.output] : [p.params.output, twin.params.output]
			found.set(cur, { src, out, nodes: [p, twin], w, h, reader, fmt, tff: p.params.tff ? 1 : 0, skip: p.params.skipSpatial ? 1 : 0 })
		}
		// one launch per group of windows that share size, field order and Loader recipe
		const groups = []
		for (const e of found.values()) {
			let g = groups.find((v) => v.length < 8 && v[0].w === e.w && v[0].h === e.h && v[0].tff === e.tff && v[0].skip === e.skip && v[0].fmt === e.fmt &&
				Deferral.sameRecipe(v[0].reader, e.reader))
			if (!g) groups.push((g = []))
			g.push(e)
		}
		for (const g of groups) {
			const e0 = g[0]
			const params = Object.assign({ tff: e0.tff, skipSpatial: e0.skip }, e0.reader)
			if (PACKING[e0.fmt]) params.packing = PACKING[e0.fmt]
			if (packed) params.packedRgb = 1
			g.forEach((e, i) => {
				;['Prev', 'Cur', 'Next'].forEach((which, f) => {
					params[`l${i}${which}`] = e.src[f][0]
					if (e.src[f].length >= 2) params[`l${i}${which}U`] = e.src[f][1] // (nv12: the interleaved CbCr plane)
					if (e.src[f].length === 3) params[`l${i}${which}V`] = e.src[f][2]
				})
				params[`l${i}Out0`] = e.out[0]; params[`l${i}Out1`] = e.out[1]
			})
			if (!this._try(this._program(`v210_yadif_pair_${g.length}`, e0.w, e0.h), params, e0.nodes[0].queue)) continue
			this.stats.fused++
			this.stats.fusedNodes += 2 * g.length
			for (const e of g) {
				this.fieldTwin.set(e.out[0], e.out[1])
				this.fieldTwin.set(e.out[1], e.out[0])
				if (packed) e.out[0]._packed = e.out[1]._packed = e0.nodes[0].queue
				if (this.timed) for (const y of e.nodes) this.timed.push(y)
				for (const y of e.nodes) this._retire(y, 'done')
			}
		}
	}

	// The OTHER field of a de-interlaced frame, if its chain is recorded too: a pending `write` of the same kind whose frame is
	// [combine_N of] transforms - the same placements - of the twin images of this frame's layers (fieldTwin).  Both fields then go to
	// the device as ONE compose_up_write_v210 launch (ph_compose_up_write_v210_pair: one table load, one partly filled last round of
	// wave steps instead of two).  layers: this frame's, as _fused found them ({ source, matrix }).
	_twinWrite(node, layers) {
		const n = layers.length
		const first = this.fieldTwin.get(layers[0].source)
		if (!first || !first._readers) return null
		// the twin images must be what the pair launch made: one that has a producer again (re-recorded since) or whose producer failed
		// is not a finished field image (ADVICE r4: the launch would have read it before - or instead of - its producer's run)
		for (const l of layers) { const t = this.fieldTwin.get(l.source); if (!t || t._producer || t._failed) return null }
		const sameWrite = (w) => w !== node && w.state === 'pending' && w.program.name === 'write' && w.program.format === node.program.format &&
			w.program.workItemsPerGroup === node.program.workItemsPerGroup && w.program.globalWorkItems[0] === node.program.globalWorkItems[0] &&
			(w.params.interlace || 0) === (node.params.interlace || 0) && w.params.output && w.params.output !== node.params.output &&
			Deferral.same(w.params.colMatrix, node.params.colMatrix) && Deferral.same(w.params.gammaLut, node.params.gammaLut)
		const placedTwin = (img, l) => { // img: a layer of the other frame; l: this frame's layer in that position
			const p = img && img._producer
			return p && p.state === 'pending' && p.program.name === 'transform' && p.params.input === this.fieldTwin.get(l.source) &&
				Deferral.same(p.params.transformMatrix, l.matrix) ? p : null
		}
		for (const t of first._readers) {
			if (!placedTwin(t.params && t.params.output, layers[0])) continue
			for (const c of t.params.output._readers) {
				let write = null
				let images = null
				if (n === 1 && sameWrite(c)) { write = c; images = [t.params.output] }
				else if (n > 1 && c.state === 'pending' && c.program.name === `combine_${n}` && c.params.output) {
					images = LAYER_PREFIX.slice(0, n).map((p) => c.params[sourceKeys(p).In])
					for (const w of c.params.output._readers) if (sameWrite(w) && w.params.input === c.params.output) write = w
				}
				if (!write || images.some((im, i) => !placedTwin(im, layers[i]))) continue
				return { node: write, output: write.params.output, sources: layers.map((l) => this.fieldTwin.get(l.source)) }
			}
		}
		return null
	}

	// node: a pending wire-format `write`.  true = the frame has been produced by one fused launch
	_fused(node) {
		const plan = this._plan(node)
		return plan ? this._commit(plan) : false
	}
	// what one fused launch would stand in for: null = the chain does not fold (or there is nothing to gain); else the candidates,
	// best first, and the nodes they replace.  Making a layer real may launch other recorded jobs (never this write).
	// A pending wire-format `write`'s frame, as far as its shape goes: the Writer's geometry against the image's, the output planes, the
	// images of its layers ([combine_N of] whatever).  null: not a frame a fused launch could make.
	_writeFrame(node) {
		// FromRGBA with a Writer whose frame the channel kernel can make: v210 (SDI), yuv422p8 / yuv422p10 (an encoder), rgba8 / bgra8 (the screen)
		const OUT = { v210: 0, yuv422p10: 1, yuv422p8: 2, yuv420p: 3, nv12: 4, rgba8: 5, bgra8: 6 }
		if (node.program.name !== 'write' || OUT[node.program.format] === undefined) return null
		const outFmt = OUT[node.program.format]
		const outRgb8 = outFmt >= 5
		const image = node.params.input
		const dims = image && image.imageDims
		const top = image && image._producer
		if (!dims || !top || top.state !== 'pending') return null
		const width = dims.width
		const height = dims.height
		const interlace = node.params.interlace || 0
		const geo = Deferral._frameOf(node)
		// (a 4:2:0 Writer's work groups are line PAIRS whatever the field mode: yuv420p.ts:381 - its geometry gives the whole height)
		if (geo.width !== width || geo.lines !== (interlace && outFmt !== 3 && outFmt !== 4 ? height / 2 : height)) return null
		const planarOut = outFmt >= 1 && outFmt <= 4 // planes: Y, U, V - nv12: Y and the interleaved CbCr plane (nv12.ts:374)
		const output = planarOut ? node.params.outputY : node.params.output
		if (!output || (!outRgb8 && !node.params.colMatrix) || !node.params.gammaLut) return null
		if (planarOut && (outFmt === 4 ? !node.params.outputC : !node.params.outputU || !node.params.outputV)) return null

		let layerImages = [image]
		const m = /^combine_(\d+)$/.exec(top.program.name)
		if (m) {
			layerImages = []
			for (let i = 0; i < Number(m[1]); ++i) {
				const l = top.params[`l${i}In`]
				if (!l) return null
				layerImages.push(l)
			}
		}
		if (layerImages.length > 8) return null
		// Will this frame be made by the 2 x 2-block compositor from de-interlaced fields alone?  Every layer [transform of] a pending yadif
		// output, placed as that compositor takes it (decided from the matrices' host copies, as `enlarged` in _plan): then the fields may
		// be written packed.
		const fieldLayer = (img) => {
			const t = img._producer
			if (!t || t.state !== 'pending' || t.program.name !== 'transform' || !t.params.input || !t.params.transformMatrix ||
				t.program.globalWorkItems[0] !== width || t.program.globalWorkItems[1] !== height) return false
			const f = t.params.input, y = f._producer, d = f.imageDims, mm = t.params.transformMatrix
			if (!y || y.state !== 'pending' || y.program.name !== 'yadif' || !d || !Buffer.isBuffer(mm) || mm.length < 36 || mm._producer) return false
			const q = new Float32Array(mm.buffer, mm.byteOffset, 9)
			if (q[1] !== 0 || q[3] !== 0 || !(q[0] > 0) || !(q[4] > 0)) return false
			if (!interlace && d.width === width && d.height === height && q[0] === 1 && q[4] === 1 && q[2] === 0 && q[5] === 0) return true
			return q[0] * d.width <= 0.99 * width && q[4] * d.height * (interlace ? 2 : 1) <= 0.99 * height
		}
		const packFields = this.packFields && !outFmt && width % 2 === 0 && layerImages.every(fieldLayer)
		return { outFmt, outRgb8, image, top, width, height, interlace, output, m, layerImages, packFields }
	}
	// The Yadif windows of ALL the frames about to be planned go to the device in shared launches (up to eight windows each): the reference's
	// four channels are all 1080i (src/index.ts:45-71) - their windows of a tick in one or two launches of the de-interlacing reader
	// instead of one per channel.  What each frame's own plan finds afterwards is the finished fields.
	_deinterlaceAhead(nodes) {
		const packed = []
		const plain = []
		for (const n of nodes) {
			if (n.state !== 'pending') continue
			let f = null
			try { f = this._writeFrame(n) } catch (e) { f = null }
			if (f) (f.packFields ? packed : plain).push(...f.layerImages)
		}
		if (packed.length) this._deinterlace(packed, true)
		if (plain.length) this._deinterlace(plain, false)
	}
	_plan(node) {
		const frame = this._writeFrame(node)
		if (!frame) return null
		const { outFmt, outRgb8, image, top, width, height, interlace, output, m, layerImages, packFields } = frame
		this._deinterlace(layerImages, packFields)

		// what each layer is made of.  The fused kernel applies ONE gamma table and gamut matrix (`reader`: a call is one colour
		// space) and one YCbCr matrix to its v210 sources (`packedCm`); a planar source may bring a matrix of its own (the 8-bit
		// formats' code ranges: `cm` on the source, compared with the call's when the launch is put together)
		let reader = null
		let packedCm = null
		const PLANAR = { yuv422p10: 1, yuv422p8: 2, yuv420p: 3, nv12: 4 } // PH_FMT_*
		const RGB8 = { rgba8: 5, bgra8: 6 } // stills and graphics: no YCbCr matrix, alpha in the data
		const used = new Set() // pending nodes the fused launch stands in for
		const sameSize = (img) => img.imageDims && img.imageDims.width === width && img.imageDims.height === height
		const materialised = (img) => { this.force(img); return img.imageDims ? { source: img } : null }
		// a wire-format plane the fused launch will read directly: if it is itself the pending result of a recorded job (a packed frame
		// made on the device and read back: write -> read), that job runs first - the launch reads the plane, not the image (ADVICE r3).
		// Whatever that ran is caught by the `used` re-check below.
		const realPlanes = (...planes) => { for (const b of planes) if (b && (b._producer || b._failed)) this.force(b) }
		const plainSource = (img) => { // an image as a sampled source: a pending ToRGBA of a wire-format frame, or the image itself
			const p = img._producer
			// v210 (SDI), or a planar frame as file decoders hand them over (ffmpegProducer.ts:398-412)
			const fmt = p && p.state === 'pending' && p.program.name === 'read' ? p.program.format : null
			if (RGB8[fmt]) {
				const q = p.params
				const f = Deferral._frameOf(p)
				const ok = q.input && q.gammaLut && q.gamutMatrix && img.imageDims && f.width === img.imageDims.width && f.lines === img.imageDims.height &&
					(!reader || (Deferral.same(reader.gammaLut, q.gammaLut) && Deferral.same(reader.gamutMatrix, q.gamutMatrix)))
				if (ok) {
					realPlanes(q.input)
					reader = reader || { colMatrix: null, gammaLut: q.gammaLut, gamutMatrix: q.gamutMatrix }
					used.add(p)
					return { source: q.input, packing: RGB8[fmt], width: f.width, height: f.lines, v210: true, planar: true, rgb8: true }
				}
			}
			if (fmt === 'v210' || PLANAR[fmt]) {
				const r = { colMatrix: p.params.colMatrix, gammaLut: p.params.gammaLut, gamutMatrix: p.params.gamutMatrix }
				const f = Deferral._frameOf(p)
				const q = p.params
				const planes = fmt === 'v210' ? q.input && f.width % 2 === 0 // (a line with a tail - 1280 - is the channel kernel's general instantiation)
					: q.inputY && (fmt === 'nv12' ? q.inputC : q.inputU && q.inputV) && f.width % 2 === 0 && (fmt === 'yuv422p10' || fmt === 'yuv422p8' || f.lines % 2 === 0)
				const ok = r.colMatrix && r.gammaLut && r.gamutMatrix && planes && img.imageDims && f.width === img.imageDims.width && f.lines === img.imageDims.height &&
					(!reader || (Deferral.same(reader.gammaLut, r.gammaLut) && Deferral.same(reader.gamutMatrix, r.gamutMatrix))) &&
					(fmt !== 'v210' || !packedCm || Deferral.same(packedCm, r.colMatrix))
				if (ok) {
					realPlanes(q.input, q.inputY, q.inputU, q.inputV, q.inputC)
					if (!reader) reader = r
					else if (!reader.colMatrix) reader.colMatrix = r.colMatrix
					if (fmt === 'v210') packedCm = packedCm || r.colMatrix
					used.add(p)
					return fmt === 'v210' ? { source: q.input, width: f.width, height: f.lines, v210: true }
						: { source: q.inputY, u: fmt === 'nv12' ? q.inputC : q.inputU, v: fmt === 'nv12' ? null : q.inputV, cm: r.colMatrix, packing: PLANAR[fmt], width: f.width, height: f.lines, v210: true, planar: true }
				}
			}
			return materialised(img)
		}
		const placed = (img) => { // [transform of] a plain source, shown at the output's size
			const p = img._producer
			if (p && p.state === 'pending' && p.program.name === 'transform' && p.params.input && p.params.transformMatrix &&
				p.program.globalWorkItems[0] === width && p.program.globalWorkItems[1] === height) {
				const s = plainSource(p.params.input)
				if (!s) return null
				used.add(p)
				return Object.assign(s, { matrix: p.params.transformMatrix })
			}
			if (!sameSize(img)) return null
			const s = plainSource(img)
			return s && (s.v210 ? (s.width === width && s.height === height ? s : null) : s)
		}
		const layers = []
		for (const img of layerImages) {
			const p = img._producer
			const kind = p && p.state === 'pending' ? p.program.name : ''
			// (`mixer` - mix.ts:30-45 - is transition_dissolve's arithmetic under another name: fma(in0, mix, in1 * (1 - mix)))
			if (kind === 'transition_dissolve' || kind === 'mixer' || kind === 'transition_wipe') {
				const wipe = kind === 'transition_wipe'
				if (!p.params.input0 || !p.params.input1 || (wipe && !p.params.maskIn) || !sameSize(img)) return null
				const l = placed(p.params.input0)
				const incoming = l && placed(p.params.input1)
				const mask = incoming && wipe ? placed(p.params.maskIn) : null
				if (!l || !incoming || (wipe && !mask)) return null
				used.add(p)
				l.transition = { wipe, mix: wipe ? 0 : Number(p.params.mix), incoming, mask }
				layers.push(l)
			} else {
				const l = placed(img)
				if (!l) return null
				layers.push(l)
			}
		}
		if (m) used.add(top)
		// making a layer real may have run a producer another layer was going to stand in for: look again, with that image real
		for (const u of used) if (u.state !== 'pending') return node.state === 'pending' ? this._plan(node) : null
		if (node.state !== 'pending') return null

		const n = layers.length
		const anyV210 = layers.some((l) => l.v210 || (l.transition && (l.transition.incoming.v210 || (l.transition.mask && l.transition.mask.v210))))
		if (!used.size) return null // every layer is a finished image taken as it is: the recorded write is as good
		const saver = { outGammaLut: node.params.gammaLut }
		if (!outRgb8) saver.outColMatrix = node.params.colMatrix
		if (outFmt) Object.assign(saver, { outPacking: outFmt }, outFmt === 4 ? { outputC: node.params.outputC } : outFmt < 5 ? { outputU: node.params.outputU, outputV: node.params.outputV } : {})
		const candidates = [] // [program name, params], best first; the library refuses the shapes a kernel does not take
		if (!outFmt && !interlace && layers.every((l) => l.v210 && !l.planar && !l.matrix && !l.transition)) {
			const params = Object.assign({ output }, reader, saver)
			layers.forEach((l, i) => { params[`l${i}In`] = l.source })
			candidates.push([`fused_v210_combine_${n}`, params])
		}
		// finished images, placed: enlarged ones share their taps.  What the 2 x 2-block compositor takes (ph_kernels_up.hip compose_up_eligible:
		// no rotation, no mirroring, less than 0.99 source texels per output pixel and written row) is decided here from the matrices' host
		// copies - a refused launch is an exception through the addon, and a field shown at its own size is the everyday case (1080i on 1080)
		const enlarged = (l) => {
			const m = l.matrix, d = l.source.imageDims
			if (!m || !d || m.length < 36) return false
			const f = new Float32Array(m.buffer, m.byteOffset, 9)
			if (f[1] !== 0 || f[3] !== 0 || !(f[0] > 0) || !(f[4] > 0)) return false
			if (!interlace && d.width === width && d.height === height && f[0] === 1 && f[4] === 1 && f[2] === 0 && f[5] === 0) return true // the default fill of a frame-size image
			return f[0] * d.width <= 0.99 * width && f[4] * d.height * (interlace ? 2 : 1) <= 0.99 * height
		}
		if (!outFmt && !anyV210 && width % 2 === 0 && layers.every((l) => l.matrix && !l.transition && enlarged(l))) {
			const params = Object.assign({ output, interlace }, saver)
			// packed field images: all of them, or none (the compositor takes one image format per launch)
			const packedLayers = layers.every((l) => l.source._packed != null)
			if (!packedLayers) for (const l of layers) if (l.source._packed != null) this._unpack(l.source)
			if (packedLayers) params.packedRgb = 1
			layers.forEach((l, i) => {
				params[`l${i}In`] = l.source; params[`l${i}Matrix`] = l.matrix
				if (packedLayers) { params[`l${i}Width`] = l.source.imageDims.width; params[`l${i}Height`] = l.source.imageDims.height }
			})
			let twin = this._twinWrite(node, layers) // the frame's other field, recorded too: both in one launch
			if (twin && !twin.sources.every((im) => (im._packed != null) === packedLayers)) twin = null
			if (twin) {
				const both = Object.assign({ output2: twin.output }, params)
				twin.sources.forEach((im, i) => { both[`l${i}In2`] = im })
				candidates.push([`compose_up_write_v210_${n}`, both, twin])
			}
			candidates.push([`compose_up_write_v210_${n}`, params])
		}
		// the channel kernel wants a whole Loader recipe even if no layer turns out to need its YCbCr matrix (packed RGB and image layers only)
		const anyCm = packedCm || (reader && reader.colMatrix) || (this.lastReader && this.lastReader.colMatrix)
		const loader = reader ? (anyCm ? { colMatrix: anyCm, gammaLut: reader.gammaLut, gamutMatrix: reader.gamutMatrix } : null) : this.lastReader
		if (loader) {
			const params = Object.assign({ output, interlace }, loader, saver)
			const put = (prefix, s) => {
				const key = sourceKeys(prefix)
				params[key.In] = s.source
				if (s.planar) params[key.Packing] = s.packing
				if (s.planar && !s.rgb8) {
					params[key.InU] = s.u
					if (s.v) params[key.InV] = s.v
					if (!Deferral.same(s.cm, loader.colMatrix)) params[key.ColMatrix] = s.cm
				}
				if (s.matrix) params[key.Matrix] = s.matrix
				if (s.v210) { params[key.Width] = s.width; params[key.Height] = s.height }
			}
			layers.forEach((l, i) => {
				const li = LAYER_PREFIX[i]
				put(li, l)
				if (l.transition) {
					const key = sourceKeys(li)
					params[key.Transition] = l.transition.wipe ? 2 : 1
					if (!l.transition.wipe) params[key.Mix] = l.transition.mix
					put(LAYER_INCOMING[i], l.transition.incoming)
					if (l.transition.wipe) put(LAYER_MASK[i], l.transition.mask)
				}
			})
			candidates.push([`chan_compose_v210_${n}`, params])
		}
		if (!candidates.length) return null
		// one call can take it together with other channels' frames: plain reads of the output's size (the headline kernel's batch form),
		// or the channel kernel as the only candidate making a v210 frame (the batch kernel for v210 / image sources)
		// (frames from planar / packed-RGB clips go along in the same call: the library runs those it cannot put into a shared launch in their turn,
		// and makes the ones of enlarged clips of one shape together)
		// ... or frames of the 2 x 2-block compositor (de-interlaced fields at their own size or enlarged: several 1080i channels in a tick)
		const up = candidates[0][0].startsWith('compose_up_write_v210_')
		const batchable = up || candidates[0][0].startsWith('fused_v210_combine_') || (candidates.length === 1 && candidates[0][0].startsWith('chan_compose_v210_') && !outFmt)
		return { node, candidates, used, n, width, height, batchable, up, loader, saver }
	}
	// launch the first candidate the library takes; false = none (nothing was launched)
	_commit(plan) {
		if (!this._fresh(plan)) return plan.node.state === 'pending' ? this._fused(plan.node) : true
		for (const [name, params, twin] of plan.candidates) {
			if (!this._try(this._program(name, plan.width, plan.height), params, plan.node.queue)) continue
			this._done(plan, twin)
			return true
		}
		return false
	}
	_done(plan, twin) {
		if (this.timed) { for (const u of plan.used) this.timed.push(u); this.timed.push(plan.node) }
		if (twin && twin.node.state === 'pending') { // the other field's frame came out of the same launch
			this.stats.fusedNodes += 1 + (plan.n > 1 ? 1 : 0) + plan.n
			this._retire(twin.node, 'done')
		}
		this.stats.fused++
		this.stats.fusedNodes += plan.used.size + 1
		this._retire(plan.node, 'done') // the producers it stood in for stay recipes until nobody can ask for their images
	}
}

module.exports = { Deferral }
