'use strict'
// node/index.js - the nodencl-shaped JS surface over the N-API addon (node/ph_napi.c), i.e.
// what phaneron's src/index.ts:94-108 gets from `new nodenCLContext(...)`: clContext with
// initialise / getPlatformInfo / createBuffer / createProgram / runProgram / waitFinish /
// queue / logBuffers, and OpenCLBuffer objects that ARE node Buffers with hostAccess /
// addRef / release and free-form timestamp fields.  Call sites: SURVEY.md 8(b).
const path = require('path')
const { Deferral } = require('./defer.js')

let addon = null
function loadAddon() {
	if (!addon) {
		try {
			addon = require(path.join(__dirname, 'phaneron_napi.node'))
		} catch (e) {
			throw new Error(`phaneron HIP addon not available (${e.message}); build it with python node/build.py - there is no OpenCL or CPU fallback`)
		}
	}
	return addon
}

const ACCESS = { readonly: 0, writeonly: 1, readwrite: 2 }
const SVM = { none: 0, coarse: 1, fine: 2 }
const HOSTDIR = { readonly: 0, writeonly: 1, none: 2 }

// The OpenCLBuffer members the reference uses, on the node Buffer the addon returned (the pinned host mirror of the device buffer).
// The methods are shared functions on `this`, reached through a per-context prototype (bufferPrototype below); the per-buffer fields
// are plain assignments: the reference makes a fresh destination per job and frame (io.ts:64-72), and five closures plus an
// Object.defineProperty per buffer showed in the profile of the recording context (node/test/defer_host_bench.js).
function hostAccess(dir, queue, src) {
	if (!(dir in HOSTDIR)) return Promise.reject(new Error(`hostAccess: unknown direction '${dir}'`))
	if (Buffer.isBuffer(queue)) { src = queue; queue = 0 } // hostAccess(dir, src)
	if (this._dead) return Promise.reject(new Error('hostAccess on a released buffer'))
	if (this._deferral) {
		try {
			this._deferral.touch(this, dir, queue || 0)
		} catch (e) { return Promise.reject(e) }
	}
	this._unsettled = true // (its mirror has been in use: a parked buffer's next owner asks the library to settle it, see createBuffer)
	return this._native.hostAccess(this._handle, HOSTDIR[dir], queue || 0, src)
}
// Reference counting is done HERE: a buffer holds one native reference from createBuffer to the moment its last owner AND the last
// recorded job that names it (node/defer.js: buf._held) have let go.  addRef / release / refCount were a call into the addon each -
// 35 calls per 4-layer frame of the recording context, 20 of them made by the recording itself.
// _dead: let go for good (a later use is refused with the messages the addon has for a released handle).
function addRef() { if (this._dead) throw new Error('addRef on a released buffer'); ++this._refs }
function release() {
	if (this._dead || this._refs <= 0) throw new Error('release on a released buffer')
	--this._refs
	if (this._held > 0) { if (this._deferral) this._deferral.released(this); return } // recorded jobs still need it: the last of them lets it go
	if (this._refs === 0) this._free()
}
function refCount() { return this._dead ? 0 : this._refs }
// The buffer goes.  Frames and images are PARKED instead - the JS object, its handle, its device block and its pinned mirror stay as
// they are, and the next createBuffer of the same size, kind and image dimensions takes them over whole: the reference makes a fresh
// destination per job and frame (io.ts:64-72, mixer.ts:196, combiner.ts:230), five to ten a frame, and a createBuffer is 3.5 us
// (pool lookup, node Buffer, external handle, decoration) where taking a parked one is 0.1.  What is parked is what the library's
// own pools would hold otherwise; clContext.trim() gives it back to them.
function free() {
	this._dead = true
	const park = this._park
	if (this._parkKey && park.on) {
		park.live -= this.length
		const room = Math.max(park.budget, park.peak)
		// over the budget the shapes nobody has asked for longest make room (a format change: 1080 -> 720 -> 2160 leaves lots of the
		// old sizes parked; the Map keeps its keys in the order they were last used - createBuffer moves a key to the end)
		if (park.parked + this.length > room) {
			for (const [key, list] of park.lists) {
				if (key === this._parkKey) continue
				while (list.length && park.parked + this.length > room) {
					const old = list.pop()
					park.parked -= old.length
					park.count--
					this._native.bufRelease(old._handle)
				}
				if (!list.length) park.lists.delete(key)
				if (park.parked + this.length <= room) break
			}
		}
		if (park.parked + this.length <= room) {
			let list = park.lists.get(this._parkKey)
			if (!list) { park.lists.set(this._parkKey, (list = [])); park.newest = this._parkKey } // (a new key goes in last)
			list.push(this)
			park.parked += this.length
			park.count++
			return
		}
	}
	this._native.bufRelease(this._handle)
}
// staging extension (not nodencl): device -> mirror on `queue` without a host wait; the bytes are
// valid after waitFinish(queue) or after an event recorded behind it has been awaited
function downloadAsync(queue) {
	if (this._dead) throw new Error('downloadAsync on a released buffer')
	if (this._deferral) this._deferral.touch(this, 'readonly', queue === undefined ? 2 : queue)
	this._unsettled = true
	return this._native.downloadAsync(this._handle, queue === undefined ? 2 : queue)
}
// What every buffer of a context shares lives on ONE prototype object per context (between the buffer and Buffer.prototype): the
// methods, the addon, the recording and the parking lot.  They are then neither own nor enumerable - console.log / deepEqual of a
// buffer do not walk into the context (ADVICE r4) - and a fresh buffer costs one Object.setPrototypeOf (0.08 us) instead of seven
// assignments; Object.defineProperty per field was measured at 3.8 us per buffer, five buffers a frame.
function bufferPrototype(native, deferral, park) {
	return Object.create(Buffer.prototype, {
		hostAccess: { value: hostAccess }, addRef: { value: addRef }, release: { value: release }, refCount: { value: refCount },
		downloadAsync: { value: downloadAsync }, _free: { value: free }, _native: { value: native }, _deferral: { value: deferral }, _park: { value: park }
	})
}
const newPark = (on, budgetMb) => ({ on, lists: new Map(), parked: 0, count: 0, live: 0, peak: 0, budget: budgetMb * 1048576,
	newest: null /* the key that is last in `lists` */, memo: { bytes: -1, w: 0, h: 0, dir: '', type: '', key: '' } })
function makeOpenCLBuffer(proto, created, numBytes, imageDims, owner, deferral, parkKey) {
	const buf = created.buffer
	Object.setPrototypeOf(buf, proto)
	buf._handle = created.handle
	buf._refs = 1
	buf._dead = false
	buf._parkKey = parkKey
	buf.numBytes = numBytes
	buf.owner = owner || ''
	buf.imageDims = imageDims
	buf.timestamp = 0
	buf.loadstamp = 0
	buf.creationTime = process.hrtime()
	if (deferral) Deferral.adopt(buf, true)
	return buf
}

// strictHandles (createBuffer): a parked buffer's block, mirror and native handle under a fresh JS object; the old one keeps `_dead`
function rewrap(proto, old) {
	const buf = Buffer.from(old.buffer, old.byteOffset, old.length) // (a view of the same memory: the addon's finalizer hangs on the ArrayBuffer)
	Object.setPrototypeOf(buf, proto)
	buf._handle = old._handle
	buf._parkKey = old._parkKey
	buf._gen = old._gen
	buf._unsettled = false
	buf.numBytes = old.numBytes
	old._handle = null
	return buf
}

class clContext {
	constructor(params) {
		params = params || {}
		this.platformIndex = params.platformIndex || 0
		this.deviceIndex = params.deviceIndex || 0
		this.overlapping = params.overlapping !== false
		this.profile = !!params.profile // true: runProgram records hipEvents and returns real RunTimings
		// extension: waitFinish first polls the queue on the JS thread for up to this many microseconds
		// before handing the wait to the libuv pool (a hand-off costs ~30 us; 0 = always hand off)
		this.spinWaitMicros = params.spinWaitMicros || 0
		// The recording context (node/defer.js): runProgram records instead of launching, and a frame's recorded operator chain
		// reaches the device as one fused kernel when its result is asked for.  It is the DEFAULT since round 4 - the reference
		// constructs the context itself (src/index.ts:94-107) and should get the fused kernels without asking; `deferred: false`
		// or PHANERON_DEFERRED=0 gives the launch-as-posted context (every job its own kernel, RunTimings measured when
		// `profile` is set, waitFinish(queue.process) a real wait)
		this.deferred = params.deferred === undefined ? process.env.PHANERON_DEFERRED !== '0' : !!params.deferred
		this._deferral = null
		// recording context: launch a frame's fused chain at the end of the tick that posted its terminal `write` instead of when somebody
		// asks for the frame (node/defer.js: for hosts whose consumers map their frames long after posting them)
		this.earlyLaunch = params.earlyLaunch === undefined ? process.env.PHANERON_EARLY_LAUNCH === '1' : !!params.earlyLaunch
		// released frames and images are parked and taken over whole by the next createBuffer of the same shape (free() above);
		// `recycleBuffers: false` or PHANERON_RECYCLE=0 returns every buffer to the library at once.  parkMb: what may stay parked
		// (default 4096 MiB - or as much as was ever in use at once, if that is more)
		this.recycleBuffers = params.recycleBuffers === undefined ? process.env.PHANERON_RECYCLE !== '0' : !!params.recycleBuffers
		this.parkMb = params.parkMb === undefined ? 4096 : params.parkMb
		// a parked buffer is taken over as the SAME JS object (0.1 us): a reference its previous owner kept past release() then aliases the new
		// owner's buffer.  strictHandles: true / PHANERON_STRICT_HANDLES=1 hands out a fresh object per takeover (+ ~1 us) and keeps the old one
		// refused for good - for running an application under test
		this.strictHandles = params.strictHandles === undefined ? process.env.PHANERON_STRICT_HANDLES === '1' : !!params.strictHandles
		this._addon = params.addon || null // tests: a stand-in for the N-API addon (node/test/defer_host_bench.js counts the calls the JS layer makes)
		this.queue = this.overlapping ? { load: 0, process: 1, unload: 2 } : { load: 1, process: 1, unload: 1 }
		this._ctx = null
		this._native = null
	}

	async initialise() {
		this._native = this._addon || loadAddon()
		this._ctx = this._native.createContext(this.deviceIndex)
		if (this.deferred) this._deferral = new Deferral(this)
		this._park = newPark(this.recycleBuffers, this.parkMb)
		this._bufferProto = bufferPrototype(this._native, this._deferral, this._park)
	}

	_need() {
		if (!this._ctx) throw new Error('clContext has not been initialised')
		return this._native
	}

	getPlatformInfo() {
		const info = this._need().contextInfo(this._ctx)
		const devices = []
		devices[this.deviceIndex] = { type: 'CL_DEVICE_TYPE_GPU', name: info.device, vendor: info.vendor }
		return { vendor: info.vendor, name: 'AMD HIP (gfx950)', devices }
	}

	async createBuffer(numBytes, bufDir, bufType, imageDims, owner) {
		const native = this._need()
		if (!(bufDir in ACCESS)) throw new Error(`createBuffer: unknown direction '${bufDir}'`)
		if (!(bufType in SVM)) throw new Error(`createBuffer: unknown svm type '${bufType}'`)
		const w = imageDims ? imageDims.width : 0
		const h = imageDims ? imageDims.height : 0
		// frames and images (not parameter buffers: a gamma table's registered LDS form goes with its native buffer) may be parked ones
		const park = this._park
		// (the key of the call before is kept: a channel asks for the same shape five to ten times a frame)
		let key = null
		if (park.on && (imageDims || numBytes >= 1048576)) {
			const m = park.memo
			if (m.bytes === numBytes && m.w === w && m.h === h && m.dir === bufDir && m.type === bufType) key = m.key
			else {
				key = `${numBytes}|${w}|${h}|${bufDir}|${bufType}`
				m.bytes = numBytes; m.w = w; m.h = h; m.dir = bufDir; m.type = bufType; m.key = key
			}
		}
		if (key) {
			park.live += numBytes
			if (park.live > park.peak) park.peak = park.live
			const list = park.lists.get(key)
			if (list && park.lists.size > 1 && park.newest !== key) { park.lists.delete(key); park.lists.set(key, list); park.newest = key } // most recently used last
			if (list && list.length) {
				let buf = list.pop()
				park.parked -= numBytes
				park.count--
				// Taken over whole, without a call into the library - unless its mirror or a ROUTE transfer may still be busy with it: a
				// download in flight when it was released (release after downloadAsync, before its waitFinish) would land in the next
				// owner's fill, RCCL may still be reading the device block on the communication stream, and a mirror the previous owner
				// filled but never handed back would be uploaded over the next owner's frame.  ph_buf_reuse settles all three (ADVICE r5).
				if (buf._unsettled) { native.bufReuse(buf._handle); buf._unsettled = false }
				// strictHandles: the next owner gets a NEW object over the same memory and the parked one stays dead - a reference the previous
				// owner kept is then refused ('... already released') as nodencl's are, instead of aliasing the new owner's buffer (ADVICE r5)
				if (this.strictHandles) buf = rewrap(this._bufferProto, buf)
				buf._gen = (buf._gen || 0) + 1
				buf._refs = 1
				buf._dead = false
				buf.owner = owner || ''
				buf.imageDims = imageDims
				buf.timestamp = 0
				buf.loadstamp = 0
				buf.creationTime = process.hrtime()
				if (this._deferral) Deferral.adopt(buf, true)
				return buf
			}
		}
		const created = native.createBuffer(this._ctx, numBytes, ACCESS[bufDir], SVM[bufType], w, h, owner || '')
		return makeOpenCLBuffer(this._bufferProto, created, numBytes, imageDims, owner, this._deferral, key)
	}
	// give the parked buffers back to the library (its own pools keep the blocks): before counting live buffers, or to shed memory
	trim() {
		const park = this._park
		if (!park) return
		for (const list of park.lists.values()) for (const buf of list) this._native.bufRelease(buf._handle)
		park.lists.clear()
		park.newest = null
		park.parked = 0
		park.count = 0
	}

	async createProgram(kernel, options) {
		const native = this._need()
		if (!options || !options.name) throw new Error('createProgram requires a kernel name')
		const gwi = options.globalWorkItems === undefined ? [] :
			typeof options.globalWorkItems === 'number' ? [options.globalWorkItems] : Array.from(options.globalWorkItems)
		const handle = native.createProgram(this._ctx, String(kernel), options.name, gwi, options.workItemsPerGroup || 0)
		const program = { name: options.name, globalWorkItems: gwi, workItemsPerGroup: options.workItemsPerGroup || 0, _handle: handle }
		// which wire format a `read` / `write` is for: the recording layer recognises the v210 ends of a channel's chain
		if (this._deferral && (options.name === 'read' || options.name === 'write')) program.format = native.resolveProgram(String(kernel), options.name).format
		return program
	}

	// The recording context's runProgram without the promise: node/jobs.js posts a batch's jobs one after another and an `await` per
	// job is a microtask turn each (24 a tick on a 1080i channel).  null = not a recording context, or a `profile` one whose terminal
	// write has to be launched and timed (runProgram does that); throws what runProgram would reject with.
	recordProgram(program, params, queue) {
		if (!this._deferral || this.profile) return null
		this._need()
		return this._deferral.record(program, params, queue === undefined ? this.queue.process : queue)
	}

	async runProgram(program, params, queue) {
		const native = this._need()
		if (this._deferral) {
			const q = queue === undefined ? this.queue.process : queue
			const zeros = this._deferral.record(program, params, q)
			// `profile` on a recording context: a frame's jobs reach the device as one launch, when its terminal `write` is posted (or
			// later).  That job is launched here and now and gets the launch's device time - the whole chain's - as its RunTimings; the
			// jobs folded into it keep their zeros (the reference prints what runProgram returns: clJobQueue.ts:159-215)
			if (this.profile && program.name === 'write' && program.format !== undefined && params.output) return this._timedWrite(params, q, zeros)
			return zeros
		}
		const names = []
		const values = []
		for (const name of Object.keys(params)) {
			const v = params[name]
			if (v === undefined || v === null) continue
			names.push(name)
			if (Buffer.isBuffer(v)) {
				if (!v._handle) throw new Error(`runProgram: parameter '${name}' is a plain Buffer, not an OpenCLBuffer`)
				if (v._dead) throw new Error('runProgram: a buffer argument has already been released')
				values.push(v._handle)
			} else {
				values.push(typeof v === 'boolean' ? (v ? 1 : 0) : v)
			}
		}
		const q = queue === undefined ? this.queue.process : queue
		return native.runProgram(this._ctx, program._handle, names, values, q, this.profile)
	}

	async _timedWrite(params, q, zeros) {
		const native = this._native
		const t0 = process.hrtime.bigint()
		const from = native.eventRecord(this._ctx, q, true)
		this._deferral.timedBegin()
		let kernelExec = 0
		try {
			this._deferral.touch(params.outputY || params.output, 'readonly', q) // runs the frame's chain (a failure rejects runProgram, as on a plain context)
			const to = native.eventRecord(this._ctx, q, true)
			await native.eventWait(to)
			kernelExec = native.eventElapsed(from, to)
		} finally {
			// the launch's device time over the jobs of the frame: each job's RunTimings object (handed out when it was recorded) gets its share
			this._deferral.timedEnd(kernelExec, null)
		}
		// the write's own row: its share (zeros' object is the write job's), the host time of the whole call as totalTime
		zeros.totalTime = Math.max(zeros.kernelExec, Number((process.hrtime.bigint() - t0) / 1000n))
		return zeros
	}

	async waitFinish(queue) {
		const native = this._need()
		const q = queue === undefined ? this.queue.process : queue
		// A recording context has nothing to wait for on the queue its jobs are recorded for: what was asked of it has not been
		// launched, and what has been launched (when a result was asked for) is ordered in front of whoever reads the result
		// on the device (defer.js touch) - waiting here would only stall the host behind the previous frame's kernel.
		// Uploads and downloads (the other queues) are waited for as ever.
		if (this._deferral && q === this.queue.process && this.queue.process !== this.queue.load) return
		if (this.spinWaitMicros > 0 && native.waitFinishSpin(this._ctx, q, this.spinWaitMicros)) return
		return native.waitFinish(this._ctx, q)
	}

	// deferred contexts: run whatever is still recorded (results nobody has asked for yet); returns the recording's counters
	flushDeferred() {
		if (this._deferral) this._deferral.forceAll()
		return this._deferral ? Object.assign({ pending: this._deferral.pending.size }, this._deferral.stats) : null
	}
	// deferred contexts: make these buffers' contents real now, as a consumer on the device would need them (a no-op otherwise)
	realise(...bufs) { if (this._deferral) for (const b of bufs) this._deferral.touch(b, 'readonly', this.queue.process) }
	// deferred contexts: wait until everything launched so far on `queue` has finished (what waitFinish does on a plain context)
	async drain(queue) {
		const native = this._need()
		return native.waitFinish(this._ctx, queue === undefined ? this.queue.process : queue)
	}
	deferredStats() { return this._deferral ? Object.assign({ pending: this._deferral.pending.size }, this._deferral.stats) : null }
	// extension (ph_trace_begin / ph_trace_end): which kernels the calls made between the two - on this thread - launched, '+'-joined
	// ("fused_v210_combine_lds", "chan_compose_batch<0>x4", "v210_yadif_pair+compose_up_write_v210" ...); dryRun: everything is chosen and
	// checked, nothing is enqueued (the frames are not made)
	traceBegin(dryRun) { this._need().traceBegin(!!dryRun) }
	traceEnd() { return this._need().traceEnd() }

	// ---- staging extensions (not nodencl; SURVEY 8f-3, node/staging.js) -----------------------------
	// later work on `waiter` starts only after everything enqueued so far on `signal` has finished
	queueWaitQueue(waiter, signal) {
		this._need().queueWaitQueue(this._ctx, waiter, signal)
	}

	// a point in `queue`: { wait(): Promise<void>, done(): boolean }
	recordEvent(queue) {
		const native = this._need()
		const ev = native.eventRecord(this._ctx, queue === undefined ? this.queue.process : queue)
		// wait(): like waitFinish, first polled on the JS thread for up to spinWaitMicros (a hand-off to the libuv pool and back costs tens of
		// microseconds, now and then milliseconds - a paced producer waiting for its ring slot sees every one of them)
		const spin = this.spinWaitMicros
		return {
			wait: () => {
				if (spin > 0) {
					const until = process.hrtime.bigint() + BigInt(spin) * 1000n
					do { if (native.eventDone(ev)) return Promise.resolve() } while (process.hrtime.bigint() < until)
				}
				return native.eventWait(ev)
			},
			done: () => native.eventDone(ev)
		}
	}

	// ---- ROUTE across GPUs (not nodencl; routeProducer.ts:63-126 with source and sink on different GPUs) ------
	// id: clContext.routeUniqueId() made on ONE rank and handed to the others by the application.
	// Returns { send(buf, peer), recv(buf, peer), group(fn), afterQueue(q), queueAfter(q), wait() }: RCCL send /
	// recv on a communication stream of its own; afterQueue / queueAfter order it against a queue ON THE DEVICE.
	openRoute(id, rank, world) {
		const native = this._need()
		const h = native.routeInit(this._ctx, id, rank, world)
		const q = (queue) => (queue === undefined ? this.queue.process : queue)
		return {
			rank, world,
			group: (fn) => { native.routeOp(h, 0); try { fn() } finally { native.routeOp(h, 1) } },
			afterQueue: (queue) => native.routeOp(h, 2, q(queue)),
			queueAfter: (queue) => native.routeOp(h, 3, q(queue)),
			wait: () => native.routeOp(h, 4),
			// (recording context: the frame's producer may be launched only now - after the caller's afterQueue() - so the
			// communication stream is ordered behind the process queue once more before the transfer is enqueued)
			send: (buf, peer) => {
				if (this._deferral) { this._deferral.touch(buf, 'readonly', this.queue.process); native.routeOp(h, 2, this.queue.process) }
				buf._unsettled = true
				return native.routeOp(h, 5, buf._handle, peer)
			},
			recv: (buf, peer) => {
				if (this._deferral) { this._deferral.touch(buf, 'writeonly', this.queue.process); native.routeOp(h, 2, this.queue.process) } // (readers of the old contents were launched just now)
				buf._unsettled = true
				return native.routeOp(h, 6, buf._handle, peer)
			}
		}
	}
	static routeUniqueId() { return loadAddon().routeUniqueId() }

	// library options (include/phaneron_hip.h ph_ctx_set_option): 'lds_lut', 'stream_images', 'stream_threshold_mb', 'host_pool_mb'
	setOption(name, value) { this._need().setOption(this._ctx, String(name), value | 0) }
	// the library's buffer counters, less what is parked here (nobody's: index.js free()): { liveBuffers, liveBytes, pooledBytes,
	// parkedBuffers, parkedBytes, pinnedInUse, pinnedPooled, pinnedPeak, pins }
	bufferStats() {
		const s = this._need().bufferStats(this._ctx)
		s.liveBuffers -= this._park.count
		s.liveBytes -= this._park.parked
		s.parkedBuffers = this._park.count
		s.parkedBytes = this._park.parked
		return s
	}
	logBuffers() {
		const s = this.bufferStats()
		console.log(`phaneron HIP buffers: ${s.liveBuffers} live (${s.liveBytes} bytes), ${s.pooledBytes} bytes pooled`)
		return s
	}
}

// ---- device-free helpers (no context, no GPU) ----------------------------------------------------
// which precompiled kernel createProgram(kernelSrc, {name}) selects: { kernel, format, how }
function resolveProgram(kernelSrc, name) {
	return loadAddon().resolveProgram(String(kernelSrc), name)
}

// the library's host colour maths (the numbers src/process/colourMaths.ts gives the reference's Loader / Saver)
const colour = {
	gamma2linearLUT: (spec) => loadAddon().gammaLut('gamma2linear', spec),
	linear2gammaLUT: (spec) => loadAddon().gammaLut('linear2gamma', spec),
	ycbcr2rgbMatrix: (spec, numBits = 10, lumaBlack = 64, lumaWhite = 940, chromaRange = 896) =>
		loadAddon().colourMatrix('ycbcr2rgb', spec, numBits, lumaBlack, lumaWhite, chromaRange),
	rgb2ycbcrMatrix: (spec, numBits = 10, lumaBlack = 64, lumaWhite = 940, chromaRange = 896) =>
		loadAddon().colourMatrix('rgb2ycbcr', spec, numBits, lumaBlack, lumaWhite, chromaRange),
	rgb2rgbMatrix: (srcSpec, dstSpec) => loadAddon().colourMatrix('rgb2rgb', srcSpec, dstSpec),
	// p: { flipH, flipV, anchorX, anchorY, scaleX, scaleY, offsetX, offsetY, rotate } as transform.ts:119-171 takes them
	transformMatrix: (width, height, p = {}) => loadAddon().transformMatrix(width, height, p.flipH ? 1 : 0, p.flipV ? 1 : 0,
		p.anchorX || 0, p.anchorY || 0, p.scaleX === undefined ? 1 : p.scaleX, p.scaleY === undefined ? 1 : p.scaleY,
		p.offsetX || 0, p.offsetY || 0, p.rotate || 0)
}

const FORMATS = ['v210', 'yuv422p10', 'yuv422p8', 'yuv420p', 'nv12', 'rgba8', 'bgra8']
// bytes per plane of a frame in `format`: the numBytes of the reference's Readers / Writers
function planeBytes(format, width, height) {
	const f = FORMATS.indexOf(format)
	if (f < 0) throw new Error(`unknown pack format '${format}'`)
	return loadAddon().planeBytes(f, width, height)
}

module.exports = { clContext, resolveProgram, colour, planeBytes, FORMATS, bufferPrototype, newPark }
