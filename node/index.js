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

// Decorate the node Buffer the addon returned (pinned host mirror of the device buffer) with the OpenCLBuffer members the
// reference uses.  The methods are shared functions on `this` and the fields plain assignments: the reference makes a fresh
// destination per job and frame (io.ts:64-72), and five closures plus an Object.defineProperty per buffer showed in the profile
// of the recording context (node/test/defer_host_bench.js).
function hostAccess(dir, queue, src) {
	if (!(dir in HOSTDIR)) return Promise.reject(new Error(`hostAccess: unknown direction '${dir}'`))
	if (Buffer.isBuffer(queue)) { src = queue; queue = 0 } // hostAccess(dir, src)
	if (this._deferral) {
		try {
			this._deferral.touch(this, dir, queue || 0)
		} catch (e) { return Promise.reject(e) }
	}
	return this._native.hostAccess(this._handle, HOSTDIR[dir], queue || 0, src)
}
function addRef() { this._native.bufAddRef(this._handle) }
// deferred contexts: recorded jobs hold ONE reference of their own while any of them needs the buffer (buf._held counts the jobs);
// the owner sees only its own
function release() { this._native.bufRelease(this._handle); if (this._deferral && this._held) this._deferral.released(this) }
function refCount() { return this._native.bufRefCount(this._handle) - (this._held > 0 ? 1 : 0) }
// staging extension (not nodencl): device -> mirror on `queue` without a host wait; the bytes are
// valid after waitFinish(queue) or after an event recorded behind it has been awaited
function downloadAsync(queue) {
	if (this._deferral) this._deferral.touch(this, 'readonly', queue === undefined ? 2 : queue)
	return this._native.downloadAsync(this._handle, queue === undefined ? 2 : queue)
}
function makeOpenCLBuffer(native, created, numBytes, imageDims, owner, deferral) {
	const buf = created.buffer
	buf._handle = created.handle
	buf._native = native
	buf._deferral = deferral
	buf.numBytes = numBytes
	buf.owner = owner || ''
	buf.imageDims = imageDims
	buf.timestamp = 0
	buf.loadstamp = 0
	buf.creationTime = process.hrtime()
	buf.hostAccess = hostAccess
	buf.addRef = addRef
	buf.release = release
	buf.refCount = refCount
	buf.downloadAsync = downloadAsync
	if (deferral) Deferral.adopt(buf)
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
		this._addon = params.addon || null // tests: a stand-in for the N-API addon (node/test/defer_host_bench.js counts the calls the JS layer makes)
		this.queue = this.overlapping ? { load: 0, process: 1, unload: 2 } : { load: 1, process: 1, unload: 1 }
		this._ctx = null
		this._native = null
	}

	async initialise() {
		this._native = this._addon || loadAddon()
		this._ctx = this._native.createContext(this.deviceIndex)
		if (this.deferred) this._deferral = new Deferral(this)
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
		const created = native.createBuffer(this._ctx, numBytes, ACCESS[bufDir], SVM[bufType], w, h, owner || '')
		return makeOpenCLBuffer(native, created, numBytes, imageDims, owner, this._deferral)
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

	async runProgram(program, params, queue) {
		const native = this._need()
		if (this._deferral) return this._deferral.record(program, params, queue === undefined ? this.queue.process : queue)
		const names = []
		const values = []
		for (const name of Object.keys(params)) {
			const v = params[name]
			if (v === undefined || v === null) continue
			names.push(name)
			if (Buffer.isBuffer(v)) {
				if (!v._handle) throw new Error(`runProgram: parameter '${name}' is a plain Buffer, not an OpenCLBuffer`)
				values.push(v._handle)
			} else {
				values.push(typeof v === 'boolean' ? (v ? 1 : 0) : v)
			}
		}
		const q = queue === undefined ? this.queue.process : queue
		return native.runProgram(this._ctx, program._handle, names, values, q, this.profile)
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

	// ---- staging extensions (not nodencl; SURVEY 8f-3, node/staging.js) -----------------------------
	// later work on `waiter` starts only after everything enqueued so far on `signal` has finished
	queueWaitQueue(waiter, signal) {
		this._need().queueWaitQueue(this._ctx, waiter, signal)
	}

	// a point in `queue`: { wait(): Promise<void>, done(): boolean }
	recordEvent(queue) {
		const native = this._need()
		const ev = native.eventRecord(this._ctx, queue === undefined ? this.queue.process : queue)
		return { wait: () => native.eventWait(ev), done: () => native.eventDone(ev) }
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
				return native.routeOp(h, 5, buf._handle, peer)
			},
			recv: (buf, peer) => {
				if (this._deferral) { this._deferral.touch(buf, 'writeonly', this.queue.process); native.routeOp(h, 2, this.queue.process) } // (readers of the old contents were launched just now)
				return native.routeOp(h, 6, buf._handle, peer)
			}
		}
	}
	static routeUniqueId() { return loadAddon().routeUniqueId() }

	// library options (include/phaneron_hip.h ph_ctx_set_option): 'lds_lut', 'stream_images', 'stream_threshold_mb', 'host_pool_mb'
	setOption(name, value) { this._need().setOption(this._ctx, String(name), value | 0) }
	logBuffers() {
		const s = this._need().bufferStats(this._ctx)
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

module.exports = { clContext, resolveProgram, colour, planeBytes, FORMATS }
