'use strict'
// A recording stand-in for a nodencl-shaped clContext: no device, every call is appended to a trace.
// scenario.js runs the REFERENCE's own operator + dispatcher code (type-stripped, build container only)
// against it; the trace - buffers, programs, kernel parameters by argument name, refcounts, order - is
// committed as tests/golden/host_trace.json and replayed on the real addon by replay.js.
// options.resolve(kernelSrc, name): if given, its result is recorded with every createProgram (the
// addon's device-free kernel selection applied to the text the reference really passes).
const crypto = require('crypto')

function makeMock(mockOptions = {}) {
	const trace = []
	let nextBuf = 0
	let nextProg = 0
	const live = new Map()
	const hash = (b) => crypto.createHash('sha256').update(b).digest('hex').slice(0, 16)

	const describe = (v) => {
		if (Buffer.isBuffer(v) && v._mockId !== undefined) return { buf: v._mockId, refs: v._refs, ts: v.timestamp }
		if (typeof v === 'number') return Number.isInteger(v) ? v : Math.fround(v)
		if (typeof v === 'boolean') return v
		return String(v)
	}

	const ctx = {
		queue: { load: 0, process: 1, unload: 2 },
		trace,
		live,
		async initialise() {},
		getPlatformInfo() { return { vendor: 'mock', devices: [{ type: 'mock' }] } },
		async createBuffer(numBytes, dir, type, dims, owner) {
			const buf = Buffer.alloc(numBytes)
			buf._mockId = nextBuf++
			buf._refs = 1
			buf.timestamp = 0
			buf.loadstamp = 0
			buf.numBytes = numBytes
			buf.hostAccess = async (d, q, src) => {
				trace.push({ op: 'hostAccess', buf: buf._mockId, dir: d, queue: q === undefined ? null : q, src: src ? { bytes: src.length, sha: hash(src) } : null })
				if (src) src.copy(buf)
				// mapped for writing and filled by plain Buffer writes afterwards (blackSilence.ts:133-134): with
				// options.recordMappedWrites the bytes found there when a kernel first uses the buffer are recorded
				else if (d === 'writeonly' && mockOptions.recordMappedWrites) buf._mapped = true
			}
			buf.addRef = () => { buf._refs++; trace.push({ op: 'addRef', buf: buf._mockId, refs: buf._refs }) }
			buf.release = () => {
				buf._refs--
				trace.push({ op: 'release', buf: buf._mockId, refs: buf._refs })
				if (buf._refs === 0) live.delete(buf._mockId)
			}
			live.set(buf._mockId, buf)
			trace.push({ op: 'createBuffer', buf: buf._mockId, numBytes, dir, type, dims: dims ? { width: dims.width, height: dims.height } : null, owner: owner || null })
			return buf
		},
		async createProgram(kernel, options) {
			const prog = { id: nextProg++, name: options.name }
			const resolved = mockOptions.resolve ? { resolved: mockOptions.resolve(kernel, options.name) } : {}
			trace.push(Object.assign(resolved, {
				op: 'createProgram', prog: prog.id, name: options.name, srcSha: hash(Buffer.from(String(kernel))),
				globalWorkItems: options.globalWorkItems === undefined ? null : (typeof options.globalWorkItems === 'number' ? options.globalWorkItems : Array.from(options.globalWorkItems)),
				workItemsPerGroup: options.workItemsPerGroup === undefined ? null : options.workItemsPerGroup
			}))
			return prog
		},
		async runProgram(program, params, queue) {
			const p = {}
			const data = {}
			const hex = {}
			for (const k of Object.keys(params)) {
				const b = params[k]
				if (Buffer.isBuffer(b) && b._mapped) {
					b._mapped = false
					const w = { op: 'hostWrite', buf: b._mockId, bytes: b.length, sha: hash(b), allZero: b.every((x) => x === 0) }
					if (b.length <= 64) w.hex = b.toString('hex') // matrices (loadSave.ts:76-99 fills them through the mapped mirror)
					trace.push(w)
				}
				p[k] = describe(params[k])
				// small read-only operands (matrices, LUTs, flip vectors): pin their contents too
				if (Buffer.isBuffer(params[k]) && params[k].length <= 262144 && !/^(output|input|prev|cur|next|l\dIn|input\d|maskIn)$/.test(k)) {
					data[k] = hash(params[k])
					if (params[k].length <= 64) hex[k] = params[k].toString('hex') // matrices, flip vectors: the values themselves
				}
			}
			trace.push({ op: 'runProgram', prog: program.id, name: program.name, queue, params: p, data, hex })
			return { dataToKernel: 0, kernelExec: 0, totalTime: 0 }
		},
		async waitFinish(queue) { trace.push({ op: 'waitFinish', queue: queue === undefined ? null : queue }) },
		logBuffers() {}
	}
	return ctx
}

module.exports = { makeMock }
