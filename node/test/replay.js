'use strict'
// Replays tests/golden/host_trace.json on the REAL addon.  The trace is what the reference's own operator layer
// and dispatcher (src/process/*, src/clJobQueue.ts - unchanged, type-stripped, run in the build container
// against a recording clContext) asked nodencl to do: every createBuffer / hostAccess / createProgram /
// runProgram (arguments by OpenCL name) / waitFinish / addRef / release, in order, with the reference counts it
// observed.  Here the same calls go to node/index.js + ph_napi.c + libphaneron_hip.so on the GPU:
//   - every call must be accepted (argument names, buffer sizes, work-group geometry as the reference sends them),
//   - reference counts must match the recorded ones at every step,
//   - every frame the reference maps for reading (saveFrame, io.ts:166-174) is dumped for the caller to check.
// Kernel text cannot travel to the GPU box; a `read` / `write` program's format comes from the recorded
// fingerprint of the text (tests/golden/kernel_text_sha.json).  Frame data comes from <dir>/<sha16>.bin (the
// caller regenerates the reference's test patterns); matrices are in the trace; LUTs are recognised by hash
// among the tables the library's own colour maths produces - a LUT it cannot produce fails the replay.
// usage: node replay.js <trace.json> <kernel_text_sha.json> <dir>
const fs = require('fs')
const path = require('path')
const crypto = require('crypto')
const { clContext, colour } = require('../index.js')

const [tracePath, shaPath, dir] = process.argv.slice(2)
const trace = JSON.parse(fs.readFileSync(tracePath))
const textSha = JSON.parse(fs.readFileSync(shaPath))
const sha16 = (b) => crypto.createHash('sha256').update(b).digest('hex').slice(0, 16)

async function main() {
	const ctx = new clContext({ deviceIndex: 0, overlapping: true })
	await ctx.initialise()
	const luts = new Map()
	for (const spec of ['601-625', '601_525', '709', '2020', 'sRGB', 'bogus'])
		for (const t of [colour.gamma2linearLUT(spec), colour.linear2gammaLUT(spec)]) {
			const bytes = Buffer.from(t.buffer, t.byteOffset, t.byteLength)
			luts.set(sha16(bytes), bytes)
		}
	const bufs = new Map() // trace id -> { buf, content }
	const progs = new Map()
	const problems = []
	const dumps = []
	let calls = 0
	const refsOf = (id) => { const b = bufs.get(id); return b && !b.dead ? b.buf.refCount() : 0 }

	for (let i = 0; i < trace.length; ++i) {
		const e = trace[i]
		try {
			switch (e.op) {
				case 'createBuffer': {
					const buf = await ctx.createBuffer(e.numBytes, e.dir, e.type, e.dims || undefined, e.owner || undefined)
					bufs.set(e.buf, { buf, content: null, dead: false })
					calls++
					break
				}
				case 'hostAccess': {
					const b = bufs.get(e.buf)
					let src
					if (e.src) {
						const file = path.join(dir, `${e.src.sha}.bin`)
						if (fs.existsSync(file)) {
							src = fs.readFileSync(file)
							if (src.length !== e.src.bytes) problems.push({ i, what: `blob ${e.src.sha} has ${src.length} bytes, trace says ${e.src.bytes}` })
							b.content = e.src.sha
						} else if (e.src.bytes > 64) problems.push({ i, what: `no data for hostAccess source ${e.src.sha} (${e.src.bytes} bytes)` })
					}
					if (e.dir === 'readonly') {
						await b.buf.hostAccess('readonly', e.queue === null ? undefined : e.queue)
						const name = `dump_${i}_buf${e.buf}.bin`
						fs.writeFileSync(path.join(dir, name), b.buf)
						dumps.push({ i, buf: e.buf, file: name, bytes: b.buf.length })
					} else if (src) await b.buf.hostAccess(e.dir, e.queue === null ? undefined : e.queue, src)
					else await b.buf.hostAccess(e.dir, e.queue === null ? undefined : e.queue)
					calls++
					break
				}
				case 'hostWrite': {  // the reference filled a buffer it had mapped for writing (Buffer.copy into the mirror): do the same
					const b = bufs.get(e.buf)
					const file = path.join(dir, `${e.sha}.bin`)
					// frames come from files, matrices are in the trace, tables are recognised among the library's own
					const bytes = e.allZero ? Buffer.alloc(e.bytes) : fs.existsSync(file) ? fs.readFileSync(file) : e.hex ? Buffer.from(e.hex, 'hex') : luts.get(e.sha) || null
					if (!bytes) throw new Error(`no data for the mapped write ${e.sha} (${e.bytes} bytes)`)
					bytes.copy(b.buf)
					b.content = e.sha
					break
				}
				case 'createProgram': {
					let source = `phaneron:${e.name}`
					if (e.name === 'read' || e.name === 'write') {
						const fmt = textSha[e.srcSha]
						if (!fmt) throw new Error(`kernel text fingerprint ${e.srcSha} is not one of the reference's pack formats`)
						source = `phaneron:${fmt}`
					}
					const options = { name: e.name }
					if (e.globalWorkItems !== null) options.globalWorkItems = e.globalWorkItems
					if (e.workItemsPerGroup !== null) options.workItemsPerGroup = e.workItemsPerGroup
					progs.set(e.prog, await ctx.createProgram(source, options))
					calls++
					break
				}
				case 'runProgram': {
					const params = {}
					let uploaded = false
					for (const k of Object.keys(e.params)) {
						const v = e.params[k]
						if (v !== null && typeof v === 'object') {
							const b = bufs.get(v.buf)
							if (!b || b.dead) throw new Error(`parameter ${k}: buffer ${v.buf} is not alive`)
							if (b.buf.refCount() !== v.refs) problems.push({ i, what: `${e.name}.${k}: buffer ${v.buf} has ${b.buf.refCount()} references, the reference saw ${v.refs}` })
							// small read-only operands: make the device copy hold what the reference's host code had written
							const want = e.hex && e.hex[k] ? e.hex[k] : (e.data && e.data[k]) || null
							if (want && b.content !== want) {
								const bytes = e.hex && e.hex[k] ? Buffer.from(e.hex[k], 'hex') : luts.get(want)
								if (!bytes) throw new Error(`parameter ${k}: no table with hash ${want} among the library's LUTs`)
								await b.buf.hostAccess('writeonly', ctx.queue.load, bytes)
								b.content = want
								uploaded = true
							}
							params[k] = b.buf
						} else params[k] = v
					}
					if (uploaded) await ctx.waitFinish(ctx.queue.load)
					await ctx.runProgram(progs.get(e.prog), params, e.queue)
					calls++
					break
				}
				case 'waitFinish':
					await ctx.waitFinish(e.queue === null ? undefined : e.queue)
					calls++
					break
				case 'addRef': {
					bufs.get(e.buf).buf.addRef()
					if (refsOf(e.buf) !== e.refs) problems.push({ i, what: `addRef(${e.buf}) -> ${refsOf(e.buf)}, the reference saw ${e.refs}` })
					calls++
					break
				}
				case 'release': {
					const b = bufs.get(e.buf)
					b.buf.release()
					if (e.refs === 0) b.dead = true
					if (refsOf(e.buf) !== e.refs) problems.push({ i, what: `release(${e.buf}) -> ${refsOf(e.buf)}, the reference saw ${e.refs}` })
					calls++
					break
				}
				case 'liveBuffers': {
					const alive = Array.from(bufs.entries()).filter(([, b]) => !b.dead).map(([id]) => id).sort((a, b) => a - b)
					if (JSON.stringify(alive) !== JSON.stringify(e.ids)) problems.push({ i, what: 'live buffer set differs', alive })
					const refs = alive.map((id) => refsOf(id))
					if (JSON.stringify(refs) !== JSON.stringify(e.refs)) problems.push({ i, what: 'final reference counts differ', refs })
					break
				}
				default: // notes, expected-throw records and geometry dumps of the scenario itself
			}
		} catch (err) {
			problems.push({ i, op: e.op, name: e.name || null, what: String(err && err.message || err) })
		}
	}
	await ctx.waitFinish(ctx.queue.process)
	const stats = ctx.logBuffers ? ctx.bufferStats() : null
	const deferred = ctx.deferredStats() // PHANERON_DEFERRED=1: the same calls through the recording context (node/defer.js)
	fs.writeFileSync(path.join(dir, 'replay.json'), JSON.stringify({ events: trace.length, calls, problems, dumps, stats, deferred, device: ctx.getPlatformInfo().devices[0].name }))
}

main().catch((e) => { process.stderr.write(String(e && e.stack || e) + '\n'); process.exit(1) })
