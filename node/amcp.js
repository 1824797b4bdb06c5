'use strict'
// Control-plane smoke (SURVEY 8 f4): a handful of AMCP commands driving synthetic-source channels end to end on the
// addon - command text in, v210 frames out.  Not the reference's server (src/AMCP/*: TCP listener, consumer and
// producer registries, FFmpeg / Decklink / route sources): only what is needed to show that a channel built from
// this repository's pieces is OPERABLE - sources -> ToRGBA -> placement -> transition -> combine -> FromRGBA -
// under the commands an operator would send, with the reference's semantics for the ones implemented:
//   PLAY    <ch>-<layer> [<source> [MIX <frames>]]   background clip (loading it first if given) to foreground; with
//                                                    MIX the outgoing clip keeps playing under a dissolve (basicCmds.ts:132-145)
//   LOADBG  <ch>-<layer> <source> [MIX <frames>]     prepare the layer's next clip (basicCmds.ts:113-115)
//   STOP    <ch>-<layer>                             remove the foreground clip (basicCmds.ts:166-171)
//   CLEAR   <ch>[-<layer>]                           remove foreground and background of one layer or of all (:177-182)
//   MIXER   <ch>-<layer> FILL x y xScale yScale | ANCHOR x y | ROTATION degrees          (mixerCmds.ts, layer.ts:256-301)
// Responses follow the server's shape: "202 <CMD> OK", "400 ERROR\r\n<line> NOT IMPLEMENTED", "404 <CMD> ERROR".
// Sources are synthetic v210 generators ("NOISE:<seed>", "RAMP"); a layer number orders the layers
// bottom to top, as the combiner does.
const { Channel, DEFAULT_PLACEMENT } = require('./channel.js')
const { planeBytes } = require('./index.js')

// legal-range v210 noise, reproducible from (seed, frame): word i = three 10-bit fields 64 + (hash % 877)
function noiseFrame(bytes, seed, k) {
	const out = Buffer.alloc(bytes)
	let s = (Math.imul(seed | 0, 0x9E3779B1) ^ Math.imul((k + 1) | 0, 0x85EBCA6B)) >>> 0
	for (let i = 0; i < bytes; i += 4) {
		let w = 0
		for (let f = 0; f < 3; ++f) {
			s = (Math.imul(s, 1664525) + 1013904223) >>> 0
			w |= (64 + ((s >>> 8) % 877)) << (10 * f)
		}
		out.writeUInt32LE(w >>> 0, i)
	}
	return out
}
function rampFrame(bytes, width, k) {
	const out = Buffer.alloc(bytes)
	for (let i = 0; i < bytes; i += 4) {
		const v = 64 + ((((i >> 2) * 3 + k * 16) % (width * 2)) * 876 / (width * 2) | 0)
		out.writeUInt32LE(((512 << 20) | (v << 10) | 512) >>> 0, i)
	}
	return out
}

class Server {
	constructor(rig, { width, height, channels = 1, readSpec = '709', writeSpec = '709' }) {
		Object.assign(this, { rig, width, height, readSpec, writeSpec })
		this.bytes = planeBytes('v210', width, height)[0]
		this.channels = []
		for (let c = 0; c < channels; ++c) this.channels.push({ layers: new Map(), chan: null, frame: 0 })
	}
	async init() {
		const { rig, width: w, height: h } = this
		this.read = await rig.unpack('v210', w, h, this.readSpec, this.writeSpec)
		this.write = await rig.pack('v210', w, h, this.writeSpec, false)
	}

	_source(name) {
		const m = /^NOISE:(\d+)$/i.exec(name)
		if (m) return (k) => noiseFrame(this.bytes, +m[1], k)
		if (/^RAMP$/i.test(name)) return (k) => rampFrame(this.bytes, this.width, k)
		return null
	}
	_layer(ch, num) {
		if (!ch.layers.has(num)) ch.layers.set(num, { id: `L${num}`, clips: [], background: null, placement: JSON.parse(JSON.stringify(DEFAULT_PLACEMENT)) })
		return ch.layers.get(num)
	}
	// a clip for channel.js: frame(k) hands over an RGBA image prepared by _prepare for channel frame start + k
	_clip(ch, layer, gen, start, transition) {
		const ready = new Map()
		return { start, gen, ready, placement: layer.placement, transition, frame: (k) => { const v = ready.get(k); ready.delete(k); return v || null } }
	}
	async _prepare(clip, k, stamp) {
		const { rig, width: w, height: h } = this
		const src = await rig.planes('v210', w, h, 'readonly')
		await rig.upload(src[0], clip.gen(k))
		await rig.sync(rig.ctx.queue.load)
		const img = await rig.image(w, h, 'clip frame')
		img.timestamp = stamp
		await rig.run(this.read(src, img))
		src[0].release()
		clip.ready.set(k, img)
	}

	// one command line -> response string
	async execute(line) {
		const cmd = line.trim().split(/\s+/)
		const fail = () => `400 ERROR\r\n${cmd.join(' ')} NOT IMPLEMENTED`
		if (!cmd[0]) return fail()
		const verb = cmd[0].toUpperCase()
		const m = /^(\d+)(?:-(\d+))?$/.exec(cmd[1] || '')
		if (!m) return fail()
		const ch = this.channels[+m[1] - 1]
		if (!ch) return `404 ${verb} ERROR`
		const layerNum = m[2] === undefined ? undefined : +m[2]
		const mix = (args) => { const i = args.findIndex((a) => /^MIX$/i.test(a)); return i >= 0 ? { type: 'dissolve', length: +args[i + 1] || 0 } : undefined }
		const load = (layer, args) => {
			const gen = this._source(args[0] || '')
			if (!gen) return false
			layer.background = { gen, transition: mix(args) }
			return true
		}
		if (verb === 'LOADBG') {
			if (layerNum === undefined || !load(this._layer(ch, layerNum), cmd.slice(2))) return `404 ${verb} ERROR`
			return '202 LOADBG OK'
		}
		if (verb === 'PLAY') {
			if (layerNum === undefined) return `404 ${verb} ERROR`
			const layer = this._layer(ch, layerNum)
			if (cmd.length > 2 && !load(layer, cmd.slice(2))) return `404 ${verb} ERROR`
			if (!layer.background) return `404 ${verb} ERROR`
			const t = layer.clips.length ? layer.background.transition : undefined // nothing to dissolve from: a cut
			if (!t) layer.clips.length = 0                                         // a cut replaces the foreground at once
			layer.clips.push(this._clip(ch, layer, layer.background.gen, ch.frame, t))
			layer.background = null
			return '202 PLAY OK'
		}
		if (verb === 'STOP') {
			if (layerNum === undefined || !ch.layers.has(layerNum)) return `404 ${verb} ERROR`
			this._drop(ch.layers.get(layerNum))
			return '202 STOP OK'
		}
		if (verb === 'CLEAR') {
			for (const [num, layer] of ch.layers) if (layerNum === undefined || num === layerNum) { this._drop(layer); layer.background = null }
			if (layerNum === undefined) ch.layers.clear(); else ch.layers.delete(layerNum)
			return '202 CLEAR OK'
		}
		if (verb === 'MIXER') {
			if (layerNum === undefined) return `404 ${verb} ERROR`
			const layer = this._layer(ch, layerNum)
			const what = (cmd[2] || '').toUpperCase()
			const v = cmd.slice(3).map(Number)
			if (what === 'FILL' && v.length === 4) layer.placement.fill = { xOffset: v[0], yOffset: v[1], xScale: v[2], yScale: v[3] } // layer.ts:285-299
			else if (what === 'ANCHOR' && v.length === 2) layer.placement.anchor = { x: v[0], y: v[1] }
			else if (what === 'ROTATION' && v.length === 1) layer.placement.rotation = v[0]
			else return fail()
			return `202 MIXER OK`
		}
		return fail()
	}
	_drop(layer) {
		for (const c of layer.clips) for (const img of c.ready.values()) img.release()
		layer.clips.length = 0
	}

	// one output period of channel `c` (1-based): returns the v210 frame as a Buffer
	async tick(c = 1) {
		const ch = this.channels[c - 1]
		const { rig, width: w, height: h } = this
		const order = [...ch.layers.keys()].sort((a, b) => a - b)
		const layers = order.map((n) => ch.layers.get(n))
		const f = ch.frame
		for (const layer of layers) {
			// a finished transition leaves only the incoming clip
			const last = layer.clips[layer.clips.length - 1]
			if (last && last.transition && f - last.start >= last.transition.length) layer.clips.splice(0, layer.clips.length - 1)
			for (let i = 0; i < layer.clips.length; ++i) // distinct stamps per layer and clip: they key the JobBoard's batches
				await this._prepare(layer.clips[i], f - layer.clips[i].start, 100000 * order[layers.indexOf(layer)] + 1000 * i + (f % 1000))
		}
		if (!ch.chan) { ch.chan = new Channel(rig, w, h, layers, `chan${c}`); await ch.chan.init() }
		ch.chan.layers = layers
		const rgba = await ch.chan.compose(f)
		const out = await rig.planes('v210', w, h, 'writeonly')
		await rig.run(this.write(rgba, out, 0))
		await rig.sync()
		await rig.download(out[0])
		const bytes = Buffer.from(out[0])
		out[0].release()
		rgba.release()
		ch.frame++
		return bytes
	}
	close() {
		for (const ch of this.channels) {
			for (const layer of ch.layers.values()) this._drop(layer)
			if (ch.chan) ch.chan.close()
		}
	}
}

module.exports = { Server, noiseFrame, rampFrame }
