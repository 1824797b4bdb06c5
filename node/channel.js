'use strict'
// Channel - the video side of one phaneron channel as a plain per-frame function, for this repository's
// tests and benchmarks (SURVEY 8 a15 / f2).  The reference builds the same thing out of redioactive pipes and
// valves (producer/mixer.ts, transitioner.ts, combiner.ts, blackSilence.ts); here there are no streams, only
//   frame n of every layer  ->  place it (transform)  ->  [transition between two clips]  ->  combine  ->  output
// with the reference's rules for WHAT is computed, so that a frame composed here equals the frame the reference's
// valve graph would hand its consumers:
//   placement    anchor - 0.5, scale = fill.x/yScale, rotate = -rotation / 360, offset = -fill.x/yOffset,
//                applied to EVERY source frame, default placement included        (mixer.ts:209-223)
//   transition   the outgoing clip keeps playing while the incoming one fades in; a dissolve of `length` frames
//                uses mix = 1 - k / (length - 1) on its k-th frame (0 when length < 2); a wipe takes its factor from
//                the red channel of a mask image; the result carries the incoming frame's timestamp;
//                a cut is a transition of no frames                               (transitioner.ts:143-176,262-270)
//   layers       a layer with nothing to show contributes a transparent black frame; zero layers give black,
//                one layer passes through untouched, two or more go through combine_N; outputs are numbered
//                0, 1, 2 ... whatever the sources' timestamps were               (combiner.ts:211-254, blackSilence.ts:96-158)
// Buffers: every intermediate is created for the frame and released once the frame's kernels have run, through the
// JobBoard's completion callbacks - the reference's reference-counting discipline, observable in buffer stats.

const DEFAULT_PLACEMENT = { anchor: { x: 0, y: 0 }, rotation: 0, fill: { xOffset: 0, yOffset: 0, xScale: 1, yScale: 1 } }

// mixer.ts:209-223
function placementToTransform(p) {
	p = p || DEFAULT_PLACEMENT
	return {
		flipH: false, flipV: false,
		anchorX: p.anchor.x - 0.5, anchorY: p.anchor.y - 0.5,
		scaleX: p.fill.xScale, scaleY: p.fill.yScale,
		rotate: -p.rotation / 360.0,
		offsetX: -p.fill.xOffset, offsetY: -p.fill.yOffset
	}
}

// transitioner.ts:170,269
function dissolveMix(k, length) {
	const steps = length > 0 ? length - 1 : 0
	return steps > 0 ? 1.0 - k / steps : 0.0
}

class Channel {
	// layers: [{ id, clips: [{ start, frame(k) -> image | null, placement?, transition?: { type: 'dissolve' | 'wipe', length, mask? } }] }]
	// a clip plays from channel frame `start`; frame(k) returns the clip's k-th source frame (an RGBA f32 image the
	// Channel then owns one reference of) or null when the clip has ended
	constructor(rig, width, height, layers, name = 'chan1') {
		this.rig = rig
		this.width = width
		this.height = height
		this.layers = layers
		this.name = name
		this.count = 0
		this.black = null
		this.stages = null
	}

	async init() {
		const { rig, width: w, height: h } = this
		this.stages = { transform: await rig.transform(w, h), dissolve: await rig.two('transition_dissolve', w, h), wipe: await rig.two('transition_wipe', w, h), combine: new Map() }
		this.black = await rig.image(w, h, `${this.name} black`)
		await this.black.hostAccess('writeonly', rig.ctx.queue.load) // blackSilence.ts:129-135: a zero-filled RGBA frame, mapped, filled, uploaded once
		this.black.fill(0)
		await this.black.hostAccess('none', rig.ctx.queue.load)
		await rig.sync(rig.ctx.queue.load)
	}

	async _combine(n) {
		if (!this.stages.combine.has(n)) this.stages.combine.set(n, await this.rig.combine(n, this.width, this.height))
		return this.stages.combine.get(n)
	}

	// the clip of `layer` showing at channel frame f, and the one before it while a transition runs
	_clipsAt(layer, f) {
		let cur = -1
		layer.clips.forEach((c, i) => { if (c.start <= f) cur = i })
		if (cur < 0) return {}
		const clip = layer.clips[cur]
		const k = f - clip.start
		const t = clip.transition
		const fading = t && cur > 0 && k < t.length ? layer.clips[cur - 1] : null
		return { clip, k, fading, fadingK: fading ? f - fading.start : 0 }
	}

	// place one source frame: a new consumer-size image under the source's job key
	async _place(layerId, clip, k, jobs) {
		const src = clip.frame(k)
		if (!src) return null
		const { rig, width: w, height: h } = this
		const out = await rig.image(w, h, `mixer ${layerId} ${src.timestamp}`)
		out.timestamp = src.timestamp
		const matrix = await this.stages.transform.matrix(placementToTransform(clip.placement))
		const id = { source: `${layerId} mix`, timestamp: src.timestamp }
		rig.post(id, this.stages.transform(src, out, matrix), () => src.release())
		jobs.push(id)
		return out
	}

	// compose channel frame `f`; resolves to an RGBA f32 image with timestamp = running output count (one reference, the caller's)
	async compose(f) {
		const { rig, width: w, height: h } = this
		const ids = []
		const shown = []
		for (const layer of this.layers) {
			const { clip, k, fading, fadingK } = this._clipsAt(layer, f)
			let img = clip ? await this._place(layer.id, clip, k, ids) : null
			if (fading) {
				const old = await this._place(layer.id, fading, fadingK, ids)
				if (old && img) {
					const t = clip.transition
					const mixed = await rig.image(w, h, `${layer.id} ${img.timestamp}`)
					mixed.timestamp = img.timestamp
					const id = { source: layer.id, timestamp: img.timestamp }
					const a = old
					const b = img
					const job = t.type === 'wipe' ? this.stages.wipe(a, b, t.mask, mixed) : this.stages.dissolve(a, b, dissolveMix(k, t.length), mixed)
					rig.post(id, job, () => { a.release(); b.release() })
					ids.push(id)
					img = mixed
				} else if (old && !img) img = old // the incoming clip has nothing yet: keep showing the outgoing one
			}
			shown.push(img) // null: nothing to show on this layer
		}
		let out
		const live = shown.filter((s) => s)
		if (shown.length === 0) {
			out = this.black
			out.addRef()
		} else if (shown.length === 1) {
			out = live.length ? live[0] : (this.black.addRef(), this.black) // a single layer passes through (combiner.ts:222-228)
		} else {
			const inputs = shown.map((s) => s || this.black)
			out = await rig.image(w, h, `combine ${this.name} ${this.count}`)
			const id = { source: `${this.name} combine`, timestamp: this.count }
			rig.post(id, (await this._combine(inputs.length))(inputs, out), () => live.forEach((s) => s.release()))
			ids.push(id)
		}
		await Promise.all(ids.map((id) => rig.board.flush(id)))
		out.timestamp = this.count++
		return out
	}

	close() {
		if (this.black) this.black.release()
		this.black = null
	}
}

module.exports = { Channel, DEFAULT_PLACEMENT, placementToTransform, dissolveMix }
