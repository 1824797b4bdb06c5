'use strict'
// GPU run of the channel compositor (node/channel.js): three layers - a full-frame source, a picture-in-picture
// source that dissolves into another clip, and an empty layer - composed frame by frame; every output frame is
// written for tests/test_node_boundary.py to compare with the oracle's chain.  usage: node channel_run.js <dir>
const fs = require('fs')
const path = require('path')
const { Rig } = require('../device.js')
const { Channel } = require('../channel.js')

const dir = process.argv[2]
const job = JSON.parse(fs.readFileSync(path.join(dir, 'job.json')))

async function main() {
	const rig = await Rig.open({ deviceIndex: 0 })
	const { width: w, height: h } = job
	// a clip's k-th frame: loaded from <name>_<k>.bin, stamped like a producer would (base + k)
	const clip = (name, base) => (k) => {
		const file = path.join(dir, `${name}_${k}.bin`)
		if (!fs.existsSync(file)) return null
		return { pending: file, timestamp: base + k }
	}
	// sources are uploaded on demand: Channel asks for an image, so wrap the loader
	const uploaded = (loader) => {
		const cache = new Map()
		return {
			prepare: async (k) => {
				const d = loader(k)
				if (!d) return cache.set(k, null)
				const img = await rig.image(w, h, `src ${d.pending}`)
				await rig.upload(img, fs.readFileSync(d.pending))
				img.timestamp = d.timestamp
				cache.set(k, img)
			},
			frame: (k) => { const v = cache.get(k); cache.delete(k); return v || null }
		}
	}
	const A = uploaded(clip('A', 100))
	const B0 = uploaded(clip('B0', 200))
	const B1 = uploaded(clip('B1', 300))
	const layers = [
		{ id: 'L1', clips: [{ start: 0, frame: A.frame }] },
		{ id: 'L2', clips: [{ start: 0, frame: B0.frame, placement: job.pip },
			{ start: job.dissolveAt, frame: B1.frame, transition: { type: 'dissolve', length: job.dissolveLen } }] },
		{ id: 'L3', clips: [] }
	]
	const chan = new Channel(rig, w, h, layers)
	await chan.init()
	const base = rig.ctx.bufferStats().liveBuffers - rig.constants.size // cached parameter buffers are not frames
	const stamps = []
	for (let f = 0; f < job.frames; ++f) {
		await A.prepare(f)
		if (f < job.dissolveAt + job.dissolveLen) await B0.prepare(f)
		if (f >= job.dissolveAt) await B1.prepare(f - job.dissolveAt)
		await rig.sync(rig.ctx.queue.load)
		const out = await chan.compose(f)
		stamps.push(out.timestamp)
		await rig.download(out)
		fs.writeFileSync(path.join(dir, `out_${f}.bin`), out)
		out.release()
	}
	const leaked = rig.ctx.bufferStats().liveBuffers - rig.constants.size - base
	chan.close()
	fs.writeFileSync(path.join(dir, 'result.json'), JSON.stringify({ stamps, leaked, board: rig.board.stats }))
	rig.close()
}

main().catch((e) => { process.stderr.write(String(e && e.stack || e) + '\n'); process.exit(1) })
