'use strict'
// GPU run of the control-plane smoke (node/amcp.js): job.json holds a script - command lines and { tick: n } entries;
// every response is recorded and every output frame written for tests/test_node_boundary.py to compare with the
// oracle's chain.  usage: node amcp_run.js <dir>
const fs = require('fs')
const path = require('path')
const { Rig } = require('../device.js')
const { Server } = require('../amcp.js')

const dir = process.argv[2]
const job = JSON.parse(fs.readFileSync(path.join(dir, 'job.json')))

async function main() {
	const rig = await Rig.open({ deviceIndex: 0 })
	const server = new Server(rig, { width: job.width, height: job.height, channels: 1, readSpec: job.readSpec, writeSpec: job.writeSpec })
	await server.init()
	const base = rig.ctx.bufferStats().liveBuffers - rig.constants.size
	const responses = []
	let n = 0
	for (const step of job.script) {
		if (typeof step === 'string') responses.push(await server.execute(step))
		else for (let i = 0; i < step.tick; ++i) fs.writeFileSync(path.join(dir, `out_${n++}.bin`), await server.tick(1))
	}
	server.close()
	const leaked = rig.ctx.bufferStats().liveBuffers - rig.constants.size - base
	rig.close()
	fs.writeFileSync(path.join(dir, 'result.json'), JSON.stringify({ responses, frames: n, leaked }))
}

main().catch((e) => { process.stderr.write(String(e && e.stack || e) + '\n'); process.exit(1) })
