'use strict'
// A stand-in for `redioactive` with just enough shape for the reference's valve-building code (producer/mixer.ts,
// transitioner.ts, combiner.ts, blackSilence.ts) to construct its pipes under node 12: nothing streams.  A pipe records
// how it was made - `redio(generator)`, `.zipEach(pipes)`, `.valve(fn)` - so that valve_scenario.js can take the
// reference's OWN valve closures off the pipes and call them frame by frame in the order the graph would.
const end = Object.freeze({ redio: 'end' })
const nil = Object.freeze({ redio: 'nil' })
const isEnd = (v) => v === end
const isNil = (v) => v === nil
const isValue = (v) => v !== end && v !== nil

class MockPipe {
	constructor(kind, props) { Object.assign(this, { kind }, props) }
	valve(fn, options) { return new MockPipe('valve', { up: this, fn, options: options || null }) }
	zipEach(others) { return new MockPipe('zip', { up: this, others }) }
	fork() { return new MockPipe('fork', { up: this }) }
	unfork() {}
	each(fn) { return new MockPipe('each', { up: this, fn }) }
	spout(fn) { return new MockPipe('spout', { up: this, fn }) }
	doto(fn) { return new MockPipe('doto', { up: this, fn }) }
	root() { let p = this; while (p.up) p = p.up; return p }
}

function redio(generator, options) { return new MockPipe('source', { generator, options: options || null }) }

module.exports = Object.assign(redio, { default: redio, end, nil, isEnd, isNil, isValue, MockPipe,
	RedioPipe: undefined, RedioEnd: undefined, RedioNil: undefined, Valve: undefined })
