'use strict'
// A stand-in for `beamcoder`: the audio side of the reference's valve-building code asks for filter graphs and silent
// frames while it sets the video side up; here they are inert (audio is out of scope, SURVEY 8).
const filterer = async (spec) => ({ spec, graph: { filters: [], dump: () => '' }, filter: async () => [{ frames: [] }] })
const frame = (o) => Object.assign({}, o)
module.exports = { filterer, frame, Frame: undefined, Filterer: undefined, AudioInputParam: undefined }
