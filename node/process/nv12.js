'use strict'
// nv12 Reader / Writer / fillBuf (reference: src/process/nv12.ts) - see packFormats.js
module.exports = require('./packFormats').makeFormat('nv12')
