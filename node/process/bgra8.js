'use strict'
// bgra8 Reader / Writer / fillBuf (reference: src/process/bgra8.ts) - see packFormats.js
module.exports = require('./packFormats').makeFormat('bgra8')
