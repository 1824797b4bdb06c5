'use strict'
// rgba8 Reader / Writer / fillBuf (reference: src/process/rgba8.ts) - see packFormats.js
module.exports = require('./packFormats').makeFormat('rgba8')
