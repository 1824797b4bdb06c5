'use strict'
// yuv420p Reader / Writer / fillBuf (reference: src/process/yuv420p.ts) - see packFormats.js
module.exports = require('./packFormats').makeFormat('yuv420p')
