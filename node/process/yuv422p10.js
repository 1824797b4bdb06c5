'use strict'
// yuv422p10 Reader / Writer / fillBuf (reference: src/process/yuv422p10.ts) - see packFormats.js
module.exports = require('./packFormats').makeFormat('yuv422p10')
