'use strict'
// yuv422p8 Reader / Writer / fillBuf (reference: src/process/yuv422p8.ts) - see packFormats.js
module.exports = require('./packFormats').makeFormat('yuv422p8')
