'use strict'
// The reference's video valve graph re-hosted on the HIP addon (SURVEY 8f-2):
// producer -> Mixer -> Transitioner -> Combiner -> consumer.
module.exports = Object.assign({}, require('./redio'), require('./black'), require('./mixer'), require('./transitioner'), require('./combiner'))
