// ph_lut.h - exact two-level compression of a 65536-entry f32 gamma LUT so that it fits in
// the 160 KiB LDS of one gfx950 CU (DESIGN.md "LUT placement").
//
// The reference's read/write kernels do 3 data-dependent lookups per pixel into a 256 KiB
// table (v210.ts:68-70,148-150).  From global memory that is ~300 G lookups/s on MI355X
// (one cache line per lane); from LDS it is ~4800 G/s.  The table does not fit in LDS as f32,
// but its bit patterns p[i] are locally smooth, so it is stored as
//     anchor[b]  (u32)  minimum bit pattern of block b
//     delta[i]   (u16)  p[i] - anchor[block(i)]
// with p[i] = anchor[block(i)] + delta[i].  Blocks are LOGARITHMIC in the index:
//     block(i) = (bits((float)(i + bias)) >> (23 - m)) - first      (first = the block number of i = 0)
// i.e. 2^m blocks per octave of (i + bias): single entries in the steep toe near 0, ~128
// entries per block at the top.  One float add and one shift give the block, which is what
// makes the lookup cheap (the LUT curves are power laws, so equal *relative* block widths
// bound the deltas).  Exact whenever every block's max-min < 65536; that and the decode
// formula are verified exhaustively when the table is built, otherwise the LUT stays
// "plain": the LDS kernels refuse it and the global-gather kernels are used.
#pragma once
#include <stdint.h>

namespace ph {

struct LutView {          // passed by value to kernels
  const uint32_t *blob;   // device: [anchors u32 x n_anchors][delta u16 x 65536], size % 16 == 0
  uint32_t bytes;         // LDS footprint: `hole` + the blob; 0 = not compressible
  uint32_t hole;          // the blob is loaded at this LDS byte address (4 * 2^m, see a_scale); [0, hole) stays unused
  float bias;             // even integer >= 2 (keeps round-to-nearest-even parity of the index)
  uint32_t shift;         // 23 - m
  // The anchor's LDS byte address comes out of the float (i + bias) * a_scale with NO offset to add: a_scale = 2^(1 - E),
  // E the exponent field of the bias, puts the smallest value at the exponent field 1, so (bits >> (shift - 2)) & ~3 runs
  // from 4 * 2^m = `hole` upwards in steps of 4 per block - the absolute address when the table is loaded at `hole`.
  float a_scale;
  float delta_scale;      // 2^-148 as a float (denormal): delta byte address = bits(fma(i + bias, delta_scale, delta_base))
  float delta_base;       // (delta_off - 2 * bias) * 2^-149 (denormal), delta_off counted from LDS address 0
};

}  // namespace ph
