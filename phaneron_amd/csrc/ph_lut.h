// ph_lut.h - exact two-level compression of a 65536-entry f32 gamma LUT so that it fits in
// the 160 KiB LDS of one gfx950 CU (DESIGN.md "LUT placement").
//
// The reference's read/write kernels do 3 data-dependent lookups per pixel into a 256 KiB
// table (v210.ts:68-70,148-150).  From global memory that is ~300 G lookups/s on MI355X
// (one cache line per lane); from LDS it is ~4800 G/s.  The table does not fit in LDS as f32,
// but its bit patterns p[i] are locally smooth, so it is stored as
//     anchor[b]  (u32)  b = i            for i <  T   (the steep toe: exact value)
//                       b = T + (i-T)>>S for i >= T   (minimum bit pattern of a 2^S block)
//     lo16[i]    (u16)  p[i] & 0xffff
// and p[i] = anchor[b] + ((lo16[i] - anchor[b]) & 0xffff), exact whenever every block's
// max-min < 65536 (verified when the table is built; otherwise the LUT stays "plain" and the
// kernels that need LDS tables refuse it and the gather kernels are used).
#pragma once
#include <stdint.h>

namespace ph {

// Optional arithmetic predictor for the anchor (saves one of the two LDS reads): for index i
// with r = (float)i,
//     pred(r) = r < knee ? r * toe_slope : (u*u) * q(u),  u = fma(r, a, b),  q = degree-4 Horner
// fitted so that |bits(table[i]) - bits(pred(r))| < 32768 for ALL 65536 entries (checked with
// the same IEEE operations on the host); then table[i] = bits(pred) + sext16(lo16[i] - bits(pred)).
struct LutPredictor {
  float a, b, q[5], toe_slope, knee;
  uint32_t ok;
};

struct LutView {        // passed by value to kernels
  const uint32_t *blob; // device: [anchors u32 x n_anchors][lo16 x 65536], 16-byte aligned size
  uint32_t bytes;       // multiple of 16; 0 = not compressible
  uint32_t toe;         // T
  uint32_t shift;       // S
  uint32_t lo_off;      // byte offset of lo16[] inside the blob (= 4 * n_anchors, 16-aligned)
  LutPredictor pred;    // pred.ok == 0: anchors only
};

}  // namespace ph
