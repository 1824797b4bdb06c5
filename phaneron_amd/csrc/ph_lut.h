// ph_lut.h - exact two-level compression of a 65536-entry f32 gamma LUT so that it fits in
// the 160 KiB LDS of one gfx950 CU (DESIGN.md "LUT placement").
//
// The reference's read/write kernels do 3 data-dependent lookups per pixel into a 256 KiB
// table (v210.ts:68-70,148-150).  From global memory that is ~300 G lookups/s on MI355X
// (one cache line per lane); from LDS it is ~4800 G/s.  The table does not fit in LDS as f32,
// but its bit patterns p[i] are locally smooth, so it is stored as
//     anchor[b]  (u32)  b = i              for i <  T  (the steep toe: exact value)
//                       b = T + (i-T)>>S   for i >= T  (minimum bit pattern of a 2^S block)
//     delta[i]   (u16)  p[i] - anchor[b]
// and p[i] = anchor[b] + delta[i]: exact whenever every block's max-min < 65536 (verified
// exhaustively when the table is built; otherwise the LUT stays "plain", the LDS kernels
// refuse it and the global-gather kernels are used).
#pragma once
#include <stdint.h>

namespace ph {

struct LutView {         // passed by value to kernels
  const uint32_t *blob;  // device: [anchors u32 x n_anchors][delta u16 x 65536], size % 16 == 0
  uint32_t bytes;        // 0 = not compressible
  uint32_t toe;          // T (a multiple of 2^S)
  uint32_t shift;        // S
  uint32_t delta_off;    // byte offset of delta[] inside the blob (= 4 * n_anchors, 16-aligned)
  uint32_t anchor_bias;  // 0 - (0x4B000000 << 2): passed at run time so it lives in an SGPR and folds
                         // into v_lshl_add_u32 (as a literal it costs every lookup an extra v_add_u32)
};

}  // namespace ph
