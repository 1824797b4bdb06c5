// ph_lut.cpp - host side of ph_lut.h: build (and exhaustively verify) the compressed table.
#include "ph_lut_host.h"

#include <algorithm>
#include <cstring>

namespace ph {

bool lut_compress(const float *lut, uint32_t max_bytes, std::vector<uint32_t> &blob, LutHostInfo &info) {
  uint32_t p[65536];
  std::memcpy(p, lut, sizeof p);
  bool found = false;
  for (uint32_t shift = 4; shift <= 6; ++shift) {
    const uint32_t B = 1u << shift;
    // T = first index (multiple of B) from which every block spans < 65536 bit patterns
    uint32_t toe = 0;
    for (uint32_t blk = 0; blk < 65536 / B; ++blk) {
      uint32_t lo = p[blk * B], hi = lo;
      for (uint32_t i = 1; i < B; ++i) lo = std::min(lo, p[blk * B + i]), hi = std::max(hi, p[blk * B + i]);
      if (hi - lo >= 65536u) toe = (blk + 1) * B;
    }
    uint32_t n_anchors = toe + (65536 - toe) / B;
    n_anchors = (n_anchors + 3) & ~3u;  // keep delta[] 16-byte aligned
    const uint32_t bytes = n_anchors * 4 + 65536 * 2;
    if (bytes > max_bytes) continue;
    if (found && bytes >= info.bytes) continue;
    found = true;
    info.bytes = bytes, info.toe = toe, info.shift = shift, info.delta_off = n_anchors * 4;
    blob.assign(bytes / 4, 0);
    uint16_t *delta = reinterpret_cast<uint16_t *>(blob.data() + n_anchors);
    for (uint32_t i = 0; i < toe; ++i) blob[i] = p[i], delta[i] = 0;
    for (uint32_t blk = toe / B; blk < 65536 / B; ++blk) {
      uint32_t lo = p[blk * B];
      for (uint32_t i = 1; i < B; ++i) lo = std::min(lo, p[blk * B + i]);
      blob[toe + (blk - toe / B)] = lo;
      for (uint32_t i = 0; i < B; ++i) delta[blk * B + i] = (uint16_t)(p[blk * B + i] - lo);
    }
  }
  if (!found) return false;
  // exhaustive self-check of the decode formula the kernels use
  const uint16_t *delta = reinterpret_cast<const uint16_t *>(reinterpret_cast<const uint8_t *>(blob.data()) + info.delta_off);
  for (uint32_t i = 0; i < 65536; ++i) {
    const uint32_t b = std::min(i, info.toe + ((i - info.toe) >> info.shift));
    if (blob[b] + delta[i] != p[i]) return false;
  }
  return true;
}

}  // namespace ph
