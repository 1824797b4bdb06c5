// ph_lut.cpp - host side of ph_lut.h: build (and exhaustively verify) the compressed table.
#include "ph_lut_host.h"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace ph {

namespace {
inline uint32_t f2u(float f) {
  uint32_t u;
  std::memcpy(&u, &f, 4);
  return u;
}
inline float u2f(uint32_t u) {
  float f;
  std::memcpy(&f, &u, 4);
  return f;
}
}  // namespace

LutView lut_view(const LutHostInfo &info, const void *blob_dev) {
  LutView v;
  v.blob = static_cast<const uint32_t *>(blob_dev);
  v.bytes = info.bytes;
  v.hole = info.hole;
  v.bias = info.bias;
  v.shift = info.shift;
  v.a_scale = std::ldexp(1.0f, 1 - (int)(f2u(info.bias) >> 23));  // 2^(1 - E): a denormal constant (2^-132 for a bias of 64)
  v.delta_scale = u2f(2u);  // 2 * 2^-149
  // (delta_off - 2*bias) * 2^-149: a denormal whose bit pattern is that integer
  v.delta_base = u2f(info.delta_off - 2u * (uint32_t)info.bias);
  return v;
}

bool lut_compress(const float *lut, uint32_t max_bytes, std::vector<uint32_t> &blob, LutHostInfo &info) {
  // per-call scratch: callers may compress tables from several threads (the node addon's libuv pool)
  std::vector<uint32_t> p(65536), blk(65536);
  std::memcpy(p.data(), lut, 65536 * sizeof(uint32_t));
  bool found = false;
  for (float bias : {16.0f, 32.0f, 64.0f, 128.0f}) {
    for (uint32_t m = 7; m <= 10; ++m) {
      const uint32_t shift = 23 - m;
      const uint32_t first = f2u(bias) >> shift;
      const uint32_t n_blocks = (f2u(65535.0f + bias) >> shift) - first + 1;
      const uint32_t n_anchors = (n_blocks + 3) & ~3u;  // keep delta[] 16-byte aligned
      const uint32_t hole = 4u << m;  // the anchors' LDS address is made without an add (ph_lut.h a_scale): they start at 4 * 2^m
      const uint32_t bytes = hole + n_anchors * 4 + 65536 * 2;
      if (bytes > max_bytes || (found && bytes >= info.bytes)) continue;
      std::vector<uint32_t> lo(n_blocks, 0xffffffffu), hi(n_blocks, 0);
      for (uint32_t i = 0; i < 65536; ++i) {
        blk[i] = (f2u((float)i + bias) >> shift) - first;
        lo[blk[i]] = std::min(lo[blk[i]], p[i]);
        hi[blk[i]] = std::max(hi[blk[i]], p[i]);
      }
      bool ok = true;
      for (uint32_t b = 0; b < n_blocks && ok; ++b)
        if (lo[b] != 0xffffffffu && hi[b] - lo[b] >= 65536u) ok = false;
      if (!ok) continue;
      found = true;
      info.bytes = bytes, info.shift = shift, info.first = first, info.n_anchors = n_anchors;
      info.hole = hole, info.delta_off = hole + n_anchors * 4, info.bias = bias;
      blob.assign((bytes - hole) / 4, 0);
      for (uint32_t b = 0; b < n_blocks; ++b) blob[b] = lo[b] == 0xffffffffu ? 0 : lo[b];
      uint16_t *delta = reinterpret_cast<uint16_t *>(blob.data() + n_anchors);
      for (uint32_t i = 0; i < 65536; ++i) delta[i] = (uint16_t)(p[i] - lo[blk[i]]);
    }
  }
  if (!found) return false;
  // Exhaustive self-check with the very operations the kernels use (ph_ldslut.h make_lut_k /
  // lds_lut_fetch): the magic-number add that rounds, the fma that gives (i + bias) * a_scale, the
  // shift and the mask, and the fma whose denormal result is the delta byte address - on an image of
  // the LDS as the kernels fill it (the blob at `hole`).
  const LutView v = lut_view(info, nullptr);
  const float magic = 12582912.0f;  // kRoundMagic
  const float abase = -((magic - v.bias) * v.a_scale);
  const float dbase = (v.delta_base + v.bias * v.delta_scale) - magic * v.delta_scale;
  std::vector<uint8_t> lds(info.bytes, 0);
  std::memcpy(lds.data() + info.hole, blob.data(), info.bytes - info.hole);
  for (uint32_t i = 0; i < 65536; ++i) {
    const float y = (float)i + magic;
    const float fs = std::fmaf(y, v.a_scale, abase);
    if (fs != ((float)i + v.bias) * v.a_scale) return false;
    const uint32_t a_addr = (f2u(fs) >> (v.shift - 2u)) & ~3u;
    const uint32_t d_addr = f2u(std::fmaf(y, v.delta_scale, dbase));
    if (a_addr < info.hole || a_addr + 4 > info.delta_off || d_addr != info.delta_off + 2 * i) return false;
    if (a_addr != info.hole + 4u * ((f2u((float)i + v.bias) >> v.shift) - info.first)) return false;
    uint32_t a;
    uint16_t d;
    std::memcpy(&a, lds.data() + a_addr, 4);
    std::memcpy(&d, lds.data() + d_addr, 2);
    if (a + d != p[i]) return false;
  }
  return true;
}

}  // namespace ph
