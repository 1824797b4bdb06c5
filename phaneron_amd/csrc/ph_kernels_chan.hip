// ph_kernels_chan.hip - a channel's whole video frame as ONE kernel, straight from the v210 sources.
//
// The reference's per-frame job batch of a channel is, per layer, ToRGBA (v210.ts:25-111) -> Mixer's
// `transform` (transform.ts:36-59) -> optionally the Transitioner's dissolve / wipe against a second source
// (transition.ts:54-79) -> and for the channel combine_N (combine.ts:45-65) -> FromRGBA (v210.ts:113-195): every arrow a
// full-size f32 RGBA frame written and read back.  ph_compose_write_v210 (ph_kernels_lds.hip) already folds everything
// from `transform` on into one kernel but still consumes f32 frames, so a 1080p channel moved ~390 MB per frame for
// 38.7 MB of v210 in and out.  Here the bilinear taps are taken from the v210 words themselves: each tap is unpacked,
// matrixed, looked up in the reader's gamma table and gamut-converted on the fly (the arithmetic of ToRGBA, per tap),
// filtered and combined in registers, and only the channel's v210 output leaves the chip.  For half-size insets and
// 1:1 sources the taps are (about) the source pixels, so hardly anything is converted twice; a full-frame layer at
// unit scale converts every source pixel four times, which is still cheaper than a round trip of f32 frames.
//
// Two phases, as in the headline kernel, because reader and writer table do not fit the LDS together:
//   phase 1 (reader table resident): pixel per lane, 64 consecutive pixels of a row per wave step; per layer the
//     placed sample (+ transition), combine_N, then the writer's first step - the 16-bit index sat_rte(rgb * 65535)
//     (v210.ts:148-150) - is parked in an INDEX FRAME (8 bytes per pixel) that lives in the XCD's L2 for the few
//     microseconds until
//   phase 2 (writer table swapped in): quad per lane, indices -> writer table -> RGB->YCbCr matrix -> packed words.
// A workgroup reads back only what it wrote itself, so one workgroup barrier (the table swap's) orders the phases.
// Results are bit-identical to running the separate kernels (tests/test_chan_gpu.py).
//
// Sources are v210 frames, f32 images, a file decoder's planar YCbCr frames and packed 8-bit RGB (ffmpegProducer.ts:398-442); the frame made
// is v210 or another consumer's format.  The kernel exists in several instantiations (chan_compose_v210_kernel<MODE, OUT>, picked by the
// launcher at the bottom) because an instantiation carries the code and scalar state of every sampler it contains.  Two kinds of frame do
// not come here at all: layers that are all plain reads of the output's size (the headline kernel), and frames all of whose layers are
// ENLARGED clips - those are read once per source pixel and composited by ph_kernels_up.hip (ph_api.cpp chan_compose_enlarged).
#include <cstdlib>

#include "ph_device.h"
#include "ph_kernels.h"
#include "ph_ldslut.h"

#pragma clang fp contract(off)

#ifndef PH_CHAN_BALANCE
#define PH_CHAN_BALANCE 1
#endif
// output rows per XCD band.  8, not 16: at 1080p sixteen-row bands give four XCDs 9 bands and four XCDs 8 (22.5 against 20 chunks per
// workgroup: a fifth round of wave steps on half the chip); with eight rows it is 17 against 16 (53.2 -> 51.3 us on config 2's frame)
#ifndef PH_CHAN_GROUP_ROWS
#define PH_CHAN_GROUP_ROWS 8
#endif

// PH_CHAN_PROBE builds (tools/chan_probe.py; never the shipped library): wave 0 of workgroup 0 stamps s_memtime at fixed
// points of its first turns, so the share of each part of a turn can be read off the real kernel.
#ifndef PH_CHAN_PROBE
#define PH_CHAN_PROBE 0
#endif
#ifndef PH_CHAN_ARG_PREFETCH  // 0: an A/B build without the batch kernel's argument-block prefetch
#define PH_CHAN_ARG_PREFETCH 1
#endif
#if PH_CHAN_PROBE
__device__ unsigned long long g_chan_probe[8 * 64];
__device__ unsigned long long g_chan_phase[256 * 16];  // per workgroup: s_memrealtime at the phase boundaries (wave 0)
extern "C" int ph_debug_chan_phase(unsigned long long *out, int n_words) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_chan_phase), (size_t)n_words * 8);
}
#define PH_CPHASE(point)                                                                                       \
  do {                                                                                                         \
    if (threadIdx.x == 0 && blockIdx.x < 256) g_chan_phase[blockIdx.x * 16 + (point)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
__device__ unsigned int g_chan_probe_n;
extern "C" int ph_debug_chan_probe(unsigned long long *out, int n_words) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_chan_probe), (size_t)n_words * 8);
}
#define PH_CSTAMP(point)                                                                    \
  do {                                                                                      \
    if (blockIdx.x == 0 && wave == 0 && probe_turn < 64 && lane == 0)                      \
      g_chan_probe[probe_turn * 8 + (point)] = __builtin_amdgcn_s_memtime();                \
  } while (0)
#else
#define PH_CSTAMP(point) do { } while (0)
#define PH_CPHASE(point) do { } while (0)
#endif

namespace ph {

constexpr uint32_t kChanChunk = 192;          // pixels: 3 wave steps of phase 1, 32 quads of phase 2
constexpr uint32_t kOutsideBit = 0x40000000u;  // row / column offsets of taps outside the frame: beyond any num_records (frames < 1 GiB)

// ---- a v210 column: where pixel i's Y and its pair's Cb / Cr sit inside the 16-byte quad (v210.ts:58-63) ----------
struct V210Col {
  uint32_t g16;  // byte offset of the pixel's quad inside a line
  uint32_t yo, ys, cbo, cbs, cro, crs;
};
__device__ __forceinline__ V210Col v210_col(uint32_t i) {
  // i / 6 as a full-rate 24-bit multiply (v_mul_hi_u32 issues at a quarter of the rate): exact for i < 98304, and a column
  // beyond any frame width is masked by the caller anyway
  const uint32_t g = __umul24(i, 43691u) >> 18;
  const uint32_t j = i - __umul24(g, 6u), pr = j >> 1;
  V210Col c;
  c.g16 = g << 4;
  c.yo = (0xCC8440u >> (4u * j)) & 0xCu;           // Y in word {0,1,1,2,3,3}
  c.ys = __umul24(10u, (0x201201u >> (4u * j)) & 3u);  //   at bit {10,0,20,10,0,20}
  c.cbo = 4u * pr, c.cbs = __umul24(10u, pr);          // Cb in word {0,1,2} at bit {0,10,20}
  c.cro = (0xC80u >> (4u * pr)) & 0xCu;                // Cr in word {0,2,3}
  c.crs = __umul24(10u, (0x102u >> (4u * pr)) & 3u);   //   at bit {20,0,10}
  return c;
}

// ---- sampling ------------------------------------------------------------------------------------------------------------
// A lane computes P = 2 output pixels at a time - (x, line) and (x, line + line_step), one above the other - through
// every op together.  The table fills the LDS, so a SIMD has four waves and a wave on its own runs at the pace of its
// dependency chains and LDS / memory round trips (measured with the in-kernel probe: 10 cycles per instruction);
// two independent pixel streams in one wave fill each other's gaps, and the per-op scalar work (descriptor loads,
// branches) and the per-step work are paid once per pair.
constexpr int kChanP = 2;

struct ChanTaps {  // the `u`, `v` half of the sampler (transform.ts:53-57, OpenCL 1.2 8.2)
  uint32_t i0, j0;
  float a, b;
};
__device__ __forceinline__ ChanTaps chan_taps(const ChanSrc &s, float px, float py) {
  const float sx = dot3(s.m[0], s.m[1], s.m[2], px, py, 1.0f) + 0.5f;
  const float sy = dot3(s.m[3], s.m[4], s.m[5], px, py, 1.0f) + 0.5f;
  const float u = sx * (float)(int)s.w, v = sy * (float)(int)s.h;
  const float fu = u - 0.5f, fv = v - 0.5f;
  const float flu = __builtin_floorf(fu), flv = __builtin_floorf(fv);
  return ChanTaps{(uint32_t)(int)flu, (uint32_t)(int)flv, fu - flu, fv - flv};
}

struct V210Words {
  uint32_t wy, wcb, wcr;
};
__device__ __forceinline__ V210Words v210_load(__amdgpu_buffer_rsrc_t rs, uint32_t base, const V210Col &c) {
  return V210Words{(uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rs, (int)(base + c.yo), 0, 0),
                   (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rs, (int)(base + c.cbo), 0, 0),
                   (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rs, (int)(base + c.cro), 0, 0)};
}
template <bool STD>
__device__ __forceinline__ PxPending v210_issue(const V210Words &w, const V210Col &c, const ReadK &k, const LutK &lut, float last = 1.0f) {
  const float y = (float)__builtin_amdgcn_ubfe(w.wy, c.ys, 10u);
  const float cb = (float)__builtin_amdgcn_ubfe(w.wcb, c.cbs, 10u);
  const float cr = (float)__builtin_amdgcn_ubfe(w.wcr, c.crs, 10u);
  return read_px_issue<STD>(y, cb, cr, k, lut, last);
}
__device__ __forceinline__ float4 rgba_load(__amdgpu_buffer_rsrc_t img, uint32_t off) {
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(img, (int)off, 0, 0);  // outside: 0 = the border colour
  return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

// ---- planar YCbCr sources (what file decoders hand over: ffmpegProducer.ts:398-412) -----------------------------------------------
//   kChanP10    yuv422p10le  Y [row][x], Cb / Cr [row][x / 2], 16-bit samples        yuv422p10.ts:60-72
//   kChanP8x422 yuv422p8     the same with 8-bit samples                             yuv422p8.ts
//   kChanP8x420 yuv420p      Cb / Cr [row / 2][x / 2], 8-bit                          yuv420p.ts
//   kChanNv12   nv12         Cb, Cr interleaved in one plane [row / 2][x / 2]         nv12.ts:61-74
// The reader's arithmetic from the three samples on is ToRGBA's own (read_px_issue), the sample taken as it is (the reference
// converts the whole 8- or 16-bit word).  A tap is three 1- or 2-byte buffer loads; taps outside the frame get offsets beyond
// every plane and load 0.  Everything that depends on the kind is uniform.
struct Planes {
  __amdgpu_buffer_rsrc_t y, u, v;
  uint32_t pitch_y, pitch_c;  // bytes per line of the Y plane / of a chroma plane
  uint32_t wide;              // 1: 16-bit samples
  uint32_t vshift;            // 1: a chroma line serves two luma lines
  uint32_t nv12;              // 1: u holds CbCr pairs (v is u)
};
__device__ __forceinline__ Planes planes_of(const ChanSrc &s, const void *pu, const void *pv) {
  Planes pl;
  pl.wide = s.kind == kChanP10 ? 1u : 0u, pl.vshift = (s.kind == kChanP8x420 || s.kind == kChanNv12) ? 1u : 0u, pl.nv12 = s.kind == kChanNv12 ? 1u : 0u;
  pl.pitch_y = s.pitch, pl.pitch_c = pl.nv12 ? s.pitch : s.pitch >> 1;
  const uint32_t crows = (s.h + pl.vshift) >> pl.vshift;
  pl.y = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(s.ptr), 0, (int)(pl.pitch_y * s.h), 0x00020000);
  pl.u = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(pu), 0, (int)(pl.pitch_c * crows), 0x00020000);
  pl.v = pl.nv12 ? pl.u : __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(pv), 0, (int)(pl.pitch_c * crows), 0x00020000);
  return pl;
}
__device__ __forceinline__ V210Words planar_load(const Planes &pl, uint32_t row, uint32_t col) {  // row inside the frame; col: the pixel's column, or kOutsideBit
  const uint32_t oy = __umul24(row, pl.pitch_y) + (col << pl.wide);
  // the pair's chroma sample: planar [x / 2] samples of 1 or 2 bytes, nv12 [x / 2] pairs of bytes (Cb first)
  const uint32_t ocb = __umul24(row >> pl.vshift, pl.pitch_c) + (pl.nv12 ? (col & ~1u) : ((col >> 1) << pl.wide)), ocr = ocb + pl.nv12;
  if (pl.wide)
    return V210Words{(uint32_t)(uint16_t)__builtin_amdgcn_raw_buffer_load_b16(pl.y, (int)oy, 0, 0),
                     (uint32_t)(uint16_t)__builtin_amdgcn_raw_buffer_load_b16(pl.u, (int)ocb, 0, 0),
                     (uint32_t)(uint16_t)__builtin_amdgcn_raw_buffer_load_b16(pl.v, (int)ocr, 0, 0)};
  return V210Words{(uint32_t)(uint8_t)__builtin_amdgcn_raw_buffer_load_b8(pl.y, (int)oy, 0, 0),
                   (uint32_t)(uint8_t)__builtin_amdgcn_raw_buffer_load_b8(pl.u, (int)ocb, 0, 0),
                   (uint32_t)(uint8_t)__builtin_amdgcn_raw_buffer_load_b8(pl.v, (int)ocr, 0, 0)};
}
template <bool STD>
__device__ __forceinline__ PxPending planar_issue(const V210Words &w, const ReadK &k, const LutK &lut) {
  return read_px_issue<STD>((float)w.wy, (float)w.wcb, (float)w.wcr, k, lut);
}
// (tap sharing: see chan_sample below)
struct ChanHalo {
  bool on;        // uniform
  uint32_t addr;  // LDS byte address of this step's 9 floats: rows j0, j0 + 1, j0 + 2 of column i0(lane 0), r g b each
};
__device__ __forceinline__ float wave_from_prev_lane(float mine, float lane0_gets) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(lane0_gets), __float_as_int(mine), 0x138 /* wave_shr:1: lane i reads lane i - 1 */, 0xf, 0xf, false));
}
template <bool STD, bool SHARE = false>  // SHARE: the instantiation for programs of clips at their own scale (the others carry the plain loop only)
__device__ __forceinline__ void chan_sample_planar(const ChanSrc &s, const void *pu, const void *pv, float px, const float (&py)[kChanP], uint32_t x,
                                                   const uint32_t (&line)[kChanP], const ReadK &k, const LutK &lut, float4 (&out)[kChanP],
                                                   const ChanHalo halo = ChanHalo{false, 0u}) {
  const Planes pl = planes_of(s, pu, pv);
  if (!s.sampled) {
    PxPending pend[kChanP];
#pragma unroll
    for (int p = 0; p < kChanP; ++p) pend[p] = planar_issue<STD>(planar_load(pl, line[p], x), k, lut);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int p = 0; p < kChanP; ++p) out[p] = read_px_finish(pend[p], k);
    return;
  }
  ChanTaps t[kChanP];
  bool touches = false;
#pragma unroll
  for (int p = 0; p < kChanP; ++p) {
    t[p] = chan_taps(s, px, py[p]);
    touches = touches || (t[p].i0 + 1u <= s.w && t[p].j0 + 1u <= s.h);
  }
  if (!__builtin_amdgcn_ballot_w64(touches)) {
#pragma unroll
    for (int p = 0; p < kChanP; ++p) out[p] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    return;
  }
  // Tap sharing as for v210 sources (chan_sample): a clip shown at its own scale - a file decoder's frame on a channel of its format, the
  // reference's everyday case (ffmpegProducer.ts:398-412) - has the lower pixel's upper taps on the upper pixel's lower taps (three source
  // rows for the pair, not four) and lane l's left column on lane l - 1's right one (lane 0's from the halo table): three conversions
  // per pair of pixels instead of eight.
  static_assert(kChanP == 2, "the row sharing below is written for a pair");
  auto load = [&](uint32_t row, uint32_t col) __attribute__((always_inline)) {  // a tap outside the frame loads zeros
    const bool inside = row < s.h && col < s.w;
    return planar_load(pl, inside ? row : 0u, inside ? col : kOutsideBit);  // (rows inside are below 2^24: the 24-bit products are exact)
  };
  // the filter on four converted taps: the sampler's border colour (0, 0, 0, 0) for taps outside (bit i of `in` clear), alpha 1 inside
  auto blend = [&](int p, float4 (&q)[4], uint32_t in) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bool inside = (in >> i) & 1u;
      q[i].x = inside ? q[i].x : 0.0f, q[i].y = inside ? q[i].y : 0.0f, q[i].z = inside ? q[i].z : 0.0f, q[i].w = inside ? 1.0f : 0.0f;
    }
    const float a = t[p].a, b = t[p].b, oma = 1.0f - a, omb = 1.0f - b;
    const float w00 = oma * omb, w10 = a * omb, w01 = oma * b, w11 = a * b;
    out[p].x = ((w00 * q[0].x + w10 * q[1].x) + w01 * q[2].x) + w11 * q[3].x;
    out[p].y = ((w00 * q[0].y + w10 * q[1].y) + w01 * q[2].y) + w11 * q[3].y;
    out[p].z = ((w00 * q[0].z + w10 * q[1].z) + w01 * q[2].z) + w11 * q[3].z;
    out[p].w = ((w00 * q[0].w + w10 * q[1].w) + w01 * q[2].w) + w11 * q[3].w;
  };
  auto inside_bits = [&](int p) __attribute__((always_inline)) {
    const bool ci0 = t[p].i0 < s.w, ci1 = t[p].i0 + 1u < s.w, ri0 = t[p].j0 < s.h, ri1 = t[p].j0 + 1u < s.h;
    return (ci0 && ri0 ? 1u : 0u) | (ci1 && ri0 ? 2u : 0u) | (ci0 && ri1 ? 4u : 0u) | (ci1 && ri1 ? 8u : 0u);
  };
  const bool stacked = SHARE && halo.on && __builtin_amdgcn_ballot_w64(!(t[1].i0 == t[0].i0 && t[1].j0 == t[0].j0 + 1u)) == 0;
  if (SHARE && stacked) {  // (only ops the launcher found at their own scale get here: the others keep the plain loop below as it always was)
    const uint32_t i0_first = (uint32_t)__builtin_amdgcn_readfirstlane((int)t[0].i0), j0_first = (uint32_t)__builtin_amdgcn_readfirstlane((int)t[0].j0);
    const bool shared = __builtin_amdgcn_ballot_w64(!(t[0].i0 == i0_first + (threadIdx.x & 63u) && t[0].j0 == j0_first)) == 0;
    float4 left[3], right[3];
    if (shared) {
      PxPending pend[3];
#pragma unroll
      for (int r = 0; r < 3; ++r) pend[r] = planar_issue<STD>(load(j0_first + (uint32_t)r, t[0].i0 + 1u), k, lut);
      const __attribute__((address_space(3))) float *hp = (const __attribute__((address_space(3))) float *)(uintptr_t)halo.addr;
      float h[9];
#pragma unroll
      for (int i = 0; i < 9; ++i) h[i] = hp[i];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        right[r] = read_px_finish(pend[r], k);
        left[r] = make_float4(wave_from_prev_lane(right[r].x, h[3 * r]), wave_from_prev_lane(right[r].y, h[3 * r + 1]),
                              wave_from_prev_lane(right[r].z, h[3 * r + 2]), 1.0f);
      }
    } else {  // three source rows for the pair, both columns converted by the lane itself
      PxPending pend[6];
#pragma unroll
      for (int i = 0; i < 6; ++i) pend[i] = planar_issue<STD>(load(t[0].j0 + (uint32_t)(i >> 1), t[0].i0 + (uint32_t)(i & 1)), k, lut);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int r = 0; r < 3; ++r) left[r] = read_px_finish(pend[2 * r], k), right[r] = read_px_finish(pend[2 * r + 1], k);
    }
#pragma unroll
    for (int p = 0; p < kChanP; ++p) {
      float4 q[4] = {left[p], right[p], left[p + 1], right[p + 1]};
      blend(p, q, inside_bits(p));
    }
    return;
  }
  // the plain loop, as it always was (a pixel's four taps converted and filtered before the next pixel's are touched)
#pragma unroll
  for (int p = 0; p < kChanP; ++p) {
    const bool ci[2] = {t[p].i0 < s.w, t[p].i0 + 1u < s.w}, ri[2] = {t[p].j0 < s.h, t[p].j0 + 1u < s.h};
    V210Words w[4];
    bool in[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const uint32_t row = t[p].j0 + (uint32_t)(i >> 1), col = t[p].i0 + (uint32_t)(i & 1);
      in[i] = ci[i & 1] && ri[i >> 1];
      w[i] = planar_load(pl, in[i] ? row : 0u, in[i] ? col : kOutsideBit);  // (rows inside are below 2^24: the 24-bit products are exact)
    }
    PxPending pend[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) pend[i] = planar_issue<STD>(w[i], k, lut);
    __builtin_amdgcn_sched_barrier(0);
    float4 q[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      q[i] = read_px_finish(pend[i], k);
      q[i].x = in[i] ? q[i].x : 0.0f, q[i].y = in[i] ? q[i].y : 0.0f, q[i].z = in[i] ? q[i].z : 0.0f, q[i].w = in[i] ? 1.0f : 0.0f;  // the border colour
    }
    const float a = t[p].a, b = t[p].b, oma = 1.0f - a, omb = 1.0f - b;
    const float w00 = oma * omb, w10 = a * omb, w01 = oma * b, w11 = a * b;
    out[p].x = ((w00 * q[0].x + w10 * q[1].x) + w01 * q[2].x) + w11 * q[3].x;
    out[p].y = ((w00 * q[0].y + w10 * q[1].y) + w01 * q[2].y) + w11 * q[3].y;
    out[p].z = ((w00 * q[0].z + w10 * q[1].z) + w01 * q[2].z) + w11 * q[3].z;
    out[p].w = ((w00 * q[0].w + w10 * q[1].w) + w01 * q[2].w) + w11 * q[3].w;
  }
}

// ---- packed 8-bit RGB sources: rgba8 / bgra8 (rgba8.ts:49-62, bgra8.ts) ------------------------------------------------------------------
// One dword per pixel.  Every byte b - alpha too - goes through the gamma table at index b * 65535 / 255 (= b * 257, exactly),
// then the gamut matrix on r, g, b; no YCbCr matrix.  A tap outside the frame is the border colour.
struct Rgb8Pending {
  LutPending r, g, b, a;
};
__device__ __forceinline__ Rgb8Pending rgb8_issue(uint32_t word, bool bgra, const LutK &lut) {
  const float b0 = (float)(word & 0xFFu), b1 = (float)((word >> 8) & 0xFFu), b2 = (float)((word >> 16) & 0xFFu), b3 = (float)(word >> 24);
  Rgb8Pending p;
  p.r = lds_lut_issue(lut, (bgra ? b2 : b0) * 257.0f + kRoundMagic);
  p.g = lds_lut_issue(lut, b1 * 257.0f + kRoundMagic);
  p.b = lds_lut_issue(lut, (bgra ? b0 : b2) * 257.0f + kRoundMagic);
  p.a = lds_lut_issue(lut, b3 * 257.0f + kRoundMagic);
  return p;
}
__device__ __forceinline__ float4 rgb8_finish(const Rgb8Pending &p, const ReadK &k) {
  const float r = lds_lut_finish(p.r), g = lds_lut_finish(p.g), b = lds_lut_finish(p.b);
  return make_float4(dot3(r, g, b, k.gm[0], k.gm[1], k.gm[2]), dot3(r, g, b, k.gm[3], k.gm[4], k.gm[5]), dot3(r, g, b, k.gm[6], k.gm[7], k.gm[8]),
                     lds_lut_finish(p.a));
}
// SHARE: the instantiation for graphics over v210 clips (mode 5): a graphic of the channel's size under the default fill shares its taps as
// clips do - the pair's three source rows, lane l's left column from lane l - 1, lane 0's from the halo table; a graphic has an alpha of
// its own, so its halo entry is TWO slots: r g b of the three rows in `halo.addr`, their alphas in the slot behind it (halo.addr + stride)
template <bool SHARE = false>
__device__ __forceinline__ void chan_sample_rgb8(const ChanSrc &s, float px, const float (&py)[kChanP], uint32_t x, const uint32_t (&line)[kChanP],
                                                 const ReadK &k, const LutK &lut, float4 (&out)[kChanP], const ChanHalo halo = ChanHalo{false, 0u},
                                                 uint32_t halo_stride = 0u) {
  const bool bgra = s.kind == kChanBgra8;  // uniform
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(s.ptr), 0, (int)(s.pitch * s.h), 0x00020000);
  if (!s.sampled) {
    Rgb8Pending pend[kChanP];
#pragma unroll
    for (int p = 0; p < kChanP; ++p)
      pend[p] = rgb8_issue((uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rs, (int)(__umul24(line[p], s.pitch) + (x << 2)), 0, 0), bgra, lut);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int p = 0; p < kChanP; ++p) out[p] = rgb8_finish(pend[p], k);
    return;
  }
  ChanTaps t[kChanP];
  bool touches = false;
#pragma unroll
  for (int p = 0; p < kChanP; ++p) {
    t[p] = chan_taps(s, px, py[p]);
    touches = touches || (t[p].i0 + 1u <= s.w && t[p].j0 + 1u <= s.h);
  }
  if (!__builtin_amdgcn_ballot_w64(touches)) {
#pragma unroll
    for (int p = 0; p < kChanP; ++p) out[p] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    return;
  }
  static_assert(kChanP == 2, "the row sharing below is written for a pair");
  const bool stacked = SHARE && halo.on && __builtin_amdgcn_ballot_w64(!(t[1].i0 == t[0].i0 && t[1].j0 == t[0].j0 + 1u)) == 0;
  if (SHARE && stacked) {
    auto word = [&](uint32_t row, uint32_t col) __attribute__((always_inline)) {
      const uint32_t off = row < s.h && col < s.w ? __umul24(row, s.pitch) + (col << 2) : kOutsideBit;
      return (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rs, (int)off, 0, 0);
    };
    const uint32_t i0_first = (uint32_t)__builtin_amdgcn_readfirstlane((int)t[0].i0), j0_first = (uint32_t)__builtin_amdgcn_readfirstlane((int)t[0].j0);
    const bool shared = __builtin_amdgcn_ballot_w64(!(t[0].i0 == i0_first + (threadIdx.x & 63u) && t[0].j0 == j0_first)) == 0;
    float4 left[3], right[3];
    if (shared) {
      Rgb8Pending pend[3];
#pragma unroll
      for (int r = 0; r < 3; ++r) pend[r] = rgb8_issue(word(j0_first + (uint32_t)r, t[0].i0 + 1u), bgra, lut);
      const __attribute__((address_space(3))) float *hp = (const __attribute__((address_space(3))) float *)(uintptr_t)halo.addr;
      const __attribute__((address_space(3))) float *ha = (const __attribute__((address_space(3))) float *)(uintptr_t)(halo.addr + halo_stride);
      float h[12];
#pragma unroll
      for (int i = 0; i < 9; ++i) h[i] = hp[i];
#pragma unroll
      for (int i = 0; i < 3; ++i) h[9 + i] = ha[i];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        right[r] = rgb8_finish(pend[r], k);
        left[r] = make_float4(wave_from_prev_lane(right[r].x, h[3 * r]), wave_from_prev_lane(right[r].y, h[3 * r + 1]),
                              wave_from_prev_lane(right[r].z, h[3 * r + 2]), wave_from_prev_lane(right[r].w, h[9 + r]));
      }
    } else {
      Rgb8Pending pend[6];
#pragma unroll
      for (int i = 0; i < 6; ++i) pend[i] = rgb8_issue(word(t[0].j0 + (uint32_t)(i >> 1), t[0].i0 + (uint32_t)(i & 1)), bgra, lut);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int r = 0; r < 3; ++r) left[r] = rgb8_finish(pend[2 * r], k), right[r] = rgb8_finish(pend[2 * r + 1], k);
    }
    const bool ci0 = t[0].i0 < s.w, ci1 = t[0].i0 + 1u < s.w;
#pragma unroll
    for (int p = 0; p < kChanP; ++p) {
      const bool ri0 = t[0].j0 + (uint32_t)p < s.h, ri1 = t[0].j0 + (uint32_t)p + 1u < s.h;
      float4 q[4] = {left[p], right[p], left[p + 1], right[p + 1]};
      const bool in[4] = {ci0 && ri0, ci1 && ri0, ci0 && ri1, ci1 && ri1};
#pragma unroll
      for (int i = 0; i < 4; ++i) q[i].x = in[i] ? q[i].x : 0.0f, q[i].y = in[i] ? q[i].y : 0.0f, q[i].z = in[i] ? q[i].z : 0.0f, q[i].w = in[i] ? q[i].w : 0.0f;
      const float a = t[p].a, b = t[p].b, oma = 1.0f - a, omb = 1.0f - b;
      const float w00 = oma * omb, w10 = a * omb, w01 = oma * b, w11 = a * b;
      out[p].x = ((w00 * q[0].x + w10 * q[1].x) + w01 * q[2].x) + w11 * q[3].x;
      out[p].y = ((w00 * q[0].y + w10 * q[1].y) + w01 * q[2].y) + w11 * q[3].y;
      out[p].z = ((w00 * q[0].z + w10 * q[1].z) + w01 * q[2].z) + w11 * q[3].z;
      out[p].w = ((w00 * q[0].w + w10 * q[1].w) + w01 * q[2].w) + w11 * q[3].w;
    }
    return;
  }
#pragma unroll
  for (int p = 0; p < kChanP; ++p) {
    const bool ci[2] = {t[p].i0 < s.w, t[p].i0 + 1u < s.w}, ri[2] = {t[p].j0 < s.h, t[p].j0 + 1u < s.h};
    Rgb8Pending pend[4];
    bool in[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      in[i] = ci[i & 1] && ri[i >> 1];
      const uint32_t off = in[i] ? __umul24(t[p].j0 + (uint32_t)(i >> 1), s.pitch) + ((t[p].i0 + (uint32_t)(i & 1)) << 2) : kOutsideBit;
      pend[i] = rgb8_issue((uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rs, (int)off, 0, 0), bgra, lut);
    }
    __builtin_amdgcn_sched_barrier(0);
    float4 q[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      q[i] = rgb8_finish(pend[i], k);
      q[i].x = in[i] ? q[i].x : 0.0f, q[i].y = in[i] ? q[i].y : 0.0f, q[i].z = in[i] ? q[i].z : 0.0f, q[i].w = in[i] ? q[i].w : 0.0f;  // the border colour
    }
    const float a = t[p].a, b = t[p].b, oma = 1.0f - a, omb = 1.0f - b;
    const float w00 = oma * omb, w10 = a * omb, w01 = oma * b, w11 = a * b;
    out[p].x = ((w00 * q[0].x + w10 * q[1].x) + w01 * q[2].x) + w11 * q[3].x;
    out[p].y = ((w00 * q[0].y + w10 * q[1].y) + w01 * q[2].y) + w11 * q[3].y;
    out[p].z = ((w00 * q[0].z + w10 * q[1].z) + w01 * q[2].z) + w11 * q[3].z;
    out[p].w = ((w00 * q[0].w + w10 * q[1].w) + w01 * q[2].w) + w11 * q[3].w;
  }
}

// the source's samples for the lane's pixels (x, line[p]): 1:1, or through the transform matrix and the bilinear filter
// TAILS: v210 sources may have a width that is not a multiple of 6 (1280 x 720).  The pixels of such a line's tail are unpacked from
// the same bit positions, but the reference's reader converts them without the matrix's offset column (v210.ts:88-93, `last` of
// read_px_issue): a tap in a column >= tail_from passes 0.  (The other instantiation has no such sources: the launcher sees to it.)
// Tap sharing.  A layer shown at its own scale and unrotated (the Mixer's default fill of a full-frame source) has lane l's taps in
// columns c + l and c + l + 1: every column is converted by two lanes.  With `halo.on` (the launcher found the placement, the
// pass in front of phase 1 left the step's first left column in the LDS) and the pattern confirmed for this step (ballots), a lane
// converts its RIGHT column only - three rows for its two stacked pixels - and takes its left column from the lane before it (one
// DPP wave shift per value; lane 0 from the LDS): 3 conversions and 9 loads per pixel pair instead of 6 and 18.  The values are
// the same conversions of the same words, so nothing changes in the result.
template <bool STD, bool TAILS, bool V210 = true>  // V210 = false: an instantiation whose programs have no v210 sources (images only come here)
__device__ __forceinline__ void chan_sample(const ChanSrc &s, float px, const float (&py)[kChanP], uint32_t x, const uint32_t (&line)[kChanP],
                                            const ReadK &k, const LutK &lut, float4 (&out)[kChanP], const ChanHalo halo = ChanHalo{false, 0u}) {
  const bool is_v210 = V210 && s.kind == kChanV210;  // uniform
  auto last_of = [&](uint32_t column) __attribute__((always_inline)) { return TAILS ? (column < s.tail_from ? 1.0f : 0.0f) : 1.0f; };
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(s.ptr), 0, (int)(s.pitch * s.h), 0x00020000);
  if (!s.sampled) {  // uniform: the source has the output's size, pixel for pixel
    if (is_v210) {
      const V210Col c = v210_col(x);
      V210Words w[kChanP];
#pragma unroll
      for (int p = 0; p < kChanP; ++p) w[p] = v210_load(rs, __umul24(line[p], s.pitch) + c.g16, c);
      PxPending pend[kChanP];
#pragma unroll
      for (int p = 0; p < kChanP; ++p) pend[p] = v210_issue<STD>(w[p], c, k, lut, last_of(x));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int p = 0; p < kChanP; ++p) out[p] = read_px_finish(pend[p], k);
    } else {
#pragma unroll
      for (int p = 0; p < kChanP; ++p) out[p] = rgba_load(rs, __umul24(line[p], s.pitch) + (x << 4));
    }
    return;
  }
  ChanTaps t[kChanP];
  bool touches = false;
#pragma unroll
  for (int p = 0; p < kChanP; ++p) {
    t[p] = chan_taps(s, px, py[p]);
    touches = touches || (t[p].i0 + 1u <= s.w && t[p].j0 + 1u <= s.h);  // columns i0, i0 + 1 / rows j0, j0 + 1: any inside
  }
  // a wave none of whose taps touches the source (the outside of a picture-in-picture inset) gets the border value
  // without a load: every product is w * 0
  if (!__builtin_amdgcn_ballot_w64(touches)) {
#pragma unroll
    for (int p = 0; p < kChanP; ++p) out[p] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    return;
  }
  float4 tap[kChanP][4];
  if (is_v210) {
    V210Words w[kChanP][4];
    V210Col col[kChanP][2];
    uint32_t in[kChanP];
    // A layer shown at its own scale (the Mixer's default fill: every full-frame layer) puts the lower pixel's upper taps on the
    // upper pixel's lower taps - same columns, next row.  When that holds for every lane (uniform) the pair needs three source
    // rows, not four: six conversions instead of eight.
    static_assert(kChanP == 2, "the row sharing below is written for a pair");
    const bool stacked = __builtin_amdgcn_ballot_w64(!(t[1].i0 == t[0].i0 && t[1].j0 == t[0].j0 + 1u)) == 0;
    // neighbouring lanes on neighbouring columns, all on the same rows (and lane 0 where the halo pass put it: it ran this very arithmetic)
    const uint32_t i0_first = (uint32_t)__builtin_amdgcn_readfirstlane((int)t[0].i0), j0_first = (uint32_t)__builtin_amdgcn_readfirstlane((int)t[0].j0);
    const bool shared = halo.on && stacked &&
                        __builtin_amdgcn_ballot_w64(!(t[0].i0 == i0_first + (threadIdx.x & 63u) && t[0].j0 == j0_first)) == 0;
    if (shared) {
      const uint32_t c_right = t[0].i0 + 1u;
      const V210Col cr = v210_col(c_right);
      const uint32_t c1 = c_right < s.w ? cr.g16 : kOutsideBit;
      V210Words w3[3];
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const uint32_t row = j0_first + (uint32_t)r;
        w3[r] = v210_load(rs, (row < s.h ? __umul24(row, s.pitch) : kOutsideBit) + c1, cr);
      }
      PxPending pend[3];
      const float last_r = last_of(c_right);
#pragma unroll
      for (int r = 0; r < 3; ++r) pend[r] = v210_issue<STD>(w3[r], cr, k, lut, last_r);
      // the halo of this step: nine floats at a uniform address (a broadcast read), asked for while the table reads are on their way
      const __attribute__((address_space(3))) float *hp = (const __attribute__((address_space(3))) float *)(uintptr_t)halo.addr;
      float h[9];
#pragma unroll
      for (int i = 0; i < 9; ++i) h[i] = hp[i];
      __builtin_amdgcn_sched_barrier(0);
      float4 right[3], left[3];
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        right[r] = read_px_finish(pend[r], k);
        left[r] = make_float4(wave_from_prev_lane(right[r].x, h[3 * r]), wave_from_prev_lane(right[r].y, h[3 * r + 1]),
                              wave_from_prev_lane(right[r].z, h[3 * r + 2]), 1.0f);
      }
      const bool ci0 = t[0].i0 < s.w, ci1 = c_right < s.w;
#pragma unroll
      for (int p = 0; p < kChanP; ++p) {
        const bool ri0 = j0_first + (uint32_t)p < s.h, ri1 = j0_first + (uint32_t)p + 1u < s.h;
        tap[p][0] = left[p], tap[p][1] = right[p], tap[p][2] = left[p + 1], tap[p][3] = right[p + 1];
        in[p] = (ci0 && ri0 ? 1u : 0u) | (ci1 && ri0 ? 2u : 0u) | (ci0 && ri1 ? 4u : 0u) | (ci1 && ri1 ? 8u : 0u);
      }
    } else {
#pragma unroll
    for (int p = 0; p < kChanP; ++p) {
      col[p][0] = v210_col(t[p].i0), col[p][1] = v210_col(t[p].i0 + 1u);
      const uint32_t c0 = t[p].i0 < s.w ? col[p][0].g16 : kOutsideBit, c1 = t[p].i0 + 1u < s.w ? col[p][1].g16 : kOutsideBit;
      // rows inside the frame are below 2^24 and so is a pitch: the 24-bit multiply is exact where its result is used
      const uint32_t r0 = t[p].j0 < s.h ? __umul24(t[p].j0, s.pitch) : kOutsideBit;
      const uint32_t r1 = t[p].j0 + 1u < s.h ? __umul24(t[p].j0 + 1u, s.pitch) : kOutsideBit;
      in[p] = 0;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const uint32_t base = ((i & 2) ? r1 : r0) + ((i & 1) ? c1 : c0);  // >= kOutsideBit when the row or the column is outside: the loads return 0
        if (!(p == 1 && i < 2 && stacked)) w[p][i] = v210_load(rs, base, col[p][i & 1]);
        in[p] |= (base < kOutsideBit ? 1u : 0u) << i;
      }
    }
    float last[kChanP][2];
#pragma unroll
    for (int p = 0; p < kChanP; ++p) last[p][0] = last_of(t[p].i0), last[p][1] = last_of(t[p].i0 + 1u);
    // one pixel's taps are converted together: 24 table reads in flight before the first is consumed
    {
      PxPending pend[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) pend[i] = v210_issue<STD>(w[0][i], col[0][i & 1], k, lut, last[0][i & 1]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 4; ++i) tap[0][i] = read_px_finish(pend[i], k);
    }
    if (stacked) {
      PxPending pend[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) pend[i] = v210_issue<STD>(w[1][2 + i], col[1][i], k, lut, last[1][i]);
      __builtin_amdgcn_sched_barrier(0);
      tap[1][0] = tap[0][2], tap[1][1] = tap[0][3];
#pragma unroll
      for (int i = 0; i < 2; ++i) tap[1][2 + i] = read_px_finish(pend[i], k);
    } else {
      PxPending pend[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) pend[i] = v210_issue<STD>(w[1][i], col[1][i & 1], k, lut, last[1][i & 1]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 4; ++i) tap[1][i] = read_px_finish(pend[i], k);
    }
    }
    bool some_outside = false;
#pragma unroll
    for (int p = 0; p < kChanP; ++p) some_outside = some_outside || in[p] != 0xFu;
    if (__builtin_amdgcn_ballot_w64(some_outside)) {  // the sampler's border colour (0, 0, 0, 0)
#pragma unroll
      for (int p = 0; p < kChanP; ++p)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const bool inside = (in[p] >> i) & 1u;
          float4 &v = tap[p][i];
          v.x = inside ? v.x : 0.0f, v.y = inside ? v.y : 0.0f, v.z = inside ? v.z : 0.0f, v.w = inside ? 1.0f : 0.0f;
        }
    }
  } else {
#pragma unroll
    for (int p = 0; p < kChanP; ++p) {
      const uint32_t c0 = t[p].i0 < s.w ? t[p].i0 << 4 : kOutsideBit, c1 = t[p].i0 + 1u < s.w ? (t[p].i0 + 1u) << 4 : kOutsideBit;
      const uint32_t r0 = t[p].j0 < s.h ? __umul24(t[p].j0, s.pitch) : kOutsideBit;
      const uint32_t r1 = t[p].j0 + 1u < s.h ? __umul24(t[p].j0 + 1u, s.pitch) : kOutsideBit;
#pragma unroll
      for (int i = 0; i < 4; ++i) tap[p][i] = rgba_load(rs, ((i & 2) ? r1 : r0) + ((i & 1) ? c1 : c0));
    }
  }
#pragma unroll
  for (int p = 0; p < kChanP; ++p) {
    const float a = t[p].a, b = t[p].b, oma = 1.0f - a, omb = 1.0f - b;
    const float w00 = oma * omb, w10 = a * omb, w01 = oma * b, w11 = a * b;
    const float4(&q)[4] = tap[p];
    out[p].x = ((w00 * q[0].x + w10 * q[1].x) + w01 * q[2].x) + w11 * q[3].x;
    out[p].y = ((w00 * q[0].y + w10 * q[1].y) + w01 * q[2].y) + w11 * q[3].y;
    out[p].z = ((w00 * q[0].z + w10 * q[1].z) + w01 * q[2].z) + w11 * q[3].z;
    out[p].w = ((w00 * q[0].w + w10 * q[1].w) + w01 * q[2].w) + w11 * q[3].w;
  }
}

// ---- the workgroup's share of the frame ----------------------------------------------------------------------------------
// Unit of work: a CHUNK of 192 pixels x 2 rows (the last chunk of a row may be short: widths are multiples of 48; row pair rp: output rows 2 rp and 2 rp + 1 of the field being written; a
// field with an odd number of rows repeats its last row as its own partner) = 3 wave steps of phase 1, 64 quads of
// phase 2.  Chunks are dealt out XCD-aware as the f32 compositor does (ph_kernels_lds.hip compose_taps_body): groups of
// PH_CHAN_GROUP_ROWS output rows belong to one XCD (blockIdx % 8), so a source row is pulled through one XCD's L2, and
// the chunks of a group go round the XCD's workgroups so that partial-frame layers load them evenly.  Everything here
// is uniform, and the divisions are multiplications by reciprocals the launcher worked out (ChanArgs::magic_*).
struct ChanShare {
  uint32_t chunks, cpg, cpr, xcd, v0, vstep, vend, slots, groups;
  bool banded;
};
template <class A>
__device__ __forceinline__ ChanShare chan_share(const A &a) {
  ChanShare s;
  s.cpr = (a.out_w + kChanChunk - 1u) / kChanChunk;  // chunks per row pair: a chunk never leaves its rows; a row's last chunk may be short (out_w % 48 == 0)
  s.chunks = s.cpr * ((a.lines + 1u) / 2u);
  s.cpg = (uint32_t)(PH_CHAN_GROUP_ROWS / 2) * s.cpr;
  s.banded = (gridDim.x & 7u) == 0;
  s.xcd = 0, s.v0 = blockIdx.x, s.vstep = gridDim.x, s.vend = s.chunks;
  s.groups = (s.chunks + s.cpg - 1u) / s.cpg;
  if (s.banded) {
    s.xcd = blockIdx.x & 7u;
    const uint32_t groups = s.groups, mine = (groups + 7u - s.xcd) / 8u;
    s.v0 = blockIdx.x >> 3, s.vstep = gridDim.x >> 3, s.vend = mine * s.cpg;
  }
  s.slots = s.v0 < s.vend ? (s.vend - s.v0 + s.vstep - 1u) / s.vstep : 0u;  // chunks of this workgroup (the last may be empty)
  return s;
}
template <class A>
__device__ __forceinline__ uint32_t chan_chunk(const A &a, const ChanShare &s, uint32_t slot) {
  const uint32_t v = s.v0 + slot * s.vstep;
  if (!s.banded) return v < s.chunks ? v : ~0u;
  const uint32_t gi = __umulhi(v, a.magic_cpg);  // v / cpg
  const uint32_t chunk = (gi * 8u + s.xcd) * s.cpg + (v - gi * s.cpg);
  return chunk < s.chunks ? chunk : ~0u;
}
// chunk -> its row pair and first column
template <class A>
__device__ __forceinline__ void chan_place(const A &a, const ChanShare &s, uint32_t chunk, uint32_t &rp, uint32_t &x0) {
  rp = s.cpr == 1u ? chunk : __umulhi(chunk, a.magic_cpr);  // chunk / cpr
  x0 = (chunk - rp * s.cpr) * kChanChunk;
}

// what a sample does to the pixel being built (transition.ts:54-79, combine.ts:45-65)
struct ChanAcc {
  float r, g, b;
  float4 hold, in1;
};
__device__ __forceinline__ void chan_apply(const ChanOp &op, const float4 v, ChanAcc &c) {
  const uint32_t act = op.action & 0xFFu;  // uniform
  float4 t = v;
  if (act == kChanActHold) {
    c.hold = v;
    return;
  }
  if (act == kChanActIncoming) {
    c.in1 = v;
    return;
  }
  if (act == kChanActDissolve) {  // fma(in0, mix, in1 * (1 - mix))
    const float m = op.mix, rm = 1.0f - m;
    t.x = fma_rn(c.hold.x, m, v.x * rm), t.y = fma_rn(c.hold.y, m, v.y * rm), t.z = fma_rn(c.hold.z, m, v.z * rm), t.w = fma_rn(c.hold.w, m, v.w * rm);
  } else if (act == kChanActWipe) {  // fma(in1, mask.r, in0 * (1 - mask.r))
    const float m = v.x, rm = 1.0f - m;
    t.x = fma_rn(c.in1.x, m, c.hold.x * rm), t.y = fma_rn(c.in1.y, m, c.hold.y * rm), t.z = fma_rn(c.in1.z, m, c.hold.z * rm), t.w = fma_rn(c.in1.w, m, c.hold.w * rm);
  }
  if (op.action & kChanActFirst) {
    c.r = t.x, c.g = t.y, c.b = t.z;
  } else {  // the result's alpha is never used by the writer
    const float kk = 1.0f - t.w;
    c.r = fma_rn(c.r, kk, t.x), c.g = fma_rn(c.g, kk, t.y), c.b = fma_rn(c.b, kk, t.z);
  }
}

// The pass in front of phase 1 for the ops that share taps (ChanHalo): one lane per (wave step of this workgroup, row) converts the
// LEFT column of the step's first lane - the one value per row no lane of the step converts - with the arithmetic phase 1 itself
// uses for that lane (same step -> chunk -> pixel mapping, same chan_taps), and parks r, g, b in the LDS behind the table.
template <bool STD, bool TAILS, int SRC = 0>
__device__ __forceinline__ void chan_halo_pass(const ChanArgs &a, const ChanShare &sh, const ReadK &rk, const LutK &rlut) {
  if (!a.halo_steps) return;
  const uint32_t steps = 3u * sh.slots < a.halo_steps ? 3u * sh.slots : a.halo_steps;
  const float fow = (float)(int)a.out_w, foh = (float)(int)a.out_h;
  for (int k = 0; k < a.n_ops; ++k) {
    const ChanOp op = a.op[k];
    if (!(op.action & kChanActShare)) continue;  // uniform
    const ChanSrc &s = op.src;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(s.ptr), 0, (int)(s.pitch * s.h), 0x00020000);
    float *const area = reinterpret_cast<float *>(g_lds + a.halo_off) + ((op.action >> kChanActShareShift) & 7u) * a.halo_steps * 9u;
    for (uint32_t item = threadIdx.x; item < 3u * steps; item += kLdsBlock) {
      const uint32_t n = item / 3u, r = item - 3u * n;
      const uint32_t slot = n / 3u, sub = n - 3u * slot;
      const uint32_t chunk = chan_chunk(a, sh, slot);
      float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      if (chunk != ~0u) {
        uint32_t rp, x0;
        chan_place(a, sh, chunk, rp, x0);
        const uint32_t x = x0 + sub * 64u;  // the step's first lane
        const uint32_t line = a.first_line + 2u * rp * a.line_step;
        const ChanTaps t = chan_taps(s, (float)(int)x / fow - 0.5f, (float)(int)line / foh - 0.5f);
        const uint32_t col = t.i0, row = t.j0 + r;
        if (col < s.w && row < s.h) {
          if (SRC == 3 && s.kind >= kChanRgba8) {  // uniform: a packed-RGB graphic (its alpha goes to the slot behind: below)
            const uint32_t word = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rs, (int)(__umul24(row, s.pitch) + (col << 2)), 0, 0);
            v = rgb8_finish(rgb8_issue(word, s.kind == kChanBgra8, rlut), rk);
          } else if (SRC >= 1 && SRC != 3 && s.kind >= kChanP10) {  // uniform: a planar YCbCr clip, converted as chan_sample_planar converts it (its own Loader matrix if it has one)
            const V210Words w = planar_load(planes_of(s, a.plane_u[k], a.plane_v[k]), row, col);
            if (STD) {  // (no op of a launch that takes the short form has a matrix of its own: ChanArgs::any_cm)
              v = read_px_finish(planar_issue<true>(w, rk, rlut), rk);
            } else {
              const float *cm = a.cm_op[k];
              const ReadK own = load_read_k(cm ? cm : a.rd_cm, a.rd_gm);
              v = read_px_finish(planar_issue<false>(w, own, rlut), own);
            }
          } else if (SRC != 2) {
            const V210Col c = v210_col(col);
            const V210Words w = v210_load(rs, __umul24(row, s.pitch) + c.g16, c);
            const PxPending pend = v210_issue<STD>(w, c, rk, rlut, TAILS ? (col < s.tail_from ? 1.0f : 0.0f) : 1.0f);
            v = read_px_finish(pend, rk);
          }
        }
      }
      area[9u * n + 3u * r] = v.x, area[9u * n + 3u * r + 1u] = v.y, area[9u * n + 3u * r + 2u] = v.z;
      if (SRC == 3 && s.kind >= kChanRgba8) area[a.halo_steps * 9u + 9u * n + r] = v.w;  // the graphic's second slot
    }
  }
  __syncthreads();
}

// SRC: what the program's sources may be - 0: v210 frames and f32 images; 1: anything (planar, packed RGB too); 3: v210 frames, packed-RGB
// graphics and f32 images; 2: planar YCbCr frames and f32
// images only (a file decoder's clips, ffmpegProducer.ts:398-412: an instantiation without the v210 and packed-RGB samplers' code and scalar
// state); TAILS: v210 frames (sources, output) may have lines that end in a tail
template <bool STD, int SRC, bool TAILS, bool PSHARE = false>
__device__ __forceinline__ void chan_phase1(const ChanArgs &a, const ChanShare &sh, const ReadK &rk, const LutK &rlut) {
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  uint2 *const index = reinterpret_cast<uint2 *>(a.index);
  const float fow = (float)(int)a.out_w, foh = (float)(int)a.out_h;
  // (taking the steps from an LDS counter as the waves come free, instead of dealing them in turn, measured the same: 51.9 against 51.3 us)
  for (uint32_t n = wave; n < 3u * sh.slots; n += kLdsBlock / 64) {  // 64-column steps, dealt round the waves
#if PH_CHAN_BALANCE
    // the four waves of a SIMD are served oldest first: left alone the oldest races ahead and then idles at the phase barrier
    // while the youngest still works alone at a wave's own pace.  A wave LOWERS its priority as it advances (priority
    // outranks age), so whoever is behind is served first and the waves reach the barrier together (as the headline kernel does)
    {
      const uint32_t quarter = (4u * n) / (3u * sh.slots + 1u);
      if (quarter == 0) __builtin_amdgcn_s_setprio(3);
      else if (quarter == 1) __builtin_amdgcn_s_setprio(2);
      else if (quarter == 2) __builtin_amdgcn_s_setprio(1);
      else __builtin_amdgcn_s_setprio(0);
    }
#endif
    const uint32_t slot = n / 3u, sub = n - 3u * slot;
    const uint32_t chunk = chan_chunk(a, sh, slot);
    if (chunk == ~0u) continue;  // uniform
    uint32_t rp, x0;
    chan_place(a, sh, chunk, rp, x0);
    const uint32_t x = x0 + sub * 64u + lane;
    uint32_t li[kChanP], line[kChanP];
    float py[kChanP];
#pragma unroll
    for (int p = 0; p < kChanP; ++p) {
      li[p] = 2u * rp + (uint32_t)p < a.lines ? 2u * rp + (uint32_t)p : 2u * rp;  // an odd field's last row is its own partner
      line[p] = a.first_line + li[p] * a.line_step;
      py[p] = (float)(int)line[p] / foh - 0.5f;  // transform.ts:53
    }
    const float px = (float)(int)x / fow - 0.5f;
    ChanAcc acc[kChanP];
#pragma unroll
    for (int p = 0; p < kChanP; ++p) acc[p] = ChanAcc{0.0f, 0.0f, 0.0f, make_float4(0.0f, 0.0f, 0.0f, 0.0f), make_float4(0.0f, 0.0f, 0.0f, 0.0f)};
#pragma unroll 1  // one copy of the sampling code whatever the program's length; an op's descriptor is one 64-byte scalar load
    for (int k = 0; k < a.n_ops; ++k) {
      const ChanOp op = a.op[k];
      float4 v[kChanP];
      if ((SRC == 1 || SRC == 3) && op.src.kind >= kChanRgba8) {  // uniform
        ChanHalo halo{false, 0u};
        if (SRC == 3 && (op.action & kChanActShare) && n < a.halo_steps)  // uniform
          halo = ChanHalo{true, a.halo_off + (((op.action >> kChanActShareShift) & 7u) * a.halo_steps + n) * 36u};
        chan_sample_rgb8<SRC == 3>(op.src, px, py, x, line, rk, rlut, v, halo, a.halo_steps * 36u);
      } else if ((SRC == 1 || SRC == 2) && op.src.kind >= kChanP10) {
        // a source with code ranges of its own (8-bit) brings its Loader matrix: the general dot products serve any matrix and give
        // the same bits as the short form where that applies (its missing terms are exact zeros)
        // (ONE copy of the planar sampler per instantiation: a launch in which some source brings a matrix takes the general form throughout -
        // the kernel's any_cm test - so the short form never meets one)
        ChanHalo halo{false, 0u};
        if ((op.action & kChanActShare) && n < a.halo_steps)  // uniform
          halo = ChanHalo{true, a.halo_off + (((op.action >> kChanActShareShift) & 7u) * a.halo_steps + n) * 36u};
        if (STD) {
          chan_sample_planar<true, PSHARE>(op.src, a.plane_u[k], a.plane_v[k], px, py, x, line, rk, rlut, v, halo);
        } else {
          const float *cm = a.cm_op[k];
          chan_sample_planar<false, PSHARE>(op.src, a.plane_u[k], a.plane_v[k], px, py, x, line, load_read_k(cm ? cm : a.rd_cm, a.rd_gm), rlut, v, halo);
        }
      } else {
        ChanHalo halo{false, 0u};
        if ((op.action & kChanActShare) && n < a.halo_steps)  // uniform
          halo = ChanHalo{true, a.halo_off + (((op.action >> kChanActShareShift) & 7u) * a.halo_steps + n) * 36u};
        chan_sample<STD, TAILS, SRC != 2>(op.src, px, py, x, line, rk, rlut, v, halo);
      }
#pragma unroll
      for (int p = 0; p < kChanP; ++p) chan_apply(op, v[p], acc[p]);
    }
    // the writer's first step needs no table (v210.ts:148-150): park the three 16-bit indices
    // (in the tail of a v210 line whose width is not a multiple of 6 the writer truncates its index: v210.ts:176-178)
    const bool trunc_idx = TAILS && x >= a.out_tail_from;
#pragma unroll
    for (int p = 0; p < kChanP; ++p) {
      auto index_of = [&](float t) __attribute__((always_inline)) {
        return __float_as_uint(TAILS ? lds_lut_index_unit_tail(t, trunc_idx) : lds_lut_index_unit(t)) & 0xFFFFu;
      };
      const uint32_t ir = index_of(acc[p].r), ig = index_of(acc[p].g), ib = index_of(acc[p].b);
      if (x < a.out_w) index[li[p] * a.out_w + x] = make_uint2(ir | (ig << 16), ib);  // (lanes beyond a short last chunk have nothing to park)
    }
  }
}

// PLANAR: the program has planar sources (an instantiation of its own: the v210 / image kernel is not touched by them)
// the writer's phase for frames that are not v210 (FromRGBA with the Writers of the reference's other consumers: rgba8 for the screen,
// screenConsumer.ts:131; yuv422p8 for an encoder, ffmpegConsumer.ts:144; yuv422p10).  Same indices, same table; what differs is the packing.
template <int OUT>
__device__ __forceinline__ void chan_phase2_other(const ChanArgs &a, const ChanShare &sh, const WriteK &wk, const LutK &wlut) {
  const uint2 *const index = reinterpret_cast<const uint2 *>(a.index);
  auto idx = [](uint32_t bits) __attribute__((always_inline)) { return __uint_as_float((bits & 0xFFFFu) | 0x4B400000u); };  // M + idx (ph_ldslut.h)
  if (OUT == 5 || OUT == 6) {  // rgba8.ts:69-101: one pixel per lane, alpha 255
    for (uint32_t t = threadIdx.x; t < 2u * kChanChunk * sh.slots; t += kLdsBlock) {
      const uint32_t slot = t / (2u * kChanChunk), within = t - slot * (2u * kChanChunk);
      const uint32_t chunk = chan_chunk(a, sh, slot);
      if (chunk == ~0u) continue;
      uint32_t rp, x0;
      chan_place(a, sh, chunk, rp, x0);
      const uint32_t li = 2u * rp + (within >= kChanChunk ? 1u : 0u), x = x0 + (within >= kChanChunk ? within - kChanChunk : within);
      if (li >= a.lines || x >= a.out_w) continue;
      const uint2 e = index[li * a.out_w + x];
      const PxPending pend = write_px_issue(idx(e.x), idx(e.x >> 16), idx(e.y), wlut);
      const float r = lds_lut_finish(pend.r), g = lds_lut_finish(pend.g), b = lds_lut_finish(pend.b);
      const uint32_t r8 = sat_u8_rte(r * 255.0f), g8 = sat_u8_rte(g * 255.0f), b8 = sat_u8_rte(b * 255.0f);
      const uint32_t line = a.first_line + li * a.line_step;
      reinterpret_cast<uint32_t *>(a.out)[(size_t)line * a.out_pitch + x] = OUT == 5 ? (r8 | g8 << 8 | b8 << 16 | 0xff000000u) : (b8 | g8 << 8 | r8 << 16 | 0xff000000u);
    }
    return;
  }
  // planar 4:2:2 (yuv422p10.ts:140-189, yuv422p8.ts): eight pixels per lane - chroma from the even pixels, codes rounded to 16 bits and
  // cut to the sample width on store (widths are multiples of 8 here: no tail group).
  // 4:2:0 (yuv420p.ts:150-216, nv12.ts:139-196): the same luma; a chroma line serves a line PAIR and is taken from the pair's upper
  // line only (`if (l == 0)`) - a field write makes one line per pair, which then is the one that gives the chroma; nv12 keeps Cb and
  // Cr interleaved in one plane (out_u).
  constexpr bool WIDE = OUT == 1, V420 = OUT == 3 || OUT == 4, NV12 = OUT == 4;
  constexpr uint32_t kGroups = kChanChunk / 8u;  // per chunk row
  for (uint32_t t = threadIdx.x; t < 2u * kGroups * sh.slots; t += kLdsBlock) {
    const uint32_t slot = t / (2u * kGroups), within = t - slot * (2u * kGroups);
    const uint32_t chunk = chan_chunk(a, sh, slot);
    if (chunk == ~0u) continue;
    uint32_t rp, x0;
    chan_place(a, sh, chunk, rp, x0);
    const uint32_t li = 2u * rp + (within >= kGroups ? 1u : 0u), x = x0 + 8u * (within >= kGroups ? within - kGroups : within);
    if (li >= a.lines || x >= a.out_w) continue;
    const uint4 *const e4 = reinterpret_cast<const uint4 *>(index + li * a.out_w + x);
    const uint4 w0 = load_stream(e4), w1 = load_stream(e4 + 1), w2 = load_stream(e4 + 2), w3 = load_stream(e4 + 3);
    const uint32_t pk[16] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w, w2.x, w2.y, w2.z, w2.w, w3.x, w3.y, w3.z, w3.w};
    uint32_t y[8], u[4], v[4];
#pragma unroll
    for (int half = 0; half < 2; ++half) {  // four pixels' twelve table reads in flight at a time
      PxPending pend[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) pend[j] = write_px_issue(idx(pk[2 * (4 * half + j)]), idx(pk[2 * (4 * half + j)] >> 16), idx(pk[2 * (4 * half + j) + 1]), wlut);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int p = 4 * half + j;
        const float gr = lds_lut_finish(pend[j].r), gg = lds_lut_finish(pend[j].g), gb = lds_lut_finish(pend[j].b);
        y[p] = sat_u16_rte(dot4(gr, gg, gb, 1.0f, wk.y));
        if (!(p & 1)) u[p >> 1] = sat_u16_rte(dot4(gr, gg, gb, 1.0f, wk.u)), v[p >> 1] = sat_u16_rte(dot4(gr, gg, gb, 1.0f, wk.v));
      }
    }
    const uint32_t line = a.first_line + li * a.line_step;
    const size_t o8 = ((size_t)line * a.out_pitch + x) >> 3;
    if (V420) {
      uint2 wy;
      wy.x = (y[0] & 0xff) | (y[1] & 0xff) << 8 | (y[2] & 0xff) << 16 | y[3] << 24;
      wy.y = (y[4] & 0xff) | (y[5] & 0xff) << 8 | (y[6] & 0xff) << 16 | y[7] << 24;
      reinterpret_cast<uint2 *>(a.out)[o8] = wy;
      if (a.line_step == 2u || !(line & 1u)) {  // the line that gives its pair the chroma
        const size_t c8 = ((size_t)(line >> 1) * a.out_pitch + x) >> 3;
        if (NV12) {
          uint2 wc;
          wc.x = (u[0] & 0xff) | (v[0] & 0xff) << 8 | (u[1] & 0xff) << 16 | v[1] << 24;
          wc.y = (u[2] & 0xff) | (v[2] & 0xff) << 8 | (u[3] & 0xff) << 16 | v[3] << 24;
          reinterpret_cast<uint2 *>(a.out_u)[c8] = wc;
        } else {
          reinterpret_cast<uint32_t *>(a.out_u)[c8] = (u[0] & 0xff) | (u[1] & 0xff) << 8 | (u[2] & 0xff) << 16 | u[3] << 24;
          reinterpret_cast<uint32_t *>(a.out_v)[c8] = (v[0] & 0xff) | (v[1] & 0xff) << 8 | (v[2] & 0xff) << 16 | v[3] << 24;
        }
      }
    } else if (WIDE) {
      uint4 wy;
      wy.x = (y[0] & 0xffff) | y[1] << 16, wy.y = (y[2] & 0xffff) | y[3] << 16, wy.z = (y[4] & 0xffff) | y[5] << 16, wy.w = (y[6] & 0xffff) | y[7] << 16;
      store_stream(reinterpret_cast<uint4 *>(a.out) + o8, wy);
      uint2 wu, wv;
      wu.x = (u[0] & 0xffff) | u[1] << 16, wu.y = (u[2] & 0xffff) | u[3] << 16;
      wv.x = (v[0] & 0xffff) | v[1] << 16, wv.y = (v[2] & 0xffff) | v[3] << 16;
      reinterpret_cast<uint2 *>(a.out_u)[o8] = wu;
      reinterpret_cast<uint2 *>(a.out_v)[o8] = wv;
    } else {  // uchar = (uchar)ushort keeps the low 8 bits (yuv422p8.ts:166-168)
      uint2 wy;
      wy.x = (y[0] & 0xff) | (y[1] & 0xff) << 8 | (y[2] & 0xff) << 16 | y[3] << 24;
      wy.y = (y[4] & 0xff) | (y[5] & 0xff) << 8 | (y[6] & 0xff) << 16 | y[7] << 24;
      reinterpret_cast<uint2 *>(a.out)[o8] = wy;
      reinterpret_cast<uint32_t *>(a.out_u)[o8] = (u[0] & 0xff) | (u[1] & 0xff) << 8 | (u[2] & 0xff) << 16 | u[3] << 24;
      reinterpret_cast<uint32_t *>(a.out_v)[o8] = (v[0] & 0xff) | (v[1] & 0xff) << 8 | (v[2] & 0xff) << 16 | v[3] << 24;
    }
  }
}

// MODE: 0 = v210 frames whose lines end on a 48-pixel block and f32 images (the fast instantiation); 1 = the same with lines that may end
// in a tail (1280 x 720: sources and / or output); 2 = everything (planar and packed-RGB sources, frames other than v210); 3 = planar YCbCr
// clips and f32 images into a v210 frame (what file playback is: ffmpegProducer.ts:398-412); 4 = the same with the clips' taps shared
// (clips at their own scale: a program that is mostly such clips - the tap-sharing paths cost the plain loop's register allocation); 5 = v210
// clips, packed-RGB graphics and f32 images, the graphics' taps shared too (a logo or a lower third over a live source).  An
// instantiation carries the scalar state of every path it contains, whether a launch takes it or not - the channel kernel's op loop
// spills scalars to VGPR lanes (6 in mode 0, 76 in mode 2) - so 1280-wide v210 channels get one of their own.
// OUT: the packed frame's format (PH_FMT_*: 0 v210; 2 yuv422p8 and 5 rgba8 - the other consumers' - with the lean modes too; the rest only with MODE 2)
template <int MODE, int OUT = 0>
__global__ __launch_bounds__(kLdsBlock) void chan_compose_v210_kernel(ChanArgs a) {
  constexpr int SRC = MODE == 2 ? 1 : MODE == 5 ? 3 : MODE >= 3 ? 2 : 0;
  constexpr bool TAILS = MODE >= 1, PSHARE = MODE == 4;
  const ReadK rk = load_read_k(a.rd_cm, a.rd_gm);
  const LutK rlut = make_lut_k(a.rd);
  const ChanShare sh = chan_share(a);
  PH_CPHASE(0);
  if (!a.images_only) lds_lut_load(a.rd);  // (uniform; a program of f32 images converts nothing)
  __syncthreads();
  PH_CPHASE(1);
  if (ycbcr_matrix_is_standard(rk) && !(SRC >= 1 && a.any_cm)) {
    chan_halo_pass<true, TAILS, SRC>(a, sh, rk, rlut);
    chan_phase1<true, SRC, TAILS, PSHARE>(a, sh, rk, rlut);
  } else {
    chan_halo_pass<false, TAILS, SRC>(a, sh, rk, rlut);
    chan_phase1<false, SRC, TAILS, PSHARE>(a, sh, rk, rlut);
  }
  PH_CPHASE(2);
  __syncthreads();  // every index of this workgroup has been stored (the barrier drains the stores) and nobody reads the reader table any more
  PH_CPHASE(3);
  lds_lut_load(a.wr);
  __syncthreads();
  PH_CPHASE(4);
  // (the writer's constants are loaded HERE, not at the top: seventeen scalar registers that phase 1 - which spills scalars to
  // VGPR lanes as it is - does not have to keep alive)
  const WriteK wk = load_write_k(a.wr_cm);
  const LutK wlut = make_lut_k(a.wr);
  if (OUT != 0) {
    chan_phase2_other<OUT>(a, sh, wk, wlut);
    return;
  }
  // phase 2: one quad per lane.  The indices were written by other waves of THIS workgroup: read past the L1.
  // A line has out_qpitch quad slots.  Widths that are multiples of 48 fill them all; otherwise (the wire-format instantiation
  // only) `full` whole quads are followed by the tail quad (v210.ts:169-194) and by slots the reference's writer clears (:131-136).
  const uint32_t qpl = a.out_qpitch, full = a.out_w / 6u, remain = a.out_w - 6u * full;
  const uint4 *const index = reinterpret_cast<const uint4 *>(a.index);
  for (uint32_t q = threadIdx.x; q < 64u * sh.slots; q += kLdsBlock) {
    const uint32_t chunk = chan_chunk(a, sh, q >> 6);
    if (chunk == ~0u) continue;
    uint32_t rp, x0;
    chan_place(a, sh, chunk, rp, x0);
    const uint32_t li = 2u * rp + ((q >> 5) & 1u);  // quads 0..31 of a chunk: its upper row, 32..63: its lower row
    const uint32_t g = x0 / 6u + (q & 31u);
    if (li >= a.lines || g >= qpl) continue;
    const uint32_t line = a.first_line + li * a.line_step;
    uint4 *const dst = reinterpret_cast<uint4 *>(a.out) + (size_t)line * qpl + g;
    if (TAILS && g > full - (remain ? 0u : 1u)) {  // past the line's pixels
      store_stream(dst, make_uint4(0u, 0u, 0u, 0u));
      continue;
    }
    const uint32_t first_px = li * a.out_w + x0 + (q & 31u) * 6u;
    const uint4 w0 = load_stream(index + (first_px >> 1)), w1 = load_stream(index + (first_px >> 1) + 1), w2 = load_stream(index + (first_px >> 1) + 2);
    const uint32_t pk[12] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w, w2.x, w2.y, w2.z, w2.w};
    float yi[18];
#pragma unroll
    for (int j = 0; j < 6; ++j) {  // M + idx: the index ORed into the mantissa of 1.5 * 2^23 (ph_ldslut.h)
      yi[3 * j] = __uint_as_float((pk[2 * j] & 0xFFFFu) | 0x4B400000u);
      yi[3 * j + 1] = __uint_as_float((pk[2 * j] >> 16) | 0x4B400000u);
      yi[3 * j + 2] = __uint_as_float((pk[2 * j + 1] & 0xFFFFu) | 0x4B400000u);
    }
    if (TAILS && g == full) store_stream(dst, write_quad_idx_lds_tail(yi, wk, wlut, remain));  // (the index frame is padded: the tail's loads stay inside)
    else store_stream(dst, write_quad_idx_lds(yi, wk, wlut));
  }
  PH_CPHASE(5);
}

// ---- several channels' frames per launch ------------------------------------------------------------------------------------------
// A 1080p frame is 63 or 66 wave steps per workgroup for sixteen waves: four rounds of steps of very different price, the last
// of which decides when the workgroup reaches its barrier, and two table loads - 4.9 + 3.1 of 46 us (profiles/r04_chan_probe.jsonl).
// The reference runs four such channels in one context through one queue (src/index.ts:45-71,156-160).  Here the jobs of a launch
// (frames of one geometry and colour recipe, each with its own layers) share the workgroups: every workgroup takes chunks of every
// frame, the tables are loaded once, and the steps of all jobs together are dealt to the waves - sixteen rounds instead of four
// in front of ONE barrier.  Which workgroup has a chunk more than its neighbours, and which XCD a band more, rotates from job
// to job (ChanBatchArgs::job_rot), so the shares add up evenly.  The words of the halo columns (ChanHalo) are asked for before
// the table is there and converted once it is.  Everything a step computes is what the one-job kernel computes for it.
//
// What was built on top of this and measured NOT to pay (round 5, profiles/r05_chan_sched_probe.txt, commit c6043a8): the workgroup
// pricing its steps (which ops' boxes a step touches) and handing them out at run time from a sorted list, cheapest last - the
// wait for the slowest wave does shrink (4.8 -> 2.6 us per launch), but a workgroup's phase 1 as a whole only by 0.5 - 1.2 us,
// and making the list costs 3 us per launch even beside the table's DMA (four barriers' worth of LDS round trips while the DMA
// fills the LDS); dearest-first over ALL steps slows phase 1 itself (like steps side by side use the pipes worse than a mix).
constexpr uint32_t kSchedOut = 16, kSchedIndex = kSchedOut + 8 * kMaxChanJobs, kSchedFirst = kSchedIndex + 8 * kMaxChanJobs,
                   kSchedShare = kSchedFirst + 4 * kMaxChanJobs, kSchedBytes = kSchedShare + 16 * kMaxChanJobs;
// byte offsets inside the batch kernel's LDS area behind the table: the jobs' pointers, first lines and shares, indexable per lane

// One aligned dword-vector of the argument block per lane, through a vector load (the scalar loads the compiler makes of plain
// accesses are one cold miss after the other where several entries are read in a loop)
template <class T>
__device__ __forceinline__ T chan_arg_lane(size_t offset, uint32_t index) {
  static_assert(sizeof(T) % 4 == 0, "dword vectors");
  T v;
  const __attribute__((address_space(1))) uint32_t *p =
      (const __attribute__((address_space(1))) uint32_t *)((uintptr_t)__builtin_amdgcn_kernarg_segment_ptr() + offset + index * sizeof(T));
#pragma unroll
  for (size_t i = 0; i < sizeof(T) / 4; ++i) reinterpret_cast<uint32_t *>(&v)[i] = p[i];
  return v;
}

// a job's share of this workgroup: as chan_share, with the workgroup's place among its XCD's workgroups and the XCD's place among
// the bands rotated by the job (so that the workgroups / XCDs with one chunk / band more are different ones for every job)
struct ChanJobShare {
  uint32_t v0, xcd, vend;
};
__device__ __forceinline__ ChanJobShare chan_job_share(const ChanBatchArgs &a, const ChanShare &sh, uint32_t j) {  // j uniform
  ChanJobShare s;
  s.xcd = sh.banded ? (sh.xcd + j) & 7u : 0u;
  s.vend = sh.banded ? ((sh.groups + 7u - s.xcd) >> 3) * sh.cpg : sh.chunks;
  const uint32_t word = a.job_rot[j][s.xcd >> 1];  // two 16-bit values per word
  s.v0 = sh.v0 + ((s.xcd & 1u) ? word >> 16 : word & 0xFFFFu);
  s.v0 = s.v0 >= sh.vstep ? s.v0 - sh.vstep : s.v0;
  return s;
}
__device__ __forceinline__ uint32_t chan_chunk_of(const ChanBatchArgs &a, const ChanShare &sh, uint32_t v0, uint32_t xcd, uint32_t vend, uint32_t slot) {
  const uint32_t v = v0 + slot * sh.vstep;
  if (v >= vend) return ~0u;
  if (!sh.banded) return v;
  const uint32_t gi = __umulhi(v, a.magic_cpg);  // v / cpg
  const uint32_t chunk = (gi * 8u + xcd) * sh.cpg + (v - gi * sh.cpg);
  return chunk < sh.chunks ? chunk : ~0u;
}

// The halo tables (ChanHalo) as in the one-job kernel, the sharing ops side by side: waves [s * wps, (s + 1) * wps) fill op s's table
// (what depends on the op is then uniform, and every op's loads are in flight at once).
// (Asking for the columns' words BEFORE the table's DMA and converting them after it was built and measured: the table then takes
// 3.7 us to arrive instead of 1.6 - its wait is for the words too, which come late in the chip-wide rush for the table - 6.0 us of
// prologue against 4.3.)
template <bool STD, bool TAILS>
__device__ __forceinline__ void chan_halo_pass_batch(const ChanBatchArgs &a, const ChanShare &sh, const ReadK &rk, const LutK &rlut) {
  if (!a.n_share) return;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63u;
  const uint32_t wps = (uint32_t)(kLdsBlock / 64) / a.n_share, s = wave / wps, ws = wave - s * wps;
  if (s >= a.n_share) return;
  const uint32_t packed = a.share_op[s];  // op | its job << 8 | the job's first line << 16 (the launcher)
  const ChanOp op = a.op[packed & 0xFFu];
  const ChanSrc &src = op.src;
  const uint32_t first_line = packed >> 16;
  const uint4 jsv = reinterpret_cast<const uint4 *>(g_lds + a.sched_off + kSchedShare)[(packed >> 8) & 0xFFu];
  const uint32_t v0 = __builtin_amdgcn_readfirstlane(jsv.x), xcd = __builtin_amdgcn_readfirstlane(jsv.y), vend = __builtin_amdgcn_readfirstlane(jsv.z);
  const float fow = (float)(int)a.out_w, foh = (float)(int)a.out_h;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(src.ptr), 0, (int)(src.pitch * src.h), 0x00020000);
  float *const area = reinterpret_cast<float *>(g_lds + a.halo_off) + s * a.halo_steps * 9u;
  for (uint32_t item = ws * 64u + lane; item < 3u * a.halo_steps; item += wps * 64u) {
    const uint32_t n = item / 3u, r = item - 3u * n;
    const uint32_t slot = n / 3u, sub = n - 3u * slot;
    const uint32_t chunk = chan_chunk_of(a, sh, v0, xcd, vend, slot);
    float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (chunk != ~0u) {
      uint32_t rp, x0;
      chan_place(a, sh, chunk, rp, x0);
      const uint32_t x = x0 + sub * 64u;
      const uint32_t line = first_line + 2u * rp * a.line_step;
      const ChanTaps t = chan_taps(src, (float)(int)x / fow - 0.5f, (float)(int)line / foh - 0.5f);
      const uint32_t col = t.i0, row = t.j0 + r;
      if (col < src.w && row < src.h) {
        const V210Col c = v210_col(col);
        const V210Words w = v210_load(rs, __umul24(row, src.pitch) + c.g16, c);
        const PxPending pend = v210_issue<STD>(w, c, rk, rlut, TAILS ? (col < src.tail_from ? 1.0f : 0.0f) : 1.0f);
        v = read_px_finish(pend, rk);
      }
    }
    area[3u * item] = v.x, area[3u * item + 1u] = v.y, area[3u * item + 2u] = v.z;  // (9 n + 3 r = 3 item)
  }
}

template <bool STD, bool TAILS, bool PLANAR = false>
__device__ __forceinline__ void chan_phase1_batch(const ChanBatchArgs &a, const ChanShare &sh, const ReadK &rk, const LutK &rlut) {
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t S = a.steps, total = a.jobs * S;  // S: the most wave steps any workgroup has per job (a step beyond this workgroup's own has no chunk)
  const float fow = (float)(int)a.out_w, foh = (float)(int)a.out_h;
  uint32_t *const counter = reinterpret_cast<uint32_t *>(g_lds + a.sched_off);
  // The steps of all jobs, one job after the other, TAKEN by the waves as they come free (an LDS counter).  Dealt in turn (wave w:
  // steps w, w + 16, ...) four jobs' steps left the workgroup waiting 21 of 132 us for its slowest wave: a step's price goes with
  // its place in the frame, and a fixed stride meets the same places again and again.
  for (;;) {
    uint32_t i = 0u;
    if (lane == 0u) i = atomicAdd(counter, 1u);
    i = __builtin_amdgcn_readfirstlane(i);
    if (i >= total) break;
#if PH_CHAN_BALANCE
    {  // as in the one-job kernel: a wave lowers its priority as it advances, so the SIMD serves whoever is behind
      const uint32_t quarter = (4u * i) / (total + 1u);
      if (quarter == 0) __builtin_amdgcn_s_setprio(3);
      else if (quarter == 1) __builtin_amdgcn_s_setprio(2);
      else if (quarter == 2) __builtin_amdgcn_s_setprio(1);
      else __builtin_amdgcn_s_setprio(0);
    }
#endif
    const uint32_t j = __umulhi(i, a.magic_spj), n = i - j * S;
    const uint4 jsv = reinterpret_cast<const uint4 *>(g_lds + a.sched_off + kSchedShare)[j];  // the job's share (chan_job_share), left here by the prologue
    const uint32_t slot = n / 3u, sub = n - 3u * slot;
    const uint32_t chunk = chan_chunk_of(a, sh, __builtin_amdgcn_readfirstlane(jsv.x), __builtin_amdgcn_readfirstlane(jsv.y),
                                         __builtin_amdgcn_readfirstlane(jsv.z), slot);
    if (chunk == ~0u) continue;  // uniform
    const ChanJob jb = a.job[j];
    uint2 *const index = reinterpret_cast<uint2 *>(jb.index);
    uint32_t rp, x0;
    chan_place(a, sh, chunk, rp, x0);
    const uint32_t x = x0 + sub * 64u + lane;
    uint32_t li[kChanP], line[kChanP];
    float py[kChanP];
#pragma unroll
    for (int p = 0; p < kChanP; ++p) {
      li[p] = 2u * rp + (uint32_t)p < a.lines ? 2u * rp + (uint32_t)p : 2u * rp;
      line[p] = jb.first_line + li[p] * a.line_step;
      py[p] = (float)(int)line[p] / foh - 0.5f;
    }
    const float px = (float)(int)x / fow - 0.5f;
    ChanAcc acc[kChanP];
#pragma unroll
    for (int p = 0; p < kChanP; ++p) acc[p] = ChanAcc{0.0f, 0.0f, 0.0f, make_float4(0.0f, 0.0f, 0.0f, 0.0f), make_float4(0.0f, 0.0f, 0.0f, 0.0f)};
#pragma unroll 1
    for (uint32_t k = jb.first_op; k < jb.first_op + jb.n_ops; ++k) {
      const ChanOp op = a.op[k];
      float4 v[kChanP];
      ChanHalo halo{false, 0u};
      if ((op.action & kChanActShare) && n < a.halo_steps)  // uniform
        halo = ChanHalo{true, a.halo_off + (((op.action >> kChanActShareShift) & 7u) * a.halo_steps + n) * 36u};
      // PLANAR: a decoder's planes / a packed-RGB graphic among the sources - the one-job kernel's "everything" sampling (chan_phase1, SRC == 1)
      if (PLANAR && op.src.kind >= kChanRgba8) {  // uniform
        chan_sample_rgb8<false>(op.src, px, py, x, line, rk, rlut, v, ChanHalo{false, 0u}, 0u);
      } else if (PLANAR && op.src.kind >= kChanP10) {
        if (STD) {
          chan_sample_planar<true, false>(op.src, a.plane_u[k], a.plane_v[k], px, py, x, line, rk, rlut, v, ChanHalo{false, 0u});
        } else {
          const uint32_t ci = a.cm_idx[k];
          chan_sample_planar<false, false>(op.src, a.plane_u[k], a.plane_v[k], px, py, x, line, load_read_k(ci ? a.cm_tab[ci - 1u] : a.rd_cm, a.rd_gm), rlut, v, ChanHalo{false, 0u});
        }
      } else {
        chan_sample<STD, TAILS>(op.src, px, py, x, line, rk, rlut, v, halo);
      }
#pragma unroll
      for (int p = 0; p < kChanP; ++p) chan_apply(op, v[p], acc[p]);
    }
    const bool trunc_idx = TAILS && x >= a.out_tail_from;
#pragma unroll
    for (int p = 0; p < kChanP; ++p) {
      auto index_of = [&](float t) __attribute__((always_inline)) {
        return __float_as_uint(TAILS ? lds_lut_index_unit_tail(t, trunc_idx) : lds_lut_index_unit(t)) & 0xFFFFu;
      };
      const uint32_t ir = index_of(acc[p].r), ig = index_of(acc[p].g), ib = index_of(acc[p].b);
      if (x < a.out_w) index[li[p] * a.out_w + x] = make_uint2(ir | (ig << 16), ib);
    }
  }
}

template <bool TAILS, bool PLANAR = false>
__global__ __launch_bounds__(kLdsBlock) void chan_compose_batch_kernel(ChanBatchArgs a) {
  const ReadK rk = load_read_k(a.rd_cm, a.rd_gm);
  const LutK rlut = make_lut_k(a.rd);
  const ChanShare sh = chan_share(a);
  PH_CPHASE(0);
  // The jobs' shares (chan_job_share, in vector code: lane j for job j) and pointers where the steps and phase 2 can index them per
  // lane.  Through ONE vector load per lane: a scalar load per job is a chain of cold misses of the argument block, a microsecond each
  // (measured: the workgroup waited 3.8 us for the wave that made four jobs' tables that way).
  if (threadIdx.x < a.jobs) {
    const uint32_t j = threadIdx.x;
    const uint4 jw0 = chan_arg_lane<uint4>(offsetof(ChanBatchArgs, job), 2u * j), jw1 = chan_arg_lane<uint4>(offsetof(ChanBatchArgs, job), 2u * j + 1u);
    const uint4 rot = chan_arg_lane<uint4>(offsetof(ChanBatchArgs, job_rot), j);
    const uint32_t xcd = sh.banded ? (sh.xcd + j) & 7u : 0u;
    const uint32_t vend = sh.banded ? ((sh.groups + 7u - xcd) >> 3) * sh.cpg : sh.chunks;
    const uint32_t word = (xcd >> 1) == 0u ? rot.x : (xcd >> 1) == 1u ? rot.y : (xcd >> 1) == 2u ? rot.z : rot.w;
    uint32_t v0 = sh.v0 + ((xcd & 1u) ? word >> 16 : word & 0xFFFFu);
    v0 = v0 >= sh.vstep ? v0 - sh.vstep : v0;
    reinterpret_cast<uint4 *>(g_lds + a.sched_off + kSchedShare)[j] = make_uint4(v0, xcd, vend, 0u);
    reinterpret_cast<uint2 *>(g_lds + a.sched_off + kSchedOut)[j] = make_uint2(jw0.x, jw0.y);    // ChanJob::out
    reinterpret_cast<uint2 *>(g_lds + a.sched_off + kSchedIndex)[j] = make_uint2(jw0.z, jw0.w);  // ChanJob::index
    reinterpret_cast<uint32_t *>(g_lds + a.sched_off + kSchedFirst)[j] = jw1.z;                  // ChanJob::first_line
  }
  if (threadIdx.x == 64u) *reinterpret_cast<uint32_t *>(g_lds + a.sched_off) = 0u;  // the counter the waves take their steps from
  // The ops' descriptors are 64 bytes each that nobody has touched yet: the halo pass and every wave's first step would each wait a
  // cold miss for theirs.  The last wave asks for the whole argument block at once (one lane per 64 bytes; vector loads fill the L2
  // the scalar cache misses into) while the table comes in.
  uint32_t touched = 0u;
  if (PH_CHAN_ARG_PREFETCH && threadIdx.x >= (uint32_t)kLdsBlock - 64u && (threadIdx.x & 63u) * 64u < (uint32_t)sizeof(ChanBatchArgs))
    touched = *(const __attribute__((address_space(1))) uint32_t *)((uintptr_t)__builtin_amdgcn_kernarg_segment_ptr() + (threadIdx.x & 63u) * 64u);
  PH_CPHASE(7);
  if (!a.images_only) lds_lut_load(a.rd);  // (uniform)
  if (touched == 0x9E3779B9u && a.jobs == 0xFFFFFFFFu) reinterpret_cast<uint32_t *>(g_lds + a.sched_off)[3] = touched;  // (never: keeps the loads above)
  __syncthreads();
  PH_CPHASE(1);
  if (ycbcr_matrix_is_standard(rk) && !(PLANAR && a.any_cm)) {
    chan_halo_pass_batch<true, TAILS>(a, sh, rk, rlut);
    __syncthreads();
    PH_CPHASE(6);
    chan_phase1_batch<true, TAILS, PLANAR>(a, sh, rk, rlut);
  } else {
    chan_halo_pass_batch<false, TAILS>(a, sh, rk, rlut);
    __syncthreads();
    PH_CPHASE(6);
    chan_phase1_batch<false, TAILS, PLANAR>(a, sh, rk, rlut);
  }
  PH_CPHASE(2);
  __syncthreads();
  PH_CPHASE(3);
  lds_lut_load(a.wr);
  __syncthreads();
  PH_CPHASE(4);
  const WriteK wk = load_write_k(a.wr_cm);
  const LutK wlut = make_lut_k(a.wr);
  // phase 2 over the quads of all jobs in one sweep; a quad's job by division, that job's share and pointers from the LDS
  const uint32_t qpl = a.out_qpitch, full = a.out_w / 6u, remain = a.out_w - 6u * full;
  const uint32_t qpj = (a.steps / 3u) * 64u;  // (slots beyond this workgroup's own have no chunk)
  for (uint32_t t = threadIdx.x; t < a.jobs * qpj; t += kLdsBlock) {
    const uint32_t j = __umulhi(t, a.magic_qpj), q = t - j * qpj;
    const uint4 js = reinterpret_cast<const uint4 *>(g_lds + a.sched_off + kSchedShare)[j];
    const uint32_t chunk = chan_chunk_of(a, sh, js.x, js.y, js.z, q >> 6);
    if (chunk == ~0u) continue;
    uint32_t rp, x0;
    chan_place(a, sh, chunk, rp, x0);
    const uint32_t li = 2u * rp + ((q >> 5) & 1u);
    const uint32_t g = x0 / 6u + (q & 31u);
    if (li >= a.lines || g >= qpl) continue;
    const uint32_t line = reinterpret_cast<const uint32_t *>(g_lds + a.sched_off + kSchedFirst)[j] + li * a.line_step;
    uint4 *const dst = reinterpret_cast<uint4 *>((uintptr_t) reinterpret_cast<const uint64_t *>(g_lds + a.sched_off + kSchedOut)[j]) + (size_t)line * qpl + g;
    if (TAILS && g > full - (remain ? 0u : 1u)) {
      store_stream(dst, make_uint4(0u, 0u, 0u, 0u));
      continue;
    }
    const uint4 *const index = reinterpret_cast<const uint4 *>((uintptr_t) reinterpret_cast<const uint64_t *>(g_lds + a.sched_off + kSchedIndex)[j]);
    const uint32_t first_px = li * a.out_w + x0 + (q & 31u) * 6u;
    const uint4 w0 = load_stream(index + (first_px >> 1)), w1 = load_stream(index + (first_px >> 1) + 1), w2 = load_stream(index + (first_px >> 1) + 2);
    const uint32_t pk[12] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w, w2.x, w2.y, w2.z, w2.w};
    float yi[18];
#pragma unroll
    for (int jj = 0; jj < 6; ++jj) {
      yi[3 * jj] = __uint_as_float((pk[2 * jj] & 0xFFFFu) | 0x4B400000u);
      yi[3 * jj + 1] = __uint_as_float((pk[2 * jj] >> 16) | 0x4B400000u);
      yi[3 * jj + 2] = __uint_as_float((pk[2 * jj + 1] & 0xFFFFu) | 0x4B400000u);
    }
    if (TAILS && g == full) store_stream(dst, write_quad_idx_lds_tail(yi, wk, wlut, remain));
    else store_stream(dst, write_quad_idx_lds(yi, wk, wlut));
  }
  PH_CPHASE(5);
}

// the most slots (chunks) any workgroup of a `grid`-workgroup launch has (chan_share)
static uint32_t chan_max_slots(uint32_t out_w, uint32_t lines, uint32_t grid) {
  const uint32_t cpr = (out_w + kChanChunk - 1u) / kChanChunk, cpg = (uint32_t)(PH_CHAN_GROUP_ROWS / 2) * cpr;
  const uint32_t chunks = cpr * ((lines + 1u) / 2u);
  if ((grid & 7u) == 0) {
    const uint32_t groups = (chunks + cpg - 1u) / cpg, mine = (groups + 7u) / 8u, vstep = grid >> 3;
    return (mine * cpg + vstep - 1u) / vstep;
  }
  return (chunks + grid - 1u) / grid;
}
static uint32_t chan_grid(uint32_t out_w, uint32_t lines, uint32_t num_cus) {
  const uint32_t cpr = (out_w + kChanChunk - 1u) / kChanChunk, chunks = cpr * ((lines + 1u) / 2u);
  return chunks < num_cus ? chunks : num_cus;
}
hipError_t launch_chan_compose_batch(hipStream_t s, const ChanBatchArgs &a, uint32_t num_cus) {
  if (!a.lines || !a.jobs) return hipSuccess;
  if (a.jobs > (uint32_t)kMaxChanJobs || a.n_ops > (uint32_t)kMaxChanBatchOps) return hipErrorInvalidValue;
  const uint32_t lds = a.rd.bytes > a.wr.bytes ? a.rd.bytes : a.wr.bytes;
  const uint32_t cpr = (a.out_w + kChanChunk - 1u) / kChanChunk, cpg = (uint32_t)(PH_CHAN_GROUP_ROWS / 2) * cpr;
  const uint32_t grid = chan_grid(a.out_w, a.lines, num_cus);
  const uint32_t chunks = cpr * ((a.lines + 1u) / 2u);
  const uint32_t slots = chan_max_slots(a.out_w, a.lines, grid), steps = 3u * slots;
  ChanBatchArgs b = a;
  b.magic_cpr = cpr > 1 ? (uint32_t)(((1ull << 32) + cpr - 1) / cpr) : 0u;
  b.magic_cpg = (uint32_t)(((1ull << 32) + cpg - 1) / cpg);
  b.steps = steps;
  // job j's share starts rot[j][x] places further round the workgroups of XCD x (x: the XCD the job's rotation gives this workgroup's
  // bands to): the workgroups that have a chunk more than the others are then different ones for every job
  for (uint32_t j = 0; j < b.jobs; ++j)
    for (uint32_t x = 0; x < 8; ++x) {
      uint32_t rot;
      if ((grid & 7u) == 0) {
        const uint32_t groups = (chunks + cpg - 1u) / cpg, mine = (groups + 7u - x) / 8u, vstep = grid >> 3;
        rot = (j * ((mine * cpg) % vstep)) % vstep;
      } else {
        rot = (j * (chunks % grid)) % grid;
      }
      if (x & 1u) b.job_rot[j][x >> 1] |= rot << 16;
      else b.job_rot[j][x >> 1] = rot;
    }
  b.magic_spj = (uint32_t)(((1ull << 32) + steps - 1) / steps);
  b.magic_qpj = (uint32_t)(((1ull << 32) + 64u * slots - 1) / (64u * slots));
  b.sched_off = (lds + 15u) & ~15u;
  b.halo_off = b.sched_off + kSchedBytes;
  if (b.halo_off > 160u * 1024u) return hipErrorInvalidValue;
  // tap sharing as in the one-job launcher; each sharing op needs 36 bytes per wave step and job behind the table, ops that do not fit go without
  b.halo_steps = 0, b.n_share = 0, b.images_only = 1;
  for (uint32_t k = 0; k < b.n_ops; ++k) b.images_only &= b.op[k].src.kind == kChanRgba ? 1u : 0u;
  static const bool no_share = getenv("PH_CHAN_NO_SHARE") != nullptr;
  const uint32_t room = 160u * 1024u - b.halo_off;
  for (uint32_t k = 0; k < b.n_ops; ++k) {
    ChanSrc &src = b.op[k].src;
    b.op[k].action &= ~(kChanActShare | (7u << kChanActShareShift));
    if (no_share || src.kind != kChanV210 || !src.sampled || src.m[1] != 0.0f || src.m[3] != 0.0f) continue;
    const float sx = src.m[0] * (float)src.w / (float)a.out_w, sy = src.m[4] * (float)src.h / (float)(a.out_h) * (float)a.line_step;
    if (!(sx > 0.9999f && sx < 1.0001f && sy > 0.9999f && sy < 1.0001f)) continue;
    if (b.n_share == 8u || (b.n_share + 1u) * steps * 36u > room) break;
    b.op[k].action |= kChanActShare | (b.n_share << kChanActShareShift);
    const uint32_t j = b.op_job[k];
    b.share_op[b.n_share++] = k | j << 8 | b.job[j].first_line << 16;
  }
  if (b.n_share) b.halo_steps = steps;
  const uint32_t lds_total = b.halo_off + b.n_share * steps * 36u;
  auto go = [&](auto kernel) -> hipError_t {
    char name[48];
    snprintf(name, sizeof name, "chan_compose_batch<%u>x%u", a.planar ? 2u : a.tails ? 1u : 0u, a.jobs);  // (route trace: the instantiation - 0 whole blocks, 1 line tails, 2 planar sources - and the jobs sharing it)
    if (trace_launch(name)) return hipSuccess;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kMaxDynamicLds);
    if (e != hipSuccess) return e;
    kernel<<<grid, kLdsBlock, lds_total, s>>>(b);
    return hipGetLastError();
  };
  return a.planar ? go(chan_compose_batch_kernel<true, true>) : a.tails ? go(chan_compose_batch_kernel<true>) : go(chan_compose_batch_kernel<false>);
}

size_t chan_index_bytes(uint32_t out_w, uint32_t lines) { return (size_t)out_w * lines * 8u + 64u; }  // + a tail quad's reach past the last line

hipError_t launch_chan_compose_v210(hipStream_t s, const ChanArgs &a, uint32_t num_cus) {
  if (!a.lines) return hipSuccess;
  const uint32_t lds = a.rd.bytes > a.wr.bytes ? a.rd.bytes : a.wr.bytes;
  const uint32_t cpr = (a.out_w + kChanChunk - 1u) / kChanChunk, cpg = (uint32_t)(PH_CHAN_GROUP_ROWS / 2) * cpr;
  const uint32_t chunks = cpr * ((a.lines + 1u) / 2u);  // 192 pixels x 2 rows each (a row's last chunk may be short)
  const uint32_t want = chunks;                         // a workgroup per chunk at most: 6 wave steps
  ChanArgs b = a;
  // reciprocals for the kernel's uniform divisions: umulhi(v, ceil(2^32 / d)) == v / d for every v * d < 2^32 (chunk counts are far below)
  b.magic_cpr = cpr > 1 ? (uint32_t)(((1ull << 32) + cpr - 1) / cpr) : 0u;
  b.magic_cpg = (uint32_t)(((1ull << 32) + cpg - 1) / cpg);
  const uint32_t grid = want < num_cus ? want : num_cus;
  b.any_cm = 0, b.images_only = 1;
  for (int k = 0; k < b.n_ops; ++k) b.any_cm |= b.cm_op[k] ? 1u : 0u, b.images_only &= b.op[k].src.kind == kChanRgba ? 1u : 0u;
  // Tap sharing (ChanHalo): v210 sources shown at their own scale, unrotated and unmirrored - columns and rows advance by one texel
  // per output pixel (the kernel still confirms the pattern per wave step).  Each such op needs 36 bytes of LDS per wave step of a
  // workgroup behind the table; ops that do not fit any more go without.
  // Planar YCbCr clips share their taps in an instantiation of its own (mode 4): taken when the program has nothing but such clips and
  // f32 images, makes a v210 frame, and at least half of its ops are clips at their own scale (a full-frame clip; not config 2's one
  // background under three insets and a wipe: there the plain loop's better register allocation is worth more, 61.5 against 68.7 us)
  const bool lean_out = a.out_fmt == 0 || a.out_fmt == 2 || a.out_fmt == 5;  // v210 (SDI), yuv422p8 (an encoder), rgba8 (the screen): the reference's three consumers
  bool clips_only = a.planar == 2 && lean_out;
  uint32_t own_scale = 0;
  for (int k = 0; k < b.n_ops; ++k) {
    const ChanSrc &s = b.op[k].src;
    const bool clip = s.kind >= kChanP10 && s.kind <= kChanNv12;
    clips_only = clips_only && (clip || s.kind == kChanRgba);
    if (clip && s.sampled && s.m[1] == 0.0f && s.m[3] == 0.0f) {
      const float sx = s.m[0] * (float)s.w / (float)a.out_w, sy = s.m[4] * (float)s.h / (float)(a.out_h) * (float)a.line_step;
      if (sx > 0.9999f && sx < 1.0001f && sy > 0.9999f && sy < 1.0001f) ++own_scale;
    }
  }
  static const bool no_clips_kernel = getenv("PH_CHAN_NO_CLIPS_KERNEL") != nullptr;  // A/B runs (tools/chan_bench.py): mode 2 for everything planar
  if (no_clips_kernel) clips_only = false;
  const bool planar_share = clips_only && 2u * own_scale >= (uint32_t)b.n_ops;
  // v210 clips under packed-RGB graphics (and f32 images), a frame of one of the three consumers: mode 5, the graphics at their own scale share their taps
  bool graphics = a.planar == 2 && lean_out && !no_clips_kernel, any_rgb8 = false;
  for (int k = 0; k < b.n_ops; ++k) {
    const uint32_t kind = b.op[k].src.kind;
    graphics = graphics && (kind == kChanV210 || kind == kChanRgba || kind >= kChanRgba8);
    any_rgb8 = any_rgb8 || kind >= kChanRgba8;
  }
  graphics = graphics && any_rgb8;
  uint32_t lds_total = lds;
  b.halo_steps = 0, b.halo_off = (lds + 15u) & ~15u;
  static const bool no_share = getenv("PH_CHAN_NO_SHARE") != nullptr;  // A/B runs (tools/chan_bench.py)
  if (!no_share) {
    // the most slots any workgroup has (chan_share)
    uint32_t slots;
    if ((grid & 7u) == 0) {
      const uint32_t groups = (chunks + cpg - 1u) / cpg, mine = (groups + 7u) / 8u, vstep = grid >> 3;
      slots = (mine * cpg + vstep - 1u) / vstep;
    } else {
      slots = (chunks + grid - 1u) / grid;
    }
    const uint32_t steps = 3u * slots, room = 160u * 1024u > b.halo_off ? 160u * 1024u - b.halo_off : 0u;
    uint32_t n_share = 0;
    for (int k = 0; k < b.n_ops && steps; ++k) {
      const ChanSrc &s = b.op[k].src;
      const bool rgb8 = graphics && s.kind >= kChanRgba8;  // (two slots: its alpha has one of its own)
      if ((s.kind != kChanV210 && !rgb8 && !(planar_share && s.kind >= kChanP10 && s.kind <= kChanNv12)) || !s.sampled || s.m[1] != 0.0f || s.m[3] != 0.0f) continue;  // v210 clips; planar ones and graphics in their instantiations
      const float sx = s.m[0] * (float)s.w / (float)a.out_w, sy = s.m[4] * (float)s.h / (float)(a.out_h) * (float)a.line_step;
      if (!(sx > 0.9999f && sx < 1.0001f && sy > 0.9999f && sy < 1.0001f)) continue;
      const uint32_t slots_needed = rgb8 ? 2u : 1u;
      if (n_share + slots_needed > 8u || (n_share + slots_needed) * steps * 36u > room) break;
      b.op[k].action |= kChanActShare | (n_share << kChanActShareShift);
      n_share += slots_needed;
    }
    if (n_share) b.halo_steps = steps, lds_total = b.halo_off + n_share * steps * 36u;
  }
  // (route trace: the instantiation is part of the route - <phase-1 mode, output format>)
  auto go = [&](auto kernel, int mode, int out_fmt) -> hipError_t {
    char name[48];
    snprintf(name, sizeof name, "chan_compose_v210<%d,%d>", mode, out_fmt);
    if (trace_launch(name)) return hipSuccess;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kMaxDynamicLds);
    if (e != hipSuccess) return e;
    kernel<<<grid, kLdsBlock, lds_total, s>>>(b);
    return hipGetLastError();
  };
#define PH_CHAN_GO(M, O) go(chan_compose_v210_kernel<M, O>, M, O)
  // The writer's phase is independent of the readers': the lean instantiations of phase 1 (v210 / image programs on whole 48-pixel blocks;
  // planar clips; planar clips with shared taps) exist for the frames of the reference's three consumers - v210 (macadamConsumer.ts:165),
  // yuv422p8 (ffmpegConsumer.ts:144), rgba8 (screenConsumer.ts:131); the other formats and v210 lines with tails take the "everything" one
  auto lean = [&](auto out_tag) -> hipError_t {
    constexpr int O = decltype(out_tag)::value;
    if (a.planar == 2)
      return planar_share ? PH_CHAN_GO(4, O) : clips_only ? PH_CHAN_GO(3, O) : graphics ? PH_CHAN_GO(5, O) : PH_CHAN_GO(2, O);
    if (a.planar == 1) return O == 0 ? PH_CHAN_GO(1, 0) : PH_CHAN_GO(2, O);
    return PH_CHAN_GO(0, O);
  };
  switch (a.out_fmt) {
    case 0: return lean(std::integral_constant<int, 0>{});
    case 1: return PH_CHAN_GO(2, 1);
    case 2: return lean(std::integral_constant<int, 2>{});
    case 3: return PH_CHAN_GO(2, 3);
    case 4: return PH_CHAN_GO(2, 4);
    case 5: return lean(std::integral_constant<int, 5>{});
    case 6: return PH_CHAN_GO(2, 6);
  }
#undef PH_CHAN_GO
  return hipErrorInvalidValue;
}

}  // namespace ph
