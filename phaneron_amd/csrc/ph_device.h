// ph_device.h - device-side arithmetic shared by every kernel (gfx950 only).
//
// Parity rules (SURVEY.md 7 "hard parts"): the reference's OpenCL C runs on AMD's device
// library, where dot(float4) = fma(a3,b3, fma(a2,b2, fma(a1,b1, a0*b0))), dot(float3) the
// 3-term analogue, and convert_ushort_sat_rte = rint -> max 0 -> min 65535 -> fptoui.  A
// one-bit difference before a LUT index picks a different LUT entry, so these chains are
// spelled out with explicit fused ops and contraction is off for everything else.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#pragma clang fp contract(off)

namespace ph {

__device__ __forceinline__ float fma_rn(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

__device__ __forceinline__ float dot4(float a0, float a1, float a2, float a3, const float4 m) {
  return fma_rn(a3, m.w, fma_rn(a2, m.z, fma_rn(a1, m.y, a0 * m.x)));
}
__device__ __forceinline__ float dot3(float a0, float a1, float a2, float m0, float m1, float m2) {
  return fma_rn(a2, m2, fma_rn(a1, m1, a0 * m0));
}

// convert_ushort_sat_rte(x) = clamp(rint(x), 0, 65535) = rint(clamp(x, 0, 65535)) (integer bounds,
// rint monotone).  Evaluated as v_med3_f32 (NaN -> 0), then ONE f32 add of 1.5 * 2^23: the sum has
// ulp 1, so the add itself rounds to nearest-even and the integer sits in the low mantissa bits
// (v_and).  One op and two slow-class ops (v_rndne, v_cvt) less than rint / clamp / convert -
// tools/opbench2.hip has the per-instruction costs.
constexpr float kRoundMagic = 12582912.0f;  // 1.5 * 2^23 = 0x4B400000
__device__ __forceinline__ uint32_t sat_u16_rte(float x) {
  x = __builtin_fminf(__builtin_fmaxf(x, 0.0f), 65535.0f);
  return __float_as_uint(x + kRoundMagic) & 0xFFFFu;
}
// convert_ushort_sat / _rtz: truncating
__device__ __forceinline__ uint32_t sat_u16_trunc(float x) {
  x = __builtin_fmaxf(x, 0.0f);
  x = __builtin_fminf(x, 65535.0f);
  return (uint32_t)x;
}

// convert_uchar_sat_rte (rgba8.ts / bgra8.ts writers): rint -> max 0 -> min 255 -> fptoui, as the device library
__device__ __forceinline__ uint32_t sat_u8_rte(float x) {
  return (uint32_t)__builtin_fminf(__builtin_fmaxf(__builtin_rintf(x), 0.f), 255.f);
}

// Reader-side constants (v210.ts:42-52): 3x4 YCbCr->RGB matrix, 3x3 gamut matrix.
struct ReadK {
  float4 r, g, b;
  float gm[9];
};
__device__ __forceinline__ ReadK load_read_k(const float *__restrict__ cm, const float *__restrict__ gm) {
  ReadK k;
  k.r = make_float4(cm[0], cm[1], cm[2], cm[3]);
  k.g = make_float4(cm[4], cm[5], cm[6], cm[7]);
  k.b = make_float4(cm[8], cm[9], cm[10], cm[11]);
#pragma unroll
  for (int i = 0; i < 9; ++i) k.gm[i] = gm[i];
  return k;
}
// Writer-side constants (v210.ts:138-140): 3x4 RGB->YCbCr matrix.
struct WriteK {
  float4 y, u, v;
};
__device__ __forceinline__ WriteK load_write_k(const float *__restrict__ cm) {
  WriteK k;
  k.y = make_float4(cm[0], cm[1], cm[2], cm[3]);
  k.u = make_float4(cm[4], cm[5], cm[6], cm[7]);
  k.v = make_float4(cm[8], cm[9], cm[10], cm[11]);
  return k;
}

// One pixel of the v210 read kernel (v210.ts:65-78; tail :96-109 passes last = 0).
__device__ __forceinline__ float4 read_px(float y, float cb, float cr, float last, const ReadK &k,
                                          const float *__restrict__ lut) {
  const float r = lut[sat_u16_rte(dot4(y, cb, cr, last, k.r) * 65535.0f)];
  const float g = lut[sat_u16_rte(dot4(y, cb, cr, last, k.g) * 65535.0f)];
  const float b = lut[sat_u16_rte(dot4(y, cb, cr, last, k.b) * 65535.0f)];
  return make_float4(dot3(r, g, b, k.gm[0], k.gm[1], k.gm[2]), dot3(r, g, b, k.gm[3], k.gm[4], k.gm[5]),
                     dot3(r, g, b, k.gm[6], k.gm[7], k.gm[8]), 1.0f);
}

// The six (Y,Cb,Cr) triples of a v210 word quad (v210.ts:58-63).
struct Yuv6 {
  float y[6], cb[3], cr[3];
};
__device__ __forceinline__ Yuv6 unpack_quad(const uint4 w) {
  Yuv6 q;
  q.cb[0] = (float)(w.x & 0x3ff), q.y[0] = (float)((w.x >> 10) & 0x3ff), q.cr[0] = (float)((w.x >> 20) & 0x3ff);
  q.y[1] = (float)(w.y & 0x3ff), q.cb[1] = (float)((w.y >> 10) & 0x3ff), q.y[2] = (float)((w.y >> 20) & 0x3ff);
  q.cr[1] = (float)(w.z & 0x3ff), q.y[3] = (float)((w.z >> 10) & 0x3ff), q.cb[2] = (float)((w.z >> 20) & 0x3ff);
  q.y[4] = (float)(w.w & 0x3ff), q.cr[2] = (float)((w.w >> 10) & 0x3ff), q.y[5] = (float)((w.w >> 20) & 0x3ff);
  return q;
}

// One pixel of the v210 write kernel (v210.ts:145-156): linear RGB -> gamma LUT -> code values.
struct Yuv1 {
  uint32_t y, u, v;
};
__device__ __forceinline__ Yuv1 write_px(float r, float g, float b, const WriteK &k,
                                         const float *__restrict__ lut) {
  const float gr = lut[sat_u16_rte(r * 65535.0f)];
  const float gg = lut[sat_u16_rte(g * 65535.0f)];
  const float gb = lut[sat_u16_rte(b * 65535.0f)];
  Yuv1 o;
  o.y = sat_u16_rte(dot4(gr, gg, gb, 1.0f, k.y));
  o.u = sat_u16_rte(dot4(gr, gg, gb, 1.0f, k.u));
  o.v = sat_u16_rte(dot4(gr, gg, gb, 1.0f, k.v));
  return o;
}
// luma only: odd pixels contribute no chroma to the packed words (v210.ts:159-162)
__device__ __forceinline__ uint32_t write_px_luma(float r, float g, float b, const WriteK &k,
                                                  const float *__restrict__ lut) {
  const float gr = lut[sat_u16_rte(r * 65535.0f)];
  const float gg = lut[sat_u16_rte(g * 65535.0f)];
  const float gb = lut[sat_u16_rte(b * 65535.0f)];
  return sat_u16_rte(dot4(gr, gg, gb, 1.0f, k.y));
}
// tail variant (v210.ts:173-184): truncating LUT index, round-half-away outputs
__device__ __forceinline__ Yuv1 write_px_tail(float r, float g, float b, const WriteK &k,
                                              const float *__restrict__ lut) {
  const float gr = lut[sat_u16_trunc(r * 65535.0f)];
  const float gg = lut[sat_u16_trunc(g * 65535.0f)];
  const float gb = lut[sat_u16_trunc(b * 65535.0f)];
  Yuv1 o;
  o.y = sat_u16_trunc(__builtin_roundf(dot4(gr, gg, gb, 1.0f, k.y)));
  o.u = sat_u16_trunc(__builtin_roundf(dot4(gr, gg, gb, 1.0f, k.u)));
  o.v = sat_u16_trunc(__builtin_roundf(dot4(gr, gg, gb, 1.0f, k.v)));
  return o;
}

__device__ __forceinline__ uint4 pack_quad(const uint32_t y[6], const uint32_t u[3], const uint32_t v[3]) {
  uint4 w;  // v210.ts:159-162 (16-bit fields shifted in 32-bit registers, as the reference)
  w.x = v[0] << 20 | y[0] << 10 | u[0];
  w.y = y[2] << 20 | u[1] << 10 | y[1];
  w.z = u[2] << 20 | y[3] << 10 | v[1];
  w.w = y[5] << 20 | v[2] << 10 | y[4];
  return w;
}

// Streaming stores.  Every kernel of the path writes each output byte exactly once and never reads it
// back, so outputs are stored non-temporally: a plain store makes the L2 / Infinity Cache allocate the
// line (and fetch it for ownership), which costs the write-heavy kernels a third of their bandwidth
// (v210 read 2160p: 45 us -> 31 us).  PH_NT_STORE=0 builds the plain stores for comparison.
#ifndef PH_NT_STORE
#define PH_NT_STORE 1
#endif
typedef float ph_f4v __attribute__((ext_vector_type(4)));
typedef uint32_t ph_u4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_stream(float4 *p, const float4 v) {
#if PH_NT_STORE
  __builtin_nontemporal_store(ph_f4v{v.x, v.y, v.z, v.w}, reinterpret_cast<ph_f4v *>(p));
#else
  *p = v;
#endif
}
// f32 IMAGE outputs.  An image is an intermediate: the next operator of the channel reads it back within a few
// kernels, and for the frame sizes of the path (33 MB at 1080p, 133 MB at 2160p) a stream-past-the-caches store sends
// that consumer to HBM.  Measured on whole chains (tools/config_bench.py, all routes): plain stores make config 2
// 10 % and the reference-shaped config 3 batch 8 % faster, although each producer alone is slower (v210 read 2160p
// 31 -> 45 us).  So images are stored plainly unless the context asks for streaming (ph_ctx_set_option
// "stream_images", for a caller that knows nothing on the device reads the image soon); wire-format outputs, which
// leave the device, always stream.  `nt` is a kernel argument (uniform): one scalar branch per store.
__device__ __forceinline__ void store_image(float4 *p, const float4 v, uint32_t nt) {
  if (nt)
    __builtin_nontemporal_store(ph_f4v{v.x, v.y, v.z, v.w}, reinterpret_cast<ph_f4v *>(p));
  else
    *p = v;
}
// Streaming loads for kernels that pull three or more full frames per output frame (combine_N with
// N >= 3, transition_wipe): measured +4 % there, -4 % on the two-input kernels, so those stay plain.
__device__ __forceinline__ float4 load_stream(const float4 *p) {
  const ph_f4v v = __builtin_nontemporal_load(reinterpret_cast<const ph_f4v *>(p));
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ uint4 load_stream(const uint4 *p) {  // fused kernel inputs: read once, 16 B per lane, +1 %
  const ph_u4v v = __builtin_nontemporal_load(reinterpret_cast<const ph_u4v *>(p));
  return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void store_stream(uint4 *p, const uint4 v) {
#if PH_NT_STORE
  __builtin_nontemporal_store(ph_u4v{v.x, v.y, v.z, v.w}, reinterpret_cast<ph_u4v *>(p));
#else
  *p = v;
#endif
}

// ------------------------------------------------------------------------------------------
// bilinear sampler (OpenCL 1.2 s8.2: NORMALIZED | CLAMP (border 0) | LINEAR).  The f32
// evaluation order is fixed: weights first, then ((w00*t00 + w10*t10) + w01*t01) + w11*t11,
// plain mul/add, no fma (DESIGN.md "sampler").
// Branch-free: the four taps are loaded unconditionally from coordinates forced into the image and
// zeroed afterwards where the real coordinate was outside (border colour 0).  A per-tap `if` makes
// four dependent branch-and-load round trips per sample; this way the four loads are in flight
// together.  (float -> int conversion saturates and NaN converts to 0, so wild coordinates stay safe;
// the index arithmetic is unsigned 32-bit: images have fewer than 2^32 pixels.)
// ------------------------------------------------------------------------------------------
// `direct` (uniform): skip the filter and return the texel (dx, dy) itself - through the same four
// loads and a final select, so that a compositor can run sampled and 1:1 layers without a branch.
template <bool MAY_BE_DIRECT = false>
__device__ __forceinline__ float4 sample_linear(const float4 *__restrict__ img, int w, int h, float s, float t,
                                                bool direct = false, uint32_t dx = 0, uint32_t dy = 0) {
  const float u = s * (float)w, v = t * (float)h;
  const float fu = u - 0.5f, fv = v - 0.5f;
  const float flu = __builtin_floorf(fu), flv = __builtin_floorf(fv);
  uint32_t i0 = (uint32_t)(int)flu, j0 = (uint32_t)(int)flv;
  if (MAY_BE_DIRECT) i0 = direct ? dx : i0, j0 = direct ? dy : j0;
  const uint32_t i1 = i0 + 1u, j1 = j0 + 1u;
  const float a = fu - flu, b = fv - flv;
  const float oma = 1.0f - a, omb = 1.0f - b;
  const float w00 = oma * omb, w10 = a * omb, w01 = oma * b, w11 = a * b;
  const bool xa = i0 < (uint32_t)w, xb = i1 < (uint32_t)w, ya = j0 < (uint32_t)h, yb = j1 < (uint32_t)h;
  const uint32_t r0 = (ya ? j0 : 0u) * (uint32_t)w, r1 = (yb ? j1 : 0u) * (uint32_t)w;
  const uint32_t c0 = xa ? i0 : 0u, c1 = xb ? i1 : 0u;
  float4 t00 = img[r0 + c0], t10 = img[r0 + c1], t01 = img[r1 + c0], t11 = img[r1 + c1];
  // component-wise selects (v_cndmask): selecting between whole float4 objects goes through scratch
  const bool k00 = xa && ya, k10 = xb && ya, k01 = xa && yb, k11 = xb && yb;
#define PH_KEEP(T, K) T.x = (K) ? T.x : 0.f, T.y = (K) ? T.y : 0.f, T.z = (K) ? T.z : 0.f, T.w = (K) ? T.w : 0.f
  PH_KEEP(t00, k00), PH_KEEP(t10, k10), PH_KEEP(t01, k01), PH_KEEP(t11, k11);
#undef PH_KEEP
  float4 r;
  r.x = ((w00 * t00.x + w10 * t10.x) + w01 * t01.x) + w11 * t11.x;
  r.y = ((w00 * t00.y + w10 * t10.y) + w01 * t01.y) + w11 * t11.y;
  r.z = ((w00 * t00.z + w10 * t10.z) + w01 * t01.z) + w11 * t11.z;
  r.w = ((w00 * t00.w + w10 * t10.w) + w01 * t01.w) + w11 * t11.w;
  if (MAY_BE_DIRECT) r.x = direct ? t00.x : r.x, r.y = direct ? t00.y : r.y, r.z = direct ? t00.z : r.z, r.w = direct ? t00.w : r.w;
  return r;
}

}  // namespace ph
