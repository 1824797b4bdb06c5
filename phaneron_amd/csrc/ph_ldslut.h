// ph_ldslut.h - device side of the LDS-resident gamma LUT (ph_lut.h): table load, lookup and the
// per-kernel constants.  Shared by ph_kernels_lds.hip (v210 / fused kernels) and ph_kernels_fmt.hip
// (the other pack formats).
#pragma once
#include "ph_device.h"
#include "ph_lut.h"

#pragma clang fp contract(off)

namespace ph {

constexpr int kLdsBlock = 1024;

// PH_ABLATE builds timing experiments of the fused kernel (WRONG results, never shipped): bit 0 = no
// LDS reads, bit 1 = no table loads, bit 2 = anchor read only, bit 3 = both reads but conflict-free.
// DESIGN.md section 4 quotes what they measured.
#ifndef PH_ABLATE
#define PH_ABLATE 0
#endif

extern __shared__ __attribute__((aligned(16))) unsigned char g_lds[];

// all lanes of the workgroup copy the table blob global -> LDS (16 bytes per lane per step)
// WAIT = false: the caller waits (s_waitcnt vmcnt(0)) and synchronises itself, having done something else meanwhile
template <int BS = kLdsBlock, bool WAIT = true>
__device__ __forceinline__ void lds_lut_load(const LutView &v) {
#if PH_ABLATE & 2  // timing experiment only: no table loads
  return;
#endif
  const uint32_t n = (v.bytes - v.hole) / 16;
  unsigned char *const dst = g_lds + v.hole;
  // LDS-DMA: each wave instruction moves 1 KiB global -> LDS (wave-uniform LDS base + lane*16)
  // without touching VGPRs; all of a wave's pieces are in flight before the single wait.
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const uint4 *src = reinterpret_cast<const uint4 *>(v.blob);
  for (uint32_t base = wave * 64; base < n; base += BS) {
    const uint32_t i = base + lane;
    if (i < n)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + i),
                                       (__attribute__((address_space(3))) void *)(dst + 16 * base), 16, 0, 0);
  }
  if (WAIT) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// table[clamp(rint(x), 0, 65535)] for x = (gamma- or linear-domain value) * 65535, see ph_lut.h:
// 4 float ops, 3 integer ops - every one of them of the 2-cycle class - and 2 LDS reads.
//   * round FIRST, then + bias (exact on integers).  Adding the bias before rounding would round
//     twice: x + bias has a coarser ulp than x just above a power of two and can manufacture a tie;
//   * rounding is the magic-number add of ph_device.h: y = x + 1.5*2^23 is M + idx exactly;
//   * the float (idx + bias) * a_scale = fma(y, a_scale, -(M - bias) * a_scale) (exact: a power-of-two multiple of a
//     24-bit integer) carries the logarithmic block number in its exponent / top mantissa bits, and a_scale is chosen so
//     that (bits >> (shift - 2)) & ~3 IS the anchor's byte address in the LDS: no base to add, no left shift
//     (v_lshl_add_u32 issues at half the rate of v_lshrrev / v_and on gfx950).
//     This needs the table at a fixed place - the blob is loaded at LutView::hole = 4 * 2^m, g_lds at LDS address 0
//     (checked once per kernel; these kernels declare no other shared memory);
//   * the delta address is produced by an fma whose result is a DENORMAL: (2*(i+bias) + base)
//     * 2^-149 has exactly that integer as its bit pattern, so no int multiply/add is needed
//     (f32 denormals are enabled in HIP kernels and cost nothing extra on gfx950); it comes straight from y
//     (the -M is folded into the fma's addend);
//   * both addresses are ABSOLUTE LDS addresses dereferenced as address-space-3 pointers made from integers: going
//     through `g_lds + offset` costs one v_add_u32 per read, because the symbol's address is only known at link time.
struct LutK {
  float a_scale, a_base;            // (idx + bias) * a_scale = fma(y, a_scale, a_base)
  float delta_scale, delta_base;    // delta_base already holds -M*scale, +2*bias and the LDS base
  uint32_t a_shift;                 // shift - 2
};
typedef const __attribute__((address_space(3))) uint32_t *lds_u32_ptr;
typedef const __attribute__((address_space(3))) uint16_t *lds_u16_ptr;
__device__ __forceinline__ LutK make_lut_k(const LutView &v) {
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)g_lds;
  // (lds0 is 0: these kernels declare no other shared memory, and the anchor address has no base to add - ph_api.cpp checks it once per context)
  // d_addr = 2*(idx + bias) + base + lds0 = bits(fma(M + idx, 2^-148, B)) with
  // B = (base + lds0 + 2*bias) * 2^-149 - M * 2^-148.  Every term is a multiple of 2^-148 below 2^24
  // of them (base, lds0 are even), so each float operation here is exact.
  const float small = (v.delta_base + __uint_as_float(lds0)) + v.bias * v.delta_scale;  // denormal sums
  const float b = small - kRoundMagic * v.delta_scale;
  return LutK{v.a_scale, -((kRoundMagic - v.bias) * v.a_scale), v.delta_scale, b, v.shift - 2u};
}
__device__ __forceinline__ uint32_t lds_lut_anchor_addr(const LutK &k, float y) {
  return (__float_as_uint(fma_rn(y, k.a_scale, k.a_base)) >> k.a_shift) & ~3u;
}
// y = M + idx (idx already clamped and rounded by the add that produced y)
__device__ __forceinline__ float lds_lut_fetch(const LutK &k, float y) {
  const uint32_t a_addr = lds_lut_anchor_addr(k, y);
  const uint32_t d_addr = __float_as_uint(fma_rn(y, k.delta_scale, k.delta_base));
#if PH_ABLATE & 8  // timing experiment only (wrong results): both reads issued, but conflict-free
  const uint32_t a = *(lds_u32_ptr)(a_addr & 4u);
  const uint32_t d = *(lds_u16_ptr)(d_addr & 2u);
#elif PH_ABLATE & 1  // timing experiment only (wrong results): no LDS reads
  const uint32_t a = a_addr, d = d_addr;
#elif PH_ABLATE & 4  // timing experiment only: anchor read only
  const uint32_t a = *(lds_u32_ptr)a_addr, d = d_addr;
#else
  const uint32_t a = *(lds_u32_ptr)a_addr;
  const uint32_t d = *(lds_u16_ptr)d_addr;
#endif
#if PH_ABLATE & 16
  // timing experiment only (results unchanged): the five instructions a COMPACT table form would add to every lookup - anchors
  // interpolated between neighbouring blocks plus a one-byte residual, small enough for reader and writer table to be resident
  // together: v_and_or_b32 (the position inside the block), v_fma_f32 (its scale), v_sub_f32 (the anchors'
  // difference), v_fma_f32 (the prediction), and the residual's v_add_u32 on top of the one that is there.  DESIGN.md 5 has what it cost.
  {
    uint32_t t;
    asm volatile("v_and_or_b32 %0, %1, %2, %3\n\tv_fma_f32 %0, %0, %0, %0\n\tv_sub_f32 %0, %0, %1\n\tv_fma_f32 %0, %0, %1, %0\n\tv_add_u32 %0, %0, %2"
                 : "=&v"(t)
                 : "v"(a), "v"(d), "v"(a_addr));
  }
#endif
  return __uint_as_float(a + d);
}
// The same lookup in two halves, for software-pipelined callers: `issue` computes both addresses and starts the
// two LDS reads, `finish` adds anchor and delta.  One wave alone can start a VALU instruction only every ~5
// cycles and an LDS read takes 64+ cycles to come back (128+ with bank conflicts), so a caller that consumes a
// lookup right after issuing it leaves its SIMD to the other three waves - and all four stall the same way.
// Issuing pixel j+1's reads BEFORE consuming pixel j's puts a whole pixel of arithmetic between every read and
// its use (tools/opbench3.hip).
struct LutPending {
  uint32_t a, d;
};
__device__ __forceinline__ LutPending lds_lut_issue(const LutK &k, float y) {
  const uint32_t a_addr = lds_lut_anchor_addr(k, y);
  const uint32_t d_addr = __float_as_uint(fma_rn(y, k.delta_scale, k.delta_base));
#if PH_ABLATE & 8
  return LutPending{*(lds_u32_ptr)(a_addr & 4u), *(lds_u16_ptr)(d_addr & 2u)};
#elif PH_ABLATE & 1
  return LutPending{a_addr, d_addr};
#else
  return LutPending{*(lds_u32_ptr)a_addr, *(lds_u16_ptr)d_addr};
#endif
}
__device__ __forceinline__ float lds_lut_finish(const LutPending &p) {
#if PH_ABLATE & 16  // timing experiment only: the compact form's five more instructions per lookup (see lds_lut_fetch)
  {
    uint32_t t;
    asm volatile("v_and_or_b32 %0, %1, %2, %1\n\tv_fma_f32 %0, %0, %0, %0\n\tv_sub_f32 %0, %0, %1\n\tv_fma_f32 %0, %0, %1, %0\n\tv_add_u32 %0, %0, %2"
                 : "=&v"(t)
                 : "v"(p.a), "v"(p.d));
  }
#endif
  return __uint_as_float(p.a + p.d);
}

// The rounded, clamped index of a unit-range value as the float M + idx (idx = its low 16 bits).
__device__ __forceinline__ float lds_lut_index_unit(float t) {
  t = __builtin_fminf(__builtin_fmaxf(t, 0.0f), 1.0f);
  return t * 65535.0f + kRoundMagic;
}
// The same with the TAIL's index (v210.ts:176-178 convert_ushort_sat_rtz): truncated instead of rounded where `trunc` is set
// (per lane).  The clamped product is not negative, so floor is the truncation, and M + an integer is exact.
__device__ __forceinline__ float lds_lut_index_unit_tail(float t, bool trunc) {
  t = __builtin_fminf(__builtin_fmaxf(t, 0.0f), 1.0f);
  const float x = t * 65535.0f;
  return (trunc ? __builtin_floorf(x) : x) + kRoundMagic;
}
// table[sat_rte(x)], x in table-index units
__device__ __forceinline__ float lds_lut_at(const LutK &k, float x) {
  x = __builtin_fminf(__builtin_fmaxf(x, 0.0f), 65535.0f);  // v_med3_f32; NaN -> 0 like the reference
  return lds_lut_fetch(k, x + kRoundMagic);
}
// table[sat_rte(t * 65535)] for a unit-range t: clamping t to [0,1] BEFORE the multiply gives the same
// index (the product and its rounding are monotone, 0 and 1 map to the bounds) and is free - the
// clamp becomes the output modifier of whatever instruction produced t.
__device__ __forceinline__ float lds_lut_at_unit(const LutK &k, float t) {
  return lds_lut_fetch(k, lds_lut_index_unit(t));
}

// One packed v210 quad from eighteen ready-made writer-table indices (floats M + idx): gamma table, RGB->YCbCr
// matrix, rounding, packing (v210.ts:145-162).  Phase 2 of the two-phase kernels (fused channel, field pipeline).
__device__ __forceinline__ uint4 write_quad_idx_lds(const float (&yi)[18], const WriteK &wk, const LutK &lut) {
  uint32_t y[6], u[3], v[3];
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const float gr = lds_lut_fetch(lut, yi[3 * j]), gg = lds_lut_fetch(lut, yi[3 * j + 1]);
    const float gb = lds_lut_fetch(lut, yi[3 * j + 2]);
    y[j] = sat_u16_rte(dot4(gr, gg, gb, 1.0f, wk.y));
    if ((j & 1) == 0) {
      u[j >> 1] = sat_u16_rte(dot4(gr, gg, gb, 1.0f, wk.u));
      v[j >> 1] = sat_u16_rte(dot4(gr, gg, gb, 1.0f, wk.v));
    }
  }
  return pack_quad(y, u, v);
}

// The tail quad of a line whose width is not a multiple of 6 (v210.ts:169-194): `remain` = 2 or 4 pixels from indices that were
// TRUNCATED (lds_lut_index_unit_tail), code values rounded half away from zero (round(), then the truncating convert_ushort_sat),
// the words the reference does not set left 0.  Chroma comes from the even pixels as everywhere.
__device__ __forceinline__ uint4 write_quad_idx_lds_tail(const float (&yi)[18], const WriteK &wk, const LutK &lut, uint32_t remain) {
  uint32_t y[4], u[2], v[2];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float gr = lds_lut_fetch(lut, yi[3 * j]), gg = lds_lut_fetch(lut, yi[3 * j + 1]);
    const float gb = lds_lut_fetch(lut, yi[3 * j + 2]);
    y[j] = sat_u16_trunc(__builtin_roundf(dot4(gr, gg, gb, 1.0f, wk.y)));
    if ((j & 1) == 0) {
      u[j >> 1] = sat_u16_trunc(__builtin_roundf(dot4(gr, gg, gb, 1.0f, wk.u)));
      v[j >> 1] = sat_u16_trunc(__builtin_roundf(dot4(gr, gg, gb, 1.0f, wk.v)));
    }
  }
  uint4 w = make_uint4(v[0] << 20 | y[0] << 10 | u[0], 0u, 0u, 0u);
  if (remain == 2u) {
    w.y = y[1];
  } else if (remain == 4u) {
    w.y = y[2] << 20 | u[1] << 10 | y[1];
    w.z = y[3] << 10 | v[1];
  }
  return w;
}

// A gamma LUT as the kernels see it: either the compressed table in LDS or the plain f32 table
// in global memory (tables that do not compress, or the "lds_lut" option switched off).
struct LutInLds {
  LutK k;
  __device__ __forceinline__ float at(float x) const { return lds_lut_at(k, x); }
  __device__ __forceinline__ float at_unit(float t) const { return lds_lut_at_unit(k, t); }
};
struct LutInGlobal {
  const float *__restrict__ t;
  __device__ __forceinline__ float at(float x) const { return t[sat_u16_rte(x)]; }
  __device__ __forceinline__ float at_unit(float u) const { return t[sat_u16_rte(u * 65535.0f)]; }
};

// Every YCbCr->RGB matrix colourMaths.ts:276-332 can produce has the same shape: one luma gain in all
// three rows, no Cb term in R, no Cr term in B (SURVEY a4 goldens: y2r709 = 3a95a025 00000000 ... /
// 3a95a025 b95b.. ba08.. / 3a95a025 3b07.. 00000000 ...).  With that shape (checked on the device, bit
// for bit) Y*m0 is computed once per pixel and the two fma by zero are skipped: fma(c, +-0, p) == p
// for the finite, non-negative code value c - except that it can turn p = -0 into +0, which the
// clamp / round that follows maps to the same table index.  8 operations per pixel instead of 12.
__device__ __forceinline__ bool ycbcr_matrix_is_standard(const ReadK &k) {
  return k.r.y == 0.0f && k.b.z == 0.0f && k.r.x == k.g.x && k.g.x == k.b.x;
}
// `last`: the fourth component of the reference's (Y, Cb, Cr, 1) vector.  It is 1 everywhere except in the tail of a line whose
// width is not a multiple of 6, where the reference's reader builds its vectors with a 0 there (v210.ts:88-93) and so drops the
// matrix's offset column; callers that serve such widths pass 0.0f / 1.0f per lane, everybody else the constant.
template <bool STD = false>
__device__ __forceinline__ float4 read_px_lds(float y, float cb, float cr, const ReadK &k, const LutK &lut, float last = 1.0f) {
  float tr, tg, tb;
  if (STD) {
    const float ym = y * k.r.x;
    tr = fma_rn(last, k.r.w, fma_rn(cr, k.r.z, ym));
    tg = fma_rn(last, k.g.w, fma_rn(cr, k.g.z, fma_rn(cb, k.g.y, ym)));
    tb = fma_rn(last, k.b.w, fma_rn(cb, k.b.y, ym));
  } else {
    tr = dot4(y, cb, cr, last, k.r), tg = dot4(y, cb, cr, last, k.g), tb = dot4(y, cb, cr, last, k.b);
  }
  const float r = lds_lut_at_unit(lut, tr);
  const float g = lds_lut_at_unit(lut, tg);
  const float b = lds_lut_at_unit(lut, tb);
  return make_float4(dot3(r, g, b, k.gm[0], k.gm[1], k.gm[2]), dot3(r, g, b, k.gm[3], k.gm[4], k.gm[5]),
                     dot3(r, g, b, k.gm[6], k.gm[7], k.gm[8]), 1.0f);
}

// ---- the same two pixel stages split at the LDS reads (ph_ldslut.h lds_lut_issue / lds_lut_finish), for the
// software-pipelined fused kernel: `issue` does everything up to and including the start of the six reads,
// `finish` everything from their results on.
struct PxPending {
  LutPending r, g, b;
};
template <bool STD>
__device__ __forceinline__ PxPending read_px_issue(float y, float cb, float cr, const ReadK &k, const LutK &lut, float last = 1.0f) {
  float tr, tg, tb;
  if (STD) {
    const float ym = y * k.r.x;
    tr = fma_rn(last, k.r.w, fma_rn(cr, k.r.z, ym));
    tg = fma_rn(last, k.g.w, fma_rn(cr, k.g.z, fma_rn(cb, k.g.y, ym)));
    tb = fma_rn(last, k.b.w, fma_rn(cb, k.b.y, ym));
  } else {
    tr = dot4(y, cb, cr, last, k.r), tg = dot4(y, cb, cr, last, k.g), tb = dot4(y, cb, cr, last, k.b);
  }
  PxPending p;
  p.r = lds_lut_issue(lut, lds_lut_index_unit(tr));
  p.g = lds_lut_issue(lut, lds_lut_index_unit(tg));
  p.b = lds_lut_issue(lut, lds_lut_index_unit(tb));
  return p;
}
__device__ __forceinline__ float4 read_px_finish(const PxPending &p, const ReadK &k) {
  const float r = lds_lut_finish(p.r), g = lds_lut_finish(p.g), b = lds_lut_finish(p.b);
  return make_float4(dot3(r, g, b, k.gm[0], k.gm[1], k.gm[2]), dot3(r, g, b, k.gm[3], k.gm[4], k.gm[5]),
                     dot3(r, g, b, k.gm[6], k.gm[7], k.gm[8]), 1.0f);
}
// writer side, from ready-made indices (floats M + idx)
__device__ __forceinline__ PxPending write_px_issue(float yr, float yg, float yb, const LutK &lut) {
  PxPending p;
  p.r = lds_lut_issue(lut, yr), p.g = lds_lut_issue(lut, yg), p.b = lds_lut_issue(lut, yb);
  return p;
}

__device__ __forceinline__ Yuv1 write_px_lds(float r, float g, float b, const WriteK &k, const LutK &lut) {
  const float gr = lds_lut_at_unit(lut, r);
  const float gg = lds_lut_at_unit(lut, g);
  const float gb = lds_lut_at_unit(lut, b);
  Yuv1 o;
  o.y = sat_u16_rte(dot4(gr, gg, gb, 1.0f, k.y));
  o.u = sat_u16_rte(dot4(gr, gg, gb, 1.0f, k.u));
  o.v = sat_u16_rte(dot4(gr, gg, gb, 1.0f, k.v));
  return o;
}
__device__ __forceinline__ uint32_t write_px_luma_lds(float r, float g, float b, const WriteK &k, const LutK &lut) {
  const float gr = lds_lut_at_unit(lut, r);
  const float gg = lds_lut_at_unit(lut, g);
  const float gb = lds_lut_at_unit(lut, b);
  return sat_u16_rte(dot4(gr, gg, gb, 1.0f, k.y));
}

}  // namespace ph
