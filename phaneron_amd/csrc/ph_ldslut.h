// ph_ldslut.h - device side of the LDS-resident gamma LUT (ph_lut.h): table load, lookup and the
// per-kernel constants.  Shared by ph_kernels_lds.hip (v210 / fused kernels) and ph_kernels_fmt.hip
// (the other pack formats).
#pragma once
#include "ph_device.h"
#include "ph_lut.h"

#pragma clang fp contract(off)

namespace ph {

constexpr int kLdsBlock = 1024;

// A/B knobs (profiles/): PH_SCHED_LEVEL 0 = no scheduling fences, 1 = one per layer / quad,
// 2 = one per pixel pair.  PH_TABLE_DMA 1 = table swaps by global_load_lds (LDS-DMA, no VGPRs).
#ifndef PH_SCHED_LEVEL
#define PH_SCHED_LEVEL 0
#endif
#ifndef PH_TABLE_DMA
#define PH_TABLE_DMA 1
#endif
#define PH_FENCE(level)                                         \
  do {                                                          \
    if (PH_SCHED_LEVEL >= (level)) __builtin_amdgcn_sched_barrier(0); \
  } while (0)

extern __shared__ __attribute__((aligned(16))) unsigned char g_lds[];

// all lanes of the workgroup copy the table blob global -> LDS (16 bytes per lane per step)
template <int BS = kLdsBlock>
__device__ __forceinline__ void lds_lut_load(const LutView &v) {
  const uint32_t n = v.bytes / 16;
#if PH_TABLE_DMA
  // LDS-DMA: each wave instruction moves 1 KiB global -> LDS (wave-uniform LDS base + lane*16)
  // without touching VGPRs; all of a wave's pieces are in flight before the single wait.
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const uint4 *src = reinterpret_cast<const uint4 *>(v.blob);
  for (uint32_t base = wave * 64; base < n; base += BS) {
    const uint32_t i = base + lane;
    if (i < n)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + i),
                                       (__attribute__((address_space(3))) void *)(g_lds + 16 * base), 16, 0, 0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
  const uint4 *src = reinterpret_cast<const uint4 *>(v.blob);
  uint4 *dst = reinterpret_cast<uint4 *>(g_lds);
  for (uint32_t i = threadIdx.x; i < n; i += BS) dst[i] = src[i];
#endif
}

// table[clamp(rint(x), 0, 65535)] for x = (gamma- or linear-domain value) * 65535, see ph_lut.h:
// 5 float ops, 3 integer ops, 2 LDS reads.
//   * v_rndne FIRST, then + bias (exact on integers).  Adding the bias before rounding would round
//     twice: x + bias has a coarser ulp than x just above a power of two and can manufacture a tie;
//   * the float's own exponent/mantissa bits are the logarithmic block number: one shift;
//   * the delta address is produced by an fma whose result is a DENORMAL: (2*(i+bias) + base)
//     * 2^-149 has exactly that integer as its bit pattern, so no int multiply/add is needed
//     (f32 denormals are enabled in HIP kernels and cost nothing extra on gfx950).
struct LutK {
  float bias, delta_scale, delta_base;
  uint32_t shift, anchor_off;
};
__device__ __forceinline__ LutK make_lut_k(const LutView &v) {
  return LutK{v.bias, v.delta_scale, v.delta_base, v.shift, v.anchor_off};
}
__device__ __forceinline__ float lds_lut_at(const LutK &k, float x) {
  x = __builtin_fminf(__builtin_fmaxf(x, 0.0f), 65535.0f);  // v_med3_f32; NaN -> 0 like the reference
  const float fb = __builtin_rintf(x) + k.bias;               // (float)(idx + bias), exact
  const uint32_t a_addr = ((__float_as_uint(fb) >> k.shift) << 2) + k.anchor_off;
  const uint32_t d_addr = __float_as_uint(fma_rn(fb, k.delta_scale, k.delta_base));
  const uint32_t a = *reinterpret_cast<const uint32_t *>(g_lds + a_addr);
  const uint32_t d = *reinterpret_cast<const uint16_t *>(g_lds + d_addr);
  return __uint_as_float(a + d);
}


// A gamma LUT as the kernels see it: either the compressed table in LDS or the plain f32 table
// in global memory (tables that do not compress, or the "lds_lut" option switched off).
struct LutInLds {
  LutK k;
  __device__ __forceinline__ float at(float x) const { return lds_lut_at(k, x); }
};
struct LutInGlobal {
  const float *__restrict__ t;
  __device__ __forceinline__ float at(float x) const { return t[sat_u16_rte(x)]; }
};

}  // namespace ph
