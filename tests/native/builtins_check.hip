// builtins_check.hip - runs AMD's OpenCL device library (the bitcode ROCm's OpenCL links into every
// kernel: /opt/rocm/amdgcn/bitcode/opencl.bc + ocml.bc) NATIVELY on the MI355X next to the product's
// own arithmetic primitives (phaneron_amd/csrc/ph_device.h, ph_ldslut.h) and counts disagreements.
//
// TEST INFRASTRUCTURE (tests/test_builtins_gpu.py builds the argument list and reads the JSON line).
// Why: the reference's kernels are OpenCL C; their results hinge on what `dot`, `fma`,
// `convert_ushort_sat_rte/_rtz`, `convert_uchar_sat_rte` and `round` do (v210.ts:68-77,148-155,176-183).
// The product spells those out by hand (explicit fma chains, magic-number rounding); this program proves
// the spelling against the real library on the real target:
//   converts : ALL 2^32 float bit patterns
//   dot3/dot4: 2^28 hash-generated operand sets (every exponent, denormals, infinities, NaNs; plus
//              code-value x coefficient shaped operands)
// and dumps device-library results for a seeded sample so the x86-64 retarget of the same bitcode
// (oracle/_ref, devlib_builtins.py) can be compared with the gfx950 execution.
// Build (see __graft_entry__.build): hipcc --offload-arch=gfx950 -O3 -ffp-contract=off
//   -Xarch_device -mlink-builtin-bitcode opencl.bc ocml.bc
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../phaneron_amd/csrc/ph_device.h"
#include "../../phaneron_amd/csrc/ph_ldslut.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); exit(2);} } while (0)

typedef float f4v __attribute__((ext_vector_type(4)));
typedef float f3v __attribute__((ext_vector_type(3)));
// the OpenCL built-ins, by their mangled names in opencl.bc
__device__ float ocl_dot4(f4v, f4v) asm("_Z3dotDv4_fS_");
__device__ float ocl_dot3(f3v, f3v) asm("_Z3dotDv3_fS_");
__device__ float ocl_fma(float, float, float) asm("_Z3fmafff");
__device__ float ocl_round(float) asm("_Z5roundf");
__device__ unsigned short ocl_us_sat_rte(float) asm("_Z22convert_ushort_sat_rtef");
__device__ unsigned short ocl_us_sat_rtz(float) asm("_Z22convert_ushort_sat_rtzf");
__device__ unsigned short ocl_us_sat(float) asm("_Z18convert_ushort_satf");
__device__ unsigned char ocl_uc_sat_rte(float) asm("_Z21convert_uchar_sat_rtef");

enum { C_RTE, C_RTZ, C_SAT, C_UCHAR, C_ROUND_RTZ, C_LDS_INDEX, C_CTRL_TRUNC, C_DOT4, C_DOT3, C_FMA, C_CTRL_UNFUSED, C_COUNT };
static const char *kNames[C_COUNT] = {
    "sat_u16_rte == convert_ushort_sat_rte", "sat_u16_trunc == convert_ushort_sat_rtz",
    "sat_u16_trunc == convert_ushort_sat", "sat_u8_rte == convert_uchar_sat_rte",
    "sat_u16_trunc(roundf) == convert_ushort_sat_rtz(round)", "lds_lut_index_unit(t) == convert_ushort_sat_rte(t * 65535)",
    "CONTROL (must disagree): sat_u16_trunc vs convert_ushort_sat_rte",
    "dot4 == dot(float4)", "dot3 == dot(float3)", "fma_rn == fma",
    "CONTROL (must disagree): unfused mul/add chain vs dot(float4)"};

struct Report {
  unsigned long long tested[C_COUNT], bad[C_COUNT];
  unsigned first_bad[C_COUNT];  // operand bits (converts) / operand-set index (dot) of one disagreement
};

__device__ __forceinline__ void note(Report *r, int c, bool ok, unsigned what) {
  if (!ok) {
    atomicAdd(&r->bad[c], 1ull);
    r->first_bad[c] = what;
  }
}

// every float bit pattern in [first, first + n)
__global__ void sweep_converts(Report *r, unsigned first, unsigned long long n) {
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (unsigned long long)gridDim.x * blockDim.x) {
    const unsigned bits = first + (unsigned)i;
    const float x = __uint_as_float(bits);
    note(r, C_RTE, ph::sat_u16_rte(x) == ocl_us_sat_rte(x), bits);
    note(r, C_RTZ, ph::sat_u16_trunc(x) == ocl_us_sat_rtz(x), bits);
    note(r, C_SAT, ph::sat_u16_trunc(x) == ocl_us_sat(x), bits);
    note(r, C_UCHAR, ph::sat_u8_rte(x) == ocl_uc_sat_rte(x), bits);
    note(r, C_ROUND_RTZ, ph::sat_u16_trunc(__builtin_roundf(x)) == ocl_us_sat_rtz(ocl_round(x)), bits);
    note(r, C_CTRL_TRUNC, ph::sat_u16_trunc(x) == ocl_us_sat_rte(x), bits);  // negative control
    // the LDS kernels' index: clamp to [0,1], * 65535, + 1.5 * 2^23, low 16 bits (ph_ldslut.h)
    note(r, C_LDS_INDEX, (__float_as_uint(ph::lds_lut_index_unit(x)) & 0xFFFFu) == ocl_us_sat_rte(x * 65535.0f), bits);
  }
}

__device__ __forceinline__ unsigned mix32(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
// operand k of set i: a third raw bit patterns, a third "code value" integers 0..1023 / unit-range values,
// a third small coefficients - the shapes the colour matrices see
__device__ __forceinline__ float operand(unsigned long long i, unsigned k, unsigned seed) {
  const unsigned h = mix32((unsigned)i * 0x9E3779B9u ^ mix32((unsigned)(i >> 32) + k * 0x85EBCA6Bu + seed));
  const unsigned kind = mix32(h ^ 0xA5A5A5A5u) % 3u;
  if (kind == 0) return __uint_as_float(h);
  if (kind == 1) return (h & 1) ? (float)(h >> 22) : (float)(h >> 8) * (1.0f / 16777216.0f);
  return ((float)(int)(h >> 8) - 8388608.0f) * (1.0f / 4194304.0f) * ((h & 2) ? 0.01f : 1.0f);
}
__device__ __forceinline__ bool same_f32(float a, float b) {  // bit-equal, or both NaN (payloads are not compared)
  return __float_as_uint(a) == __float_as_uint(b) || (a != a && b != b);
}
__global__ void sweep_dots(Report *r, unsigned long long n, unsigned seed, float *dump4, float *dump3, unsigned n_dump) {
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (unsigned long long)gridDim.x * blockDim.x) {
    float a[4], b[4];
    for (unsigned k = 0; k < 4; ++k) a[k] = operand(i, k, seed), b[k] = operand(i, k + 4, seed);
    const f4v va = {a[0], a[1], a[2], a[3]}, vb = {b[0], b[1], b[2], b[3]};
    const f3v va3 = {a[0], a[1], a[2]}, vb3 = {b[0], b[1], b[2]};
    const float d4 = ocl_dot4(va, vb), d3 = ocl_dot3(va3, vb3);
    note(r, C_DOT4, same_f32(ph::dot4(a[0], a[1], a[2], a[3], make_float4(b[0], b[1], b[2], b[3])), d4), (unsigned)i);
    note(r, C_DOT3, same_f32(ph::dot3(a[0], a[1], a[2], b[0], b[1], b[2]), d3), (unsigned)i);
    note(r, C_CTRL_UNFUSED, same_f32(((a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]) + a[3] * b[3], d4), (unsigned)i);  // negative control
    note(r, C_FMA, same_f32(ph::fma_rn(a[0], b[0], a[1]), ocl_fma(a[0], b[0], a[1])), (unsigned)i);
    if (i < n_dump) {  // operands + the device library's results, for the x86 retarget to reproduce
      for (unsigned k = 0; k < 4; ++k) dump4[i * 9 + k] = a[k], dump4[i * 9 + 4 + k] = b[k];
      dump4[i * 9 + 8] = d4;
      dump3[i] = d3;
    }
  }
}
__global__ void dump_converts(const unsigned *bits, unsigned n, unsigned short *out) {  // out[which * n + i]
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float x = __uint_as_float(bits[i]);
  out[0 * n + i] = ocl_us_sat_rte(x), out[1 * n + i] = ocl_us_sat_rtz(x), out[2 * n + i] = ocl_us_sat(x);
  out[3 * n + i] = ocl_uc_sat_rte(x), out[4 * n + i] = ocl_us_sat_rtz(ocl_round(x));
}

int main(int argc, char **argv) {
  // usage: builtins_check [log2 of dot operand sets = 28] [dump file]
  const int dot_log2 = argc > 1 ? atoi(argv[1]) : 28;
  const char *dump_path = argc > 2 ? argv[2] : nullptr;
  const unsigned n_dump = 1u << 20;
  Report *dr, hr;
  CK(hipMalloc(&dr, sizeof(Report)));
  CK(hipMemset(dr, 0, sizeof(Report)));
  float *dump4, *dump3;
  CK(hipMalloc(&dump4, (size_t)n_dump * 9 * sizeof(float)));
  CK(hipMalloc(&dump3, (size_t)n_dump * sizeof(float)));
  // converts: all 2^32 patterns, in 16 launches (keeps each launch short)
  for (unsigned part = 0; part < 16; ++part) sweep_converts<<<4096, 256>>>(dr, part << 28, 1ull << 28);
  const unsigned long long n_dot = 1ull << dot_log2;
  sweep_dots<<<4096, 256>>>(dr, n_dot, 0x5EED0000u, dump4, dump3, n_dump);
  CK(hipDeviceSynchronize());
  CK(hipMemcpy(&hr, dr, sizeof hr, hipMemcpyDeviceToHost));
  for (int c = 0; c < C_COUNT; ++c) hr.tested[c] = c < C_DOT4 ? (1ull << 32) : n_dot;
  if (dump_path) {
    // layout: u32 n | n x 9 f32 (a0..3, b0..3, dot4) | n f32 dot3 | n u32 convert inputs | 5 x n u16 convert outputs
    std::vector<float> h4((size_t)n_dump * 9), h3(n_dump);
    CK(hipMemcpy(h4.data(), dump4, h4.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(h3.data(), dump3, h3.size() * 4, hipMemcpyDeviceToHost));
    std::vector<unsigned> bits(n_dump);
    for (unsigned i = 0; i < n_dump; ++i) {  // ties, bounds and random patterns
      unsigned x = i * 2654435761u;
      x ^= x >> 15;
      if (i % 4 == 0) { const float t = (float)(x % 70000u) - 2000.0f + 0.5f * (float)((x >> 20) & 3); memcpy(&bits[i], &t, 4); }
      else bits[i] = x;
    }
    unsigned *dbits; unsigned short *dout;
    CK(hipMalloc(&dbits, n_dump * 4)); CK(hipMalloc(&dout, (size_t)n_dump * 5 * 2));
    CK(hipMemcpy(dbits, bits.data(), n_dump * 4, hipMemcpyHostToDevice));
    dump_converts<<<(n_dump + 255) / 256, 256>>>(dbits, n_dump, dout);
    std::vector<unsigned short> hout((size_t)n_dump * 5);
    CK(hipMemcpy(hout.data(), dout, hout.size() * 2, hipMemcpyDeviceToHost));
    FILE *f = fopen(dump_path, "wb");
    if (!f) { fprintf(stderr, "cannot write %s\n", dump_path); return 2; }
    fwrite(&n_dump, 4, 1, f);
    fwrite(h4.data(), 4, h4.size(), f); fwrite(h3.data(), 4, h3.size(), f);
    fwrite(bits.data(), 4, bits.size(), f); fwrite(hout.data(), 2, hout.size(), f);
    fclose(f);
  }
  hipDeviceProp_t props; CK(hipGetDeviceProperties(&props, 0));
  printf("{\"device\":\"%s\",\"checks\":[", props.gcnArchName);
  unsigned long long total_bad = 0;
  for (int c = 0; c < C_COUNT; ++c) {
    const bool control = c == C_CTRL_TRUNC || c == C_CTRL_UNFUSED;
    printf("%s{\"check\":\"%s\",\"control\":%s,\"tested\":%llu,\"bad\":%llu,\"example\":\"0x%08x\"}", c ? "," : "", kNames[c],
           control ? "true" : "false", hr.tested[c], hr.bad[c], hr.first_bad[c]);
    total_bad += control ? (hr.bad[c] ? 0 : 1) : hr.bad[c];  // a control that agrees everywhere means the checker is blind
  }
  printf("]}\n");
  return total_bad ? 1 : 0;
}
