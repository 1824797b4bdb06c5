// opbench.hip - issue cost (cycles per wave64 instruction per SIMD) of the VALU / LDS-address ops
// the lookup path is made of.  8 independent chains per lane, 16 waves per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

#define OPKERNEL(NAME, TYPE, INIT, EXPR)                                                   \
  __global__ void NAME(TYPE* out, TYPE a, TYPE b, int iters) {                             \
    TYPE x[8];                                                                             \
    for (int k = 0; k < 8; ++k) x[k] = INIT;                                               \
    for (int i = 0; i < iters; ++i) {                                                      \
      _Pragma("unroll") for (int k = 0; k < 8; ++k) { TYPE v = x[k]; x[k] = (EXPR); }      \
    }                                                                                      \
    TYPE s = x[0];                                                                         \
    for (int k = 1; k < 8; ++k) s = s + x[k];                                              \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                        \
  }

typedef unsigned int u32;
OPKERNEL(k_fma, float, (float)(threadIdx.x + k), __builtin_fmaf(v, a, b))
OPKERNEL(k_mul, float, (float)(threadIdx.x + k) * 1e-3f + 1.0f, v * a)
OPKERNEL(k_addf, float, (float)(threadIdx.x + k), v + a)
OPKERNEL(k_rndne, float, (float)(threadIdx.x + k) * 0.37f, __builtin_rintf(v) + a)       /* 2 ops */
OPKERNEL(k_med3, float, (float)(threadIdx.x + k), __builtin_fminf(__builtin_fmaxf(v, a), b) + a) /* med3 + add */
OPKERNEL(k_floor, float, (float)(threadIdx.x + k) * 0.37f, __builtin_floorf(v) + a)
OPKERNEL(k_cvt_u32_f32, float, (float)(threadIdx.x + k), (float)(u32)v)                   /* cvt + cvt */
OPKERNEL(k_addu, u32, threadIdx.x + k, v + a)
OPKERNEL(k_lshl_add, u32, threadIdx.x + k, (v << 2) + a)
OPKERNEL(k_lshr, u32, threadIdx.x * 977 + k, (v >> 1) ^ a)                                /* lshr + xor */
OPKERNEL(k_minu, u32, threadIdx.x * 977 + k, (v < a ? v : a) + b)                         /* min + add */
OPKERNEL(k_and_or, u32, threadIdx.x * 977 + k, (v & a) | b)
OPKERNEL(k_bfe, u32, threadIdx.x * 977 + k, ((v >> 3) & 0x3ff) + a)                        /* bfe + add */
OPKERNEL(k_mad24, u32, threadIdx.x + k, __umul24(v, a) + b)
OPKERNEL(k_cndmask, u32, threadIdx.x * 977 + k, (v & 16) ? (v + a) : (v + b))

template <typename K, typename T>
static void run(const char* name, K kern, T a, T b, double ops_per_iter) {
  const int blocks = 256 * 4, thr = 1024, iters = 2048;
  T* out; CK(hipMalloc(&out, (size_t)blocks * thr * sizeof(T)));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  kern<<<blocks, thr>>>(out, a, b, iters); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0)); for (int r = 0; r < 5; ++r) kern<<<blocks, thr>>>(out, a, b, iters); CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
  const double wave_iters = (double)blocks * thr / 64 * iters * 8;  // chain steps
  printf("{\"op\":\"%s\",\"ns_per_chain_step_per_simd\":%.3f,\"steps_T_per_s\":%.2f,\"ops_per_step\":%.0f}\n", name,
         ms * 1e6 / (wave_iters / 1024), wave_iters * 64 / ms / 1e9, ops_per_iter);
  CK(hipFree(out));
}

int main() {
  run("v_fma_f32", k_fma, 1.0001f, 0.5f, 1);
  run("v_mul_f32", k_mul, 1.0001f, 0.f, 1);
  run("v_add_f32", k_addf, 1.5f, 0.f, 1);
  run("v_rndne_f32+add", k_rndne, 0.25f, 0.f, 2);
  run("v_med3_f32+add", k_med3, 1.0f, 65535.f, 2);
  run("v_floor_f32+add", k_floor, 0.25f, 0.f, 2);
  run("v_cvt_u32_f32+v_cvt_f32_u32", k_cvt_u32_f32, 0.f, 0.f, 2);
  run("v_add_u32", k_addu, 12345u, 0u, 1);
  run("v_lshl_add_u32", k_lshl_add, 12345u, 0u, 1);
  run("v_lshrrev+xor", k_lshr, 0x9e3779b9u, 0u, 2);
  run("v_min_u32+add", k_minu, 0x7fffffffu, 3u, 2);
  run("v_and_or_b32", k_and_or, 0xfffff0ffu, 0x100u, 1);
  run("v_bfe_u32+add", k_bfe, 77u, 0u, 2);
  run("v_mad_u32_u24", k_mad24, 3u, 5u, 1);
  run("v_cndmask path", k_cndmask, 3u, 5u, 3);
  return 0;
}
