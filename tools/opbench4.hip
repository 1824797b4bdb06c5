// opbench4.hip - why does a REAL VALU stream cost more per instruction than opbench3's independent chains?
// Same harness (s_memtime / s_memrealtime per wave, 4 waves per SIMD, every CU busy), three suspects:
//   banks   : v_fma_f32 whose three VGPR sources sit in the same register bank (vN, vN+4, vN+8) vs four apart
//   depend  : ONE dependent chain per wave (each instruction needs the previous result) vs 2 / 4 / 8 chains
//   literal : 8-byte encodings (32-bit literal operand) and SGPR operands, as the colour maths uses them
//   waitcnt : an s_waitcnt lgkmcnt(0) (nothing outstanding) after every 15 VALU, as the LUT code has them
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
struct Stamp { unsigned long long cyc, real; };

#define KERNEL(NAME, BODY, PER_ITER)                                                              \
  __global__ __launch_bounds__(1024) void NAME(Stamp* st, float* out, float a, float b, int iters) { \
    asm volatile("v_mov_b32 v8, %0\n v_mov_b32 v9, %1\n v_mov_b32 v10, %0\n v_mov_b32 v11, %1\n"      \
                 "v_mov_b32 v12, %0\n v_mov_b32 v13, %1\n v_mov_b32 v14, %0\n v_mov_b32 v15, %1\n"    \
                 "v_mov_b32 v16, %0\n v_mov_b32 v17, %1\n v_mov_b32 v18, %0\n v_mov_b32 v19, %1\n"    \
                 "v_mov_b32 v20, %0\n v_mov_b32 v21, %1\n v_mov_b32 v22, %0\n v_mov_b32 v23, %1\n"    \
                 :: "v"(a + threadIdx.x), "v"(b) : "v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23"); \
    __syncthreads();                                                                              \
    unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();  \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                            \
    for (int i = 0; i < iters; ++i) {                                                             \
      asm volatile(BODY BODY BODY BODY ::: "v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23"); \
    }                                                                                             \
    unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();  \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                            \
    float s;                                                                                      \
    asm volatile("v_add_f32 %0, v8, v12\n v_add_f32 %0, %0, v16\n v_add_f32 %0, %0, v20" : "=v"(s)); \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                               \
    if ((threadIdx.x & 63) == 0) st[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = Stamp{c1 - c0, r1 - r0}; \
  }                                                                                               \
  static const int NAME##_per_iter = 4 * (PER_ITER);

// 8 instructions per BODY unless noted
// sources in FOUR different banks-of-4 (v8, v9, v10 ...): d = d * vA + vB with A, B, d in distinct banks
#define B_SPREAD "v_fma_f32 v8, v8, v13, v18\n v_fma_f32 v9, v9, v14, v19\n v_fma_f32 v10, v10, v15, v16\n v_fma_f32 v11, v11, v12, v17\n" \
                 "v_fma_f32 v20, v20, v13, v18\n v_fma_f32 v21, v21, v14, v19\n v_fma_f32 v22, v22, v15, v16\n v_fma_f32 v23, v23, v12, v17\n"
// all three sources in the SAME bank (register numbers equal mod 4)
#define B_SAME   "v_fma_f32 v8, v8, v12, v16\n v_fma_f32 v9, v9, v13, v17\n v_fma_f32 v10, v10, v14, v18\n v_fma_f32 v11, v11, v15, v19\n" \
                 "v_fma_f32 v20, v20, v12, v16\n v_fma_f32 v21, v21, v13, v17\n v_fma_f32 v22, v22, v14, v18\n v_fma_f32 v23, v23, v15, v19\n"
// two of three sources in the same bank
#define B_TWO    "v_fma_f32 v8, v8, v12, v17\n v_fma_f32 v9, v9, v13, v18\n v_fma_f32 v10, v10, v14, v19\n v_fma_f32 v11, v11, v15, v16\n" \
                 "v_fma_f32 v20, v20, v12, v17\n v_fma_f32 v21, v21, v13, v18\n v_fma_f32 v22, v22, v14, v19\n v_fma_f32 v23, v23, v15, v16\n"
#define D_1 "v_fma_f32 v8, v8, v13, v18\n v_fma_f32 v8, v8, v13, v18\n v_fma_f32 v8, v8, v13, v18\n v_fma_f32 v8, v8, v13, v18\n" \
            "v_fma_f32 v8, v8, v13, v18\n v_fma_f32 v8, v8, v13, v18\n v_fma_f32 v8, v8, v13, v18\n v_fma_f32 v8, v8, v13, v18\n"
#define D_2 "v_fma_f32 v8, v8, v13, v18\n v_fma_f32 v9, v9, v14, v19\n v_fma_f32 v8, v8, v13, v18\n v_fma_f32 v9, v9, v14, v19\n" \
            "v_fma_f32 v8, v8, v13, v18\n v_fma_f32 v9, v9, v14, v19\n v_fma_f32 v8, v8, v13, v18\n v_fma_f32 v9, v9, v14, v19\n"
#define D_4 "v_fma_f32 v8, v8, v13, v18\n v_fma_f32 v9, v9, v14, v19\n v_fma_f32 v10, v10, v15, v16\n v_fma_f32 v11, v11, v12, v17\n" \
            "v_fma_f32 v8, v8, v13, v18\n v_fma_f32 v9, v9, v14, v19\n v_fma_f32 v10, v10, v15, v16\n v_fma_f32 v11, v11, v12, v17\n"
// the colour-maths shapes: literal multiplier, SGPR coefficient, clamp modifier
#define L_LIT "v_mul_f32 v8, 0x477fff00, v8\n v_add_f32 v9, 0x4b400000, v9\n v_mul_f32 v10, 0x477fff00, v10\n v_add_f32 v11, 0x4b400000, v11\n" \
              "v_mul_f32 v20, 0x477fff00, v20\n v_add_f32 v21, 0x4b400000, v21\n v_mul_f32 v22, 0x477fff00, v22\n v_add_f32 v23, 0x4b400000, v23\n"
#define L_SGPR "v_fma_f32 v8, s4, v8, v13\n v_fmac_f32 v9, s5, v14\n v_fma_f32 v10, s4, v10, v15\n v_fmac_f32 v11, s5, v12\n" \
               "v_fma_f32 v20, s4, v20, v13\n v_fmac_f32 v21, s5, v14\n v_fma_f32 v22, s4, v22, v15\n v_fmac_f32 v23, s5, v12\n"
#define W_WAIT B_SPREAD B_SPREAD "s_waitcnt lgkmcnt(0)\n"
#define W_NOWAIT B_SPREAD B_SPREAD

KERNEL(k_spread, B_SPREAD, 8)
KERNEL(k_same, B_SAME, 8)
KERNEL(k_two, B_TWO, 8)
KERNEL(k_d1, D_1, 8)
KERNEL(k_d2, D_2, 8)
KERNEL(k_d4, D_4, 8)
KERNEL(k_lit, L_LIT, 8)
KERNEL(k_sgpr, L_SGPR, 8)
KERNEL(k_wait, W_WAIT, 16)
KERNEL(k_nowait, W_NOWAIT, 16)

template <typename K>
static void run(const char* name, K kern, int per_iter, int waves_per_simd) {
  const int iters = 2048, cus = 256;
  const int thr = 256 * waves_per_simd, blocks = cus, waves = blocks * thr / 64;
  float* out; Stamp* st;
  CK(hipMalloc(&out, (size_t)blocks * thr * 4)); CK(hipMalloc(&st, waves * sizeof(Stamp)));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  kern<<<blocks, thr>>>(st, out, 1.0003f, 0.5f, iters); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0)); kern<<<blocks, thr>>>(st, out, 1.0003f, 0.5f, iters); CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<Stamp> h(waves); CK(hipMemcpy(h.data(), st, waves * sizeof(Stamp), hipMemcpyDeviceToHost));
  double cmax = 0, ghz = 0;
  for (auto& s : h) { cmax = std::max(cmax, (double)s.cyc); ghz += (double)s.cyc / (double)s.real * 0.1; }
  const double instrs = (double)iters * per_iter;
  printf("{\"case\":\"%s\",\"waves_per_simd\":%d,\"cyc_per_valu_per_simd\":%.3f,\"ghz\":%.3f,\"ns_wall\":%.3f}\n", name, waves_per_simd,
         cmax / (instrs * waves_per_simd), ghz / waves, ms * 1e6 / (instrs * waves_per_simd));
  fflush(stdout);
  CK(hipFree(out)); CK(hipFree(st));
}
#define R(n, k) for (int w : {1, 2, 4}) run(n, k, k##_per_iter, w);
int main() {
  R("fma, sources in different banks", k_spread) R("fma, two sources in one bank", k_two) R("fma, three sources in one bank", k_same)
  R("fma, 1 dependent chain per wave", k_d1) R("fma, 2 chains", k_d2) R("fma, 4 chains", k_d4)
  R("mul/add with 32-bit literal", k_lit) R("fma/fmac with SGPR coefficient", k_sgpr)
  R("16 fma, no waitcnt", k_nowait) R("16 fma + s_waitcnt lgkmcnt(0)", k_wait)
  return 0;
}
