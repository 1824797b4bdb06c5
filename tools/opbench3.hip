// opbench3.hip - cycle-accurate issue cost of gfx950 VALU / LDS instructions.
//
// opbench2 divided wall time by instruction count and so could not tell "the instruction takes 3
// cycles" from "the chip clocked down".  Here every wave brackets its loop with s_memtime (shader
// cycles) AND s_memrealtime (constant 100 MHz), so each line reports
//   cyc   = shader cycles per wave64 instruction per SIMD  (delta_memtime * 1 / (instrs * waves_per_simd))
//   ghz   = sustained shader clock during the loop           (delta_memtime / delta_realtime * 0.1)
//   ns    = wall time per instruction per SIMD               (hipEvent, cross-check)
// at 1, 2, 4 and 8 waves per SIMD, with every CU busy (grid = 256 CUs) or a single CU busy.
// Build: hipcc --offload-arch=gfx950 -O3 tools/opbench3.hip -o tools/opbench3
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

struct Stamp { unsigned long long cyc, real, r0, r1; };

#define CHAINS 8
#define UNROLL 16
#define ASMK(NAME, TEXT)                                                                     \
  __global__ __launch_bounds__(1024) void NAME(Stamp* st, unsigned* out, unsigned a, unsigned b, int iters) { \
    unsigned x[CHAINS];                                                                      \
    for (int k = 0; k < CHAINS; ++k) x[k] = threadIdx.x * 977u + k * 131u + 0x3f800000u;     \
    unsigned va = a + (threadIdx.x & 1), vb = b;                                             \
    asm volatile("" : "+v"(va), "+v"(vb));                                                   \
    __syncthreads();                                                                         \
    unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime(); \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                       \
    for (int i = 0; i < iters; ++i) {                                                        \
      _Pragma("unroll") for (int u = 0; u < UNROLL; ++u)                                          \
      _Pragma("unroll") for (int k = 0; k < CHAINS; ++k)                                     \
        asm volatile(TEXT : "+v"(x[k]) : "v"(va), "v"(vb));                                  \
    }                                                                                        \
    unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime(); \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                       \
    unsigned s = x[0];                                                                       \
    for (int k = 1; k < CHAINS; ++k) s ^= x[k];                                              \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                          \
    if ((threadIdx.x & 63) == 0) st[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = Stamp{c1 - c0, r1 - r0, r0, r1}; \
  }

ASMK(k_fma, "v_fma_f32 %0, %0, %1, %2")
ASMK(k_fmac, "v_fmac_f32 %0, %1, %2")
ASMK(k_mul, "v_mul_f32 %0, %0, %1")
ASMK(k_add, "v_add_f32 %0, %0, %1")
ASMK(k_add_clamp, "v_add_f32_e64 %0, %0, %1 clamp")
ASMK(k_sub, "v_sub_f32 %0, %0, %1")
ASMK(k_max, "v_max_f32 %0, %0, %1")
ASMK(k_med3, "v_med3_f32 %0, %0, %1, %2")
ASMK(k_rndne, "v_rndne_f32 %0, %0")
ASMK(k_cvt_f32_u32, "v_cvt_f32_u32 %0, %0")
ASMK(k_cvt_u32_f32, "v_cvt_u32_f32 %0, %0")
ASMK(k_cvt_ubyte0, "v_cvt_f32_ubyte0 %0, %0")
ASMK(k_addu, "v_add_u32 %0, %0, %1")
ASMK(k_lshl_add, "v_lshl_add_u32 %0, %0, 2, %1")
ASMK(k_lshr, "v_lshrrev_b32 %0, 1, %0")
ASMK(k_lshl, "v_lshlrev_b32 %0, 1, %0")
ASMK(k_and, "v_and_b32 %0, %0, %1")
ASMK(k_or, "v_or_b32 %0, %0, %1")
ASMK(k_bfe, "v_bfe_u32 %0, %0, 3, 10")
ASMK(k_perm, "v_perm_b32 %0, %0, %1, %2")
ASMK(k_mov, "v_mov_b32 %0, %1")
ASMK(k_and_or, "v_and_or_b32 %0, %0, %1, %2")
ASMK(k_mad24, "v_mad_u32_u24 %0, %0, %1, %2")
ASMK(k_sdwa_or, "v_or_b32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1")
ASMK(k_exp, "v_exp_f32 %0, %0")
// two instructions per slot: do the classes overlap (dual pipe) or add?
ASMK(k_fma_then_cvt, "v_fma_f32 %0, %0, %1, %2\n\tv_cvt_f32_u32 %0, %0")
ASMK(k_fma_then_lshl_add, "v_fma_f32 %0, %0, %1, %2\n\tv_lshl_add_u32 %0, %0, 2, %1")
ASMK(k_fma_then_fma, "v_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2")
ASMK(k_fma_dpp, "v_fma_f32 %0, %0, %1, %2\n\tv_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
// neighbour moves: inside a row of 16 lanes against across the whole wave (s_nop: the VALU-write -> DPP-read hazard)
ASMK(k_fma_rowshr, "v_fma_f32 %0, %0, %1, %2\n\ts_nop 1\n\tv_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf")
ASMK(k_fma_waveshr, "v_fma_f32 %0, %0, %1, %2\n\ts_nop 1\n\tv_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf")
ASMK(k_fma_nop, "v_fma_f32 %0, %0, %1, %2\n\ts_nop 1")

// packed f32
__global__ __launch_bounds__(1024) void k_pk(Stamp* st, unsigned* out, unsigned a, unsigned b, int iters) {
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 x[CHAINS], aa = {__uint_as_float(a), __uint_as_float(a)}, bb = {__uint_as_float(b), __uint_as_float(b)};
  for (int k = 0; k < CHAINS; ++k) x[k] = f2{(float)(threadIdx.x + k), (float)k};
  asm volatile("" : "+v"(aa), "+v"(bb));
  __syncthreads();
  unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < UNROLL; ++u)
#pragma unroll
      for (int k = 0; k < CHAINS; ++k) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x[k]) : "v"(aa), "v"(bb));
  }
  unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  float s = 0;
  for (int k = 0; k < CHAINS; ++k) s += x[k].x + x[k].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = __float_as_uint(s);
  if ((threadIdx.x & 63) == 0) st[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = Stamp{c1 - c0, r1 - r0, r0, r1};
}

// LDS gathers: MODE 0 random b32 over 128 KiB, 1 random u16, 2 conflict-free b32 (lane-linear), 3 b64 random
template <int MODE>
__global__ __launch_bounds__(1024) void k_lds(Stamp* st, unsigned* out, unsigned a, unsigned b, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  for (int i = threadIdx.x; i < 32768; i += blockDim.x) reinterpret_cast<unsigned*>(smem)[i] = i * 2654435761u;
  unsigned x[CHAINS];
  for (int k = 0; k < CHAINS; ++k) x[k] = (threadIdx.x * 977u + k * 131071u) * 2654435761u;
  __syncthreads();
  unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      unsigned v[CHAINS];
#pragma unroll
      for (int k = 0; k < CHAINS; ++k) {
        unsigned addr;
        if (MODE == 2) addr = ((threadIdx.x & 63) * 4 + (x[k] & 0x1ff00u)) & 0x1fffcu;
        else if (MODE == 3) addr = x[k] & 0x1fff8u;
        else if (MODE == 1) addr = x[k] & 0x1fffeu;
        else addr = x[k] & 0x1fffcu;
        if (MODE == 1) asm volatile("ds_read_u16 %0, %1" : "=v"(v[k]) : "v"(addr));
        else if (MODE == 3) { unsigned long long t; asm volatile("ds_read_b64 %0, %1" : "=v"(t) : "v"(addr)); v[k] = (unsigned)t; asm volatile("" : "+v"(v[k])); }
        else asm volatile("ds_read_b32 %0, %1" : "=v"(v[k]) : "v"(addr));
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int k = 0; k < CHAINS; ++k) x[k] = x[k] * 1664525u + v[k];
    }
  }
  unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  unsigned s = x[0];
  for (int k = 1; k < CHAINS; ++k) s ^= x[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) st[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = Stamp{c1 - c0, r1 - r0, r0, r1};
}

template <typename K>
static void run(const char* name, K kern, int instr_per_slot, int waves_per_simd, int cus, size_t lds = 0) {
  const int iters = 1024;
  // waves_per_simd w: one block of 256*w threads per CU (w <= 4), two blocks of 1024 for w = 8
  int thr = 256 * waves_per_simd, blocks = cus;
  if (waves_per_simd == 8) thr = 1024, blocks = 2 * cus;
  if (lds && waves_per_simd == 8) return;  // one 128 KiB table per CU
  const int waves = blocks * thr / 64;
  unsigned* out; Stamp* st;
  CK(hipMalloc(&out, (size_t)blocks * thr * sizeof(unsigned)));
  CK(hipMalloc(&st, waves * sizeof(Stamp)));
  if (lds) CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  kern<<<blocks, thr, lds>>>(st, out, 0x3f800347u, 0x3f000000u, iters); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0)); kern<<<blocks, thr, lds>>>(st, out, 0x3f800347u, 0x3f000000u, iters); CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<Stamp> h(waves);
  CK(hipMemcpy(h.data(), st, waves * sizeof(Stamp), hipMemcpyDeviceToHost));
  // Waves of one SIMD are served oldest-first, so a median wave finishes early: the SIMD's cost per
  // instruction is the LONGEST wave's cycles / (instructions * waves per SIMD).  The realtime counter
  // (s_memrealtime, 100 MHz) gives the span first-start .. last-end, cross-checked against the hipEvent time.
  double cmax = 0, ghz_sum = 0; unsigned long long rmin = ~0ull, rmax = 0;
  for (int i = 0; i < waves; ++i) {
    cmax = std::max(cmax, (double)h[i].cyc); ghz_sum += (double)h[i].cyc / (double)h[i].real * 0.1;
    rmin = std::min(rmin, h[i].r0); rmax = std::max(rmax, h[i].r1);
  }
  const double instrs = (double)iters * UNROLL * CHAINS * instr_per_slot;  // per wave
  const int wps = waves_per_simd == 8 ? 8 : waves_per_simd;
  printf("{\"instr\":\"%s\",\"waves_per_simd\":%d,\"cus\":%d,\"cyc_per_instr_per_simd\":%.3f,\"ghz\":%.3f,\"ns_wall_per_instr_per_simd\":%.3f,\"span_us_realtime\":%.1f,\"event_us\":%.1f}\n",
         name, waves_per_simd, cus, cmax / (instrs * wps), ghz_sum / waves, ms * 1e6 / (instrs * wps), (rmax - rmin) / 100.0, ms * 1e3);
  fflush(stdout);
  CK(hipFree(out)); CK(hipFree(st));
}

int main(int argc, char** argv) {
  if (argc > 1 && !strcmp(argv[1], "dpp")) {  // only the neighbour-move rows
    run("pair fma+mov_dpp quad_perm", k_fma_dpp, 2, 4, 256);
    run("pair fma+s_nop", k_fma_nop, 1, 4, 256);
    run("pair fma+s_nop+mov_dpp row_shr:1", k_fma_rowshr, 2, 4, 256);
    run("pair fma+s_nop+mov_dpp wave_shr:1", k_fma_waveshr, 2, 4, 256);
    return 0;
  }
  const bool quick = argc > 1;
  const int wlist[] = {1, 2, 4, 8};
#define R(n, k, ips) for (int w : wlist) { run(n, k, ips, w, 256); } if (!quick) run(n, k, ips, 4, 1);
  R("v_fma_f32", k_fma, 1) R("v_fmac_f32", k_fmac, 1) R("v_mul_f32", k_mul, 1) R("v_add_f32", k_add, 1)
  R("v_add_f32 clamp", k_add_clamp, 1) R("v_sub_f32", k_sub, 1) R("v_max_f32", k_max, 1) R("v_med3_f32", k_med3, 1)
  R("v_rndne_f32", k_rndne, 1) R("v_cvt_f32_u32", k_cvt_f32_u32, 1) R("v_cvt_u32_f32", k_cvt_u32_f32, 1)
  R("v_cvt_f32_ubyte0", k_cvt_ubyte0, 1) R("v_add_u32", k_addu, 1) R("v_lshl_add_u32", k_lshl_add, 1)
  R("v_lshrrev_b32", k_lshr, 1) R("v_lshlrev_b32", k_lshl, 1) R("v_and_b32", k_and, 1) R("v_or_b32", k_or, 1)
  R("v_bfe_u32", k_bfe, 1) R("v_perm_b32", k_perm, 1) R("v_mov_b32", k_mov, 1) R("v_and_or_b32", k_and_or, 1)
  R("v_mad_u32_u24", k_mad24, 1) R("v_or_b32_sdwa", k_sdwa_or, 1) R("v_exp_f32", k_exp, 1) R("v_pk_fma_f32", k_pk, 1)
  R("pair fma+cvt", k_fma_then_cvt, 2) R("pair fma+lshl_add", k_fma_then_lshl_add, 2) R("pair fma+fma", k_fma_then_fma, 2)
  R("pair fma+mov_dpp", k_fma_dpp, 2)
  const int wl[] = {1, 2, 4};
  for (int w : wl) run("ds_read_b32 random (+1 mad per read)", k_lds<0>, 1, w, 256, 131072);
  for (int w : wl) run("ds_read_u16 random (+1 mad per read)", k_lds<1>, 1, w, 256, 131072);
  for (int w : wl) run("ds_read_b32 conflict-free (+1 mad per read)", k_lds<2>, 1, w, 256, 131072);
  for (int w : wl) run("ds_read_b64 random (+1 mad per read)", k_lds<3>, 1, w, 256, 131072);
  return 0;
}
