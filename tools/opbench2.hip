// opbench2.hip - issue cost of single gfx950 VALU instructions (ns and cycles per wave64 instruction
// per SIMD), each measured alone through inline asm: 8 independent chains per lane, 16 waves per CU.
// Output feeds DESIGN.md section 4 (which ops are "full rate" and which are not).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

#define ASMK(NAME, TEXT)                                                                    \
  __global__ void NAME(unsigned* out, unsigned a, unsigned b, int iters) {                  \
    unsigned x[8];                                                                          \
    for (int k = 0; k < 8; ++k) x[k] = threadIdx.x * 977u + k * 131u + 0x3f800000u;         \
    for (int i = 0; i < iters; ++i) {                                                       \
      _Pragma("unroll") for (int k = 0; k < 8; ++k)                                         \
        asm volatile(TEXT : "+v"(x[k]) : "v"(a), "v"(b));                                   \
    }                                                                                       \
    unsigned s = x[0];                                                                      \
    for (int k = 1; k < 8; ++k) s ^= x[k];                                                  \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                         \
  }

ASMK(k_fma, "v_fma_f32 %0, %0, %1, %2")
ASMK(k_fmac, "v_fmac_f32 %0, %1, %2")
ASMK(k_mul, "v_mul_f32 %0, %0, %1")
ASMK(k_add, "v_add_f32 %0, %0, %1")
ASMK(k_add_clamp, "v_add_f32_e64 %0, %0, %1 clamp")
ASMK(k_max, "v_max_f32 %0, %0, %1")
ASMK(k_max_clamp, "v_max_f32_e64 %0, %0, %0 clamp")
ASMK(k_med3, "v_med3_f32 %0, %0, %1, %2")
ASMK(k_rndne, "v_rndne_f32 %0, %0")
ASMK(k_floor, "v_floor_f32 %0, %0")
ASMK(k_cvt_f32_u32, "v_cvt_f32_u32 %0, %0")
ASMK(k_cvt_u32_f32, "v_cvt_u32_f32 %0, %0")
ASMK(k_cvt_ubyte0, "v_cvt_f32_ubyte0 %0, %0")
ASMK(k_addu, "v_add_u32 %0, %0, %1")
ASMK(k_subu, "v_sub_u32 %0, %0, %1")
ASMK(k_add3, "v_add3_u32 %0, %0, %1, %2")
ASMK(k_lshl_add, "v_lshl_add_u32 %0, %0, 2, %1")
ASMK(k_lshl_or, "v_lshl_or_b32 %0, %0, 10, %1")
ASMK(k_and_or, "v_and_or_b32 %0, %0, %1, %2")
ASMK(k_lshr, "v_lshrrev_b32 %0, 1, %0")
ASMK(k_lshl, "v_lshlrev_b32 %0, 1, %0")
ASMK(k_and, "v_and_b32 %0, %0, %1")
ASMK(k_or, "v_or_b32 %0, %0, %1")
ASMK(k_xor, "v_xor_b32 %0, %0, %1")
ASMK(k_bfe, "v_bfe_u32 %0, %0, 3, 10")
ASMK(k_bfi, "v_bfi_b32 %0, %1, %0, %2")
ASMK(k_perm, "v_perm_b32 %0, %0, %1, %2")
ASMK(k_alignbit, "v_alignbit_b32 %0, %0, %1, 7")
ASMK(k_mad24, "v_mad_u32_u24 %0, %0, %1, %2")
ASMK(k_mul24, "v_mul_u32_u24 %0, %0, %1")
ASMK(k_mullo, "v_mul_lo_u32 %0, %0, %1")
ASMK(k_minu, "v_min_u32 %0, %0, %1")
ASMK(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
ASMK(k_mov, "v_mov_b32 %0, %1")
ASMK(k_ldexp, "v_ldexp_f32 %0, %0, %1")
ASMK(k_frexp_exp, "v_frexp_exp_i32_f32 %0, %0")
ASMK(k_cvt_pknorm, "v_cvt_pknorm_u16_f32 %0, %0, %1")
ASMK(k_cvt_pk_u16_u32, "v_cvt_pk_u16_u32 %0, %0, %1")
ASMK(k_exp, "v_exp_f32 %0, %0")
ASMK(k_log, "v_log_f32 %0, %0")
ASMK(k_rcp, "v_rcp_f32 %0, %0")

// packed f32: 2-register operands
__global__ void k_pk(unsigned* out, unsigned a, unsigned b, int iters) {
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 x[8], aa = {__uint_as_float(a), __uint_as_float(a)}, bb = {__uint_as_float(b), __uint_as_float(b)};
  for (int k = 0; k < 8; ++k) x[k] = f2{(float)(threadIdx.x + k), (float)k};
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 8; ++k) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x[k]) : "v"(aa), "v"(bb));
  }
  float s = 0;
  for (int k = 0; k < 8; ++k) s += x[k].x + x[k].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = __float_as_uint(s);
}

template <typename K>
static void run(const char* name, K kern, unsigned a, unsigned b, double clock_ghz) {
  const int blocks = 256 * 4, thr = 1024, iters = 2048;
  unsigned* out; CK(hipMalloc(&out, (size_t)blocks * thr * sizeof(unsigned)));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  kern<<<blocks, thr>>>(out, a, b, iters); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0)); for (int r = 0; r < 5; ++r) kern<<<blocks, thr>>>(out, a, b, iters); CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
  const double wave_instrs = (double)blocks * thr / 64 * iters * 8;
  const double ns = ms * 1e6 / (wave_instrs / 1024);
  printf("{\"instr\":\"%s\",\"ns_per_wave_instr_per_simd\":%.3f,\"cycles_at_%.1fGHz\":%.2f}\n", name, ns, clock_ghz, ns * clock_ghz);
  CK(hipFree(out));
}

int main() {
  int khz = 0; CK(hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0));
  const double ghz = khz / 1e6;
  const unsigned one = 0x3f800347u, half = 0x3f000000u;
#define R(n, k) run(n, k, one, half, ghz)
  R("v_fma_f32", k_fma); R("v_fmac_f32", k_fmac); R("v_mul_f32", k_mul); R("v_add_f32", k_add);
  R("v_add_f32 clamp", k_add_clamp); R("v_max_f32", k_max); R("v_max_f32 clamp", k_max_clamp); R("v_med3_f32", k_med3);
  R("v_rndne_f32", k_rndne); R("v_floor_f32", k_floor); R("v_cvt_f32_u32", k_cvt_f32_u32); R("v_cvt_u32_f32", k_cvt_u32_f32);
  R("v_cvt_f32_ubyte0", k_cvt_ubyte0); R("v_pk_fma_f32", k_pk);
  R("v_add_u32", k_addu); R("v_sub_u32", k_subu); R("v_add3_u32", k_add3); R("v_lshl_add_u32", k_lshl_add);
  R("v_lshl_or_b32", k_lshl_or); R("v_and_or_b32", k_and_or); R("v_lshrrev_b32", k_lshr); R("v_lshlrev_b32", k_lshl);
  R("v_and_b32", k_and); R("v_or_b32", k_or); R("v_xor_b32", k_xor); R("v_bfe_u32", k_bfe); R("v_bfi_b32", k_bfi);
  R("v_perm_b32", k_perm); R("v_alignbit_b32", k_alignbit); R("v_mad_u32_u24", k_mad24); R("v_mul_u32_u24", k_mul24);
  R("v_mul_lo_u32", k_mullo); R("v_min_u32", k_minu); R("v_cndmask_b32", k_cndmask); R("v_mov_b32", k_mov);
  R("v_ldexp_f32", k_ldexp); R("v_frexp_exp_i32_f32", k_frexp_exp); R("v_cvt_pknorm_u16_f32", k_cvt_pknorm);
  R("v_cvt_pk_u16_u32", k_cvt_pk_u16_u32); R("v_exp_f32", k_exp); R("v_log_f32", k_log); R("v_rcp_f32", k_rcp);
  return 0;
}
