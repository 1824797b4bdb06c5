// mfma_probe.hip - v_mfma_f32_4x4x1_16b_f32 on gfx950: what it computes and what it costs.
//
// Part 1 (semantics).  16 blocks of a (4x1)*(1x4) outer product: lane l supplies A[block l/4][row l%4] and
// B[block l/4][col l%4]; result VGPR v of lane l is row v, column l%4 of block l/4:
//     D[v][l] = A[4*(l/4) + v] * B[l] + C[v][l]
// so with A = M[l%4][k] (a per-lane constant) and B = the lane's own pixel component, VGPR v is row v of the matrix
// product for the lane's own pixel - pixel-per-lane, no shuffles.  The probe checks that layout and whether ONE step
// equals fmaf(a, b, c) bit for bit: seeded random operands over the whole f32 range, denormal operands / products /
// sums, signed zeros, infinities and NaN, and the products the CSC matrices really see (code values x coefficients).
// Part 2 (price).  s_memtime / s_memrealtime around loops of independent MFMA chains, of v_fma_f32, and of both
// interleaved in one wave, at 4 waves per SIMD with all 256 CUs busy - the fused kernels' geometry.
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_probe.hip -o tools/mfma_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

typedef float f4v __attribute__((ext_vector_type(4)));

// one step per wave: d = mfma(a, b, c)
__global__ void k_step(const float *a, const float *b, const float *c, float *d, int n_waves) {
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (wave >= n_waves) return;
  const size_t o = (size_t)wave * 64 + lane;
  f4v acc = {c[o * 4 + 0], c[o * 4 + 1], c[o * 4 + 2], c[o * 4 + 3]};
  acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[o], b[o], acc, 0, 0, 0);
  d[o * 4 + 0] = acc[0], d[o * 4 + 1] = acc[1], d[o * 4 + 2] = acc[2], d[o * 4 + 3] = acc[3];
}
// a chain of four steps as a 3x4 matrix row would use it: x[k] per lane, m[k] per lane (row l%4), started from C = c0
__global__ void k_chain(const float *m, const float *x, float *d, int n_waves, float c0) {
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (wave >= n_waves) return;
  const size_t o = (size_t)wave * 64 + lane;
  f4v acc = {c0, c0, c0, c0};
#pragma unroll
  for (int k = 0; k < 4; ++k) acc = __builtin_amdgcn_mfma_f32_4x4x1f32(m[o * 4 + k], x[o * 4 + k], acc, 0, 0, 0);
  d[o * 4 + 0] = acc[0], d[o * 4 + 1] = acc[1], d[o * 4 + 2] = acc[2], d[o * 4 + 3] = acc[3];
}

static uint64_t g_s = 0x5EED0006ull;
static uint64_t sm64() { uint64_t z = (g_s += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
static float bits_f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static uint32_t f_bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static bool same(float x, float y) { return f_bits(x) == f_bits(y) || (std::isnan(x) && std::isnan(y)); }

// operand classes: 0 any bit pattern, 1 denormal, 2 tiny normal, 3 moderate, 4 code value 0..1023, 5 unit range, 6 special
static float gen(int cls) {
  const uint64_t r = sm64();
  switch (cls) {
    case 0: return bits_f((uint32_t)r);
    case 1: return bits_f(((uint32_t)r & 0x807FFFFFu));
    case 2: return bits_f(((uint32_t)r & 0x80FFFFFFu) | 0x00800000u * (1 + (r >> 40) % 3));
    case 3: return bits_f(((uint32_t)r & 0x807FFFFFu) | ((uint32_t)(100 + (r >> 40) % 56) << 23));
    case 4: return (float)((r >> 20) % 1024);
    case 5: return (float)((r >> 11) * (1.0 / 9007199254740992.0));
    default: {
      static const uint32_t sp[] = {0u, 0x80000000u, 0x7F800000u, 0xFF800000u, 0x7FC00000u, 0x3F800000u, 0xBF800000u, 0x00000001u, 0x80000001u, 0x007FFFFFu, 0x00800000u, 0x7F7FFFFFu};
      return bits_f(sp[r % 12]);
    }
  }
}

struct Stamp { unsigned long long cyc, real; };
#define NCH 6
// MODE 0: NCH independent MFMA chains; 1: NV v_fma_f32 per group; 2: both interleaved (NCH MFMA + NV VALU per group)
template <int MODE, int NV>
__global__ __launch_bounds__(1024) void k_rate(Stamp *st, float *out, float a, float b, int iters) {
  f4v acc[NCH];
  float x[8];
  for (int k = 0; k < NCH; ++k) acc[k] = f4v{(float)threadIdx.x, 1.f, 2.f, (float)k};
  for (int k = 0; k < 8; ++k) x[k] = (float)(threadIdx.x + k);
  float va = a + (threadIdx.x & 1), vb = b;
  asm volatile("" : "+v"(va), "+v"(vb));
  __syncthreads();
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (MODE == 0) {
#pragma unroll
        for (int k = 0; k < NCH; ++k) asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+v"(acc[k]) : "v"(va), "v"(vb));
      } else if (MODE == 1) {
#pragma unroll
        for (int k = 0; k < NV; ++k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[k & 7]) : "v"(va), "v"(vb));
      } else if (MODE == 3) {
        // the NCH MFMAs back to back, then the NV VALU
#pragma unroll
        for (int k = 0; k < NCH; ++k) asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+v"(acc[k]) : "v"(va), "v"(vb));
#pragma unroll
        for (int k = 0; k < NV; ++k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[k & 7]) : "v"(va), "v"(vb));
      } else if (MODE == 4) {
        // pairs of MFMAs with the VALU between the pairs
#pragma unroll
        for (int k = 0; k < NCH; k += 2) {
          asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+v"(acc[k]) : "v"(va), "v"(vb));
          asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+v"(acc[k + 1]) : "v"(va), "v"(vb));
#pragma unroll
          for (int v = 0; v < NV / (NCH / 2); ++v) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[v & 7]) : "v"(va), "v"(vb));
        }
      } else if (MODE == 5) {
        // like 2, but the VALU are the slow class (v_med3_f32), which leaves the fast VALU half idle
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
          asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+v"(acc[k]) : "v"(va), "v"(vb));
#pragma unroll
          for (int v = 0; v < NV / NCH; ++v) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(x[(k * (NV / NCH) + v) & 7]) : "v"(va), "v"(vb));
        }
      } else {
        // NV VALU spread evenly between the NCH MFMAs
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
          asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+v"(acc[k]) : "v"(va), "v"(vb));
#pragma unroll
          for (int v = 0; v < NV / NCH; ++v) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[(k * (NV / NCH) + v) & 7]) : "v"(va), "v"(vb));
        }
      }
    }
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  float s = 0.f;
  for (int k = 0; k < NCH; ++k) s += acc[k][0] + acc[k][1] + acc[k][2] + acc[k][3];
  for (int k = 0; k < 8; ++k) s += x[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) st[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = Stamp{c1 - c0, r1 - r0};
}

template <int MODE, int NV>
static void rate(const char *name, int threads, int n_mfma, int n_valu) {
  const int blocks = 256, iters = 2000;
  Stamp *st;
  float *out;
  CK(hipMalloc(&st, sizeof(Stamp) * blocks * 16));
  CK(hipMalloc(&out, 4 * blocks * 1024));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k_rate<MODE, NV>), dim3(blocks), dim3(threads), 0, 0, st, out, 1.0f, 0.5f, 10);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((k_rate<MODE, NV>), dim3(blocks), dim3(threads), 0, 0, st, out, 1.0f, 0.5f, iters);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<Stamp> h(blocks * threads / 64);
  CK(hipMemcpy(h.data(), st, sizeof(Stamp) * h.size(), hipMemcpyDeviceToHost));
  double cyc = 0, real = 0;
  for (auto &s : h) cyc += s.cyc, real += s.real;
  cyc /= h.size(), real /= h.size();
  const int waves_per_simd = threads / 256;
  const double groups = (double)iters * 4 * waves_per_simd;  // groups executed per SIMD
  printf("{\"probe\": \"rate\", \"name\": \"%s\", \"waves_per_simd\": %d, \"mfma_per_group\": %d, \"valu_per_group\": %d, \"cycles_per_group_per_simd\": %.2f, "
         "\"ghz\": %.3f, \"ns_per_group_per_simd\": %.3f}\n",
         name, waves_per_simd, n_mfma, n_valu, cyc / groups, cyc / real * 0.1, ms * 1e6 / groups);
  CK(hipFree(st)); CK(hipFree(out));
}

int main() {
  // ---- part 1: semantics
  const int n_waves = 1 << 14;
  const size_t n = (size_t)n_waves * 64;
  std::vector<float> a(n), b(n), c(n * 4), d(n * 4);
  float *da, *db, *dc, *dd;
  CK(hipMalloc(&da, n * 4)); CK(hipMalloc(&db, n * 4)); CK(hipMalloc(&dc, n * 16)); CK(hipMalloc(&dd, n * 16));
  // class triples (a, b, c); the last rows are the CSC products: coefficient x code value + unit-range, unit x unit + unit
  const int combos[][3] = {{0, 0, 0}, {1, 3, 1}, {1, 1, 1}, {2, 5, 2}, {2, 2, 1}, {3, 3, 3}, {6, 6, 6}, {6, 0, 0}, {0, 6, 6}, {3, 4, 5}, {5, 5, 5}, {3, 5, 5}, {1, 4, 5}, {2, 4, 1}};
  const int n_combos = sizeof(combos) / sizeof(combos[0]);
  long long total = 0, bad_layout = 0, bad_bits = 0, denorm_in = 0, denorm_out = 0, bad_denorm = 0;
  for (int pass = 0; pass < n_combos; ++pass) {
    for (size_t i = 0; i < n; ++i) {
      a[i] = gen(combos[pass][0]), b[i] = gen(combos[pass][1]);
      for (int v = 0; v < 4; ++v) c[i * 4 + v] = gen(combos[pass][2]);
    }
    CK(hipMemcpy(da, a.data(), n * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(db, b.data(), n * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dc, c.data(), n * 16, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_step, dim3(n_waves / 4), dim3(256), 0, 0, da, db, dc, dd, n_waves);
    CK(hipMemcpy(d.data(), dd, n * 16, hipMemcpyDeviceToHost));
    long long bad_here = 0;
    for (size_t w = 0; w < (size_t)n_waves; ++w)
      for (int l = 0; l < 64; ++l)
        for (int v = 0; v < 4; ++v) {
          const size_t o = w * 64 + l;
          const float av = a[w * 64 + 4 * (l / 4) + v], bv = b[o], cv = c[o * 4 + v];
          const float want = fmaf(av, bv, cv), got = d[o * 4 + v];
          ++total;
          const bool den = (std::fpclassify(av) == FP_SUBNORMAL) || (std::fpclassify(bv) == FP_SUBNORMAL) || (std::fpclassify(cv) == FP_SUBNORMAL);
          const bool den_o = std::fpclassify(want) == FP_SUBNORMAL;
          denorm_in += den, denorm_out += den_o;
          if (!same(want, got)) {
            ++bad_bits, ++bad_here;
            if (den || den_o) ++bad_denorm;
            // is it at least the right lanes?  (a wrong layout would be wrong almost everywhere)
            if (bad_here <= 3) printf("{\"probe\": \"mismatch\", \"pass\": %d, \"a\": \"%08x\", \"b\": \"%08x\", \"c\": \"%08x\", \"want\": \"%08x\", \"got\": \"%08x\"}\n", pass, f_bits(av), f_bits(bv), f_bits(cv), f_bits(want), f_bits(got));
          }
        }
    if (pass == 5 && bad_here > (long long)n * 2) bad_layout = bad_here;  // moderate normals: any mismatch there is layout or arithmetic, not denormals
    printf("{\"probe\": \"step\", \"classes\": [%d, %d, %d], \"checked\": %lld, \"mismatches\": %lld}\n", combos[pass][0], combos[pass][1], combos[pass][2], (long long)n * 4, bad_here);
  }
  printf("{\"probe\": \"step_total\", \"checked\": %lld, \"mismatches\": %lld, \"with_denormal_operand\": %lld, \"with_denormal_result\": %lld, \"mismatches_involving_denormals\": %lld, \"layout_ok\": %s}\n",
         total, bad_bits, denorm_in, denorm_out, bad_denorm, bad_layout ? "false" : "true");
  // chain of four against AMD's dot(): fma(a3,b3,fma(a2,b2,fma(a1,b1,a0*b0))).  Started from C = +0 the first step is
  // a0*b0 + (+0), which turns a product of -0 into +0; started from C = -0 it is a0*b0 for every product.
  for (int start = 0; start < 2; ++start) {
    std::vector<float> m(n * 4), x(n * 4);
    long long bad = 0, bad_sign_only = 0;
    for (int pass = 0; pass < 4; ++pass) {
      for (size_t i = 0; i < n * 4; ++i) {
        m[i] = pass == 0 ? gen(3) : pass == 1 ? gen(5) * 0.01f - 0.003f : pass == 2 ? gen(0) : ((sm64() & 3) ? gen(3) : 0.0f * (sm64() & 1 ? -1.f : 1.f));
        x[i] = pass == 0 ? gen(3) : pass == 1 ? gen(4) : pass == 2 ? gen(0) : ((sm64() & 3) ? gen(4) : 0.0f);
      }
      CK(hipMemcpy(dc, m.data(), n * 16, hipMemcpyHostToDevice)); CK(hipMemcpy(dd, x.data(), n * 16, hipMemcpyHostToDevice));
      float *dout;
      CK(hipMalloc(&dout, n * 16));
      hipLaunchKernelGGL(k_chain, dim3(n_waves / 4), dim3(256), 0, 0, dc, dd, dout, n_waves, start ? -0.0f : 0.0f);
      CK(hipMemcpy(d.data(), dout, n * 16, hipMemcpyDeviceToHost));
      CK(hipFree(dout));
      for (size_t w = 0; w < (size_t)n_waves; ++w)
        for (int l = 0; l < 64; ++l)
          for (int v = 0; v < 4; ++v) {
            const size_t o = w * 64 + l, ao = w * 64 + 4 * (l / 4) + v;
            float want = m[ao * 4 + 0] * x[o * 4 + 0];
            for (int k = 1; k < 4; ++k) want = fmaf(m[ao * 4 + k], x[o * 4 + k], want);
            if (!same(want, d[o * 4 + v])) {
              ++bad;
              if (want == 0.0f && d[o * 4 + v] == 0.0f) ++bad_sign_only;
            }
          }
    }
    printf("{\"probe\": \"chain4\", \"start\": \"%s\", \"checked\": %lld, \"mismatches\": %lld, \"of_which_sign_of_zero_only\": %lld}\n", start ? "-0" : "+0", (long long)n * 16, bad, bad_sign_only);
  }
  // ---- part 2: price
  rate<0, 0>("mfma_only_6chains", 1024, NCH, 0);
  rate<0, 0>("mfma_only_6chains_2waves", 512, NCH, 0);
  rate<0, 0>("mfma_only_6chains_1wave", 256, NCH, 0);
  rate<1, 24>("valu_only_24", 1024, 0, 24);
  rate<1, 42>("valu_only_42", 1024, 0, 42);
  rate<2, 12>("mixed_6mfma_12valu", 1024, NCH, 12);
  rate<2, 24>("mixed_6mfma_24valu", 1024, NCH, 24);
  rate<2, 42>("mixed_6mfma_42valu", 1024, NCH, 42);
  rate<2, 42>("mixed_6mfma_42valu_2waves", 512, NCH, 42);
  rate<3, 12>("grouped_6mfma_then_12valu", 1024, NCH, 12);
  rate<3, 24>("grouped_6mfma_then_24valu", 1024, NCH, 24);
  rate<3, 42>("grouped_6mfma_then_42valu", 1024, NCH, 42);
  rate<4, 42>("pairs_6mfma_42valu", 1024, NCH, 42);
  rate<5, 24>("mixed_6mfma_24med3", 1024, NCH, 24);
  rate<3, 42>("grouped_6mfma_then_42valu_2waves", 512, NCH, 42);
  rate<3, 42>("grouped_6mfma_then_42valu_1wave", 256, NCH, 42);
  return 0;
}
