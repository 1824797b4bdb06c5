// microbench.hip - design probes for the LUT problem (DESIGN.md "LUT placement"): how fast can
// gfx950 do data-dependent 4-byte lookups from (a) a 256 KiB table in global memory (L1/L2
// served) and (b) tables resident in LDS, next to the HBM streaming ceiling.
// Build: hipcc --offload-arch=gfx950 -O3 tools/microbench.hip -o tools/microbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x;
}

// ---- HBM streaming ---------------------------------------------------------------------------
__global__ void copy16(const uint4* __restrict__ in, uint4* __restrict__ out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i];
}
__global__ void read16(const uint4* __restrict__ in, uint32_t* __restrict__ out, size_t n) {
  uint32_t acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    uint4 v = in[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x12345678) out[0] = acc;
}
// 4 streams in, 1 out (the fused kernel's access shape)
__global__ void copy4to1(const uint4* __restrict__ a, const uint4* __restrict__ b, const uint4* __restrict__ c,
                         const uint4* __restrict__ d, uint4* __restrict__ out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { uint4 x = a[i], y = b[i], z = c[i], w = d[i];
    out[i] = make_uint4(x.x ^ y.x ^ z.x ^ w.x, x.y ^ y.y ^ z.y ^ w.y, x.z ^ y.z ^ z.z ^ w.z, x.w ^ y.w ^ z.w ^ w.w); }
}

// ---- global gathers ---------------------------------------------------------------------------
// MODE 0: uniformly random index per lane; MODE 1: "image-like": lanes of a wave share a base
// index and differ by a small jitter (neighbouring pixels are similar).
template <int MODE, int PER>
__global__ void gather_global(const float* __restrict__ lut, float* __restrict__ out, uint32_t seed) {
  uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t s = mix32(tid ^ seed);
  uint32_t wave_base = mix32((tid >> 6) ^ seed) & 0xffff;
  float acc = 0.f;
#pragma unroll 4
  for (int i = 0; i < PER; ++i) {
    s = s * 1664525u + 1013904223u;
    uint32_t idx = MODE == 0 ? (s >> 16) : ((wave_base + ((s >> 24) & 0xff) + i * 37) & 0xffff);
    acc += lut[idx];
  }
  out[tid] = acc;
}

// ---- LDS gathers -------------------------------------------------------------------------------
// 64K-entry table of T resident in LDS (T = uint16_t: 128 KiB; uint8_t: 64 KiB), random index.
template <typename T, int ENTRIES, int MODE, int PER>
__global__ __launch_bounds__(1024) void gather_lds(const T* __restrict__ table, float* __restrict__ out, uint32_t seed, int rounds) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  T* t = reinterpret_cast<T*>(smem);
  for (int i = threadIdx.x; i < ENTRIES * (int)sizeof(T) / 16; i += blockDim.x)
    reinterpret_cast<uint4*>(smem)[i] = reinterpret_cast<const uint4*>(table)[i];
  __syncthreads();
  uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t acc = 0;
  for (int r = 0; r < rounds; ++r) {
    uint32_t s = mix32(tid ^ seed ^ (r * 0x9e3779b9u));
    uint32_t wave_base = mix32((tid >> 6) ^ seed ^ r) & (ENTRIES - 1);
#pragma unroll 8
    for (int i = 0; i < PER; ++i) {
      s = s * 1664525u + 1013904223u;
      uint32_t idx = MODE == 0 ? (s >> 16) & (ENTRIES - 1) : ((wave_base + ((s >> 24) & 0xff) + i * 37) & (ENTRIES - 1));
      acc += t[idx];
    }
  }
  out[tid] = (float)acc;
}

// ---- VALU reference: dependent-free fma streams -----------------------------------------------------
template <int PER>
__global__ void valu_fma(float* __restrict__ out, float a, float b) {
  uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  float x0 = tid, x1 = tid + 1, x2 = tid + 2, x3 = tid + 3, x4 = tid + 4, x5 = tid + 5, x6 = tid + 6, x7 = tid + 7;
  for (int i = 0; i < PER; ++i) {
    x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, a, b); x2 = __builtin_fmaf(x2, a, b); x3 = __builtin_fmaf(x3, a, b);
    x4 = __builtin_fmaf(x4, a, b); x5 = __builtin_fmaf(x5, a, b); x6 = __builtin_fmaf(x6, a, b); x7 = __builtin_fmaf(x7, a, b);
  }
  out[tid] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}

typedef float float2v __attribute__((ext_vector_type(2)));
template <int PER>
__global__ void valu_pk_fma(float* __restrict__ out, float a, float b) {
  uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  float2v x0 = {(float)tid, tid + 1.f}, x1 = {tid + 2.f, tid + 3.f}, x2 = {tid + 4.f, tid + 5.f}, x3 = {tid + 6.f, tid + 7.f};
  float2v x4 = x0 + 8.f, x5 = x1 + 8.f, x6 = x2 + 8.f, x7 = x3 + 8.f;
  const float2v av = {a, a}, bv = {b, b};
  for (int i = 0; i < PER; ++i) {
    x0 = __builtin_elementwise_fma(x0, av, bv); x1 = __builtin_elementwise_fma(x1, av, bv);
    x2 = __builtin_elementwise_fma(x2, av, bv); x3 = __builtin_elementwise_fma(x3, av, bv);
    x4 = __builtin_elementwise_fma(x4, av, bv); x5 = __builtin_elementwise_fma(x5, av, bv);
    x6 = __builtin_elementwise_fma(x6, av, bv); x7 = __builtin_elementwise_fma(x7, av, bv);
  }
  float2v s = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
  out[tid] = s.x + s.y;
}
// the index chain of one LUT lookup: mul, rndne, max, min, cvt (5 dependent-free streams)
template <int PER>
__global__ void valu_index_chain(uint32_t* __restrict__ out, float a) {
  uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  float x0 = tid * 1e-3f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
  uint32_t acc = 0;
  for (int i = 0; i < PER; ++i) {
    x0 += a; x1 += a; x2 += a; x3 += a;
    acc += (uint32_t)__builtin_fminf(__builtin_fmaxf(__builtin_rintf(x0 * 65535.0f), 0.f), 65535.f);
    acc += (uint32_t)__builtin_fminf(__builtin_fmaxf(__builtin_rintf(x1 * 65535.0f), 0.f), 65535.f);
    acc += (uint32_t)__builtin_fminf(__builtin_fmaxf(__builtin_rintf(x2 * 65535.0f), 0.f), 65535.f);
    acc += (uint32_t)__builtin_fminf(__builtin_fmaxf(__builtin_rintf(x3 * 65535.0f), 0.f), 65535.f);
  }
  out[tid] = acc;
}
template <int PER>
__global__ void valu_transc(float* __restrict__ out, float a) {
  uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  float x0 = 0.2f + tid * 1e-9f, x1 = x0 + 0.1f, x2 = x0 + 0.2f, x3 = x0 + 0.3f;
  for (int i = 0; i < PER; ++i) {
    x0 = __builtin_amdgcn_exp2f(a * __builtin_amdgcn_logf(x0)); x1 = __builtin_amdgcn_exp2f(a * __builtin_amdgcn_logf(x1));
    x2 = __builtin_amdgcn_exp2f(a * __builtin_amdgcn_logf(x2)); x3 = __builtin_amdgcn_exp2f(a * __builtin_amdgcn_logf(x3));
  }
  out[tid] = x0 + x1 + x2 + x3;
}
template <int PER>
__global__ void bpermute_rate(uint32_t* __restrict__ out, uint32_t seed) {
  uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t s = mix32(tid ^ seed), v = tid, acc = 0;
  for (int i = 0; i < PER; ++i) {
    s = s * 1664525u + 1013904223u;
    acc += __builtin_amdgcn_ds_bpermute((s >> 24) & 0xfc, v + i);
  }
  out[tid] = acc;
}

template <typename F> static float time_ms(F f, int reps = 10) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0)); for (int i = 0; i < reps; ++i) f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / reps;
}

int main() {
  const size_t N = (size_t)64 << 20;  // 64M uint4 = 1 GiB
  uint4 *a, *b; CK(hipMalloc(&a, N * 16)); CK(hipMalloc(&b, N * 16));
  CK(hipMemset(a, 1, N * 16)); CK(hipMemset(b, 2, N * 16));
  float *lut, *out; CK(hipMalloc(&lut, 65536 * 4)); CK(hipMalloc(&out, (size_t)64 << 20));
  std::vector<float> h(65536); for (int i = 0; i < 65536; ++i) h[i] = i * 1e-5f;
  CK(hipMemcpy(lut, h.data(), 65536 * 4, hipMemcpyHostToDevice));
  uint16_t* t16; CK(hipMalloc(&t16, 65536 * 2)); CK(hipMemset(t16, 3, 65536 * 2));

  float ms = time_ms([&] { copy16<<<2048, 256>>>(a, b, N); });
  printf("{\"probe\":\"copy16\",\"GBps\":%.1f}\n", 2.0 * N * 16 / ms / 1e6);
  ms = time_ms([&] { read16<<<2048, 256>>>(a, (uint32_t*)out, N); });
  printf("{\"probe\":\"read16\",\"GBps\":%.1f}\n", 1.0 * N * 16 / ms / 1e6);
  { size_t n = N / 4;  // four quarter-GiB inputs -> one output
    ms = time_ms([&] { copy4to1<<<(unsigned)((n + 255) / 256), 256>>>(a, a + n, a + 2 * n, a + 3 * n, b, n); });
    printf("{\"probe\":\"copy4to1\",\"GBps\":%.1f}\n", 5.0 * n * 16 / ms / 1e6); }

  const int blocks = 256 * 16, thr = 256; const double lanes = (double)blocks * thr;
  ms = time_ms([&] { gather_global<0, 256><<<blocks, thr>>>(lut, out, 1); });
  printf("{\"probe\":\"global_gather_random\",\"Glookups_per_s\":%.1f}\n", lanes * 256 / ms / 1e6);
  ms = time_ms([&] { gather_global<1, 256><<<blocks, thr>>>(lut, out, 1); });
  printf("{\"probe\":\"global_gather_imagelike\",\"Glookups_per_s\":%.1f}\n", lanes * 256 / ms / 1e6);

  { const int lb = 256 * 4, lt = 1024, rounds = 8; const double ll = (double)lb * lt * rounds;
    CK(hipFuncSetAttribute((const void*)gather_lds<uint16_t, 65536, 0, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    CK(hipFuncSetAttribute((const void*)gather_lds<uint16_t, 65536, 1, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    CK(hipFuncSetAttribute((const void*)gather_lds<float, 32768, 0, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    CK(hipFuncSetAttribute((const void*)gather_lds<uint8_t, 65536, 0, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    ms = time_ms([&] { gather_lds<uint16_t, 65536, 0, 64><<<lb, lt, 131072>>>(t16, out, 1, rounds); });
    printf("{\"probe\":\"lds_gather_u16_random_128K\",\"Glookups_per_s\":%.1f}\n", ll * 64 / ms / 1e6);
    ms = time_ms([&] { gather_lds<uint16_t, 65536, 1, 64><<<lb, lt, 131072>>>(t16, out, 1, rounds); });
    printf("{\"probe\":\"lds_gather_u16_imagelike_128K\",\"Glookups_per_s\":%.1f}\n", ll * 64 / ms / 1e6);
    ms = time_ms([&] { gather_lds<float, 32768, 0, 64><<<lb, lt, 131072>>>(lut, out, 1, rounds); });
    printf("{\"probe\":\"lds_gather_f32_random_128K\",\"Glookups_per_s\":%.1f}\n", ll * 64 / ms / 1e6);
    ms = time_ms([&] { gather_lds<uint8_t, 65536, 0, 64><<<lb, lt, 65536>>>((const uint8_t*)t16, out, 1, rounds); });
    printf("{\"probe\":\"lds_gather_u8_random_64K\",\"Glookups_per_s\":%.1f}\n", ll * 64 / ms / 1e6);
  }
  ms = time_ms([&] { valu_fma<512><<<blocks, thr>>>(out, 1.0001f, 0.5f); });
  printf("{\"probe\":\"valu_fma\",\"Gops_per_s\":%.1f}\n", lanes * 512 * 8 / ms / 1e6);
  ms = time_ms([&] { valu_pk_fma<512><<<blocks, thr>>>(out, 1.0001f, 0.5f); });
  printf("{\"probe\":\"valu_pk_fma (2 fma per instr)\",\"Gfma_per_s\":%.1f}\n", lanes * 512 * 16 / ms / 1e6);
  ms = time_ms([&] { valu_index_chain<256><<<blocks, thr>>>((uint32_t*)out, 1e-4f); });
  printf("{\"probe\":\"valu_index_chain (add,mul,rndne,max,min,cvt,iadd = 7 ops)\",\"Gchains_per_s\":%.1f}\n", lanes * 256 * 4 / ms / 1e6);
  ms = time_ms([&] { valu_transc<256><<<blocks, thr>>>(out, 0.45f); });
  printf("{\"probe\":\"valu_pow (log2,mul,exp2)\",\"Gpow_per_s\":%.1f}\n", lanes * 256 * 4 / ms / 1e6);
  ms = time_ms([&] { bpermute_rate<256><<<blocks, thr>>>((uint32_t*)out, 3); });
  printf("{\"probe\":\"ds_bpermute\",\"Gops_per_s\":%.1f}\n", lanes * 256 / ms / 1e6);
  return 0;
}
