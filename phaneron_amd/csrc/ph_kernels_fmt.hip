// ph_kernels_fmt.hip - the reference's other packed formats:
//   yuv422p10le (yuv422p10.ts), yuv422p8 (yuv422p8.ts), yuv420p (yuv420p.ts), nv12 (nv12.ts),
//   rgba8 (rgba8.ts), bgra8 (bgra8.ts) - read (-> linear f32 RGBA) and write (<- f32 RGBA).
//
// Readers: one PIXEL per lane (coalesced float4 stores; the sample loads of a wave are 64-128
// contiguous bytes).  Writers: one group of 8 pixels per lane (8-16 byte plane stores), the
// 4:2:0 writers handle the line pair of their group.  Every kernel exists in two forms chosen at
// launch: gamma LUT in LDS (persistent 1024-lane workgroups) or plain table in global memory.
#include <cstdlib>

#include "ph_kernels.h"
#include "ph_ldslut.h"

#pragma clang fp contract(off)

namespace ph {

constexpr int kFmtBlock = 256;

// formats (same numbering as PH_FMT_* in include/phaneron_hip.h)
enum { F_V210 = 0, F_YUV422P10 = 1, F_YUV422P8 = 2, F_YUV420P = 3, F_NV12 = 4, F_RGBA8 = 5, F_BGRA8 = 6 };

template <typename LUT>
__device__ __forceinline__ float4 yuv_to_rgba(float y, float u, float v, const ReadK &k, const LUT &lut) {
  const float r = lut.at_unit(dot4(y, u, v, 1.0f, k.r));  // e.g. yuv422p10.ts:74-78
  const float g = lut.at_unit(dot4(y, u, v, 1.0f, k.g));
  const float b = lut.at_unit(dot4(y, u, v, 1.0f, k.b));
  return make_float4(dot3(r, g, b, k.gm[0], k.gm[1], k.gm[2]), dot3(r, g, b, k.gm[3], k.gm[4], k.gm[5]),
                     dot3(r, g, b, k.gm[6], k.gm[7], k.gm[8]), 1.0f);
}

struct FmtReadArgs {
  const void *p0, *p1, *p2;
  float4 *out;
  uint32_t width, lines, pitch;  // pitch in luma samples (RGBA: pixels)
  const float *cm, *gm;
  uint32_t nt;  // stream the image output past the caches (ph_device.h store_image)
};

// FMT is a template parameter: plane layout and sample width are compile-time
template <int FMT, typename LUT, bool PERSISTENT>
__device__ __forceinline__ void fmt_read_body(const FmtReadArgs &a, const LUT &lut, uint32_t block = blockIdx.x, uint32_t blocks = gridDim.x) {
  ReadK k;
  if (FMT >= F_RGBA8) {  // the RGB formats carry no YCbCr matrix: gamut only (9 floats, never more)
#pragma unroll
    for (int i = 0; i < 9; ++i) k.gm[i] = a.gm[i];
  } else {
    k = load_read_k(a.cm, a.gm);
  }
  const uint32_t total = a.width * a.lines;
  const uint32_t stride = PERSISTENT ? blocks * blockDim.x : total;
  for (uint32_t p = block * blockDim.x + threadIdx.x; p < total; p += stride) {
    const uint32_t line = p / a.width, x = p - line * a.width;
    float4 o;
    if (FMT == F_RGBA8 || FMT == F_BGRA8) {  // rgba8.ts:49-62
      const uchar4 px = reinterpret_cast<const uchar4 *>(a.p0)[(size_t)line * a.pitch + x];
      const float rf = (float)(FMT == F_RGBA8 ? px.x : px.z), gf = (float)px.y, bf = (float)(FMT == F_RGBA8 ? px.z : px.x);
      const float r = lut.at(rf * 65535.0f / 255.0f), g = lut.at(gf * 65535.0f / 255.0f), b = lut.at(bf * 65535.0f / 255.0f);
      o = make_float4(dot3(r, g, b, k.gm[0], k.gm[1], k.gm[2]), dot3(r, g, b, k.gm[3], k.gm[4], k.gm[5]),
                      dot3(r, g, b, k.gm[6], k.gm[7], k.gm[8]), lut.at((float)px.w * 65535.0f / 255.0f));
    } else {
      const uint32_t cl = (FMT == F_YUV420P || FMT == F_NV12) ? line >> 1 : line;
      float y, u, v;
      if (FMT == F_YUV422P10) {
        y = (float)reinterpret_cast<const uint16_t *>(a.p0)[(size_t)line * a.pitch + x];
        u = (float)reinterpret_cast<const uint16_t *>(a.p1)[(size_t)cl * (a.pitch >> 1) + (x >> 1)];
        v = (float)reinterpret_cast<const uint16_t *>(a.p2)[(size_t)cl * (a.pitch >> 1) + (x >> 1)];
      } else if (FMT == F_NV12) {  // nv12.ts:61-74
        y = (float)reinterpret_cast<const uint8_t *>(a.p0)[(size_t)line * a.pitch + x];
        const uint8_t *c = reinterpret_cast<const uint8_t *>(a.p1) + (size_t)cl * a.pitch + (x & ~1u);  // the pair's Cb, Cr bytes (two byte loads: a 2-byte vector load measured twice as slow here)
        u = (float)c[0], v = (float)c[1];
      } else {
        y = (float)reinterpret_cast<const uint8_t *>(a.p0)[(size_t)line * a.pitch + x];
        u = (float)reinterpret_cast<const uint8_t *>(a.p1)[(size_t)cl * (a.pitch >> 1) + (x >> 1)];
        v = (float)reinterpret_cast<const uint8_t *>(a.p2)[(size_t)cl * (a.pitch >> 1) + (x >> 1)];
      }
      o = yuv_to_rgba(y, u, v, k, lut);
    }
    store_image(a.out + p, o, a.nt);
  }
}

template <int FMT>
__global__ __launch_bounds__(kLdsBlock) void fmt_read_lds_kernel(FmtReadArgs a, LutView lv) {
  const LutInLds lut{make_lut_k(lv)};
  lds_lut_load(lv);
  __syncthreads();
  fmt_read_body<FMT, LutInLds, true>(a, lut);
}
// several frames of one format, size and Loader recipe in one launch (several channels' clips of a tick: ph_pack_read_batch): the
// workgroups go round the frames (uniform per workgroup: the frame's pointers are scalar loads), the table is loaded once per workgroup as ever
struct FmtReadBatchArgs {
  const void *p0[kMaxLayers], *p1[kMaxLayers], *p2[kMaxLayers];
  float4 *out[kMaxLayers];
  uint32_t jobs;
  uint32_t width, lines, pitch;
  const float *cm, *gm;
  uint32_t nt;
};
template <int FMT>
__global__ __launch_bounds__(kLdsBlock) void fmt_read_lds_batch_kernel(FmtReadBatchArgs b, LutView lv) {
  const LutInLds lut{make_lut_k(lv)};
  lds_lut_load(lv);
  __syncthreads();
  const uint32_t per_job = gridDim.x / b.jobs, job = blockIdx.x / per_job;  // (the launcher makes the grid a multiple of jobs)
  const FmtReadArgs a{b.p0[job], b.p1[job], b.p2[job], b.out[job], b.width, b.lines, b.pitch, b.cm, b.gm, b.nt};
  fmt_read_body<FMT, LutInLds, true>(a, lut, blockIdx.x - job * per_job, per_job);
}
template <int FMT>
__global__ __launch_bounds__(kFmtBlock) void fmt_read_gather_kernel(FmtReadArgs a, const float *__restrict__ table) {
  const LutInGlobal lut{table};
  fmt_read_body<FMT, LutInGlobal, false>(a, lut);
}

// ------------------------------------------------------------------------------------------
// writers
// ------------------------------------------------------------------------------------------
struct FmtWriteArgs {
  const float4 *in;
  void *p0, *p1, *p2;
  uint32_t width, pitch;          // pitch in luma samples
  uint32_t groups;                // work groups of the reference: lines, or line pairs for 4:2:0
  uint32_t interlace;
  const float *cm;
};

template <typename LUT>
__device__ __forceinline__ void px_codes(const float4 px, const WriteK &k, const LUT &lut, bool tail, uint32_t &y,
                                         uint32_t &u, uint32_t &v) {
  const float gr = lut.at_unit(px.x), gg = lut.at_unit(px.y), gb = lut.at_unit(px.z);
  float ty = dot4(gr, gg, gb, 1.0f, k.y), tu = dot4(gr, gg, gb, 1.0f, k.u), tv = dot4(gr, gg, gb, 1.0f, k.v);
  if (tail) ty = __builtin_roundf(ty), tu = __builtin_roundf(tu), tv = __builtin_roundf(tv);  // :186-188
  y = sat_u16_rte(ty), u = sat_u16_rte(tu), v = sat_u16_rte(tv);
}

template <int FMT, typename LUT, bool PERSISTENT>
__device__ __forceinline__ void fmt_write_body(const FmtWriteArgs &a, const LUT &lut) {
  if (FMT == F_RGBA8 || FMT == F_BGRA8) {  // rgba8.ts:69-101: one pixel per lane
    const uint32_t total = a.width * a.groups;
    const uint32_t stride = PERSISTENT ? gridDim.x * blockDim.x : total;
    for (uint32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < total; p += stride) {
      const uint32_t g = p / a.width, x = p - g * a.width;
      const uint32_t line = g * (a.interlace ? 2 : 1) + ((3 == a.interlace) ? 1 : 0);
      const float4 px = a.in[(size_t)line * a.width + x];
      const float r = lut.at_unit(px.x), gg = lut.at_unit(px.y), b = lut.at_unit(px.z);
      const uint32_t r8 = sat_u8_rte(r * 255.0f);
      const uint32_t g8 = sat_u8_rte(gg * 255.0f);
      const uint32_t b8 = sat_u8_rte(b * 255.0f);
      const uint32_t w = FMT == F_RGBA8 ? (r8 | g8 << 8 | b8 << 16 | 0xff000000u) : (b8 | g8 << 8 | r8 << 16 | 0xff000000u);
      reinterpret_cast<uint32_t *>(a.p0)[(size_t)line * a.pitch + x] = w;
    }
    return;
  }
  constexpr bool V420 = (FMT == F_YUV420P || FMT == F_NV12);
  constexpr bool WIDE = (FMT == F_YUV422P10);
  const WriteK k = load_write_k(a.cm);
  const uint32_t full = a.width / 8, remain = a.width % 8, octets = full + (remain ? 1 : 0);
  const uint32_t total = octets * a.groups;
  const uint32_t stride = PERSISTENT ? gridDim.x * blockDim.x : total;
  for (uint32_t f = blockIdx.x * blockDim.x + threadIdx.x; f < total; f += stride) {
    const uint32_t g = f / octets, o8 = f - g * octets;
    const bool tail = (o8 == full);
    const uint32_t n = tail ? (remain < 6 ? remain : 6) : 8;
    // 4:2:2: one line per group (yuv422p10.ts:140-141); 4:2:0: line pair, or one field line of it
    const uint32_t first = V420 ? g * 2 + ((3 == a.interlace) ? 1 : 0)
                                : g * (a.interlace ? 2 : 1) + ((3 == a.interlace) ? 1 : 0);
    const uint32_t nlines = V420 ? (a.interlace ? 1 : 2) : 1;
    const uint32_t crow = V420 ? g : first;
    for (uint32_t l = 0; l < nlines; ++l) {
      const uint32_t line = first + l;
      const float4 *px = a.in + (size_t)line * a.width + 8 * o8;
      uint32_t y[8], u[4], v[4];
#pragma unroll
      for (int p = 0; p < 8; ++p) y[p] = WIDE ? 64u : 16u;  // tail defaults (:191-193)
#pragma unroll
      for (int p = 0; p < 4; ++p) u[p] = v[p] = WIDE ? 512u : 128u;
      if (!tail) {
#pragma unroll
        for (int p = 0; p < 8; ++p) {
          uint32_t cy, cu, cv;
          px_codes(px[p], k, lut, false, cy, cu, cv);
          y[p] = cy;
          if (!(p & 1)) u[p >> 1] = cu, v[p >> 1] = cv;
        }
      } else {  // yuv422p10.ts:190-217 / yuv420p.ts:213-251
        uint32_t cy[6] = {0, 0, 0, 0, 0, 0}, cu[6] = {0, 0, 0, 0, 0, 0}, cv[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int p = 0; p < 6; ++p)
          if ((uint32_t)p < n) px_codes(px[p], k, lut, true, cy[p], cu[p], cv[p]);
        y[0] = cy[0], y[1] = cy[1], u[0] = cu[0], v[0] = cv[0];
        if (remain > 2) {
          y[2] = cy[2], y[3] = cy[3], u[1] = cu[2], v[1] = cv[2];
          if (remain > 4) {
            y[4] = cy[4], y[5] = cy[5];
            if (V420) u[2] = cu[4], v[2] = cv[4];
            else u[1] = cu[4], v[1] = cv[4];  // the 4:2:2 writers overwrite slot 1 (yuv422p10.ts:209-210)
          }
        }
      }
      if (WIDE) {
        uint4 w;
        w.x = (y[0] & 0xffff) | y[1] << 16, w.y = (y[2] & 0xffff) | y[3] << 16;
        w.z = (y[4] & 0xffff) | y[5] << 16, w.w = (y[6] & 0xffff) | y[7] << 16;
        reinterpret_cast<uint4 *>(a.p0)[((size_t)line * a.pitch >> 3) + o8] = w;
      } else {  // uchar = (uchar)ushort keeps the low 8 bits (yuv422p8.ts:166-168)
        uint2 w;
        w.x = (y[0] & 0xff) | (y[1] & 0xff) << 8 | (y[2] & 0xff) << 16 | y[3] << 24;
        w.y = (y[4] & 0xff) | (y[5] & 0xff) << 8 | (y[6] & 0xff) << 16 | y[7] << 24;
        reinterpret_cast<uint2 *>(a.p0)[((size_t)line * a.pitch >> 3) + o8] = w;
      }
      if (l == 0) {
        if (FMT == F_NV12) {
          uint2 w;
          w.x = (u[0] & 0xff) | (v[0] & 0xff) << 8 | (u[1] & 0xff) << 16 | v[1] << 24;
          w.y = (u[2] & 0xff) | (v[2] & 0xff) << 8 | (u[3] & 0xff) << 16 | v[3] << 24;
          reinterpret_cast<uint2 *>(a.p1)[((size_t)crow * a.pitch >> 3) + o8] = w;
        } else if (WIDE) {
          uint2 wu, wv;
          wu.x = (u[0] & 0xffff) | u[1] << 16, wu.y = (u[2] & 0xffff) | u[3] << 16;
          wv.x = (v[0] & 0xffff) | v[1] << 16, wv.y = (v[2] & 0xffff) | v[3] << 16;
          reinterpret_cast<uint2 *>(a.p1)[((size_t)crow * a.pitch >> 3) + o8] = wu;
          reinterpret_cast<uint2 *>(a.p2)[((size_t)crow * a.pitch >> 3) + o8] = wv;
        } else {
          const uint32_t wu = (u[0] & 0xff) | (u[1] & 0xff) << 8 | (u[2] & 0xff) << 16 | u[3] << 24;
          const uint32_t wv = (v[0] & 0xff) | (v[1] & 0xff) << 8 | (v[2] & 0xff) << 16 | v[3] << 24;
          reinterpret_cast<uint32_t *>(a.p1)[((size_t)crow * a.pitch >> 3) + o8] = wu;
          reinterpret_cast<uint32_t *>(a.p2)[((size_t)crow * a.pitch >> 3) + o8] = wv;
        }
      }
    }
  }
}

template <int FMT>
__global__ __launch_bounds__(kLdsBlock) void fmt_write_lds_kernel(FmtWriteArgs a, LutView lv) {
  const LutInLds lut{make_lut_k(lv)};
  lds_lut_load(lv);
  __syncthreads();
  fmt_write_body<FMT, LutInLds, true>(a, lut);
}
template <int FMT>
__global__ __launch_bounds__(kFmtBlock) void fmt_write_gather_kernel(FmtWriteArgs a, const float *__restrict__ table) {
  const LutInGlobal lut{table};
  fmt_write_body<FMT, LutInGlobal, false>(a, lut);
}

// ------------------------------------------------------------------------------------------
// geometry + launchers
// ------------------------------------------------------------------------------------------
uint32_t pack_pitch(int fmt, uint32_t width) {
  if (fmt == F_RGBA8 || fmt == F_BGRA8) return width;                 // rgba8.ts:103-105
  if (fmt == F_V210) return width + 47 - ((width - 1) % 48);
  return width + 7 - ((width - 1) % 8);                              // yuv422p10.ts:221
}

int pack_plane_bytes(int fmt, uint32_t width, uint32_t height, size_t bytes[3]) {
  const size_t p = pack_pitch(fmt, width);
  bytes[0] = bytes[1] = bytes[2] = 0;
  switch (fmt) {
    case F_V210: bytes[0] = (size_t)v210_pitch_bytes(width) * height; return 1;
    case F_YUV422P10: bytes[0] = p * 2 * height, bytes[1] = bytes[2] = bytes[0] / 2; return 3;
    case F_YUV422P8: bytes[0] = p * height, bytes[1] = bytes[2] = bytes[0] / 2; return 3;
    case F_YUV420P: bytes[0] = p * height, bytes[1] = bytes[2] = bytes[0] / 4; return 3;
    case F_NV12: bytes[0] = p * height, bytes[1] = bytes[0] / 2; return 2;
    case F_RGBA8:
    case F_BGRA8: bytes[0] = p * 4 * height; return 1;
  }
  return -1;
}

template <int FMT>
static hipError_t launch_read_fmt(hipStream_t s, const FmtReadArgs &a, const float *table, const LutView *lv,
                                  uint32_t num_cus) {
  const uint32_t total = a.width * a.lines;
  if (!total) return hipSuccess;
  if (lv) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(fmt_read_lds_kernel<FMT>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, kMaxDynamicLds);
    if (e != hipSuccess) return e;
    const uint32_t want = (total + kLdsBlock - 1) / kLdsBlock;
    fmt_read_lds_kernel<FMT><<<want < num_cus ? want : num_cus, kLdsBlock, lv->bytes, s>>>(a, *lv);
  } else {
    fmt_read_gather_kernel<FMT><<<(total + kFmtBlock - 1) / kFmtBlock, kFmtBlock, 0, s>>>(a, table);
  }
  return hipGetLastError();
}

hipError_t launch_pack_read(hipStream_t s, int fmt, const void *const planes[3], void *out, uint32_t width,
                            uint32_t height, const void *cm, const void *table, const void *gm, const LutView *lv,
                            uint32_t num_cus) {
  const bool v420 = (fmt == F_YUV420P || fmt == F_NV12);
  FmtReadArgs a{planes[0], planes[1], planes[2], (float4 *)out, width, v420 ? (height / 2) * 2 : height,
                pack_pitch(fmt, width), (const float *)cm, (const float *)gm, image_nt((size_t)width * height * 16)};
  switch (fmt) {
    case F_YUV422P10: return launch_read_fmt<F_YUV422P10>(s, a, (const float *)table, lv, num_cus);
    case F_YUV422P8: return launch_read_fmt<F_YUV422P8>(s, a, (const float *)table, lv, num_cus);
    case F_YUV420P: return launch_read_fmt<F_YUV420P>(s, a, (const float *)table, lv, num_cus);
    case F_NV12: return launch_read_fmt<F_NV12>(s, a, (const float *)table, lv, num_cus);
    case F_RGBA8: return launch_read_fmt<F_RGBA8>(s, a, (const float *)table, lv, num_cus);
    case F_BGRA8: return launch_read_fmt<F_BGRA8>(s, a, (const float *)table, lv, num_cus);
    default: return hipErrorInvalidValue;
  }
}

template <int FMT>
static hipError_t launch_read_batch_fmt(hipStream_t s, const FmtReadBatchArgs &b, const LutView &lv, uint32_t num_cus) {
  const uint32_t total = b.width * b.lines;
  if (!total) return hipSuccess;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(fmt_read_lds_batch_kernel<FMT>), hipFuncAttributeMaxDynamicSharedMemorySize, kMaxDynamicLds);
  if (e != hipSuccess) return e;
  const uint32_t want = (total + kLdsBlock - 1) / kLdsBlock, share = num_cus / b.jobs ? num_cus / b.jobs : 1u, per_job = want < share ? want : share;
  fmt_read_lds_batch_kernel<FMT><<<per_job * b.jobs, kLdsBlock, lv.bytes, s>>>(b, lv);
  return hipGetLastError();
}
// n frames (1 .. kMaxLayers) of one format, size and Loader recipe; the table in its LDS form
hipError_t launch_pack_read_batch(hipStream_t s, int fmt, int n, const void *const (*planes)[3], void *const *outs, uint32_t width, uint32_t height,
                                  const void *cm, const void *gm, const LutView &lv, uint32_t num_cus) {
  const bool v420 = (fmt == F_YUV420P || fmt == F_NV12);
  FmtReadBatchArgs b{};
  for (int i = 0; i < n; ++i) b.p0[i] = planes[i][0], b.p1[i] = planes[i][1], b.p2[i] = planes[i][2], b.out[i] = (float4 *)outs[i];
  b.jobs = (uint32_t)n, b.width = width, b.lines = v420 ? (height / 2) * 2 : height, b.pitch = pack_pitch(fmt, width);
  b.cm = (const float *)cm, b.gm = (const float *)gm, b.nt = image_nt((size_t)width * height * 16);
  switch (fmt) {
    case F_YUV422P10: return launch_read_batch_fmt<F_YUV422P10>(s, b, lv, num_cus);
    case F_YUV422P8: return launch_read_batch_fmt<F_YUV422P8>(s, b, lv, num_cus);
    case F_YUV420P: return launch_read_batch_fmt<F_YUV420P>(s, b, lv, num_cus);
    case F_NV12: return launch_read_batch_fmt<F_NV12>(s, b, lv, num_cus);
    case F_RGBA8: return launch_read_batch_fmt<F_RGBA8>(s, b, lv, num_cus);
    case F_BGRA8: return launch_read_batch_fmt<F_BGRA8>(s, b, lv, num_cus);
    default: return hipErrorInvalidValue;
  }
}

template <int FMT>
static hipError_t launch_write_fmt(hipStream_t s, const FmtWriteArgs &a, const float *table, const LutView *lv,
                                   uint32_t num_cus) {
  const bool rgb = (FMT == F_RGBA8 || FMT == F_BGRA8);
  const uint32_t total = rgb ? a.width * a.groups : ((a.width + 7) / 8) * a.groups;
  if (!total) return hipSuccess;
  if (lv) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(fmt_write_lds_kernel<FMT>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, kMaxDynamicLds);
    if (e != hipSuccess) return e;
    const uint32_t want = (total + kLdsBlock - 1) / kLdsBlock;
    fmt_write_lds_kernel<FMT><<<want < num_cus ? want : num_cus, kLdsBlock, lv->bytes, s>>>(a, *lv);
  } else {
    fmt_write_gather_kernel<FMT><<<(total + kFmtBlock - 1) / kFmtBlock, kFmtBlock, 0, s>>>(a, table);
  }
  return hipGetLastError();
}

hipError_t launch_pack_write(hipStream_t s, int fmt, const void *in, void *const planes[3], uint32_t width,
                             uint32_t height, uint32_t interlace, const void *cm, const void *table, const LutView *lv,
                             uint32_t num_cus) {
  const bool v420 = (fmt == F_YUV420P || fmt == F_NV12);
  const uint32_t groups = v420 ? height / 2 : (interlace ? height / 2 : height);  // e.g. yuv422p10.ts:328, yuv420p.ts:381
  FmtWriteArgs a{(const float4 *)in, planes[0], planes[1], planes[2], width, pack_pitch(fmt, width), groups, interlace,
                 (const float *)cm};
  switch (fmt) {
    case F_YUV422P10: return launch_write_fmt<F_YUV422P10>(s, a, (const float *)table, lv, num_cus);
    case F_YUV422P8: return launch_write_fmt<F_YUV422P8>(s, a, (const float *)table, lv, num_cus);
    case F_YUV420P: return launch_write_fmt<F_YUV420P>(s, a, (const float *)table, lv, num_cus);
    case F_NV12: return launch_write_fmt<F_NV12>(s, a, (const float *)table, lv, num_cus);
    case F_RGBA8: return launch_write_fmt<F_RGBA8>(s, a, (const float *)table, lv, num_cus);
    case F_BGRA8: return launch_write_fmt<F_BGRA8>(s, a, (const float *)table, lv, num_cus);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace ph
