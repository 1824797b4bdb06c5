// ph_kernels.hip - hand-written gfx950 kernels for the phaneron per-pixel hot path.
//
// Layouts: v210 = little-endian 32-bit words, 4 words (16 B) per 6 pixels, line pitch
// ceil(width/48)*128 B; images = row-major float4 RGBA, unpadded.  One lane owns one v210
// word quad (16-byte coalesced load/store); the 6 float4 pixels that go with it are moved
// between "quad order" and "pixel order" through LDS so the f32 side is also accessed as
// contiguous 1 KiB wave transactions.
#include <cstdlib>
#include "ph_device.h"
#include "ph_kernels.h"
#include "ph_yadif.h"
#include <type_traits>

#pragma clang fp contract(off)

namespace ph {

thread_local uint32_t t_stream_images = 0;
thread_local uint32_t t_stream_threshold_mb = 64;

constexpr int kBlock = 256;
constexpr int kWave = 64;

// ------------------------------------------------------------------------------------------
// v210 read (reference v210.ts:25-111)
// ------------------------------------------------------------------------------------------
// Fast path, width % 6 == 0: quads are numbered flat over the frame (f = line*qpl_used + g),
// the float4 output of quad f starts at pixel 6f, so a wave's 64 quads produce 384 contiguous
// pixels = six 1 KiB stores after the LDS transpose.
__global__ __launch_bounds__(kBlock) void v210_read_kernel(const uint4 *__restrict__ in,
                                                           float4 *__restrict__ out, uint32_t quads_per_line_used,
                                                           uint32_t quads_per_line_pitch, uint32_t total_quads,
                                                           const float *__restrict__ cm, const float *__restrict__ lut,
                                                           const float *__restrict__ gm, uint32_t nt) {
  __shared__ float4 tile[kBlock / kWave][kWave * 6];
  const ReadK k = load_read_k(cm, gm);
  const uint32_t lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  const uint32_t f = blockIdx.x * kBlock + threadIdx.x;
  if (f < total_quads) {
    const uint32_t line = f / quads_per_line_used, g = f - line * quads_per_line_used;
    const Yuv6 q = unpack_quad(in[(size_t)line * quads_per_line_pitch + g]);
#pragma unroll
    for (int j = 0; j < 6; ++j) tile[wave][6 * lane + j] = read_px(q.y[j], q.cb[j >> 1], q.cr[j >> 1], 1.0f, k, lut);
  }
  __syncthreads();
  const uint32_t wave_f0 = blockIdx.x * kBlock + wave * kWave;  // first quad of this wave
  const size_t px0 = (size_t)wave_f0 * 6, px_end = (size_t)total_quads * 6;
#pragma unroll
  for (int s = 0; s < 6; ++s) {
    const size_t p = px0 + s * kWave + lane;
    if (p < px_end) store_image(out + p, tile[wave][s * kWave + lane], nt);
  }
}

// General path (any width): one lane per quad slot of the line, including the reference's
// tail quirk for width % 6 != 0 (4th vector component 0, v210.ts:84-110).
__global__ __launch_bounds__(kBlock) void v210_read_tail_kernel(const uint4 *__restrict__ in,
                                                                float4 *__restrict__ out, uint32_t width,
                                                                uint32_t height, uint32_t quads_per_line_pitch,
                                                                const float *__restrict__ cm,
                                                                const float *__restrict__ lut,
                                                                const float *__restrict__ gm) {
  const ReadK k = load_read_k(cm, gm);
  const uint32_t full = width / 6, remain = width % 6, slots = full + (remain ? 1 : 0);
  const uint32_t f = blockIdx.x * kBlock + threadIdx.x;
  if (f >= slots * height) return;
  const uint32_t line = f / slots, g = f - line * slots;
  const Yuv6 q = unpack_quad(in[(size_t)line * quads_per_line_pitch + g]);
  float4 *o = out + (size_t)line * width + 6 * g;
  if (g < full) {
#pragma unroll
    for (int j = 0; j < 6; ++j) o[j] = read_px(q.y[j], q.cb[j >> 1], q.cr[j >> 1], 1.0f, k, lut);
  } else {
    for (uint32_t j = 0; j < remain && j < 4; ++j) o[j] = read_px(q.y[j], q.cb[j >> 1], q.cr[j >> 1], 0.0f, k, lut);
  }
}

// ------------------------------------------------------------------------------------------
// v210 write (reference v210.ts:113-195)
// ------------------------------------------------------------------------------------------
// width % 6 == 0.  `lines` output lines are produced: line = first + idx*step (interlace).
// Within a line the six float4 loads per wave are contiguous 1 KiB reads; LDS turns them
// into quad order.  Lines are addressed by pitch (see DESIGN.md "deviations").
__global__ __launch_bounds__(kBlock) void v210_write_kernel(const float4 *__restrict__ in, uint4 *__restrict__ out,
                                                            uint32_t width, uint32_t quads_per_line_used,
                                                            uint32_t quads_per_line_pitch, uint32_t first_line,
                                                            uint32_t line_step, uint32_t blocks_per_line,
                                                            const float *__restrict__ cm,
                                                            const float *__restrict__ lut) {
  __shared__ float4 tile[kBlock / kWave][kWave * 6];
  const WriteK k = load_write_k(cm);
  const uint32_t lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  const uint32_t line_idx = blockIdx.x / blocks_per_line, blk = blockIdx.x - line_idx * blocks_per_line;
  const uint32_t line = first_line + line_idx * line_step;
  const uint32_t g0 = blk * kBlock + wave * kWave;  // first quad of this wave in the line
  const float4 *src = in + (size_t)line * width;
#pragma unroll
  for (int s = 0; s < 6; ++s) {
    const uint32_t p = g0 * 6 + s * kWave + lane;
    if (p < width) tile[wave][s * kWave + lane] = src[p];
  }
  __syncthreads();
  const uint32_t g = g0 + lane;
  if (g >= quads_per_line_used) return;
  uint32_t y[6], u[3], v[3];
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const float4 px = tile[wave][6 * lane + j];
    if ((j & 1) == 0) {
      const Yuv1 c = write_px(px.x, px.y, px.z, k, lut);
      y[j] = c.y, u[j >> 1] = c.u, v[j >> 1] = c.v;
    } else {
      y[j] = write_px_luma(px.x, px.y, px.z, k, lut);
    }
  }
  store_stream(out + (size_t)line * quads_per_line_pitch + g, pack_quad(y, u, v));
}

// Any width: one lane per quad slot; implements the tail (remain = width % 6 pixels with the
// reference's truncating / round-half-away arithmetic) and zero-fills the rest of the last
// 128-byte block of the line (v210.ts:131-136).
__global__ __launch_bounds__(kBlock) void v210_write_tail_kernel(const float4 *__restrict__ in,
                                                                 uint4 *__restrict__ out, uint32_t width,
                                                                 uint32_t lines, uint32_t quads_per_line_pitch,
                                                                 uint32_t first_line, uint32_t line_step,
                                                                 const float *__restrict__ cm,
                                                                 const float *__restrict__ lut) {
  const WriteK k = load_write_k(cm);
  const uint32_t f = blockIdx.x * kBlock + threadIdx.x;
  if (f >= quads_per_line_pitch * lines) return;
  const uint32_t line_idx = f / quads_per_line_pitch, g = f - line_idx * quads_per_line_pitch;
  const uint32_t line = first_line + line_idx * line_step;
  const uint32_t full = width / 6, remain = width % 6;
  const float4 *px = in + (size_t)line * width + 6 * g;
  uint4 w = make_uint4(0, 0, 0, 0);
  if (g < full) {
    uint32_t y[6], u[3], v[3];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const float4 p = px[j];
      const Yuv1 c = write_px(p.x, p.y, p.z, k, lut);
      y[j] = c.y;
      if ((j & 1) == 0) u[j >> 1] = c.u, v[j >> 1] = c.v;
    }
    w = pack_quad(y, u, v);
  } else if (g == full && remain) {
    Yuv1 c[4] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    for (uint32_t j = 0; j < remain && j < 4; ++j) c[j] = write_px_tail(px[j].x, px[j].y, px[j].z, k, lut);
    w.x = c[0].v << 20 | c[0].y << 10 | c[0].u;
    if (2 == remain) {
      w.y = c[1].y;
    } else if (4 == remain) {
      w.y = c[2].y << 20 | c[2].u << 10 | c[1].y;
      w.z = c[3].y << 10 | c[2].v;
    }
  } else if (width % 48 == 0) {
    return;  // no padding quads exist
  }
  // quads past the tail inside the last 48-pixel block are cleared, like the reference
  store_stream(out + (size_t)line * quads_per_line_pitch + g, w);
}

// ------------------------------------------------------------------------------------------
// Fused channel pipeline: [v210 read] x N -> combine_N -> v210 write, one quad per lane.
// The f32 RGBA intermediates of the reference's job batch live in registers only.
// ------------------------------------------------------------------------------------------
template <int N>
__global__ __launch_bounds__(kBlock) void fused_v210_combine_kernel(FusedArgs a) {
  const ReadK rk = load_read_k(a.rd_cm, a.rd_gm);
  const WriteK wk = load_write_k(a.wr_cm);
  const uint32_t f = blockIdx.x * kBlock + threadIdx.x;
  if (f >= a.total_quads) return;
  // Lines that do not end on a 48-pixel block (total_quads then counts the slots of the pitch): whole quads, the tail quad
  // (read with the fourth vector component 0, v210.ts:88-93; written with the tail arithmetic, :169-194), cleared slots (:131-136)
  const bool ragged = a.quads_per_line_used != a.quads_per_line_pitch;  // uniform
  const uint32_t per_line = ragged ? a.quads_per_line_pitch : a.quads_per_line_used;
  const uint32_t line = f / per_line, g = f - line * per_line;
  const size_t off = (size_t)line * a.quads_per_line_pitch + g;
  const bool in_tail = ragged && g == a.quads_per_line_used && a.tail_px;
  if (ragged && g >= a.quads_per_line_used && !in_tail) {
    store_stream(reinterpret_cast<uint4 *>(a.out) + off, make_uint4(0u, 0u, 0u, 0u));
    return;
  }
  const float last = in_tail ? 0.0f : 1.0f;

  uint4 w[N];
#pragma unroll
  for (int l = 0; l < N; ++l) w[l] = reinterpret_cast<const uint4 *>(a.layers[l])[off];

  float4 acc[6];
  {
    const Yuv6 q = unpack_quad(w[0]);
#pragma unroll
    for (int j = 0; j < 6; ++j) acc[j] = read_px(q.y[j], q.cb[j >> 1], q.cr[j >> 1], last, rk, a.rd_lut);
  }
#pragma unroll
  for (int l = 1; l < N; ++l) {  // combine.ts:45-65: premultiplied "over", alpha = top layer's
    const Yuv6 q = unpack_quad(w[l]);
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const float4 t = read_px(q.y[j], q.cb[j >> 1], q.cr[j >> 1], last, rk, a.rd_lut);
      const float kk = 1.0f - t.w;
      acc[j].x = fma_rn(acc[j].x, kk, t.x);
      acc[j].y = fma_rn(acc[j].y, kk, t.y);
      acc[j].z = fma_rn(acc[j].z, kk, t.z);
      acc[j].w = fma_rn(acc[j].w, 0.0f, t.w);
    }
  }
  if (in_tail) {
    Yuv1 c[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) c[j] = write_px_tail(acc[j].x, acc[j].y, acc[j].z, wk, a.wr_lut);
    uint4 o = make_uint4(c[0].v << 20 | c[0].y << 10 | c[0].u, 0u, 0u, 0u);
    if (a.tail_px == 2u) {
      o.y = c[1].y;
    } else if (a.tail_px == 4u) {
      o.y = c[2].y << 20 | c[2].u << 10 | c[1].y;
      o.z = c[3].y << 10 | c[2].v;
    }
    store_stream(reinterpret_cast<uint4 *>(a.out) + off, o);
    return;
  }
  uint32_t y[6], u[3], v[3];
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    if ((j & 1) == 0) {
      const Yuv1 c = write_px(acc[j].x, acc[j].y, acc[j].z, wk, a.wr_lut);
      y[j] = c.y, u[j >> 1] = c.u, v[j >> 1] = c.v;
    } else {
      y[j] = write_px_luma(acc[j].x, acc[j].y, acc[j].z, wk, a.wr_lut);
    }
  }
  store_stream(reinterpret_cast<uint4 *>(a.out) + off, pack_quad(y, u, v));
}

// ------------------------------------------------------------------------------------------
// yadif (reference yadifCl.ts:28-167).  grid = (ceil(w/250), ceil(h/16)); the two neighbouring `cur`
// rows of an interpolated row are staged in LDS for the 14-tap spatial predictor.
// ------------------------------------------------------------------------------------------
// The filter walks DOWN a strip of rows with the five-row windows of the three frames in registers.
// A row-per-block mapping re-reads every source row for each output row that uses it (13 loads per
// interpolated pixel, served by L2: the L2 becomes the bound, 129 us at 2160p against 94 us);
// here every row a strip needs is loaded once - five float4 loads and two stores per PAIR of output
// rows, less than the "every input once" figure because the field not being rebuilt never needs the
// even rows of one neighbour frame.  250 of a block's 256 columns produce output, the outer three on
// each side only feed the spatial predictor's x-3..x+3 taps through LDS (clamped at the image edge).
#ifndef PH_YADIF_ROWS
#define PH_YADIF_ROWS 16
#endif
constexpr int kYadifRows = PH_YADIF_ROWS, kYadifCols = kBlock - 6;
__global__ __launch_bounds__(kBlock) void yadif_rows_kernel(const float4 *__restrict__ prev, const float4 *__restrict__ cur,
                                                            const float4 *__restrict__ next, int w, int h, int parity,
                                                            int tff, int skip, float4 *__restrict__ out, uint32_t nt) {
  __shared__ float4 rows[2][2][kBlock];  // [step parity][row y-1 / row y+1][column]
  const int lane = threadIdx.x;
  const int xr = blockIdx.x * kYadifCols - 3 + lane, x = clampi(xr, 0, w - 1);  // CLAMP_TO_EDGE
  const bool emit = lane >= 3 && lane < kBlock - 3 && xr < w;
  const int y0 = blockIdx.y * kYadifRows, y_end = (y0 + kYadifRows < h) ? y0 + kYadifRows : h;
  const int second = !(parity ^ tff);  // yadifCl.ts:143
  auto row = [&](const float4 *img, int y) { return img[(size_t)clampi(y, 0, h - 1) * w + x]; };
  // the interpolated row of the first pair: y0 is even, rows with (y & 1) == parity are copied
  int yi = y0 + (parity == 0 ? 1 : 0);
  // windows around the interpolated row yi: cur rows yi-2..yi+2; prev / next rows yi-1, yi+1; and the
  // even rows yi-2, yi, yi+2 of the one neighbour frame the temporal check reads (next for the second
  // field, prev for the first: yadifCl.ts:146-151).  Plain values, never arrays addressed through a
  // select - that would move them to scratch.
  const float4 *other = second ? next : prev;
  float4 C[5], E[3];
#pragma unroll
  for (int k = 0; k < 5; ++k) C[k] = row(cur, yi - 2 + k);
#pragma unroll
  for (int k = 0; k < 3; ++k) E[k] = row(other, yi - 2 + 2 * k);
  float4 P1 = row(prev, yi - 1), P3 = row(prev, yi + 1), N1 = row(next, yi - 1), N3 = row(next, yi + 1);
  // pairs: (copy yi-1, interpolated yi) when parity == 0, (interpolated yi, copy yi+1) when parity == 1
  for (int step = 0; (parity == 0 ? yi - 1 : yi) < y_end; ++step, yi += 2) {
    const int buf = step & 1;
    rows[buf][0][lane] = C[1], rows[buf][1][lane] = C[3];
    __syncthreads();  // one barrier per pair: the buffers alternate
    if (emit) {
      const int yc = parity == 0 ? yi - 1 : yi + 1;  // the copied row of this pair (yadifCl.ts:117-121)
      if (yc < h) {
        float4 cp;
        cp.x = parity == 0 ? C[1].x : C[3].x, cp.y = parity == 0 ? C[1].y : C[3].y;
        cp.z = parity == 0 ? C[1].z : C[3].z, cp.w = parity == 0 ? C[1].w : C[3].w;
        store_image(out + (size_t)yc * w + xr, cp, nt);
      }
      if (yi < h) {
        float4 ra[7], rb[7];
#pragma unroll
        for (int t = 0; t < 7; ++t) ra[t] = rows[buf][0][lane - 3 + t], rb[t] = rows[buf][1][lane - 3 + t];
        float res[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float sp = yadif_spatial(PH_C4(ra[0], c), PH_C4(ra[1], c), PH_C4(ra[2], c), PH_C4(ra[3], c),
                                         PH_C4(ra[4], c), PH_C4(ra[5], c), PH_C4(ra[6], c), PH_C4(rb[0], c),
                                         PH_C4(rb[1], c), PH_C4(rb[2], c), PH_C4(rb[3], c), PH_C4(rb[4], c),
                                         PH_C4(rb[5], c), PH_C4(rb[6], c));
          // second field: s0 = cur, s1 = next; first field: s0 = prev, s1 = cur
          const float c0 = PH_C4(C[0], c), c2 = PH_C4(C[2], c), c4 = PH_C4(C[4], c);
          const float e0 = PH_C4(E[0], c), e1 = PH_C4(E[1], c), e2 = PH_C4(E[2], c);
          res[c] = yadif_temporal(PH_C4(P1, c), PH_C4(P3, c), second ? c0 : e0, second ? c2 : e1, second ? c4 : e2,
                                  PH_C4(C[1], c), PH_C4(C[3], c), second ? e0 : c0, second ? e1 : c2, second ? e2 : c4,
                                  PH_C4(N1, c), PH_C4(N3, c), sp, skip);
        }
        store_image(out + (size_t)yi * w + xr, make_float4(res[0], res[1], res[2], C[2].w), nt);  // :164 alpha from cur
      }
    }
    // slide the windows down two rows
    C[0] = C[2], C[1] = C[3], C[2] = C[4], C[3] = row(cur, yi + 3), C[4] = row(cur, yi + 4);
    E[0] = E[1], E[1] = E[2], E[2] = row(other, yi + 4);
    P1 = P3, P3 = row(prev, yi + 3), N1 = N3, N3 = row(next, yi + 3);
  }
}

// Both fields of one frame in one pass (the send_field mode of yadif.ts:88-145 runs the filter twice over the same
// three frames, parity 1 ^ tff then parity tff).  Every row is interpolated in exactly one of the two outputs and
// copied into the other, and between them the two runs read every row of prev, cur and next: this kernel walks a
// strip ROW BY ROW with the five-row windows of all three frames in registers, loads each source row once (three
// loads per row) and stores the row twice - 5 frames of traffic per pair of fields instead of 7.  out0 / out1 are
// exactly what yadif_rows_kernel writes for parity 0 / parity 1.  The `cur` rows are staged once each in a ring of
// four LDS rows (row y + 1 at step y; it is row y' - 1 two steps later), one barrier per row.
template <int TFF>
__global__ __launch_bounds__(kBlock) void yadif_pair_kernel(const float4 *__restrict__ prev, const float4 *__restrict__ cur,
                                                            const float4 *__restrict__ next, int w, int h, int skip,
                                                            float4 *__restrict__ out0, float4 *__restrict__ out1, uint32_t nt) {
  // component planes: the spatial predictor reads its 14 taps one component at a time (conflict-free ds_read_b32,
  // 14 live registers instead of 56 for whole float4 taps: 6 waves per SIMD instead of 4)
  __shared__ float rows[4][4][kBlock];  // [row & 3][component][column]
  const int lane = threadIdx.x;
  const int xr = blockIdx.x * kYadifCols - 3 + lane, x = clampi(xr, 0, w - 1);  // CLAMP_TO_EDGE
  const bool emit = lane >= 3 && lane < kBlock - 3 && xr < w;
  const int y0 = blockIdx.y * kYadifRows, y_end = (y0 + kYadifRows < h) ? y0 + kYadifRows : h;  // y0 is even
  auto row = [&](const float4 *img, int y) { return img[(size_t)clampi(y, 0, h - 1) * w + x]; };
  auto stage = [&](int r, const float4 v) {
    rows[r & 3][0][lane] = v.x, rows[r & 3][1][lane] = v.y, rows[r & 3][2][lane] = v.z, rows[r & 3][3][lane] = v.w;
  };
  float4 C[5], P[5], N[5];
#pragma unroll
  for (int k = 0; k < 5; ++k) C[k] = row(cur, y0 - 2 + k), P[k] = row(prev, y0 - 2 + k), N[k] = row(next, y0 - 2 + k);
  stage(y0 + 3, C[1]), stage(y0, C[2]);  // rows y0 - 1 and y0; every later row is staged as "y + 1"
  // one row: interpolated into the output whose parity is (y & 1) ^ 1, copied into the other.  SECOND
  // (yadifCl.ts:143, !(parity ^ tff)) is a compile-time constant of the row's evenness.
  auto step = [&](int y, auto second_tag, float4 *__restrict__ out_interp, float4 *__restrict__ out_copy) {
    constexpr bool second = decltype(second_tag)::value;
    stage(y + 1, C[3]);
    __syncthreads();
    if (emit) {
      store_image(out_copy + (size_t)y * w + xr, C[2], nt);  // yadifCl.ts:117-121
      float res[4];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float *ra = &rows[(y + 3) & 3][c][lane - 3], *rb = &rows[(y + 1) & 3][c][lane - 3];
        const float sp = yadif_spatial(ra[0], ra[1], ra[2], ra[3], ra[4], ra[5], ra[6], rb[0], rb[1], rb[2], rb[3], rb[4], rb[5], rb[6]);
        // second field: s0 = cur, s1 = next; first field: s0 = prev, s1 = cur (yadifCl.ts:146-151)
        const float c0 = PH_C4(C[0], c), c2 = PH_C4(C[2], c), c4 = PH_C4(C[4], c);
        const float e0 = second ? PH_C4(N[0], c) : PH_C4(P[0], c), e1 = second ? PH_C4(N[2], c) : PH_C4(P[2], c),
                    e2 = second ? PH_C4(N[4], c) : PH_C4(P[4], c);
        res[c] = yadif_temporal(PH_C4(P[1], c), PH_C4(P[3], c), second ? c0 : e0, second ? c2 : e1, second ? c4 : e2,
                                PH_C4(C[1], c), PH_C4(C[3], c), second ? e0 : c0, second ? e1 : c2, second ? e2 : c4,
                                PH_C4(N[1], c), PH_C4(N[3], c), sp, skip);
      }
      store_image(out_interp + (size_t)y * w + xr, make_float4(res[0], res[1], res[2], C[2].w), nt);  // :164 alpha from cur
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) C[k] = C[k + 1], P[k] = P[k + 1], N[k] = N[k + 1];
    C[4] = row(cur, y + 3), P[4] = row(prev, y + 3), N[4] = row(next, y + 3);  // (loading a row one step earlier measured the same)
  };
  for (int y = y0; y < y_end; y += 2) {
    // even row: interpolated in the parity-1 output, second = !(1 ^ tff) = tff
    step(y, std::integral_constant<bool, TFF != 0>{}, out1, out0);
    // odd row: interpolated in the parity-0 output, second = !(0 ^ tff) = !tff
    if (y + 1 < y_end) step(y + 1, std::integral_constant<bool, TFF == 0>{}, out0, out1);
  }
}

// transform.ts:36-59.  2-D grid; 64x4 blocks keep a wave on one output row.
__global__ __launch_bounds__(kBlock) void transform_kernel(const float4 *__restrict__ in, int iw, int ih,
                                                           const float *__restrict__ m, float4 *__restrict__ out,
                                                           int ow, int oh, uint32_t nt) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= ow || y >= oh) return;
  const float px = (float)x / (float)ow - 0.5f, py = (float)y / (float)oh - 0.5f;
  const float s = dot3(m[0], m[1], m[2], px, py, 1.0f) + 0.5f;
  const float t = dot3(m[3], m[4], m[5], px, py, 1.0f) + 0.5f;
  store_image(out + (size_t)y * ow + x, sample_linear(in, iw, ih, s, t), nt);
}

// resize.ts:35-59
__global__ __launch_bounds__(kBlock) void resize_kernel(const float4 *__restrict__ in, int iw, int ih, float scale,
                                                        float off_x, float off_y, const float *__restrict__ flip,
                                                        float4 *__restrict__ out, int ow, int oh, uint32_t nt) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= ow || y >= oh) return;
  const float cx = (-0.5f - off_x) / scale + 0.5f, cy = (-0.5f - off_y) / scale + 0.5f;
  const float ox = fma_rn(cx, flip[1], flip[0]), oy = fma_rn(cy, flip[3], flip[2]);
  const float mx = flip[1] / scale, my = flip[3] / scale;
  const float s = fma_rn((float)x / (float)ow, mx, ox), t = fma_rn((float)y / (float)oh, my, oy);
  store_image(out + (size_t)y * ow + x, sample_linear(in, iw, ih, s, t), nt);
}

// ------------------------------------------------------------------------------------------
// combine_N / transitions / mixer / wipe: one float4 per lane, grid-stride
// ------------------------------------------------------------------------------------------
template <int N>
__global__ __launch_bounds__(kBlock) void combine_kernel(CombineArgs a) {
  for (size_t p = (size_t)blockIdx.x * kBlock + threadIdx.x; p < a.npx; p += (size_t)gridDim.x * kBlock) {
    float4 acc = N >= 3 ? load_stream(reinterpret_cast<const float4 *>(a.layers[0]) + p)
                        : reinterpret_cast<const float4 *>(a.layers[0])[p];
#pragma unroll
    for (int l = 1; l < N; ++l) {
      const float4 t = N >= 3 ? load_stream(reinterpret_cast<const float4 *>(a.layers[l]) + p)
                              : reinterpret_cast<const float4 *>(a.layers[l])[p];
      const float k = 1.0f - t.w;
      acc.x = fma_rn(acc.x, k, t.x);
      acc.y = fma_rn(acc.y, k, t.y);
      acc.z = fma_rn(acc.z, k, t.z);
      acc.w = fma_rn(acc.w, 0.0f, t.w);
    }
    store_image(reinterpret_cast<float4 *>(a.out) + p, acc, a.nt);
  }
}

__global__ __launch_bounds__(kBlock) void dissolve_kernel(const float4 *__restrict__ in0, const float4 *__restrict__ in1,
                                                          float mix, size_t npx, float4 *__restrict__ out, uint32_t nt) {
  const float rmix = 1.0f - mix;
  for (size_t p = (size_t)blockIdx.x * kBlock + threadIdx.x; p < npx; p += (size_t)gridDim.x * kBlock) {
    const float4 a = in0[p], b = in1[p];
    store_image(out + p, make_float4(fma_rn(a.x, mix, b.x * rmix), fma_rn(a.y, mix, b.y * rmix), fma_rn(a.z, mix, b.z * rmix),
                        fma_rn(a.w, mix, b.w * rmix)), nt);
  }
}

__global__ __launch_bounds__(kBlock) void twipe_kernel(const float4 *__restrict__ in0, const float4 *__restrict__ in1,
                                                       const float4 *__restrict__ mask, size_t npx,
                                                       float4 *__restrict__ out, uint32_t nt) {
  for (size_t p = (size_t)blockIdx.x * kBlock + threadIdx.x; p < npx; p += (size_t)gridDim.x * kBlock) {
    const float4 a = load_stream(in0 + p), b = load_stream(in1 + p);
    const float m = load_stream(mask + p).x, rm = 1.0f - m;
    store_image(out + p, make_float4(fma_rn(b.x, m, a.x * rm), fma_rn(b.y, m, a.y * rm), fma_rn(b.z, m, a.z * rm),
                        fma_rn(b.w, m, a.w * rm)), nt);
  }
}

__global__ __launch_bounds__(kBlock) void wipe_kernel(const float4 *__restrict__ in0, const float4 *__restrict__ in1,
                                                      float wipe, int w, int h, float4 *__restrict__ out, uint32_t nt) {
  const float edge = (float)w * wipe;  // wipe.ts:44
  const size_t npx = (size_t)w * h;
  for (size_t p = (size_t)blockIdx.x * kBlock + threadIdx.x; p < npx; p += (size_t)gridDim.x * kBlock) {
    const int x = (int)(p % (size_t)w);
    store_image(out + p, ((float)x > edge) ? in1[p] : in0[p], nt);
  }
}

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------
static inline uint32_t div_up(uint64_t a, uint32_t b) { return (uint32_t)((a + b - 1) / b); }
static inline uint32_t stream_grid(size_t n) {  // memory-bound element-wise: <= 8 blocks per CU
  const uint64_t want = (n + kBlock - 1) / kBlock;
  return (uint32_t)(want < 2048 ? (want ? want : 1) : 2048);
}

uint32_t v210_pitch_bytes(uint32_t width) { return (width + 47 - ((width - 1) % 48)) * 8 / 3; }

hipError_t launch_v210_read(hipStream_t s, const void *in, void *out, uint32_t width, uint32_t height,
                            const void *cm, const void *lut, const void *gm) {
  const uint32_t qpl = v210_pitch_bytes(width) / 16;
  if (width % 6 == 0) {
    const uint32_t used = width / 6, total = used * height;
    v210_read_kernel<<<div_up(total, kBlock), kBlock, 0, s>>>((const uint4 *)in, (float4 *)out, used, qpl, total,
                                                             (const float *)cm, (const float *)lut, (const float *)gm,
                                                             image_nt((size_t)width * height * 16));
  } else {
    const uint32_t slots = width / 6 + 1;
    v210_read_tail_kernel<<<div_up((uint64_t)slots * height, kBlock), kBlock, 0, s>>>(
        (const uint4 *)in, (float4 *)out, width, height, qpl, (const float *)cm, (const float *)lut, (const float *)gm);
  }
  return hipGetLastError();
}

hipError_t launch_v210_write(hipStream_t s, const void *in, void *out, uint32_t width, uint32_t height,
                             uint32_t interlace, const void *cm, const void *lut) {
  const uint32_t qpl = v210_pitch_bytes(width) / 16;
  const uint32_t step = interlace ? 2 : 1, first = (interlace == 3) ? 1 : 0;
  const uint32_t lines = interlace ? height / 2 : height;  // v210.ts:323
  if (lines == 0) return hipSuccess;
  if (width % 48 == 0) {
    const uint32_t used = width / 6, bpl = div_up(used, kBlock);
    v210_write_kernel<<<bpl * lines, kBlock, 0, s>>>((const float4 *)in, (uint4 *)out, width, used, qpl, first, step,
                                                     bpl, (const float *)cm, (const float *)lut);
  } else {
    v210_write_tail_kernel<<<div_up((uint64_t)qpl * lines, kBlock), kBlock, 0, s>>>(
        (const float4 *)in, (uint4 *)out, width, lines, qpl, first, step, (const float *)cm, (const float *)lut);
  }
  return hipGetLastError();
}

hipError_t launch_fused_v210_combine(hipStream_t s, int n, const FusedArgs &a) {
  const uint32_t grid = div_up(a.total_quads, kBlock);
  switch (n) {
    case 1: fused_v210_combine_kernel<1><<<grid, kBlock, 0, s>>>(a); break;
    case 2: fused_v210_combine_kernel<2><<<grid, kBlock, 0, s>>>(a); break;
    case 3: fused_v210_combine_kernel<3><<<grid, kBlock, 0, s>>>(a); break;
    case 4: fused_v210_combine_kernel<4><<<grid, kBlock, 0, s>>>(a); break;
    case 5: fused_v210_combine_kernel<5><<<grid, kBlock, 0, s>>>(a); break;
    case 6: fused_v210_combine_kernel<6><<<grid, kBlock, 0, s>>>(a); break;
    case 7: fused_v210_combine_kernel<7><<<grid, kBlock, 0, s>>>(a); break;
    case 8: fused_v210_combine_kernel<8><<<grid, kBlock, 0, s>>>(a); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t launch_yadif(hipStream_t s, const void *prev, const void *cur, const void *next, int w, int h, int parity,
                        int tff, int skip, void *out) {
  dim3 grid(div_up(w, kYadifCols), div_up(h, kYadifRows));
  yadif_rows_kernel<<<grid, kBlock, 0, s>>>((const float4 *)prev, (const float4 *)cur, (const float4 *)next, w, h, parity,
                                            tff, skip, (float4 *)out, image_nt((size_t)w * h * 16));
  return hipGetLastError();
}

hipError_t launch_yadif_pair(hipStream_t s, const void *prev, const void *cur, const void *next, int w, int h, int tff,
                             int skip, void *out0, void *out1) {
  dim3 grid(div_up(w, kYadifCols), div_up(h, kYadifRows));
  if (tff)
    yadif_pair_kernel<1><<<grid, kBlock, 0, s>>>((const float4 *)prev, (const float4 *)cur, (const float4 *)next, w, h, skip,
                                                 (float4 *)out0, (float4 *)out1, image_nt((size_t)w * h * 16));
  else
    yadif_pair_kernel<0><<<grid, kBlock, 0, s>>>((const float4 *)prev, (const float4 *)cur, (const float4 *)next, w, h, skip,
                                                 (float4 *)out0, (float4 *)out1, image_nt((size_t)w * h * 16));
  return hipGetLastError();
}

hipError_t launch_transform(hipStream_t s, const void *in, int iw, int ih, const void *m9, void *out, int ow, int oh) {
  dim3 grid(div_up(ow, 64), div_up(oh, 4));
  transform_kernel<<<grid, kBlock, 0, s>>>((const float4 *)in, iw, ih, (const float *)m9, (float4 *)out, ow, oh, image_nt((size_t)ow * oh * 16));
  return hipGetLastError();
}

hipError_t launch_resize(hipStream_t s, const void *in, int iw, int ih, float scale, float ox, float oy,
                         const void *flip4, void *out, int ow, int oh) {
  dim3 grid(div_up(ow, 64), div_up(oh, 4));
  resize_kernel<<<grid, kBlock, 0, s>>>((const float4 *)in, iw, ih, scale, ox, oy, (const float *)flip4, (float4 *)out,
                                        ow, oh, image_nt((size_t)ow * oh * 16));
  return hipGetLastError();
}

hipError_t launch_combine(hipStream_t s, int n, const CombineArgs &args) {
  CombineArgs a = args;
  a.nt = image_nt(a.npx * 16);
  const uint32_t grid = stream_grid(a.npx);
  switch (n) {
    case 2: combine_kernel<2><<<grid, kBlock, 0, s>>>(a); break;
    case 3: combine_kernel<3><<<grid, kBlock, 0, s>>>(a); break;
    case 4: combine_kernel<4><<<grid, kBlock, 0, s>>>(a); break;
    case 5: combine_kernel<5><<<grid, kBlock, 0, s>>>(a); break;
    case 6: combine_kernel<6><<<grid, kBlock, 0, s>>>(a); break;
    case 7: combine_kernel<7><<<grid, kBlock, 0, s>>>(a); break;
    case 8: combine_kernel<8><<<grid, kBlock, 0, s>>>(a); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t launch_dissolve(hipStream_t s, const void *in0, const void *in1, float mix, int w, int h, void *out) {
  const size_t npx = (size_t)w * h;
  dissolve_kernel<<<stream_grid(npx), kBlock, 0, s>>>((const float4 *)in0, (const float4 *)in1, mix, npx, (float4 *)out, image_nt(npx * 16));
  return hipGetLastError();
}

hipError_t launch_twipe(hipStream_t s, const void *in0, const void *in1, const void *mask, int w, int h, void *out) {
  const size_t npx = (size_t)w * h;
  twipe_kernel<<<stream_grid(npx), kBlock, 0, s>>>((const float4 *)in0, (const float4 *)in1, (const float4 *)mask, npx,
                                                   (float4 *)out, image_nt(npx * 16));
  return hipGetLastError();
}

// packed f32 RGB (12 bytes per pixel, alpha == 1 implied: the de-interlacing reader's field images) -> f32 RGBA; src and dst do not overlap
__global__ __launch_bounds__(kBlock) void rgb_unpack_kernel(const float *__restrict__ src, float4 *__restrict__ dst, size_t npx) {
  for (size_t p = (size_t)blockIdx.x * kBlock + threadIdx.x; p < npx; p += (size_t)gridDim.x * kBlock)
    dst[p] = make_float4(src[3 * p], src[3 * p + 1], src[3 * p + 2], 1.0f);
}
hipError_t launch_rgb_unpack(hipStream_t s, const void *packed, void *rgba, size_t npx) {
  rgb_unpack_kernel<<<stream_grid(npx), kBlock, 0, s>>>((const float *)packed, (float4 *)rgba, npx);
  return hipGetLastError();
}

hipError_t launch_wipe(hipStream_t s, const void *in0, const void *in1, float wipe, int w, int h, void *out) {
  wipe_kernel<<<stream_grid((size_t)w * h), kBlock, 0, s>>>((const float4 *)in0, (const float4 *)in1, wipe, w, h,
                                                           (float4 *)out, image_nt((size_t)w * h * 16));
  return hipGetLastError();
}

}  // namespace ph
