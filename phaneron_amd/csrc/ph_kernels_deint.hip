// ph_kernels_deint.hip - v210 unpack + yadif for both fields of a frame, as one kernel.
//
// The per-frame job batch of a de-interlacing channel is, per layer, ToRGBA (v210.ts:25-111) on the newest frame
// and then Yadif twice over the window of three RGBA frames (yadif.ts:100-145, send_field: parity 1 ^ tff, then
// parity tff).  Run as separate kernels that is 33 MB of f32 RGBA written per frame and 3 x 33 MB read back per
// field at 1080i.  Here the window stays in v210 (5.5 MB per frame): every row the filter needs is unpacked,
// matrixed and passed through the reader's gamma table ON THE FLY - three times over a frame's life, as next, cur
// and prev - and only the two de-interlaced frames are written.  HBM traffic per layer and frame: 3 x 5.5 MB in,
// 2 x 33 MB out, instead of 5.5 + 33 + 3 x 33 + 2 x 33.  The arithmetic per value is the reader's and the
// filter's own, so out_parity0 / out_parity1 are bit-identical to ph_v210_read x 3 -> ph_yadif x 2.
//
// Shape: the reader's table fills the LDS, so one 1024-lane workgroup per CU, persistent; the unit of work is one
// WAVE walking one strip (58 columns x R rows) of one layer with the five-row windows of the three frames in
// registers (RGB only: the reader's alpha is the constant 1).  Nothing is shared between waves - the spatial
// predictor's x +- 3 taps come from the neighbouring lanes through ds_bpermute (the wave carries a 3-column halo
// on each side) - so there is no barrier after the table load.
#include "ph_device.h"
#include "ph_kernels.h"
#include "ph_ldslut.h"
#include "ph_yadif.h"

#include <type_traits>

#pragma clang fp contract(off)

namespace ph {

constexpr int kDeintCols = 64 - 6;  // columns a wave produces
#ifndef PH_DEINT_BLOCK
#define PH_DEINT_BLOCK 1024
#endif
// lanes per workgroup.  Four 1080i layers, packed-RGB fields: 1024 lanes 113 us per frame, 768 lanes (131 registers, no spill) 119 us,
// 512 lanes 128 us - half the waves cost 13 %: the kernel sits on VALU issue and the LDS pipe, not on latency
constexpr int kDeintBlock = PH_DEINT_BLOCK;

struct Rgb {
  float r, g, b;
};

// the lane's pixel of one v210 line: which 32-bit word holds its Y, its pair's Cb and Cr, and at which bit
// (v210.ts:58-63) depends on x only, so the lane loads exactly those three words of every line (three dword loads
// at lane-constant offsets from the line's start - no 16-byte load followed by seven selects)
struct LanePick {
  uint32_t off_y, off_cb, off_cr;  // byte offsets inside a line
  uint32_t sy, scb, scr;
};
__device__ __forceinline__ LanePick lane_pick(uint32_t x) {
  const uint32_t g = x / 6, j = x - 6 * g, pr = j >> 1;
  const uint32_t wy = (j == 0) ? 0u : (j < 3) ? 1u : (j == 3) ? 2u : 3u;
  const uint32_t wcb = pr, wcr = pr == 0 ? 0u : pr + 1u;
  LanePick p;
  p.off_y = 16u * g + 4u * wy, p.off_cb = 16u * g + 4u * wcb, p.off_cr = 16u * g + 4u * wcr;
  p.sy = (j == 0 || j == 3) ? 10u : (j == 1 || j == 4) ? 0u : 20u;
  p.scb = 10u * pr;
  p.scr = pr == 0 ? 20u : pr == 1 ? 0u : 10u;
  return p;
}
// a planar 4:2:2 frame (yuv422p10.ts:60-72, yuv422p8.ts): Y at [line][x], Cb / Cr at [line][x / 2] of their planes, samples of
// BPS bytes taken whole; the three "words" of a pixel are its three samples
template <int BPS>
__device__ __forceinline__ LanePick lane_pick_planar(uint32_t x) {
  LanePick p;
  p.off_y = x * BPS, p.off_cb = p.off_cr = (x >> 1) * BPS;
  p.sy = p.scb = p.scr = 0;
  return p;
}
struct FramePlanes {  // a frame of the window as buffer resources: v210 uses y only
  __amdgpu_buffer_rsrc_t y, u, v;
};
// the three words of the lane's pixel, as loaded; unpacking them is a separate step so that the loads of the next
// row can be issued BEFORE the filter of the current one and consumed after it (a frame's rows come from HBM)
struct RawPx {
  uint32_t wy, wcb, wcr;
};
// buffer loads: the line's start travels as the scalar offset, the lane's word as the vector offset - no address arithmetic
__device__ __forceinline__ RawPx load_row_px(__amdgpu_buffer_rsrc_t frame, uint32_t pitch_bytes, int line, const LanePick &p) {
  const int row = (int)((uint32_t)line * pitch_bytes);  // uniform
  return RawPx{(uint32_t)__builtin_amdgcn_raw_buffer_load_b32(frame, (int)p.off_y, row, 0),
               (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(frame, (int)p.off_cb, row, 0),
               (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(frame, (int)p.off_cr, row, 0)};
}
// PACK: 1 yuv422p10 (16-bit samples), 2 yuv422p8, 3 yuv420p (a chroma line serves two luma lines: yuv420p.ts), 4 nv12 (the same with Cb and Cr
// interleaved in one plane: nv12.ts:61-74; FramePlanes::v is u, the lane's offsets the pair's two bytes)
template <int PACK>
__device__ __forceinline__ RawPx load_row_planar(const FramePlanes &f, uint32_t pitch_y, int line, const LanePick &p) {
  const int row_y = (int)((uint32_t)line * pitch_y);  // uniform
  const int row_c = PACK >= 3 ? (int)(((uint32_t)line >> 1) * (PACK == 4 ? pitch_y : pitch_y >> 1)) : (int)((uint32_t)line * (pitch_y >> 1));
  if (PACK == 1)
    return RawPx{(uint32_t)(uint16_t)__builtin_amdgcn_raw_buffer_load_b16(f.y, (int)p.off_y, row_y, 0),
                 (uint32_t)(uint16_t)__builtin_amdgcn_raw_buffer_load_b16(f.u, (int)p.off_cb, row_c, 0),
                 (uint32_t)(uint16_t)__builtin_amdgcn_raw_buffer_load_b16(f.v, (int)p.off_cr, row_c, 0)};
  return RawPx{(uint32_t)(uint8_t)__builtin_amdgcn_raw_buffer_load_b8(f.y, (int)p.off_y, row_y, 0),
               (uint32_t)(uint8_t)__builtin_amdgcn_raw_buffer_load_b8(f.u, (int)p.off_cb, row_c, 0),
               (uint32_t)(uint8_t)__builtin_amdgcn_raw_buffer_load_b8(f.v, (int)p.off_cr, row_c, 0)};
}
template <bool STD, int PACK = 0>
__device__ __forceinline__ Rgb unpack_px(const RawPx &r, const LanePick &p, const ReadK &k, const LutK &lk) {
  // (a planar sample is converted whole, as the reference does: yuv422p10.ts:66-68)
  const float yf = PACK ? (float)r.wy : (float)((r.wy >> p.sy) & 0x3ff);
  const float cbf = PACK ? (float)r.wcb : (float)((r.wcb >> p.scb) & 0x3ff);
  const float crf = PACK ? (float)r.wcr : (float)((r.wcr >> p.scr) & 0x3ff);
  const float4 v = read_px_lds<STD>(yf, cbf, crf, k, lk);
  return Rgb{v.x, v.y, v.z};
}

__device__ __forceinline__ float lane_tap(float v, uint32_t lane, int d) {
  return __int_as_float(__builtin_amdgcn_ds_bpermute((int)((lane + (uint32_t)d) << 2), __float_as_int(v)));
}

#define PH_RGB(v, c) ((c) == 0 ? (v).r : (c) == 1 ? (v).g : (v).b)
// PH_DEINT_PRICE_INDEX builds (tools/config3_price.py; timing only, never shipped): what the de-interlacing reader would gain if a
// field's KEPT lines left as three 16-bit table indices instead of three floats (first candidate)
#ifndef PH_DEINT_PRICE_INDEX
#define PH_DEINT_PRICE_INDEX 0
#endif

template <int TFF, bool STD, int PACK = 0>
__device__ __forceinline__ void v210_yadif_pair_body(const DeintArgs &a, const ReadK &k, const LutK &lk) {
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int w = (int)a.width, h = (int)a.height;
  const uint32_t tasks = (uint32_t)a.n * a.strips * a.col_blocks;
  for (uint32_t t = blockIdx.x * (kDeintBlock / 64) + wave; t < tasks; t += gridDim.x * (kDeintBlock / 64)) {
    const uint32_t cb = t % a.col_blocks, rest = t / a.col_blocks, strip = rest % a.strips, l = rest / a.strips;
    // bytes per line: a v210 line of quads_pitch quads, or (planar) quads_pitch luma samples of 1 or 2 bytes
    const uint32_t line_bytes = PACK == 0 ? a.quads_pitch * 16u : a.quads_pitch * (PACK == 1 ? 2u : 1u);
    const int frame_bytes = (int)(line_bytes * a.height);
    auto planes = [&](const uint4 *y, const void *u, const void *v) __attribute__((always_inline)) {
      FramePlanes f;
      f.y = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4 *>(y), 0, frame_bytes, 0x00020000);
      if (PACK == 4) {  // nv12: (height + 1) / 2 lines of CbCr pairs, a luma line's bytes each
        f.u = f.v = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(u), 0, (int)(line_bytes * ((a.height + 1u) >> 1)), 0x00020000);
      } else if (PACK) {  // 4:2:2: half a luma line per line; 4:2:0: that for every other line
        const int chroma_bytes = PACK == 3 ? (int)((line_bytes >> 1) * ((a.height + 1u) >> 1)) : frame_bytes / 2;
        f.u = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(u), 0, chroma_bytes, 0x00020000);
        f.v = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(v), 0, chroma_bytes, 0x00020000);
      } else {
        f.u = f.v = f.y;
      }
      return f;
    };
    const FramePlanes prev = planes(a.prev[l], PACK ? a.prev_u[l] : nullptr, PACK ? a.prev_v[l] : nullptr);
    const FramePlanes cur = planes(a.cur[l], PACK ? a.cur_u[l] : nullptr, PACK ? a.cur_v[l] : nullptr);
    const FramePlanes next = planes(a.next[l], PACK ? a.next_u[l] : nullptr, PACK ? a.next_v[l] : nullptr);
    float4 *__restrict__ out0 = a.out0[l], *__restrict__ out1 = a.out1[l];
    const int xr = (int)(cb * kDeintCols) - 3 + (int)lane, x = clampi(xr, 0, w - 1);  // CLAMP_TO_EDGE
    const bool emit = lane >= 3 && lane < 64 - 3 && xr < w;
    LanePick pick = PACK == 0 ? lane_pick((uint32_t)x) : PACK == 1 ? lane_pick_planar<2>((uint32_t)x) : lane_pick_planar<1>((uint32_t)x);
    if (PACK == 4) pick.off_cb = (uint32_t)x & ~1u, pick.off_cr = pick.off_cb + 1u;
    const int y0 = (int)(strip * a.rows_per_strip), y_end = (y0 + (int)a.rows_per_strip < h) ? y0 + (int)a.rows_per_strip : h;
    auto raw = [&](const FramePlanes &frame, int y) {
      return PACK == 0 ? load_row_px(frame.y, line_bytes, clampi(y, 0, h - 1), pick) : load_row_planar<PACK>(frame, line_bytes, clampi(y, 0, h - 1), pick);
    };
    auto row = [&](const FramePlanes &frame, int y) { return unpack_px<STD, PACK>(raw(frame, y), pick, k, lk); };
    // rows y - 2 .. y + 2 of each frame live in a RING of five registers: the step with rotation R finds row y - 2 + i
    // in slot (R + i) % 5 and refills slot R % 5 (row y - 2, no longer needed) with row y + 3 - no window shifts
    // (36 v_mov per row otherwise).  Ten steps (lcm of the ring and of the even / odd row roles) make one loop body.
    Rgb C[5], P[5], N[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) C[i] = row(cur, y0 - 2 + i), P[i] = row(prev, y0 - 2 + i), N[i] = row(next, y0 - 2 + i);
    // one row: interpolated into the output whose parity is (y & 1) ^ 1, copied into the other; SECOND
    // (yadifCl.ts:143, !(parity ^ tff)) is a compile-time constant of the row's evenness (as yadif_pair_kernel)
    auto step = [&](int y, auto rot_tag) {
      constexpr int R = decltype(rot_tag)::value;
      constexpr bool even = (R & 1) == 0;                      // y0 is even and R counts rows from y0 (mod 10)
      constexpr bool second = even ? (TFF != 0) : (TFF == 0);  // even row: parity-1 output, !(1 ^ tff); odd: !(0 ^ tff)
      float4 *__restrict__ out_interp = even ? out1 : out0, *__restrict__ out_copy = even ? out0 : out1;
#define PH_W(A, i) A[(R + (i)) % 5]
      const RawPx rc = raw(cur, y + 3), rp = raw(prev, y + 3), rn = raw(next, y + 3);  // in flight during the filter
      float res[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float up = PH_RGB(PH_W(C, 1), c), dn = PH_RGB(PH_W(C, 3), c);
        const float sp = yadif_spatial(lane_tap(up, lane, -3), lane_tap(up, lane, -2), lane_tap(up, lane, -1), up,
                                       lane_tap(up, lane, 1), lane_tap(up, lane, 2), lane_tap(up, lane, 3),
                                       lane_tap(dn, lane, -3), lane_tap(dn, lane, -2), lane_tap(dn, lane, -1), dn,
                                       lane_tap(dn, lane, 1), lane_tap(dn, lane, 2), lane_tap(dn, lane, 3));
        // second field: s0 = cur, s1 = next; first field: s0 = prev, s1 = cur (yadifCl.ts:146-151)
        const float c0 = PH_RGB(PH_W(C, 0), c), c2 = PH_RGB(PH_W(C, 2), c), c4 = PH_RGB(PH_W(C, 4), c);
        const float e0 = second ? PH_RGB(PH_W(N, 0), c) : PH_RGB(PH_W(P, 0), c), e1 = second ? PH_RGB(PH_W(N, 2), c) : PH_RGB(PH_W(P, 2), c),
                    e2 = second ? PH_RGB(PH_W(N, 4), c) : PH_RGB(PH_W(P, 4), c);
        res[c] = yadif_temporal(PH_RGB(PH_W(P, 1), c), PH_RGB(PH_W(P, 3), c), second ? c0 : e0, second ? c2 : e1, second ? c4 : e2,
                                PH_RGB(PH_W(C, 1), c), PH_RGB(PH_W(C, 3), c), second ? e0 : c0, second ? e1 : c2, second ? e2 : c4,
                                PH_RGB(PH_W(N, 1), c), PH_RGB(PH_W(N, 3), c), sp, a.skip);
      }
      if (emit) {
        if (a.rgb12) {  // uniform: packed RGB for the 2 x 2-block compositor (ph_kernels_up.hip); alpha == 1 is implied
          typedef float ph_f3v __attribute__((ext_vector_type(3)));
          ph_f3v *const pc = reinterpret_cast<ph_f3v *>(reinterpret_cast<char *>(out_copy) + ((size_t)y * w + xr) * 12);
          ph_f3v *const pi = reinterpret_cast<ph_f3v *>(reinterpret_cast<char *>(out_interp) + ((size_t)y * w + xr) * 12);
          const ph_f3v vc = {PH_W(C, 2).r, PH_W(C, 2).g, PH_W(C, 2).b}, vi = {res[0], res[1], res[2]};
#if PH_DEINT_PRICE_INDEX  // timing only (tools/config3_price.py): the kept line stored as three 16-bit values - half the bytes
          typedef uint16_t ph_h3v __attribute__((ext_vector_type(3)));
          *reinterpret_cast<ph_h3v *>(reinterpret_cast<char *>(out_copy) + ((size_t)y * w + xr) * 6) =
              ph_h3v{(uint16_t)__float_as_uint(vc.x), (uint16_t)__float_as_uint(vc.y), (uint16_t)__float_as_uint(vc.z)};
          if (a.nt) __builtin_nontemporal_store(vi, pi);
          else *pi = vi;
#else
          if (a.nt) __builtin_nontemporal_store(vc, pc), __builtin_nontemporal_store(vi, pi);
          else *pc = vc, *pi = vi;
#endif
        } else {
          store_image(out_copy + (size_t)y * w + xr, make_float4(PH_W(C, 2).r, PH_W(C, 2).g, PH_W(C, 2).b, 1.0f), a.nt);  // yadifCl.ts:117-121
          store_image(out_interp + (size_t)y * w + xr, make_float4(res[0], res[1], res[2], 1.0f), a.nt);                  // :164 alpha from cur
        }
      }
      PH_W(C, 0) = unpack_px<STD, PACK>(rc, pick, k, lk), PH_W(P, 0) = unpack_px<STD, PACK>(rp, pick, k, lk), PH_W(N, 0) = unpack_px<STD, PACK>(rn, pick, k, lk);
#undef PH_W
    };
    for (int y = y0; y < y_end; y += 10) {  // y0 is even
      step(y, std::integral_constant<int, 0>{});
      if (y + 1 < y_end) step(y + 1, std::integral_constant<int, 1>{});
      if (y + 2 < y_end) step(y + 2, std::integral_constant<int, 2>{});
      if (y + 3 < y_end) step(y + 3, std::integral_constant<int, 3>{});
      if (y + 4 < y_end) step(y + 4, std::integral_constant<int, 4>{});
      if (y + 5 < y_end) step(y + 5, std::integral_constant<int, 5>{});
      if (y + 6 < y_end) step(y + 6, std::integral_constant<int, 6>{});
      if (y + 7 < y_end) step(y + 7, std::integral_constant<int, 7>{});
      if (y + 8 < y_end) step(y + 8, std::integral_constant<int, 8>{});
      if (y + 9 < y_end) step(y + 9, std::integral_constant<int, 9>{});
    }
  }
}

template <int TFF, int PACK = 0>
__global__ __launch_bounds__(kDeintBlock) void v210_yadif_pair_kernel(DeintArgs a) {
  const ReadK k = load_read_k(a.cm, a.gm);
  const LutK lk = make_lut_k(a.lut);
  lds_lut_load<kDeintBlock>(a.lut);
  __syncthreads();
  if (ycbcr_matrix_is_standard(k))  // every matrix colourMaths produces (ph_ldslut.h): 8 operations per pixel instead of 12
    v210_yadif_pair_body<TFF, true, PACK>(a, k, lk);
  else
    v210_yadif_pair_body<TFF, false, PACK>(a, k, lk);
}

hipError_t launch_v210_yadif_pair(hipStream_t s, DeintArgs a, int tff, uint32_t num_cus) {
  // strip height: even, and such that the waves of the chip are filled in whole rounds - cost ~ rounds x (R + 4 halo rows)
  const uint32_t slots = num_cus * (kDeintBlock / 64);
  a.col_blocks = (a.width + kDeintCols - 1) / kDeintCols;
  uint32_t best_r = 16;
  uint64_t best_cost = ~0ull;
  for (uint32_t r = 8; r <= 64; r += 2) {
    const uint32_t strips = (a.height + r - 1) / r;
    const uint64_t tasks = (uint64_t)a.n * strips * a.col_blocks, rounds = (tasks + slots - 1) / slots;
    const uint64_t cost = rounds * (r + 4);
    if (cost < best_cost) best_cost = cost, best_r = r;
  }
  a.nt = image_nt((size_t)a.width * a.height * (a.rgb12 ? 12 : 16));
  a.rows_per_strip = best_r;
  a.strips = (a.height + best_r - 1) / best_r;
  const uint32_t tasks = (uint32_t)a.n * a.strips * a.col_blocks;
  const uint32_t want = (tasks + kDeintBlock / 64 - 1) / (kDeintBlock / 64);
  const uint32_t grid = want < num_cus ? want : num_cus;
  auto go = [&](auto kernel) -> hipError_t {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kMaxDynamicLds);
    if (e != hipSuccess) return e;
    kernel<<<grid, kDeintBlock, a.lut.bytes, s>>>(a);
    return hipGetLastError();
  };
  switch (a.pack) {
    case 0: return tff ? go(v210_yadif_pair_kernel<1, 0>) : go(v210_yadif_pair_kernel<0, 0>);
    case 1: return tff ? go(v210_yadif_pair_kernel<1, 1>) : go(v210_yadif_pair_kernel<0, 1>);
    case 2: return tff ? go(v210_yadif_pair_kernel<1, 2>) : go(v210_yadif_pair_kernel<0, 2>);
    case 3: return tff ? go(v210_yadif_pair_kernel<1, 3>) : go(v210_yadif_pair_kernel<0, 3>);
    case 4: return tff ? go(v210_yadif_pair_kernel<1, 4>) : go(v210_yadif_pair_kernel<0, 4>);
  }
  return hipErrorInvalidValue;
}

}  // namespace ph
