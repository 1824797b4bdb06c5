// ph_kernels_up.hip - the compositor for MAGNIFYING placements: [transform] x N -> combine_N -> v210 write as one kernel,
// each lane producing a 2 x 2 block of output pixels.
//
// The pixel-per-lane compositor (ph_kernels_lds.hip) loads four 16-byte taps per layer and output pixel; for BASELINE
// config 3 (four 1080 sources shown at 2160) that is 256 bytes through the texture addressers per output pixel, and
// they - not HBM, not the VALU - were what the kernel waited for.  When a layer is magnified without rotation
// (Mixer's default fill of an HD source on a UHD channel: producer/mixer.ts:209-223, transform.ts:36-59), neighbouring
// output pixels share their taps: the four pixels (x, x + 1) x (y, y + 1) with x, y even draw all sixteen taps from a
// 3 x 3 patch of the source.  A lane owns such a block, loads the nine texels once per layer and picks each pixel's
// 2 x 2 sub-patch with selects (columns, per lane) and a uniform branch (rows): 2.25 taps per pixel instead of 4, and with
// sources stored as packed RGB (12 bytes per texel: alpha == 1 is implied for a de-interlaced v210 source,
// ph_v210_yadif_pair_fmt) 27 bytes per layer and pixel instead of 64.  Arithmetic and its order are those of
// transform.ts / combine.ts / v210.ts, pixel by pixel: results are bit-identical to the separate kernels.
//
// Shape: only the writer's table is needed (one phase); one 1024-lane workgroup per CU, persistent.  A wave step is
// 126 columns x 2 rows: 63 lanes x (2 x 2) - 21 v210 quads per row, three lanes to a quad.  The lanes of a quad hand
// their code values to each other with two DPP moves per row (no LDS staging): lane A (pixels 0, 1) and lane C (4, 5)
// each store half of the packed quad, lane B (2, 3) only gives.
#include "ph_device.h"
#include "ph_kernels.h"
#include "ph_ldslut.h"

#include <type_traits>

#pragma clang fp contract(off)

#ifndef PH_UP_GROUP_ROWS
#define PH_UP_GROUP_ROWS 16
#endif

namespace ph {

#ifndef PH_UP_BLOCK
#define PH_UP_BLOCK 1024
#endif
// lanes per workgroup (one workgroup per CU: the table fills the LDS).  Measured at 2160p x 4 layers: 1024 lanes 64.7 us, 768 lanes
// (three waves per SIMD, 168 registers each) 65.7 us, 512 lanes 78 us; with the next layer's patch requested before this layer's is
// filtered (two patches in flight) 768 lanes 69 us, 512 lanes 72.5 us, 1024 lanes 92 us (31 spilled registers): not a kernel that waits for its loads
constexpr int kUpBlock = PH_UP_BLOCK;
constexpr uint32_t kUpCols = 126;              // output columns of a wave step
constexpr uint32_t kUpOutside = 0x40000000u;   // offsets of texels outside the image: beyond any num_records (images < 1 GiB)

typedef uint32_t ph_u32x3 __attribute__((ext_vector_type(3)));
typedef uint32_t ph_u32x4 __attribute__((ext_vector_type(4)));

struct UpTexel {
  float r, g, b, a;
};
template <bool RGB12>
__device__ __forceinline__ UpTexel up_load(__amdgpu_buffer_rsrc_t img, uint32_t off) {  // outside: 0 = the border colour
  if (RGB12) {
    const ph_u32x3 v = __builtin_amdgcn_raw_buffer_load_b96(img, (int)off, 0, 0);
    UpTexel t;  // .a is not part of a packed-RGB texel (UpPend::cin / rin)
    t.r = __uint_as_float(v.x), t.g = __uint_as_float(v.y), t.b = __uint_as_float(v.z);
    return t;
  }
  const ph_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(img, (int)off, 0, 0);
  return UpTexel{__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
}
__device__ __forceinline__ UpTexel up_pick(bool second, const UpTexel &t0, const UpTexel &t1) {
  return UpTexel{second ? t1.r : t0.r, second ? t1.g : t0.g, second ? t1.b : t0.b, second ? t1.a : t0.a};
}

struct UpShare {  // as ChanShare (ph_kernels_chan.hip): units dealt XCD-aware in groups of output rows, divisions by reciprocal
  uint32_t units, upg, upr, xcd, v0, vstep, vend, slots;
  bool banded;
};
__device__ __forceinline__ UpShare up_share(const UpArgs &a) {
  UpShare s;
  s.upr = (a.cover_w + kUpCols - 1u) / kUpCols;  // wave steps per row pair
  s.units = s.upr * ((a.lines + 1u) / 2u) * a.jobs;  // a further job's row pairs follow the one before's
  s.upg = (uint32_t)(PH_UP_GROUP_ROWS / 2) * s.upr;
  s.banded = (gridDim.x & 7u) == 0;
  s.xcd = 0, s.v0 = blockIdx.x * (kUpBlock / 64), s.vstep = gridDim.x * (kUpBlock / 64), s.vend = s.units;
  if (s.banded) {
    s.xcd = blockIdx.x & 7u;
    const uint32_t groups = (s.units + s.upg - 1u) / s.upg, mine = (groups + 7u - s.xcd) / 8u;
    s.v0 = (blockIdx.x >> 3) * (kUpBlock / 64), s.vstep = (gridDim.x >> 3) * (kUpBlock / 64), s.vend = mine * s.upg;
  }
  return s;
}
__device__ __forceinline__ uint32_t up_unit(const UpArgs &a, const UpShare &s, uint32_t v) {
  if (!s.banded) return v < s.units ? v : ~0u;
  const uint32_t gi = __umulhi(v, a.magic_upg);  // v / upg
  const uint32_t unit = (gi * 8u + s.xcd) * s.upg + (v - gi * s.upg);
  return unit < s.units ? unit : ~0u;
}

// one output pixel of a layer: the sampler's filter on its 2 x 2 sub-patch with the four weights of OpenCL 1.2 8.2 as DESIGN.md
// section 2 fixes them (w00 = (1 - a)(1 - b), w10 = a (1 - b), w01 = (1 - a) b, w11 = a b), then combine.ts:45-65 into the pixel's accumulator
struct UpAcc {
  float r, g, b;
};
struct UpWeights {
  float w00, w10, w01, w11;
};
__device__ __forceinline__ UpWeights up_weights(float wa, float wb) {
  const float oma = 1.0f - wa, omb = 1.0f - wb;
  return UpWeights{oma * omb, wa * omb, oma * wb, wa * wb};
}
// `INSIDE` (packed RGB, every texel of every lane's patch inside the image): each texel's alpha is exactly 1, so the filtered alpha
// is the sum of the four weights and 1 - alpha depends on the block's geometry only (UpGeo::kk, computed with it).
// A pixel of a layer in two steps: its filtered colour with what it leaves of the layers below (kk = 1 - alpha) ...
struct UpColour {
  float r, g, b, kk;
};
template <bool RGB12, bool INSIDE, bool FIRST>
__device__ __forceinline__ UpColour up_blend(const UpTexel &t00, const UpTexel &t10, const UpTexel &t01, const UpTexel &t11, const UpWeights &w,
                                             float kk_inside) {
  UpColour c;
  c.r = ((w.w00 * t00.r + w.w10 * t10.r) + w.w01 * t01.r) + w.w11 * t11.r;
  c.g = ((w.w00 * t00.g + w.w10 * t10.g) + w.w01 * t01.g) + w.w11 * t11.g;
  c.b = ((w.w00 * t00.b + w.w10 * t10.b) + w.w01 * t01.b) + w.w11 * t11.b;
  c.kk = kk_inside;  // (the result's alpha is never used by the writer)
  if (!FIRST && !INSIDE) c.kk = 1.0f - (((w.w00 * t00.a + w.w10 * t10.a) + w.w01 * t01.a) + w.w11 * t11.a);
  return c;
}
// ... then combine.ts:45-65 into the pixel's accumulator.  The bottom layer is taken as it is (the kernel peels it off the layer loop);
// above it acc = fma(acc, kk, colour) is the three-operand v_fma_f32 with the accumulator as destination, issued once per pixel
// outside any branch: the accumulators stay in their registers over the layer loop (LLVM's v_fmac into the fresh colour costs a
// v_mov per accumulator and layer, an accumulator merged from two paths another one)
template <bool FIRST>
__device__ __forceinline__ void up_over(const UpColour &c, UpAcc &acc) {
  if (FIRST) {
    acc.r = c.r, acc.g = c.g, acc.b = c.b;
  } else {
    asm volatile("v_fma_f32 %0, %0, %3, %4\n\tv_fma_f32 %1, %1, %3, %5\n\tv_fma_f32 %2, %2, %3, %6"
                 : "+v"(acc.r), "+v"(acc.g), "+v"(acc.b)
                 : "v"(c.kk), "v"(c.r), "v"(c.g), "v"(c.b));
  }
}

// where a wave step lies: 126 columns x 2 rows; the lane's block starts at (x0, line[0])
static_assert(kMaxUpJobs == 4, "up_step finds a row pair's job by three comparisons");
struct UpStep {
  uint32_t job;  // uniform: which of the launch's jobs the row pair belongs to
  uint32_t x0, li[2], line[2];
  float px[2], py[2];
  bool live;
};
// A wave's steps: round k of the share's units is [k * vstep, (k + 1) * vstep), the wave's place in a round v0 + wave.  The LAST round is
// usually partial (2160p: 8.2 rounds) and its units are dealt one per SIMD instead - workgroup first, wave second (the waves of a
// workgroup go round its four SIMDs) - so that the tail runs as single waves on otherwise idle SIMDs all over the chip, not as a few
// CUs with sixteen waves each doing a ninth step.
__device__ __forceinline__ bool up_step(const UpArgs &a, const UpShare &sh, uint32_t &base, uint32_t wave, uint32_t lane, UpStep &st) {
  while (base < sh.vend) {
    const uint32_t left = sh.vend - base, nwg = sh.vstep / (kUpBlock / 64), wg = sh.v0 / (kUpBlock / 64);
    const uint32_t pos = left >= sh.vstep ? sh.v0 + wave : wg + nwg * wave;
    const uint32_t v = base + pos;
    base += sh.vstep;
    if (pos >= left) continue;  // uniform
    const uint32_t unit = up_unit(a, sh, v);
    if (unit == ~0u) continue;  // uniform
    uint32_t rp = sh.upr == 1u ? unit : __umulhi(unit, a.magic_upr);  // unit / upr
    const uint32_t col_unit = unit - rp * sh.upr, rp_per_job = (a.lines + 1u) / 2u;
    st.job = (rp >= rp_per_job ? 1u : 0u) + (rp >= 2u * rp_per_job ? 1u : 0u) + (rp >= 3u * rp_per_job ? 1u : 0u);  // kMaxUpJobs == 4
    rp -= st.job * rp_per_job;
    st.x0 = col_unit * kUpCols + 2u * lane;  // even
    st.live = lane < 63u && st.x0 < a.out_w;               // lane 63 has no quad; the row's last step may be short
#pragma unroll
    for (int dy = 0; dy < 2; ++dy) {
      st.li[dy] = 2u * rp + (uint32_t)dy < a.lines ? 2u * rp + (uint32_t)dy : 2u * rp;  // an odd field's last row is its own partner
      st.line[dy] = a.first_line + st.li[dy] * a.line_step;
      st.py[dy] = (float)(int)st.line[dy] / (float)(int)a.out_h - 0.5f;  // transform.ts:53
    }
#pragma unroll
    for (int dx = 0; dx < 2; ++dx) st.px[dx] = (float)(int)(st.x0 + (uint32_t)dx) / (float)(int)a.out_w - 0.5f;
    return true;
  }
  return false;
}

// Where a layer's 3 x 3 patch lies for the lane's block and how each of the four pixels weighs its 2 x 2 sub-patch: everything
// that depends on the placement and the source's SIZE but not on its pixels.  Layers of one size under one placement (the
// four full-frame HD sources of a UHD channel) share it: computed for the first, reused for the rest.
struct UpGeo {
  uint32_t coff[3];  // per lane: byte offsets of the patch's columns inside a row, or kUpOutside
  uint32_t roff[3];  // uniform: byte offsets of its rows, or kUpOutside
  UpWeights w[2][2];  // [output row][output column]
  float kk[2][2];    // packed RGB with every texel inside: 1 - (the filtered alpha = the sum of the weights, each times exactly 1)
  bool d1;           // per lane: the right pixel's first tap is one texel further than the left pixel's
  bool dj;           // uniform: likewise the lower row's
  bool all_inside;   // uniform (packed RGB): every texel of every lane's patch is inside the image
  uint32_t cin;      // packed RGB, per lane: bit c = patch column c is inside the image
  uint32_t rin;      // packed RGB, uniform: bit r = patch row r is inside
};
template <bool RGB12>
__device__ __forceinline__ UpGeo up_geo(const UpLayer &L, const UpStep &st) {
  constexpr uint32_t kTexel = RGB12 ? 12u : 16u;
  UpGeo g;
  // columns (per lane): transform.ts:53-57 with m1 == 0 (host-checked): fma(py, 0, px * m0) == px * m0 for every finite py
  uint32_t i0[2];
  float wa[2], wb[2];
#pragma unroll
  for (int dx = 0; dx < 2; ++dx) {
    const float sx = dot3(L.m[0], L.m[1], L.m[2], st.px[dx], st.py[0], 1.0f) + 0.5f;
    const float fu = sx * (float)(int)L.w - 0.5f, flu = __builtin_floorf(fu);
    i0[dx] = (uint32_t)(int)flu, wa[dx] = fu - flu;
  }
  // rows (the same for every lane: m3 == 0, host-checked; computed on the lane's own operands, read from one lane)
  uint32_t j0[2];
#pragma unroll
  for (int dy = 0; dy < 2; ++dy) {
    const float sy = dot3(L.m[3], L.m[4], L.m[5], st.px[0], st.py[dy], 1.0f) + 0.5f;
    const float fv = sy * (float)(int)L.h - 0.5f, flv = __builtin_floorf(fv);
    j0[dy] = (uint32_t)__builtin_amdgcn_readfirstlane((int)flv);
    wb[dy] = __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(fv - flv)));
  }
#pragma unroll
  for (int dy = 0; dy < 2; ++dy)
#pragma unroll
    for (int dx = 0; dx < 2; ++dx) {
      g.w[dy][dx] = up_weights(wa[dx], wb[dy]);
      const UpWeights &q = g.w[dy][dx];
      g.kk[dy][dx] = RGB12 ? 1.0f - (((q.w00 + q.w10) + q.w01) + q.w11) : 0.0f;
    }
  // the 3 x 3 patch: columns i0[0] .. + 2 (i0[1] - i0[0] is 0 or 1 for a magnification of 2 or more), rows j0[0] .. + 2
  g.d1 = i0[1] != i0[0];
  g.dj = j0[1] != j0[0];
  bool cin[3], rin[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const uint32_t col = i0[0] + (uint32_t)c;
    cin[c] = col < L.w;
    g.coff[c] = cin[c] ? __umul24(col, kTexel) : kUpOutside;
  }
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const uint32_t row = j0[0] + (uint32_t)r;
    rin[r] = row < L.h;
    g.roff[r] = rin[r] ? row * L.pitch : kUpOutside;  // uniform
  }
  g.all_inside = false, g.cin = 0, g.rin = 0;
  if (RGB12) {  // alpha of a packed-RGB texel: 1 inside the image, 0 (the border colour) outside - made when the patch is filtered
    const bool mine = cin[0] && cin[1] && cin[2];
    g.all_inside = rin[0] && rin[1] && rin[2] && __builtin_amdgcn_ballot_w64(!mine) == 0;  // uniform
    g.cin = (cin[0] ? 1u : 0u) | (cin[1] ? 2u : 0u) | (cin[2] ? 4u : 0u);
    g.rin = (rin[0] ? 1u : 0u) | (rin[1] ? 2u : 0u) | (rin[2] ? 4u : 0u);
  }
  return g;
}
// one layer in two halves: its nine texels are requested, then the four pixels are filtered and combined
struct UpPatch {
  UpTexel P[3][3];
};
template <bool RGB12>
__device__ __forceinline__ void up_fetch(const UpLayer &L, const UpGeo &g, UpPatch &p) {
  const __amdgpu_buffer_rsrc_t img = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(L.ptr), 0, (int)(L.pitch * L.h), 0x00020000);
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) p.P[r][c] = up_load<RGB12>(img, g.roff[r] + g.coff[c]);
}
// PH_UP_PRICE_INDEX builds (tools/config3_price.py; timing only, WRONG pixels, never shipped): what it would cost this kernel if the
// fields' KEPT lines - half of a field's lines are copies of the source frame's - arrived as three 16-bit reader-table indices
// instead of three floats (first candidate): every such texel then needs the reader's three table lookups and its
// gamut matrix here.  Five of a patch's nine texels get that arithmetic (on average 4.5 lie on kept lines).
#ifndef PH_UP_PRICE_INDEX
#define PH_UP_PRICE_INDEX 0
#endif
template <bool RGB12, bool INSIDE, bool FIRST>
__device__ __forceinline__ void up_filter(const UpPatch &p, const UpGeo &g, UpAcc (&acc)[2][2], const LutK &price_lut) {
  UpTexel P[3][3];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      P[r][c] = p.P[r][c];
#if PH_UP_PRICE_INDEX
      if (r == 0 || (r == 2 && c < 2)) {
        const float lr = lds_lut_at_unit(price_lut, P[r][c].r), lg = lds_lut_at_unit(price_lut, P[r][c].g), lb = lds_lut_at_unit(price_lut, P[r][c].b);
        P[r][c].r = dot3(lr, lg, lb, 0.6274f, 0.3293f, 0.0433f), P[r][c].g = dot3(lr, lg, lb, 0.0691f, 0.9195f, 0.0114f);
        P[r][c].b = dot3(lr, lg, lb, 0.0164f, 0.0880f, 0.8956f);
      }
#endif
      if (RGB12) P[r][c].a = (!INSIDE && ((g.rin >> r) & 1u) && ((g.cin >> c) & 1u)) ? 1.0f : 0.0f;  // unused when all are inside
    }
  const bool d1 = g.d1;
  // upper output row: patch rows 0, 1; left pixel: columns 0, 1; right pixel: columns d1, d1 + 1
  up_over<FIRST>(up_blend<RGB12, INSIDE, FIRST>(P[0][0], P[0][1], P[1][0], P[1][1], g.w[0][0], g.kk[0][0]), acc[0][0]);
  up_over<FIRST>(up_blend<RGB12, INSIDE, FIRST>(up_pick(d1, P[0][0], P[0][1]), up_pick(d1, P[0][1], P[0][2]), up_pick(d1, P[1][0], P[1][1]),
                                                up_pick(d1, P[1][1], P[1][2]), g.w[0][1], g.kk[0][1]), acc[0][1]);
  // lower output row: patch rows dj, dj + 1 (a uniform branch: no selects)
  UpColour lo0, lo1;
  if (g.dj) {
    lo0 = up_blend<RGB12, INSIDE, FIRST>(P[1][0], P[1][1], P[2][0], P[2][1], g.w[1][0], g.kk[1][0]);
    lo1 = up_blend<RGB12, INSIDE, FIRST>(up_pick(d1, P[1][0], P[1][1]), up_pick(d1, P[1][1], P[1][2]), up_pick(d1, P[2][0], P[2][1]),
                                         up_pick(d1, P[2][1], P[2][2]), g.w[1][1], g.kk[1][1]);
  } else {
    lo0 = up_blend<RGB12, INSIDE, FIRST>(P[0][0], P[0][1], P[1][0], P[1][1], g.w[1][0], g.kk[1][0]);
    lo1 = up_blend<RGB12, INSIDE, FIRST>(up_pick(d1, P[0][0], P[0][1]), up_pick(d1, P[0][1], P[0][2]), up_pick(d1, P[1][0], P[1][1]),
                                         up_pick(d1, P[1][1], P[1][2]), g.w[1][1], g.kk[1][1]);
  }
  up_over<FIRST>(lo0, acc[1][0]);
  up_over<FIRST>(lo1, acc[1][1]);
}

// writer (v210.ts:145-162) of the lane's block: the even pixel gives Y, Cb, Cr, the odd one Y; then the quad's three lanes trade halves
// TAILS: lines that do not end on a 48-pixel block (1280: src/config.ts:43-54) - `full` whole quads, the tail quad of out_w % 6 = 2 or 4
// pixels with the reference's tail arithmetic (table indices truncated, code values through round() and a truncating convert, the words
// it does not set 0: v210.ts:166-193), then the slots it clears up to the pitch (:131-136).  A lane beyond the line's pixels contributes
// zero code values, so the same hand-overs and stores make the tail quad's unset words and the cleared slots.
template <bool TAILS>
__device__ __forceinline__ void up_write(const UpArgs &a, const UpStep &st, const UpAcc (&acc)[2][2], uint32_t role, const WriteK &wk, const LutK &lk) {
  const uint32_t qpl = TAILS ? a.out_qpitch : a.out_w / 6u;
  const uint32_t quad = st.x0 / 6u, full = a.out_w / 6u;
  const bool in_tail = TAILS && quad == full, beyond = TAILS && st.x0 >= a.out_w;  // (a tail's pixels are inside: beyond is false for them)
  // the twelve table reads of the block's four pixels are started together
  PxPending pend[2][2];
#pragma unroll
  for (int dy = 0; dy < 2; ++dy)
#pragma unroll
    for (int dx = 0; dx < 2; ++dx) {
      auto index_of = [&](float t) __attribute__((always_inline)) { return TAILS ? lds_lut_index_unit_tail(t, in_tail) : lds_lut_index_unit(t); };
      pend[dy][dx] = write_px_issue(index_of(acc[dy][dx].r), index_of(acc[dy][dx].g), index_of(acc[dy][dx].b), lk);
    }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int dy = 0; dy < 2; ++dy) {
    const float er = lds_lut_finish(pend[dy][0].r), eg = lds_lut_finish(pend[dy][0].g), eb = lds_lut_finish(pend[dy][0].b);
    const float orr = lds_lut_finish(pend[dy][1].r), og = lds_lut_finish(pend[dy][1].g), ob = lds_lut_finish(pend[dy][1].b);
    auto code = [&](float v) __attribute__((always_inline)) {
      if (!TAILS) return sat_u16_rte(v);
      const uint32_t c = in_tail ? sat_u16_trunc(__builtin_roundf(v)) : sat_u16_rte(v);
      return beyond ? 0u : c;
    };
    const uint32_t ey = code(dot4(er, eg, eb, 1.0f, wk.y)), eu = code(dot4(er, eg, eb, 1.0f, wk.u)), ev = code(dot4(er, eg, eb, 1.0f, wk.v));
    const uint32_t y1 = code(dot4(orr, og, ob, 1.0f, wk.y));
    // lane B's gifts: to A (the lane before it) Cb2 << 10 | Y2 << 20, to C (the lane after it) Cr2 | Y3 << 10
    const uint32_t to_prev = eu << 10 | ey << 20, to_next = ev | y1 << 10;
    const uint32_t from_next = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)to_prev, 0x130 /* wave_shl:1: lane i reads lane i + 1 */, 0xf, 0xf, false);
    const uint32_t from_prev = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)to_next, 0x138 /* wave_shr:1: lane i reads lane i - 1 */, 0xf, 0xf, false);
    uint2 half;
    if (role == 0u) half = make_uint2(ev << 20 | ey << 10 | eu, from_next | y1);  // w0, w1
    else half = make_uint2(eu << 20 | from_prev, y1 << 20 | ev << 10 | ey);      // w2, w3 (lane C)
    // (TAILS: every quad slot of the pitch is written - the tail quad's unset words and the cleared slots come out of the zero lanes)
    const bool store = (TAILS ? (threadIdx.x & 63u) < 63u && quad < qpl : st.live) && role != 1u && (dy == 0 || st.li[1] != st.li[0]);
    if (store) {
      // ONE eight-byte store per lane: the wave's lanes A and C then cover their row segment without a gap, every 32-byte sector
      // written whole and once.  (As two dword stores each instruction wrote every other dword and each sector went out twice,
      // half filled: WRITE_SIZE 36.7 MB for a 22.1 MB frame, profiles/r03_pmc_up.txt.)
      typedef uint32_t ph_u2v __attribute__((ext_vector_type(2)));
      ph_u2v *dst = reinterpret_cast<ph_u2v *>(reinterpret_cast<uint4 *>(st.job ? a.more_out[st.job - 1u] : a.out) + (size_t)st.line[dy] * qpl + st.x0 / 6u) + (role == 2u ? 1 : 0);
      __builtin_nontemporal_store(ph_u2v{half.x, half.y}, dst);
    }
  }
}

template <bool RGB12, bool TAILS = false>
__global__ __launch_bounds__(kUpBlock) void compose_up_write_v210_kernel(UpArgs a) {
  const WriteK wk = load_write_k(a.wr_cm);
  const LutK lk = make_lut_k(a.wr);
  lds_lut_load<kUpBlock>(a.wr);
  __syncthreads();
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const UpShare sh = up_share(a);
  const uint32_t role = lane - 3u * (lane / 3u);  // 0: pixels 0, 1 of the quad, 1: pixels 2, 3, 2: pixels 4, 5
  UpStep st;
  for (uint32_t base = 0; up_step(a, sh, base, wave, lane, st);) {
    // (lowering a wave's priority as it advances, which helps the two-phase kernels reach their barrier together, measured neutral here)
    UpAcc acc[2][2];
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) acc[dy][dx] = UpAcc{0.0f, 0.0f, 0.0f};
    // one layer at a time.  Variants built and measured slower at 2160p x 4 layers: layers in pairs with both patches in
    // flight (76 against 70 us; again with the shared geometry: 84 against 68 us, 128 registers and spills) and a patch
    // carried in flight across loop turns (85 us) - two patches plus the writer's temporaries do not fit 128 registers
    // The bottom layer is peeled off the loop (it is taken as it is; above it the accumulators are updated in place), and the loop
    // exists three times - shared geometry with every texel inside (packed RGB: no alpha arithmetic at all), shared geometry, a
    // geometry per layer - so that inside a loop nothing about an accumulator is decided by a branch
    UpGeo geo;
    auto layers = [&](auto inside_tag, auto shared_tag) __attribute__((always_inline)) {
      constexpr bool INSIDE = decltype(inside_tag)::value, SHARED = decltype(shared_tag)::value;
      {
        UpLayer L = a.layer[0];  // one 48-byte scalar load
        if (st.job) L.ptr = a.more_ptr[st.job - 1u][0];
        if (!SHARED) geo = up_geo<RGB12>(L, st);
        UpPatch p;
        up_fetch<RGB12>(L, geo, p);
        up_filter<RGB12, INSIDE, true>(p, geo, acc, lk);
      }
#pragma unroll 1
      for (int l = 1; l < a.n; ++l) {
        UpLayer L = a.layer[l];
        if (st.job) L.ptr = a.more_ptr[st.job - 1u][l];
        if (!SHARED) geo = up_geo<RGB12>(L, st);
        UpPatch p;
        up_fetch<RGB12>(L, geo, p);
        up_filter<RGB12, INSIDE, false>(p, geo, acc, lk);
      }
    };
    if (a.shared) {  // uniform
      geo = up_geo<RGB12>(a.layer[0], st);
      if (RGB12 && geo.all_inside) layers(std::true_type{}, std::true_type{});  // uniform
      else layers(std::false_type{}, std::true_type{});
    } else {
      layers(std::false_type{}, std::false_type{});
    }
    up_write<TAILS>(a, st, acc, role, wk, lk);
  }
}

// Does the 2 x 2 block scheme apply?  Unrotated, unmirrored, MAGNIFIED in both directions (then the first taps of the two
// pixels of a block are at most one texel apart, rounding noise included, and the block's taps lie in a 3 x 3 patch), images
// below 1 GiB.
bool compose_up_eligible(const UpArgs &a) {
  if (a.out_w % 2u || !a.n) return false;  // (any even width: lines with a tail take the TAILS instantiation)
  for (int l = 0; l < a.n; ++l) {
    const UpLayer &L = a.layer[l];
    if (L.m[1] != 0.0f || L.m[3] != 0.0f || !(L.m[0] > 0.0f) || !(L.m[4] > 0.0f)) return false;
    // source texels per output pixel: d(u)/dx = m0 * w / out_w; per WRITTEN row: d(v)/dy = m4 * h / out_h * line_step (a field
    // write takes every other line).  Below one texel by a margin far above the f32 noise of the coordinates (1e-4 texels at
    // 1920 columns): then the first taps of the block's two columns / rows are 0 or 1 apart; at a whole texel between them
    // the noise could make it 2.
    // ... or EXACTLY one: a source of the frame's size under the Mixer's default fill (the identity, a frame write).  Its taps sit half a
    // texel off the pixel centres (transform.ts:53 samples at x / w, not (x + 0.5) / w: fu = x - 0.5), as far from a flip as can be:
    // neighbouring pixels' first taps are exactly 1 apart
    const bool fill = L.w == a.out_w && L.h == a.out_h && a.line_step == 1u && L.m[0] == 1.0f && L.m[4] == 1.0f && L.m[2] == 0.0f && L.m[5] == 0.0f;
    if (!fill && ((double)L.m[0] * L.w > 0.99 * a.out_w || (double)L.m[4] * L.h * a.line_step > 0.99 * a.out_h)) return false;
    if ((uint64_t)L.pitch * L.h >= (1ull << 30) || L.w >= (1u << 22)) return false;
  }
  return true;
}

hipError_t launch_compose_up_write_v210(hipStream_t s, const UpArgs &a, bool rgb12, uint32_t num_cus) {
  if (!a.lines) return hipSuccess;
  const bool tails = a.out_w % 48u != 0;  // lines that end in a tail quad and / or cleared slots: an instantiation of its own
  const void *fn = tails ? (rgb12 ? reinterpret_cast<const void *>(compose_up_write_v210_kernel<true, true>)
                                  : reinterpret_cast<const void *>(compose_up_write_v210_kernel<false, true>))
                         : (rgb12 ? reinterpret_cast<const void *>(compose_up_write_v210_kernel<true>)
                                  : reinterpret_cast<const void *>(compose_up_write_v210_kernel<false>));
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, kMaxDynamicLds);
  if (e != hipSuccess) return e;
  UpArgs b = a;
  if (!b.jobs) b.jobs = 1;
  b.out_qpitch = v210_pitch_bytes(a.out_w) / 16u;
  b.cover_w = tails ? b.out_qpitch * 6u : a.out_w;
  b.shared = 1;  // every layer has the size and the placement of the first: the patch geometry and the weights are computed once per block
  for (int l = 1; l < a.n; ++l) {
    b.shared = b.shared && a.layer[l].w == a.layer[0].w && a.layer[l].h == a.layer[0].h && a.layer[l].pitch == a.layer[0].pitch;
    for (int k = 0; k < 6; ++k) b.shared = b.shared && a.layer[l].m[k] == a.layer[0].m[k];
  }
  const uint32_t upr = (b.cover_w + kUpCols - 1u) / kUpCols, upg = (uint32_t)(PH_UP_GROUP_ROWS / 2) * upr;
  const uint32_t units = upr * ((a.lines + 1u) / 2u) * b.jobs;
  // reciprocals for the kernel's uniform divisions: umulhi(v, ceil(2^32 / d)) == v / d while v * d < 2^32
  b.magic_upr = upr > 1 ? (uint32_t)(((1ull << 32) + upr - 1) / upr) : 0u;
  b.magic_upg = (uint32_t)(((1ull << 32) + upg - 1) / upg);
  const uint32_t want = (units + kUpBlock / 64 - 1) / (kUpBlock / 64);
  const uint32_t grid = want < num_cus ? want : num_cus;
  if (tails && rgb12) compose_up_write_v210_kernel<true, true><<<grid, kUpBlock, a.wr.bytes, s>>>(b);
  else if (tails) compose_up_write_v210_kernel<false, true><<<grid, kUpBlock, a.wr.bytes, s>>>(b);
  else if (rgb12) compose_up_write_v210_kernel<true><<<grid, kUpBlock, a.wr.bytes, s>>>(b);
  else compose_up_write_v210_kernel<false><<<grid, kUpBlock, a.wr.bytes, s>>>(b);
  return hipGetLastError();
}

}  // namespace ph
