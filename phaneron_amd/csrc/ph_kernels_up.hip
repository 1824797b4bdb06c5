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
  const __amdgpu_buffer_rsrc_t img = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(L.ptr), 0, (int)(L.pad ? L.pad : L.pitch * L.h), 0x00020000);
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
__device__ __forceinline__ void up_write(const UpArgs &a, const UpStep &st, const UpAcc (&acc)[2][2], uint32_t role, const WriteK &wk, const LutK &lk,
                                         void *frame = nullptr) {  // frame: the step's output frame, where the caller knows it (else by st.job)
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
      ph_u2v *dst = reinterpret_cast<ph_u2v *>(reinterpret_cast<uint4 *>(frame ? frame : st.job ? a.more_out[st.job - 1u] : a.out) + (size_t)st.line[dy] * qpl + st.x0 / 6u) + (role == 2u ? 1 : 0);
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

// ------------------------------------------------------------------------------------------------------------------------------------
// Clips straight from their wire formats: [read] x N -> [transform] x N -> combine_N -> v210 write as ONE launch.
// The two-launch route (reader of the format into an f32 image, then the kernel above) sends every clip's image out of the chip and
// back - 6 x the frame's compulsory bytes for a 720p clip on a 1080p channel - and pays launch, table load and drain twice for two
// kernels of ten microseconds each.  Here a workgroup owns a TILE of the output (tcu wave-step columns x trp row pairs).  Phase 1, the
// reader's table in the LDS: the source pixels the tile's taps fall on are converted once each (v210.ts:58-78 / yuv420p.ts ... the
// reader's own arithmetic, a wave per source row) into a rectangle of the workgroup's own scratch, which never leaves the XCD's L2.
// Phase 2, the writer's table swapped in: the kernel above, its nine-texel patches taken from the rectangle.
// ------------------------------------------------------------------------------------------------------------------------------------
// PH_CLIP_ABLATE builds (timing experiments, never shipped): bit 0 = stop after phase 1, bit 1 = phase 1 converts nothing, bit 2 = no phase 2 work
#ifndef PH_CLIP_ABLATE
#define PH_CLIP_ABLATE 0
#endif
enum { CF_V210 = 0, CF_YUV422P10 = 1, CF_YUV422P8 = 2, CF_YUV420P = 3, CF_NV12 = 4, CF_RGBA8 = 5, CF_BGRA8 = 6 };  // = PH_FMT_*

// one source pixel as the reader of its format makes it (ph_kernels_fmt.hip fmt_read_body, ph_kernels_lds.hip v210_read_lds_kernel), in two
// halves: its samples requested, then converted - a lane requests the samples of all its pixels of a round before it converts the first
// (a rectangle is a handful of pixels per lane: one at a time, every pixel would wait out its own trip to HBM)
struct ClipRaw {
  uint4 w;  // v210: the pixel's word quad; planar: .x .y .z = Y, Cb, Cr; packed RGB: .x = the pixel
};
template <int FMT>
__device__ __forceinline__ ClipRaw clip_fetch(const ClipSrc &S, uint32_t x, uint32_t line) {
  ClipRaw r;
  r.w = make_uint4(0u, 0u, 0u, 0u);
  if (FMT == CF_V210) {
    r.w = reinterpret_cast<const uint4 *>(S.p0)[(size_t)line * S.pitch + x / 6u];
  } else if (FMT == CF_RGBA8 || FMT == CF_BGRA8) {
    r.w.x = reinterpret_cast<const uint32_t *>(S.p0)[(size_t)line * S.pitch + x];
  } else {
    const uint32_t cl = (FMT == CF_YUV420P || FMT == CF_NV12) ? line >> 1 : line;
    if (FMT == CF_YUV422P10) {
      r.w.x = reinterpret_cast<const uint16_t *>(S.p0)[(size_t)line * S.pitch + x];
      r.w.y = reinterpret_cast<const uint16_t *>(S.p1)[(size_t)cl * (S.pitch >> 1) + (x >> 1)];
      r.w.z = reinterpret_cast<const uint16_t *>(S.p2)[(size_t)cl * (S.pitch >> 1) + (x >> 1)];
    } else if (FMT == CF_NV12) {  // nv12.ts:61-74
      r.w.x = reinterpret_cast<const uint8_t *>(S.p0)[(size_t)line * S.pitch + x];
      const uint8_t *c = reinterpret_cast<const uint8_t *>(S.p1) + (size_t)cl * S.pitch + (x & ~1u);
      r.w.y = c[0], r.w.z = c[1];
    } else {
      r.w.x = reinterpret_cast<const uint8_t *>(S.p0)[(size_t)line * S.pitch + x];
      r.w.y = reinterpret_cast<const uint8_t *>(S.p1)[(size_t)cl * (S.pitch >> 1) + (x >> 1)];
      r.w.z = reinterpret_cast<const uint8_t *>(S.p2)[(size_t)cl * (S.pitch >> 1) + (x >> 1)];
    }
  }
  return r;
}
template <int FMT>
__device__ __forceinline__ float4 clip_finish(const ClipRaw &raw, uint32_t width, uint32_t x, const ReadK &k, const LutK &lk) {
  if (FMT == CF_V210) {
    const uint4 w = raw.w;
    const uint32_t g = x / 6u, j = x - 6u * g, pr = j >> 1;
    const uint32_t wy = (j == 0) ? w.x : (j < 3) ? w.y : (j == 3) ? w.z : w.w;  // v210.ts:58-63, as v210_read_lds_kernel
    const uint32_t sy = (j == 0 || j == 3) ? 10u : (j == 1 || j == 4) ? 0u : 20u;
    const uint32_t wcb = pr == 0 ? w.x : pr == 1 ? w.y : w.z;
    const uint32_t wcr = pr == 0 ? w.x : pr == 1 ? w.z : w.w;
    const uint32_t scr = pr == 0 ? 20u : pr == 1 ? 0u : 10u;
    const float yf = (float)((wy >> sy) & 0x3ff), cbf = (float)((wcb >> (10u * pr)) & 0x3ff), crf = (float)((wcr >> scr) & 0x3ff);
    return read_px_lds(yf, cbf, crf, k, lk, x < width - width % 6u ? 1.0f : 0.0f);  // (a line's tail: v210.ts:88-93)
  }
  if (FMT == CF_RGBA8 || FMT == CF_BGRA8) {  // rgba8.ts:49-62
    const uint32_t p = raw.w.x;
    const float c0 = (float)(p & 0xffu), gf = (float)((p >> 8) & 0xffu), c2 = (float)((p >> 16) & 0xffu), af = (float)(p >> 24);
    const float rf = FMT == CF_RGBA8 ? c0 : c2, bf = FMT == CF_RGBA8 ? c2 : c0;
    const float r = lds_lut_at(lk, rf * 65535.0f / 255.0f), g = lds_lut_at(lk, gf * 65535.0f / 255.0f), b = lds_lut_at(lk, bf * 65535.0f / 255.0f);
    return make_float4(dot3(r, g, b, k.gm[0], k.gm[1], k.gm[2]), dot3(r, g, b, k.gm[3], k.gm[4], k.gm[5]), dot3(r, g, b, k.gm[6], k.gm[7], k.gm[8]),
                       lds_lut_at(lk, af * 65535.0f / 255.0f));
  }
  // (yuv422p10.ts:74-78: dot4 with the offset column, table, gamut - the v210 reader's arithmetic)
  return read_px_lds((float)raw.w.x, (float)raw.w.y, (float)raw.w.z, k, lk, 1.0f);
}

// The first tap (column / row) of an output pixel, exactly as up_geo computes it: monotone in the pixel's position (every step is a
// correctly rounded monotone operation), so a tile's first and last pixels bound the taps of everything between them.
__device__ __forceinline__ int clip_tap_col(const UpLayer &L, uint32_t x, uint32_t out_w) {
  const float px = (float)(int)x / (float)(int)out_w - 0.5f;
  const float sx = dot3(L.m[0], L.m[1], L.m[2], px, 0.0f, 1.0f) + 0.5f;
  return (int)__builtin_floorf(sx * (float)(int)L.w - 0.5f);
}
__device__ __forceinline__ int clip_tap_row(const UpLayer &L, uint32_t line, uint32_t out_h) {
  const float py = (float)(int)line / (float)(int)out_h - 0.5f;
  const float sy = dot3(L.m[3], L.m[4], L.m[5], 0.0f, py, 1.0f) + 0.5f;
  return (int)__builtin_floorf(sy * (float)(int)L.h - 0.5f);
}
struct ClipRect {  // uniform: the source rectangle of a tile, inside the image (w == 0: nothing of the image under this tile)
  uint32_t c0, r0, w, h;
};

// FIRST: the launch's first rectangle.  Its first round of samples is requested BEFORE the reader's table is loaded (the table load and the
// samples' trip to HBM overlap), and the table load + barrier happen in here.  Within a rectangle a round's samples are requested before
// the round before it is converted.
template <int FMT, bool RGB12, bool FIRST>
__device__ __forceinline__ void clip_convert(const ClipSrc &S, const UpLayer &L, const ClipRect &R, char *dst, uint32_t ppitch, const float *rd_gm, const LutK &rlk,
                                             const LutView &rd_view) {
  // the rectangle's pixels row by row over all the workgroup's lanes, kClipRound of them per lane and round
  constexpr int kClipRound = FMT == CF_V210 ? 4 : 6;  // (a 720p clip on a 1080p channel: 4.3 K pixels per workgroup - one round)
  const uint32_t total = R.w * R.h;  // (< 2^24: exact as a float)
  const float inv_w = 1.0f / (float)(R.w ? R.w : 1u);
  struct Round {
    ClipRaw raw[kClipRound];
    uint32_t col[kClipRound], row[kClipRound];
  };
  auto request = [&](uint32_t base, Round &q) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < kClipRound; ++i) {
      const uint32_t idx = base + (uint32_t)i * kUpBlock < total ? base + (uint32_t)i * kUpBlock : total - 1u;  // (lanes past the end redo the last pixel)
      uint32_t r = (uint32_t)((float)idx * inv_w);  // idx / R.w, off by one at most
      r -= r * R.w > idx ? 1u : 0u;
      r += (r + 1u) * R.w <= idx ? 1u : 0u;
      q.row[i] = r, q.col[i] = idx - r * R.w;
      q.raw[i] = clip_fetch<FMT>(S, R.c0 + q.col[i], R.r0 + r);
    }
  };
  Round cur;
  if (total) request(threadIdx.x, cur);  // (uniform)
  if (FIRST) {
    lds_lut_load<kUpBlock>(rd_view);
    __syncthreads();
  }
  ReadK k;
  if (FMT >= CF_RGBA8) {
#pragma unroll
    for (int i = 0; i < 9; ++i) k.gm[i] = rd_gm[i];
  } else {
    k = load_read_k(S.cm, rd_gm);
  }
  for (uint32_t base = threadIdx.x; base < total; base += kClipRound * kUpBlock) {
    Round nxt = cur;
    const uint32_t next_base = base + kClipRound * kUpBlock;
    if (next_base - threadIdx.x < total) request(next_base < total ? next_base : total - 1u, nxt);  // (uniform test: the whole workgroup takes the round or not)
#pragma unroll
    for (int i = 0; i < kClipRound; ++i) {
      const float4 t = clip_finish<FMT>(cur.raw[i], L.w, R.c0 + cur.col[i], k, rlk);
      char *at = dst + (size_t)cur.row[i] * ppitch + (size_t)cur.col[i] * (RGB12 ? 12u : 16u);
      if (RGB12) *reinterpret_cast<ph_u32x3 *>(at) = ph_u32x3{__float_as_uint(t.x), __float_as_uint(t.y), __float_as_uint(t.z)};
      else *reinterpret_cast<float4 *>(at) = t;
    }
    cur = nxt;
  }
}

// a wave step of a tile: as up_step makes it for (col_unit, row pair) of job 0
__device__ __forceinline__ void clip_step(const UpArgs &a, uint32_t job, uint32_t col_unit, uint32_t rp, uint32_t lane, UpStep &st) {
  st.job = job;
  st.x0 = col_unit * kUpCols + 2u * lane;
  st.live = lane < 63u && st.x0 < a.out_w;
#pragma unroll
  for (int dy = 0; dy < 2; ++dy) {
    st.li[dy] = 2u * rp + (uint32_t)dy < a.lines ? 2u * rp + (uint32_t)dy : 2u * rp;
    st.line[dy] = a.first_line + st.li[dy] * a.line_step;
    st.py[dy] = (float)(int)st.line[dy] / (float)(int)a.out_h - 0.5f;
  }
#pragma unroll
  for (int dx = 0; dx < 2; ++dx) st.px[dx] = (float)(int)(st.x0 + (uint32_t)dx) / (float)(int)a.out_w - 0.5f;
}

// MULTI: several frames of one shape in the launch (an instantiation of its own: a job index alive through both phases costs the
// one-frame kernel eight spilled registers and 1.3 us, measured)
template <bool RGB12, bool TAILS, bool MULTI = false>
__global__ __launch_bounds__(kUpBlock) void clip_up_write_v210_kernel(ClipUpArgs c) {
  const UpArgs &a = c.up;
  constexpr uint32_t kTexel = RGB12 ? 12u : 16u;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  // the tile
  const uint32_t upr = (a.cover_w + kUpCols - 1u) / kUpCols, rps = (a.lines + 1u) / 2u;
  const uint32_t tyj = blockIdx.x / c.gx, tx = blockIdx.x - tyj * c.gx;
  const uint32_t job = MULTI ? tyj / c.gy : 0u, ty = tyj - job * c.gy;  // (uniform) several frames of one shape: rows of tiles job after job
  const uint32_t cu0 = tx * c.tcu, rp0 = ty * c.trp;
  const uint32_t ncu = cu0 + c.tcu <= upr ? c.tcu : upr - cu0, nrp = rp0 + c.trp <= rps ? c.trp : rps - rp0;
  char *const mine = c.scratch + (size_t)blockIdx.x * c.wg_bytes;
  // ---- phase 1: the reader's table; every layer's rectangle converted
  const LutK rlk = make_lut_k(c.rd);
  uint4 *const info = reinterpret_cast<uint4 *>(g_lds + c.info_off);
#pragma unroll 1
  for (int l = 0; l < a.n; ++l) {
    const UpLayer L = a.layer[l];
    const ClipSrc S = c.src[MULTI ? job : (uint32_t)l];  // (several jobs: single-layer frames, job j's clip in src[j])
    // the tile's first and last pixels (inside the frame: columns past out_w - the padding of a line with a tail - have no taps that count)
    const uint32_t x_first = cu0 * kUpCols, x_end = (cu0 + ncu) * kUpCols < a.out_w ? (cu0 + ncu) * kUpCols : a.out_w;
    const uint32_t li_last = 2u * (rp0 + nrp) < a.lines ? 2u * (rp0 + nrp) - 1u : a.lines - 1u;
    ClipRect R{0, 0, 0, 0};
    if (x_first < x_end) {
      int c_lo = clip_tap_col(L, x_first, a.out_w), c_hi = clip_tap_col(L, x_end - 1u, a.out_w) + 2;  // (a block's patch: three columns from its left pixel's first tap)
      int r_lo = clip_tap_row(L, a.first_line + 2u * rp0 * a.line_step, a.out_h), r_hi = clip_tap_row(L, a.first_line + li_last * a.line_step, a.out_h) + 2;
      c_lo = c_lo < 0 ? 0 : c_lo, r_lo = r_lo < 0 ? 0 : r_lo;
      c_hi = c_hi >= (int)L.w ? (int)L.w - 1 : c_hi, r_hi = r_hi >= (int)L.h ? (int)L.h - 1 : r_hi;
      if (c_hi >= c_lo && r_hi >= r_lo) R = ClipRect{(uint32_t)c_lo, (uint32_t)r_lo, (uint32_t)(c_hi - c_lo + 1), (uint32_t)(r_hi - r_lo + 1)};
    }
    R.c0 = __builtin_amdgcn_readfirstlane(R.c0), R.r0 = __builtin_amdgcn_readfirstlane(R.r0);
    R.w = __builtin_amdgcn_readfirstlane(R.w), R.h = __builtin_amdgcn_readfirstlane(R.h);
    const uint32_t ppitch = R.w * kTexel;
    if (ppitch && R.h * ppitch > c.rect_cap[l]) R.h = c.rect_cap[l] / ppitch;  // (the launcher's bound holds: never taken)
    char *const dst = mine + c.rect_off[l];
#if PH_CLIP_ABLATE & 2  // timing experiment only (wrong pixels): nothing converted
    R.h = 0;
#endif
#define PH_CLIP_CASE(F)                                                                                   \
  case F:                                                                                                  \
    if (l == 0) clip_convert<F, RGB12, true>(S, L, R, dst, ppitch, c.rd_gm, rlk, c.rd);                    \
    else clip_convert<F, RGB12, false>(S, L, R, dst, ppitch, c.rd_gm, rlk, c.rd);                          \
    break;
    switch (S.fmt) {  // uniform
      PH_CLIP_CASE(CF_V210)
      PH_CLIP_CASE(CF_YUV422P10)
      PH_CLIP_CASE(CF_YUV422P8)
      PH_CLIP_CASE(CF_YUV420P)
      PH_CLIP_CASE(CF_NV12)
      PH_CLIP_CASE(CF_RGBA8)
      default:
        if (l == 0) clip_convert<CF_BGRA8, RGB12, true>(S, L, R, dst, ppitch, c.rd_gm, rlk, c.rd);
        else clip_convert<CF_BGRA8, RGB12, false>(S, L, R, dst, ppitch, c.rd_gm, rlk, c.rd);
        break;
    }
#undef PH_CLIP_CASE
    // the layer as phase 2 sees it: texel (col, row) of the IMAGE at base + row * ppitch + col * kTexel (only texels of the rectangle are ever
    // taken: everything else of a patch is outside the image, an offset beyond the buffer's records)
    if (threadIdx.x == 0) {
      const uint64_t base = (uint64_t)(uintptr_t)dst - ((uint64_t)R.r0 * ppitch + (uint64_t)R.c0 * kTexel);
      info[l] = make_uint4((uint32_t)base, (uint32_t)(base >> 32), ppitch ? ppitch : kTexel, 0u);
    }
  }
#if PH_CLIP_ABLATE & 1  // timing experiment only: phase 1 alone
  return;
#endif
  // ---- the swap: every wave's rectangle rows are in L2 (a workgroup's stores are visible to its own later loads), the writer's table comes in
  __threadfence_block();
  __syncthreads();
  const WriteK wk = load_write_k(a.wr_cm);
  const LutK lk = make_lut_k(a.wr);
  lds_lut_load<kUpBlock>(a.wr);
  __syncthreads();
  // ---- phase 2: the 2 x 2-block compositor on the tile's wave steps
  const uint32_t role = lane - 3u * (lane / 3u);
  auto layer_of = [&](int l) __attribute__((always_inline)) {
    UpLayer L = a.layer[l];
    const uint4 d = info[l];  // uniform address: one broadcast read
    const uint64_t base = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)d.x) | (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)d.y) << 32;
    L.ptr = reinterpret_cast<const void *>((uintptr_t)base);
    L.pitch = (uint32_t)__builtin_amdgcn_readfirstlane((int)d.z);
    L.pad = kUpOutside - 16u;  // the buffer's records: rows are ppitch apart but columns count from the IMAGE's left edge, so pitch * h is no bound here
    return L;
  };
#if PH_CLIP_ABLATE & 4
  const uint32_t units = 0;
#else
  const uint32_t units = ncu * nrp;
#endif
  void *const frame = MULTI && job ? a.more_out[job - 1u] : a.out;  // (uniform: the tile's frame)
  for (uint32_t u = wave; u < units; u += kUpBlock / 64) {
    const uint32_t rp_in = ncu == 1u ? u : u / ncu, cu_in = u - rp_in * ncu;
    UpStep st;
    clip_step(a, 0u, cu0 + cu_in, rp0 + rp_in, lane, st);
    UpAcc acc[2][2];
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) acc[dy][dx] = UpAcc{0.0f, 0.0f, 0.0f};
    UpGeo geo;
    auto layers = [&](auto inside_tag, auto shared_tag) __attribute__((always_inline)) {
      constexpr bool INSIDE = decltype(inside_tag)::value, SHARED = decltype(shared_tag)::value;
      {
        const UpLayer L = layer_of(0);
        if (!SHARED) geo = up_geo<RGB12>(L, st);  // (shared: layers of one size and placement have rectangles of one shape, their bases aside)
        UpPatch p;
        up_fetch<RGB12>(L, geo, p);
        up_filter<RGB12, INSIDE, true>(p, geo, acc, lk);
      }
#pragma unroll 1
      for (int l = 1; l < a.n; ++l) {
        const UpLayer L = layer_of(l);
        if (!SHARED) geo = up_geo<RGB12>(L, st);
        UpPatch p;
        up_fetch<RGB12>(L, geo, p);
        up_filter<RGB12, INSIDE, false>(p, geo, acc, lk);
      }
    };
    if (a.shared) {  // uniform
      geo = up_geo<RGB12>(layer_of(0), st);
      if (RGB12 && geo.all_inside) layers(std::true_type{}, std::true_type{});
      else layers(std::false_type{}, std::true_type{});
    } else {
      layers(std::false_type{}, std::false_type{});
    }
    up_write<TAILS>(a, st, acc, role, wk, lk, frame);
  }
}

// The tiling: tiles of tcu wave-step columns x trp row pairs, one per workgroup, as even as the frame allows (a workgroup's sixteen waves
// share its tile's wave steps); then every layer's rectangle bound - the taps of (tcu * 126) columns x (2 * trp) written rows span that
// many source texels times the magnification, + the patch's third column / row, + one for the rounding of the coordinates.
uint32_t clip_up_plan(ClipUpArgs &c, bool rgb12, uint32_t num_cus) {
  UpArgs &a = c.up;
  if (!a.lines || !a.n || !num_cus) return 0;
  if (!a.jobs) a.jobs = 1;
  if (a.jobs > (uint32_t)kMaxUpJobs || (a.jobs > 1u && a.n != 1)) return 0;
  const bool tails = a.out_w % 48u != 0;
  a.out_qpitch = v210_pitch_bytes(a.out_w) / 16u;
  a.cover_w = tails ? a.out_qpitch * 6u : a.out_w;
  a.shared = 1;
  for (int l = 1; l < a.n; ++l) {
    a.shared = a.shared && a.layer[l].w == a.layer[0].w && a.layer[l].h == a.layer[0].h;
    for (int k = 0; k < 6; ++k) a.shared = a.shared && a.layer[l].m[k] == a.layer[0].m[k];
  }
  const uint32_t upr = (a.cover_w + kUpCols - 1u) / kUpCols, rps = (a.lines + 1u) / 2u, cus_per_job = num_cus / a.jobs ? num_cus / a.jobs : 1u;
  uint32_t best = ~0u, tcu = 0, trp = 0;
  for (uint32_t cx = 1; cx <= 32u && cx <= upr && cx <= cus_per_job; cx *= 2u) {
    const uint32_t ry = cus_per_job / cx < rps ? cus_per_job / cx : rps;  // rows of tiles per job (a tile belongs to one job)
    const uint32_t t_cu = (upr + cx - 1u) / cx, t_rp = (rps + ry - 1u) / ry;
    // wave steps of the fullest workgroup, in rounds of its sixteen waves (what phase 2 takes), + its rectangle's rim (what phase 1 converts twice)
    const uint32_t rounds = (t_cu * t_rp + kUpBlock / 64 - 1u) / (kUpBlock / 64);
    const uint32_t cost = rounds * 1024u + t_cu + t_rp;
    if (cost < best) best = cost, tcu = t_cu, trp = t_rp;
  }
  c.tcu = tcu, c.trp = trp, c.gx = (upr + tcu - 1u) / tcu, c.gy = (rps + trp - 1u) / trp;
  const uint32_t grid = c.gx * c.gy * a.jobs;
  const uint32_t texel = rgb12 ? 12u : 16u;
  uint32_t off = 0;
  for (int l = 0; l < a.n; ++l) {
    const UpLayer &L = a.layer[l];
    const double sx = (double)L.m[0] * L.w / a.out_w, sy = (double)L.m[4] * L.h / a.out_h * a.line_step;
    uint32_t rw = (uint32_t)((tcu * kUpCols - 1u) * sx) + 6u, rh = (uint32_t)((2u * trp - 1u) * sy) + 6u;
    rw = rw < L.w ? rw : L.w, rh = rh < L.h ? rh : L.h;
    c.rect_off[l] = off, c.rect_cap[l] = (rw * rh * texel + 255u) & ~255u;
    off += c.rect_cap[l] + 256u;  // (+ a rim: dead lanes' patches read a few texels past a rectangle's last row)
  }
  c.wg_bytes = off;
  const uint32_t lds = c.rd.bytes > a.wr.bytes ? c.rd.bytes : a.wr.bytes;
  c.info_off = (lds + 15u) & ~15u;
  if (c.info_off + 16u * (uint32_t)a.n > (uint32_t)kMaxDynamicLds) return 0;
  return grid;
}

hipError_t launch_clip_up_write_v210(hipStream_t s, const ClipUpArgs &c, bool rgb12, uint32_t grid) {
  if (!grid) return hipErrorInvalidValue;
  char name[48];
  if (c.up.jobs > 1u) snprintf(name, sizeof name, "clip_up_write_v210<%s>x%u", rgb12 ? "rgb" : "rgba", c.up.jobs);
  else snprintf(name, sizeof name, "clip_up_write_v210<%s>", rgb12 ? "rgb" : "rgba");
  if (trace_launch(name)) return hipSuccess;
  const bool tails = c.up.out_w % 48u != 0, multi = c.up.jobs > 1u;
  const uint32_t lds = c.info_off + 16u * (uint32_t)c.up.n;
  auto go = [&](auto kernel) -> hipError_t {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kMaxDynamicLds);
    if (e != hipSuccess) return e;
    kernel<<<grid, kUpBlock, lds, s>>>(c);
    return hipGetLastError();
  };
  if (multi) {
    if (tails) return rgb12 ? go(clip_up_write_v210_kernel<true, true, true>) : go(clip_up_write_v210_kernel<false, true, true>);
    return rgb12 ? go(clip_up_write_v210_kernel<true, false, true>) : go(clip_up_write_v210_kernel<false, false, true>);
  }
  if (tails) return rgb12 ? go(clip_up_write_v210_kernel<true, true, false>) : go(clip_up_write_v210_kernel<false, true, false>);
  return rgb12 ? go(clip_up_write_v210_kernel<true, false, false>) : go(clip_up_write_v210_kernel<false, false, false>);
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
