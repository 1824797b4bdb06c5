// ph_kernels_field.hip - the field pipeline of a de-interlacing, scaling channel as ONE kernel:
//
//     for every layer:  yadif (prev, cur, next)  ->  transform (bilinear, border 0)        yadifCl.ts:105-167, transform.ts:36-59
//     combine_N  ->  gamma LUT + RGB->YCbCr + v210 pack                                      combine.ts:45-65, v210.ts:113-195
//
// i.e. BASELINE config 3's per-field job batch (yadif.ts:115-145 -> producer/mixer.ts:209-223 -> combiner.ts:219-254
// -> io.ts:152-164) without the de-interlaced frames (N x source size), the placed frames (N x output size) and the
// combined frame ever reaching HBM.  Bit-identical to ph_yadif + ph_transform + ph_combine + ph_v210_write.
//
// Scanline neighbourhoods are staged in LDS (north_star): the workgroup first builds, for one layer, the DE-INTERLACED
// source window that a slice of 384 x 16 output pixels can touch - every de-interlaced pixel is computed once and then
// sampled by up to four output pixels - then every lane filters its six output pixels from that window and folds them
// into its accumulator; next layer, same LDS.  Like the headline kernel the slice ends as eighteen 16-bit writer-table
// indices in nine registers, and after six slices the LDS is handed to the writer's gamma table for the second phase
// (table lookup, matrix, pack, one 16-byte store per lane).
//
// Limits of the fused form (the host checks them, ph_api.cpp): every layer's transform is axis-aligned and not
// mirrored (no rotation: m01 = m10 = 0, m00 > 0, m11 > 0 - fill and picture-in-picture placements), and the source
// window of a slice fits the LDS (an up-scale or 1:1; a layer shrunk below ~2/3 does not).  Anything else runs as
// separate kernels.  out_w % 384 == 0.
#include "ph_device.h"
#include "ph_kernels.h"
#include "ph_ldslut.h"
#include "ph_yadif.h"

#pragma clang fp contract(off)

#ifndef PH_PROBE
#define PH_PROBE 0
#endif
#if PH_PROBE  // tools/field_probe.py: wave 0 stamps s_memtime at the stage boundaries of its first slice
__device__ unsigned long long g_field_probe[2048 * 8];
extern "C" int ph_debug_field_probe(unsigned long long *out, int n_words) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_field_probe), (size_t)n_words * 8);
}
#define PH_FSTAMP(k, cond)                                                                      \
  do {                                                                                          \
    if (threadIdx.x == 0 && (cond)) g_field_probe[blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#else
#define PH_FSTAMP(k, cond) do { } while (0)
#endif

namespace ph {

// Stage 1 runs 512-lane workgroups: its row windows (13 x float4) next to the predictors' temporaries need more than the
// 128 registers a 1024-lane workgroup leaves a lane, and every spill is a memory round trip the whole lock-stepped
// workgroup waits for (measured: 25 000 cycles per source row with spills).  Stage 2 needs the gamma
// table in LDS and keeps the 1024-lane shape of the other table kernels.
constexpr int kFieldBlock = 512;
// The kernel is built twice.  DEINT: some layer is de-interlaced - the thirteen row windows need ~170 registers, one
// workgroup per CU (compiled for 128 registers the spills cost 2.4x the time, measured).  !DEINT: every layer is
// progressive (or was de-interlaced by ph_yadif beforehand) - only `cur` rows are copied into the windows, the kernel fits
// 128 registers and two workgroups share a CU, one's arithmetic covering the other's barriers and loads.
constexpr int kSliceW = 192, kSliceH = 16;  // output pixels per slice: 32 quads x 16 rows = one quad per lane.  Wide, so that a
                                            // layer's lane group (one lane per window column) is mostly busy
// slices per strip, walked top to bottom: 12 (192 x 192 output pixels, 240 strips at 2160p: one workgroup per CU) when a
// layer de-interlaces, 6 (460 strips: two workgroups per CU) when none does
__host__ __device__ constexpr uint32_t field_slices_per_strip(bool deint) { return deint ? 12u : 6u; }

#define PH_C3(v, c) ((c) == 0 ? (v).x : (c) == 1 ? (v).y : (v).z)

struct Window {  // the part of a layer's (de-interlaced) image one slice samples: [x0, x0 + w) x [y0, y0 + h), clipped
  int x0, y0, w, h;
};

// tap coordinates exactly as sample_linear computes them (ph_device.h)
struct Taps {
  uint32_t i0, j0;
  float a, b;
};
__device__ __forceinline__ Taps taps_from(const float *m, int lw, int lh, float px, float py);
__device__ __forceinline__ Taps taps_at(const float *m, int lw, int lh, uint32_t x, uint32_t y, uint32_t ow, uint32_t oh) {
  return taps_from(m, lw, lh, (float)(int)x / (float)(int)ow - 0.5f, (float)(int)y / (float)(int)oh - 0.5f);
}
// px, py: the pixel's normalised centre-relative position (transform.ts:53-54)
__device__ __forceinline__ Taps taps_from(const float *m, int lw, int lh, float px, float py) {
  const float s = dot3(m[0], m[1], m[2], px, py, 1.0f) + 0.5f;  // transform.ts:55-57
  const float t = dot3(m[3], m[4], m[5], px, py, 1.0f) + 0.5f;
  const float u = s * (float)lw, v = t * (float)lh;
  const float fu = u - 0.5f, fv = v - 0.5f;
  const float flu = __builtin_floorf(fu), flv = __builtin_floorf(fv);
  return Taps{(uint32_t)(int)flu, (uint32_t)(int)flv, fu - flu, fv - flv};
}

// The window of a layer that the slice [x_first, x_first + kSliceW) x [y_first, y_last] samples.  Tap coordinates grow with
// x and y (axis-aligned, not mirrored), so the corners bound them.  The window is NOT clipped to the image: it may reach one
// cell past every edge, and those cells hold the border colour (zeros) - so the filter needs no per-tap bounds test, only
// a clamp of the tap coordinates to [-1, size].
__device__ __forceinline__ int clamp_tap(int v, int size) { return v < -1 ? -1 : (v > size ? size : v); }
__device__ __forceinline__ Window slice_window(const float *m, int lw, int lh, uint32_t x_first, uint32_t y_first, uint32_t y_last,
                                               uint32_t ow, uint32_t oh) {
  const Taps lo = taps_at(m, lw, lh, x_first, y_first, ow, oh);
  const Taps hi = taps_at(m, lw, lh, x_first + kSliceW - 1, y_last, ow, oh);
  Window win;
  win.x0 = clamp_tap((int)lo.i0, lw), win.y0 = clamp_tap((int)lo.j0, lh);
  win.w = clamp_tap((int)hi.i0 + 1, lw) - win.x0 + 1, win.h = clamp_tap((int)hi.j0 + 1, lh) - win.y0 + 1;
  return win;
}

// one lane's six output pixels filtered from a layer's window in LDS and folded into its accumulator
// (transform.ts:53-59 with the OpenCL 1.2 s8.2 LINEAR formula as sample_linear evaluates it; combine.ts:45-65)
__device__ __forceinline__ void sample_and_combine(const float4 *tile, const Window &win, const float *m, int lw, int lh,
                                                   const float (&pxs)[6], float py, bool first, float (&acc)[18]) {
  // rows: the six pixels of a lane share their y, and an axis-aligned transform makes the row taps independent of x
  const Taps ty = taps_from(m, lw, lh, pxs[0], py);
  const float omb = 1.0f - ty.b;
  const uint32_t r0 = (uint32_t)(clamp_tap((int)ty.j0, lh) - win.y0) * (uint32_t)win.w;
  const uint32_t r1 = (uint32_t)(clamp_tap((int)ty.j0 + 1, lh) - win.y0) * (uint32_t)win.w;
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const Taps tx = taps_from(m, lw, lh, pxs[j], py);  // only the column part is used (the row part is dead code here)
    const float oma = 1.0f - tx.a;
    const float w00 = oma * omb, w10 = tx.a * omb, w01 = oma * ty.b, w11 = tx.a * ty.b;
    const uint32_t c0 = (uint32_t)(clamp_tap((int)tx.i0, lw) - win.x0), c1 = (uint32_t)(clamp_tap((int)tx.i0 + 1, lw) - win.x0);
    const float4 t00 = tile[r0 + c0], t10 = tile[r0 + c1], t01 = tile[r1 + c0], t11 = tile[r1 + c1];
    float4 r;
    r.x = ((w00 * t00.x + w10 * t10.x) + w01 * t01.x) + w11 * t11.x;
    r.y = ((w00 * t00.y + w10 * t10.y) + w01 * t01.y) + w11 * t11.y;
    r.z = ((w00 * t00.z + w10 * t10.z) + w01 * t01.z) + w11 * t11.z;
    r.w = ((w00 * t00.w + w10 * t10.w) + w01 * t01.w) + w11 * t11.w;
    if (first) {
      acc[3 * j] = r.x, acc[3 * j + 1] = r.y, acc[3 * j + 2] = r.z;
    } else {  // the result's alpha is never used by the writer
      const float kk = 1.0f - r.w;
      acc[3 * j] = fma_rn(acc[3 * j], kk, r.x), acc[3 * j + 1] = fma_rn(acc[3 * j + 1], kk, r.y);
      acc[3 * j + 2] = fma_rn(acc[3 * j + 2], kk, r.z);
    }
  }
}

// Stage 1 of the field pipeline: windows -> filtered, combined pixels -> the eighteen 16-bit writer-table indices of
// every output quad (the writer's first step, index = sat_rte(rgb * 65535), needs no table: v210.ts:148-150), stored
// as nine planes of 32-bit words (two indices each) - 6 bytes per pixel instead of the 16 of an f32 frame.
//
// A workgroup owns a STRIP of the output: 192 pixels wide, 6 or 12 slices of 16 rows, walked top to bottom.  `batch`
// layers have their windows in LDS at once (all of them when they fit - the host decides); each gets 512 / batch
// lanes, one lane per window COLUMN, which walks down the source rows with the row windows of all three frames in
// registers - every source row is loaded once - and shares `cur` rows y-1, y+1 through LDS for the x-3..x+3 taps of the
// spatial predictor.  With a single batch (CONTINUE) the register windows and the last rows of the LDS window carry over
// from slice to slice, so the strip reads each source row once from top to bottom: no halo is loaded twice.
template <bool CONTINUE, bool DEINT>
__global__ __launch_bounds__(kFieldBlock, DEINT ? 2 : 4) void field_indices_kernel(FieldArgs a, uint32_t *__restrict__ idx, uint32_t batch) {
  float4 *const tile = reinterpret_cast<float4 *>(g_lds);
  const uint32_t strips_x = a.out_w / kSliceW, slices_y = (a.out_h + kSliceH - 1) / kSliceH;
  constexpr uint32_t kFieldP = field_slices_per_strip(DEINT);
  const uint32_t bands = (slices_y + kFieldP - 1) / kFieldP, n_tiles = strips_x * bands;
  const uint32_t lane = threadIdx.x, qx = lane % (kSliceW / 6), qy = lane / (kSliceW / 6);  // a wave = two output rows of the slice
  const uint32_t qpl = a.out_w / 6, n_quads = qpl * a.out_h;
  const uint32_t cap = a.window_capacity, per = kFieldBlock / batch;  // lanes per layer (>= window width + 6, host-checked)
  const uint32_t grp = lane / per < batch ? lane / per : batch - 1u, col = lane - (lane / per) * per;
  const bool spare = lane / per >= batch;  // lanes left over when the workgroup is not a multiple of the batch
  float4 *const tile_g = tile + (size_t)grp * cap;
  float *const share = reinterpret_cast<float *>(tile + (size_t)batch * cap) + (size_t)grp * 6u * per;  // 2 rows x 3 planes per layer
  int *const trips_box = reinterpret_cast<int *>(reinterpret_cast<float *>(tile + (size_t)batch * cap) + (size_t)batch * 6u * per);  // `batch` ints
  for (uint32_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
    const uint32_t sx = t % strips_x, band = t / strips_x, x_first = sx * kSliceW;
    // the row windows (see the kernel comment), centred on source row next_y; prev_y0 = first row of the LDS window
    float4 C0, C1, C2, C3, C4, O0, O1, O2, O3, O4, Q1, Q2, Q3;
    C0 = C1 = C2 = C3 = C4 = O0 = O1 = O2 = O3 = O4 = Q1 = Q2 = Q3 = make_float4(0, 0, 0, 0);
    int next_y = 0, prev_y0 = 0;
#pragma unroll 1
    for (uint32_t p = 0; p < (uint32_t)kFieldP; ++p) {
      const uint32_t sy = band * kFieldP + p;
      if (sy >= slices_y) break;  // uniform
      const uint32_t y_first = sy * kSliceH;
      const uint32_t y_last = (y_first + kSliceH - 1 < a.out_h ? y_first + kSliceH : a.out_h) - 1;
      const uint32_t y = y_first + qy < a.out_h ? y_first + qy : a.out_h - 1;  // rows past the frame recompute the last one
      PH_FSTAMP(0, p == 0 && t == blockIdx.x);
      // one batch of layers: windows (stage D), then filter + combine into acc (stage S).  A lambda so that with a single
      // batch the accumulator does not exist - and occupies no registers - while the windows are being built.
      auto do_batch = [&](const uint32_t b0, float (&acc)[18]) {
        // ---- stage D: lane group `grp` builds the window of layer b0 + grp ------------------------------------------
        const uint32_t g = b0 + grp;
        const bool have = !spare && g < (uint32_t)a.n;
        // this lane's layer parameters.  Indexing the kernel arguments with the per-lane g would make the compiler copy the
        // whole argument block to scratch memory; a loop over the (uniform) layer index with selects does not.
        const float4 *prev = nullptr, *cur = nullptr, *next = nullptr;
        int lw = 1, lh = 1, mode = 0, parity = 0, skip = 0, tff = 0;
        float mg[6] = {1, 0, 0, 0, 1, 0};
#pragma unroll 1
        for (uint32_t l = b0; l < b0 + batch && l < (uint32_t)a.n; ++l) {
          const bool mine = l == g;
          prev = mine ? reinterpret_cast<const float4 *>(a.prev[l]) : prev;
          cur = mine ? reinterpret_cast<const float4 *>(a.cur[l]) : cur;
          next = mine ? reinterpret_cast<const float4 *>(a.next[l]) : next;
          lw = mine ? a.lw[l] : lw, lh = mine ? a.lh[l] : lh, mode = mine ? a.mode[l] : mode;
          parity = mine ? a.parity[l] : parity, skip = mine ? a.skip[l] : skip, tff = mine ? a.tff[l] : tff;
#pragma unroll
          for (int i = 0; i < 6; ++i) mg[i] = mine ? a.matrix[l][i] : mg[i];
        }
        if (!DEINT) mode = 0;  // compile-time: no layer de-interlaces, the neighbour-frame windows and predictors fold away
        const int second = !(parity ^ tff);  // yadifCl.ts:143
        const float4 *other = second ? next : prev, *far = second ? prev : next;
        Window win = slice_window(mg, lw, lh, x_first, y_first, y_last, a.out_w, a.out_h);
        if (!have) win.w = win.h = 0;
        const int xr = win.x0 - 3 + (int)col, x = clampi(xr, 0, lw - 1);  // CLAMP_TO_EDGE
        const bool emit = have && col >= 3u && (int)col - 3 < win.w;
        const bool x_inside = xr >= 0 && xr < lw;  // a window column one past the image edge holds the border colour
        auto row = [&](const float4 *img, int yy) { return img[(size_t)clampi(yy, 0, lh - 1) * lw + x]; };
        __syncthreads();  // whoever sampled the windows of the previous batch / slice is done
        // rows [win.y0, next_y) may already be in the LDS window from the slice above: move them up, compute the rest
        const bool carry = CONTINUE && p != 0 && next_y >= win.y0 && win.h > 0;
        if (!carry) next_y = win.y0;
        const int keep = next_y - win.y0, shift = win.y0 - prev_y0;
        if (carry && emit && shift > 0)
          for (int k = 0; k < keep && k < win.h; ++k) tile_g[(uint32_t)k * (uint32_t)win.w + (col - 3u)] = tile_g[(uint32_t)(k + shift) * (uint32_t)win.w + (col - 3u)];
        const int trips = win.h - keep > 0 ? win.h - keep : 0;
        if (col == 0 && !spare) trips_box[grp] = trips;
        if (!carry && have && win.h > 0) {
          C0 = row(cur, next_y - 2), C1 = row(cur, next_y - 1), C2 = row(cur, next_y), C3 = row(cur, next_y + 1), C4 = row(cur, next_y + 2);
          if (mode != 0) {
            O0 = row(other, next_y - 2), O1 = row(other, next_y - 1), O2 = row(other, next_y), O3 = row(other, next_y + 1), O4 = row(other, next_y + 2);
            Q1 = row(far, next_y - 1), Q2 = row(far, next_y), Q3 = row(far, next_y + 1);
          }
        }
        prev_y0 = win.y0;
        __syncthreads();
        int steps = 0;  // uniform: the most rows any layer of the batch still has to build
        for (uint32_t k = 0; k < batch; ++k) steps = trips_box[k] > steps ? trips_box[k] : steps;
        if (!DEINT) {
          // no layer de-interlaces: the window is a copy of `cur` (with its border cells) - a lane's column, no taps shared,
          // nothing to wait for between rows: all the loads of a slice are in flight together
          if (emit)
            for (int step = 0; step < trips; ++step) {
              const int yy = next_y + step;
              float4 o = row(cur, yy);
              if (!(x_inside && yy >= 0 && yy < lh)) o = make_float4(0.f, 0.f, 0.f, 0.f);
              tile_g[(uint32_t)(yy - win.y0) * (uint32_t)win.w + (col - 3u)] = o;
            }
          next_y += trips;
          __syncthreads();
        }
#pragma unroll 1
        for (int step = 0; DEINT && step < steps; ++step) {
          const bool active = step < trips;
          const int yy = next_y;
          const bool interp = active && mode != 0 && (yy & 1) != parity;
          // rows y - 1 and y + 1 of `cur`, one plane per colour component (conflict-free 4-byte reads)
          float4 nC = C4, nO = O4, nQ = Q3;  // rows y+3 (cur, other) and y+2 (far): requested now, needed at the bottom
          if (active && (CONTINUE || step + 1 < trips)) {
            nC = row(cur, yy + 3);
            if (mode != 0) nO = row(other, yy + 3), nQ = row(far, yy + 2);
          }
          share[col] = C1.x, share[per + col] = C1.y, share[2u * per + col] = C1.z;
          share[3u * per + col] = C3.x, share[4u * per + col] = C3.y, share[5u * per + col] = C3.z;
          __syncthreads();
          if (emit && active) {
            float4 o = C2;  // a line of the field being kept, or a progressive layer (yadifCl.ts:117-121)
            if (interp) {
              // the fourteen taps of the spatial predictor, one colour component at a time: fourteen registers live
              // instead of fifty-six, next to the thirteen row windows
              float sp[3];
#pragma unroll
              for (int k = 0; k < 3; ++k) {
                const float *up = share + (uint32_t)k * per + col - 3, *dn = share + (uint32_t)(3 + k) * per + col - 3;
                sp[k] = yadif_spatial(up[0], up[1], up[2], up[3], up[4], up[5], up[6], dn[0], dn[1], dn[2], dn[3], dn[4], dn[5], dn[6]);
                __builtin_amdgcn_sched_barrier(0);
              }
              float res[3];
#pragma unroll
              for (int k = 0; k < 3; ++k) {
                const float c0 = PH_C3(C0, k), c2 = PH_C3(C2, k), c4 = PH_C3(C4, k);
                const float f0 = PH_C3(O0, k), f1 = PH_C3(O2, k), f2 = PH_C3(O4, k);
                // prev / next rows y-1, y+1: from O when it is that frame, else from Q
                const float p1 = second ? PH_C3(Q1, k) : PH_C3(O1, k), p3 = second ? PH_C3(Q3, k) : PH_C3(O3, k);
                const float n1 = second ? PH_C3(O1, k) : PH_C3(Q1, k), n3 = second ? PH_C3(O3, k) : PH_C3(Q3, k);
                // second field: s0 = cur, s1 = next; first field: s0 = prev, s1 = cur (:146-151)
                res[k] = yadif_temporal(p1, p3, second ? c0 : f0, second ? c2 : f1, second ? c4 : f2, PH_C3(C1, k), PH_C3(C3, k),
                                        second ? f0 : c0, second ? f1 : c2, second ? f2 : c4, n1, n3, sp[k], skip);
              }
              o = make_float4(res[0], res[1], res[2], C2.w);  // :164 alpha from cur
            }
            if (!(x_inside && yy >= 0 && yy < lh)) o = make_float4(0.f, 0.f, 0.f, 0.f);  // border colour (transform.ts:41 CLAMP)
            tile_g[(uint32_t)(yy - win.y0) * (uint32_t)win.w + (col - 3u)] = o;
          }
          __syncthreads();  // the shared rows are rewritten by the next step
          if (active) {
            C0 = C1, C1 = C2, C2 = C3, C3 = C4, C4 = nC;
            O0 = O1, O1 = O2, O2 = O3, O3 = O4, O4 = nO;
            Q1 = Q2, Q2 = Q3, Q3 = nQ;
            next_y = yy + 1;
          }
        }
        PH_FSTAMP(1, p == 0 && t == blockIdx.x && b0 == 0);
        // ---- stage S: the batch's windows are in LDS ---------------------------------------------------------------
        // pixel-centre coordinates of this lane's six pixels (transform.ts:53-54): the same for every layer
        float pxs[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) pxs[j] = (float)(int)(x_first + 6 * qx + j) / (float)(int)a.out_w - 0.5f;
        const float py = (float)(int)y / (float)(int)a.out_h - 0.5f;
#pragma unroll 1
        for (uint32_t l = b0; l < b0 + batch && l < (uint32_t)a.n; ++l) {
          float m[6];
#pragma unroll
          for (int i = 0; i < 6; ++i) m[i] = a.matrix[l][i];
          const Window wl = slice_window(m, a.lw[l], a.lh[l], x_first, y_first, y_last, a.out_w, a.out_h);
          sample_and_combine(tile + (size_t)(l - b0) * cap, wl, m, a.lw[l], a.lh[l], pxs, py, l == 0, acc);
        }
      };
      float acc[18];
      if (CONTINUE) {
        do_batch(0u, acc);
      } else {
#pragma unroll
        for (int i = 0; i < 18; ++i) acc[i] = 0.0f;
#pragma unroll 1
        for (uint32_t b0 = 0; b0 < (uint32_t)a.n; b0 += batch) do_batch(b0, acc);
      }
      PH_FSTAMP(2, p == 0 && t == blockIdx.x);
      const uint32_t line = y_first + qy;
      if (line < a.out_h) {
        const uint32_t q = line * qpl + sx * (kSliceW / 6) + qx;
#pragma unroll
        for (int i = 0; i < 9; ++i) {
          const uint32_t lo16 = __float_as_uint(lds_lut_index_unit(acc[2 * i]));
          const uint32_t hi16 = __float_as_uint(lds_lut_index_unit(acc[2 * i + 1]));
          __builtin_nontemporal_store(__builtin_amdgcn_perm(hi16, lo16, 0x05040100u), idx + (size_t)i * n_quads + q);
        }
      }
    }
  }
}

// Stage 2: indices -> writer gamma table (LDS) -> RGB->YCbCr matrix -> v210 words (v210.ts:145-162)
__global__ __launch_bounds__(kLdsBlock) void indices_to_v210_kernel(const uint32_t *__restrict__ idx, uint4 *__restrict__ out,
                                                                    uint32_t n_quads, const float *__restrict__ cm, LutView lut) {
  const WriteK wk = load_write_k(cm);
  const LutK wlut = make_lut_k(lut);
  lds_lut_load(lut);
  __syncthreads();
  for (uint32_t q = blockIdx.x * kLdsBlock + threadIdx.x; q < n_quads; q += gridDim.x * kLdsBlock) {
    float yi[18];
#pragma unroll
    for (int i = 0; i < 9; ++i) {  // M + idx: the index ORed into the mantissa of 1.5 * 2^23
      const uint32_t v = __builtin_nontemporal_load(idx + (size_t)i * n_quads + q);
      yi[2 * i] = __uint_as_float((v & 0xFFFFu) | 0x4B400000u);
      yi[2 * i + 1] = __uint_as_float((v >> 16) | 0x4B400000u);
    }
    store_stream(out + q, write_quad_idx_lds(yi, wk, wlut));  // out_w % 48 == 0: quad index == word-quad index
  }
}

// LDS a batch of `batch` layers needs: their windows, two shared rows per layer, the trip counts
static uint32_t field_lds_bytes(uint32_t batch, uint32_t cap) { return batch * cap * 16u + batch * 6u * (kFieldBlock / batch) * 4u + 64u; }

uint32_t field_index_bytes(uint32_t out_w, uint32_t out_h) { return out_w / 6 * out_h * 36u; }

hipError_t launch_field_compose_v210(hipStream_t s, const FieldArgs &a, void *index_scratch, uint32_t num_cus) {
  // the largest batch of layers whose windows fit the LDS together and whose lane groups cover a window's width plus
  // the three columns of taps either side
  uint32_t batch = (uint32_t)a.n;
  while (batch > 1 && (field_lds_bytes(batch, a.window_capacity) > 160u * 1024u || a.window_max_width + 6u > kFieldBlock / batch)) --batch;
  if (field_lds_bytes(batch, a.window_capacity) > 160u * 1024u || a.window_max_width + 6u > kFieldBlock / batch) return hipErrorInvalidValue;
  const uint32_t lds = field_lds_bytes(batch, a.window_capacity);
  bool deint = false;
  for (int l = 0; l < a.n; ++l) deint = deint || a.mode[l] != 0;
  const bool one = batch == (uint32_t)a.n;
  auto kernel = deint ? (one ? field_indices_kernel<true, true> : field_indices_kernel<false, true>)
                      : (one ? field_indices_kernel<true, false> : field_indices_kernel<false, false>);
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  const uint32_t slices_y = (a.out_h + kSliceH - 1) / kSliceH;
  const uint32_t n_tiles = (a.out_w / kSliceW) * ((slices_y + field_slices_per_strip(deint) - 1) / field_slices_per_strip(deint));
  // two workgroups per CU when the build and the LDS allow it: every strip gets its own workgroup
  const uint32_t resident = (!deint && lds <= 80u * 1024u ? 2u : 1u) * num_cus;
  kernel<<<n_tiles < resident ? n_tiles : resident, kFieldBlock, lds, s>>>(a, reinterpret_cast<uint32_t *>(index_scratch), batch);
  e = hipGetLastError();
  if (e != hipSuccess) return e;
  e = hipFuncSetAttribute(reinterpret_cast<const void *>(indices_to_v210_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)a.wr.bytes);
  if (e != hipSuccess) return e;
  const uint32_t n_quads = a.out_w / 6 * a.out_h, want = (n_quads + kLdsBlock - 1) / kLdsBlock;
  indices_to_v210_kernel<<<want < num_cus ? want : num_cus, kLdsBlock, a.wr.bytes, s>>>(
      reinterpret_cast<const uint32_t *>(index_scratch), reinterpret_cast<uint4 *>(a.out), n_quads, a.wr_cm, a.wr);
  return hipGetLastError();
}

// the largest source window any slice of this layer samples, as the kernel computes it - on the host
void field_window_extent(const float m[6], int lw, int lh, uint32_t out_w, uint32_t out_h, uint32_t *cols_out, uint32_t *rows_out) {
  // axis-aligned, not mirrored: the first tap column of the slice's last pixel lies at most ceil(span) columns right of
  // the first pixel's (floor(b) - floor(a) <= ceil(b - a)), and one more column holds its right-hand tap; rows likewise.
  // 1e-3 covers the f32 rounding of the coordinates (they are below 2^13, so their error is below 2^-10).
  const double span_x = (double)(kSliceW - 1) * (double)m[0] * (double)lw / (double)out_w + 1e-3;
  const double span_y = (double)(kSliceH - 1) * (double)m[4] * (double)lh / (double)out_h + 1e-3;
  const double cols = __builtin_ceil(span_x) + 2.0, rows = __builtin_ceil(span_y) + 2.0;
  const double c = cols > lw + 2 ? lw + 2 : cols, r = rows > lh + 2 ? lh + 2 : rows;  // the window may reach one cell past each edge
  *cols_out = (uint32_t)c, *rows_out = (uint32_t)r;
}

}  // namespace ph
