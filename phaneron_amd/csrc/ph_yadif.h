// ph_yadif.h - the per-component yadif predictors, shared by the stand-alone yadif kernel (ph_kernels.hip), the fused
// de-interlacing reader (ph_kernels_deint.hip).
//
// yadif is FFmpeg's libavfilter/vf_yadif.c (filter_line_c): an edge-directed spatial interpolation between the lines
// above and below, limited to a band around the temporal average whose width comes from how much the neighbouring
// fields changed.  The reference runs it per f32 RGBA component (src/process/yadifCl.ts:28-167); the order of the
// operations below is the one that reproduces its results bit for bit (the golden vectors of tests/golden).
#pragma once
#include "ph_device.h"

#pragma clang fp contract(off)

namespace ph {

// Spatial candidate: vf_yadif's CHECK(j).  `up` / `dn` are the seven samples x-3..x+3 of the lines above and below.
// Direction j pairs up[3 + j] with dn[3 - j]; its score is the mismatch of the three sample pairs around that line.
struct YadifEdge {
  float score, value;
};
__device__ __forceinline__ YadifEdge yadif_edge(const float (&up)[7], const float (&dn)[7], int j) {
  YadifEdge e;
  e.score = __builtin_fabsf(up[2 + j] - dn[2 - j]) + __builtin_fabsf(up[3 + j] - dn[3 - j]) + __builtin_fabsf(up[4 + j] - dn[4 - j]);
  e.value = (up[3 + j] + dn[3 - j]) / 2.0f;
  return e;
}

// The best of five directions: straight down first, then one and two steps to the left, then one and two to the right;
// the second step of a side is tried only if its first step was taken, and every comparison is against the best so far.
__device__ __forceinline__ float yadif_spatial(float up_m3, float up_m2, float up_m1, float up_0, float up_p1, float up_p2,
                                               float up_p3, float dn_m3, float dn_m2, float dn_m1, float dn_0, float dn_p1,
                                               float dn_p2, float dn_p3) {
  const float up[7] = {up_m3, up_m2, up_m1, up_0, up_p1, up_p2, up_p3};
  const float dn[7] = {dn_m3, dn_m2, dn_m1, dn_0, dn_p1, dn_p2, dn_p3};
  YadifEdge best = yadif_edge(up, dn, 0);
#pragma unroll
  for (int side = -1; side <= 1; side += 2) {
    const YadifEdge near = yadif_edge(up, dn, side), far = yadif_edge(up, dn, 2 * side);
    const bool take_near = near.score < best.score;
    best.value = take_near ? near.value : best.value;
    best.score = take_near ? near.score : best.score;
    const bool take_far = take_near && far.score < best.score;
    best.value = take_far ? far.value : best.value;
    best.score = take_far ? far.score : best.score;
  }
  return best.value;
}

// Temporal limit (vf_yadif FILTER: c, d, e, temporal_diff0..2, and the b / f terms of the spatial check).
//   prev_up / prev_dn, next_up / next_dn : the lines above / below in the previous and next frame
//   cur_up / cur_dn                      : the same in the current frame
//   old_* / new_*                        : lines y-2, y, y+2 of the two fields that bracket this one in time
// The spatial prediction is clamped to [mid - band, mid + band], mid = the temporal average of line y.
__device__ __forceinline__ float yadif_temporal(float prev_up, float prev_dn, float old_m2, float old_0, float old_p2,
                                                float cur_up, float cur_dn, float new_m2, float new_0, float new_p2,
                                                float next_up, float next_dn, float spatial, int skip_spatial_check) {
  const float avg_m2 = (old_m2 + new_m2) / 2.0f, mid = (old_0 + new_0) / 2.0f, avg_p2 = (old_p2 + new_p2) / 2.0f;
  const float change_here = __builtin_fabsf(old_0 - new_0);
  const float change_before = (__builtin_fabsf(prev_up - cur_up) + __builtin_fabsf(prev_dn - cur_dn)) / 2.0f;
  const float change_after = (__builtin_fabsf(next_up - cur_up) + __builtin_fabsf(cur_dn - next_dn)) / 2.0f;
  float band = __builtin_fmaxf(__builtin_fmaxf(change_here, change_before), change_after);
  if (!skip_spatial_check) {
    const float over_dn = mid - cur_dn, over_up = mid - cur_up;          // vf_yadif: d - e, d - c
    const float far_up = avg_m2 - cur_up, far_dn = avg_p2 - cur_dn;      // vf_yadif: b, f
    const float hi = __builtin_fmaxf(__builtin_fmaxf(over_dn, over_up), __builtin_fminf(far_up, far_dn));
    const float lo = __builtin_fminf(__builtin_fminf(over_dn, over_up), __builtin_fmaxf(far_up, far_dn));
    band = __builtin_fmaxf(__builtin_fmaxf(band, lo), -hi);
  }
  float out = (spatial > (mid + band)) ? mid + band : spatial;
  out = (out < (mid - band)) ? mid - band : out;
  return out;
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

#define PH_C4(v, c) ((c) == 0 ? (v).x : (c) == 1 ? (v).y : (c) == 2 ? (v).z : (v).w)

}  // namespace ph
