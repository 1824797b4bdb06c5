// ph_yadif.h - the per-component yadif predictors (reference yadifCl.ts:28-103), shared by the stand-alone
// yadif kernel (ph_kernels.hip) and the fused field pipeline (ph_kernels_field.hip).
#pragma once
#include "ph_device.h"

#pragma clang fp contract(off)

namespace ph {

__device__ __forceinline__ float yadif_spatial(float a, float b, float c, float d, float e, float f, float g,
                                               float h, float i, float j, float k, float l, float m, float n) {
  float pred = (d + k) / 2.0f;
  float best = __builtin_fabsf(c - j) + __builtin_fabsf(d - k) + __builtin_fabsf(e - l);
  float score = __builtin_fabsf(b - k) + __builtin_fabsf(c - l) + __builtin_fabsf(d - m);
  bool cmp = score < best;
  pred = cmp ? (c + l) / 2.0f : pred;
  best = cmp ? score : best;
  score = cmp ? __builtin_fabsf(a - l) + __builtin_fabsf(b - m) + __builtin_fabsf(c - n) : score;
  cmp = cmp && (score < best);
  pred = cmp ? (b + m) / 2.0f : pred;
  best = cmp ? score : best;

  score = __builtin_fabsf(d - i) + __builtin_fabsf(e - j) + __builtin_fabsf(f - k);
  cmp = score < best;
  pred = cmp ? (e + j) / 2.0f : pred;
  best = cmp ? score : best;
  score = cmp ? __builtin_fabsf(e - h) + __builtin_fabsf(f - i) + __builtin_fabsf(g - j) : score;
  cmp = cmp && (score < best);
  pred = cmp ? (f + i) / 2.0f : pred;
  return pred;
}

__device__ __forceinline__ float yadif_temporal(float A, float B, float C, float D, float E, float F, float G,
                                                float H, float I, float J, float K, float L, float pred,
                                                int skip) {
  const float p0 = (C + H) / 2.0f, p1 = F, p2 = (D + I) / 2.0f, p3 = G, p4 = (E + J) / 2.0f;
  const float t0 = __builtin_fabsf(D - I);
  const float t1 = (__builtin_fabsf(A - F) + __builtin_fabsf(B - G)) / 2.0f;
  const float t2 = (__builtin_fabsf(K - F) + __builtin_fabsf(G - L)) / 2.0f;
  float diff = __builtin_fmaxf(__builtin_fmaxf(t0, t1), t2);
  if (!skip) {
    const float p2mp3 = p2 - p3, p2mp1 = p2 - p1, p0mp1 = p0 - p1, p4mp3 = p4 - p3;
    const float maxi = __builtin_fmaxf(__builtin_fmaxf(p2mp3, p2mp1), __builtin_fminf(p0mp1, p4mp3));
    const float mini = __builtin_fminf(__builtin_fminf(p2mp3, p2mp1), __builtin_fmaxf(p0mp1, p4mp3));
    diff = __builtin_fmaxf(__builtin_fmaxf(diff, mini), -maxi);
  }
  pred = (pred > (p2 + diff)) ? p2 + diff : pred;
  pred = (pred < (p2 - diff)) ? p2 - diff : pred;
  return pred;
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

#define PH_C4(v, c) ((c) == 0 ? (v).x : (c) == 1 ? (v).y : (c) == 2 ? (v).z : (v).w)

}  // namespace ph
