// ph_lut.cpp - host side of ph_lut.h: build the compressed table, keep a registry of the
// LUTs known to a context (device f32 pointer -> compressed view) so kernels launched with a
// plain `gammaLut` pointer can find the LDS form.
#include "ph_lut_host.h"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace ph {

bool lut_compress(const float *lut, uint32_t max_bytes, std::vector<uint32_t> &blob, LutHostInfo &info) {
  uint32_t p[65536];
  std::memcpy(p, lut, sizeof p);
  bool found = false;
  for (uint32_t shift = 4; shift <= 6; ++shift) {
    const uint32_t B = 1u << shift;
    // T = first index (multiple of B) from which every block spans < 65536 bit patterns
    uint32_t toe = 0;
    for (uint32_t blk = 0; blk < 65536 / B; ++blk) {
      uint32_t lo = p[blk * B], hi = lo;
      for (uint32_t i = 1; i < B; ++i) lo = std::min(lo, p[blk * B + i]), hi = std::max(hi, p[blk * B + i]);
      if (hi - lo >= 65536u) toe = (blk + 1) * B;
    }
    uint32_t n_anchors = toe + (65536 - toe) / B;
    n_anchors = (n_anchors + 3) & ~3u;  // keep lo16[] 16-byte aligned
    const uint32_t bytes = n_anchors * 4 + 65536 * 2;
    if (bytes > max_bytes) continue;
    if (found && bytes >= info.bytes) continue;
    found = true;
    info.bytes = bytes, info.toe = toe, info.shift = shift, info.lo_off = n_anchors * 4;
    blob.assign(bytes / 4, 0);
    for (uint32_t i = 0; i < toe; ++i) blob[i] = p[i];
    for (uint32_t blk = toe / B; blk < 65536 / B; ++blk) {
      uint32_t lo = p[blk * B];
      for (uint32_t i = 1; i < B; ++i) lo = std::min(lo, p[blk * B + i]);
      blob[toe + (blk - toe / B)] = lo;
    }
    uint16_t *lo16 = reinterpret_cast<uint16_t *>(blob.data() + n_anchors);
    for (uint32_t i = 0; i < 65536; ++i) lo16[i] = (uint16_t)(p[i] & 0xffff);
  }
  if (!found) return false;
  // exhaustive self-check of the decode formula the kernels use
  const uint16_t *lo16 = reinterpret_cast<const uint16_t *>(reinterpret_cast<const uint8_t *>(blob.data()) + info.lo_off);
  for (uint32_t i = 0; i < 65536; ++i) {
    const uint32_t b = std::min(i, info.toe + ((i - info.toe) >> info.shift));
    const uint32_t a = blob[b];
    const uint32_t v = a + ((lo16[i] - a) & 0xffffu);
    if (v != p[i]) return false;
  }
  return true;
}

// ---- anchor predictor ---------------------------------------------------------------------------
namespace {

// the device evaluates exactly these IEEE operations (ph_kernels_lds.hip: lds_lut_get_pred)
inline float pred_eval(const LutPredictor &p, float r) {
  const float u = std::fmaf(r, p.a, p.b);
  float q = std::fmaf(u, p.q[4], p.q[3]);
  q = std::fmaf(u, q, p.q[2]);
  q = std::fmaf(u, q, p.q[1]);
  q = std::fmaf(u, q, p.q[0]);
  const float pw = (u * u) * q;
  const float toe = r * p.toe_slope;
  return r < p.knee ? toe : pw;
}

int64_t pred_max_err(const LutPredictor &p, const float *lut) {
  int64_t worst = 0;
  for (uint32_t i = 0; i < 65536; ++i) {
    const float v = pred_eval(p, (float)i);
    uint32_t pb, tb;
    std::memcpy(&pb, &v, 4);
    std::memcpy(&tb, &lut[i], 4);
    const int64_t d = (int64_t)tb - (int64_t)pb;
    worst = std::max<int64_t>(worst, d < 0 ? -d : d);
  }
  return worst;
}

// weighted least squares, degree 4, normal equations in double (5x5 Gaussian elimination)
bool solve5(double m[5][6]) {
  for (int c = 0; c < 5; ++c) {
    int piv = c;
    for (int r = c + 1; r < 5; ++r)
      if (std::fabs(m[r][c]) > std::fabs(m[piv][c])) piv = r;
    if (std::fabs(m[piv][c]) < 1e-300) return false;
    for (int k = 0; k < 6; ++k) std::swap(m[c][k], m[piv][k]);
    for (int r = 0; r < 5; ++r) {
      if (r == c) continue;
      const double f = m[r][c] / m[c][c];
      for (int k = c; k < 6; ++k) m[r][k] -= f * m[c][k];
    }
  }
  for (int c = 0; c < 5; ++c) m[c][5] /= m[c][c];
  return true;
}

}  // namespace

bool lut_fit_predictor(const float *lut, LutPredictor &out) {
  // candidate OETF shapes: (alpha, beta, gamma, delta) of BT.601/709/2020 and sRGB
  static const double cand[2][4] = {{1.099, 0.018, 0.45, 4.5}, {1.055, 0.0031308, 1.0 / 2.4, 12.92}};
  out.ok = 0;
  for (const double *c : {cand[0], cand[1]}) {
    LutPredictor p{};
    const double alpha = c[0], knee_fi = c[1] * c[3];
    p.a = (float)(1.0 / (65535.0 * alpha));
    p.b = (float)((alpha - 1.0) / alpha);
    p.toe_slope = lut[1];
    uint32_t knee = 0;
    while (knee < 65536 && (double)knee / 65535.0 < knee_fi) ++knee;
    p.knee = (float)knee;
    if (knee < 2 || knee > 60000) continue;
    // fit q(u) ~ table / u^2 on the power segment, minimising relative error; a few Lawson
    // re-weightings push the least-squares fit towards minimax
    std::vector<double> w(65536, 1.0);
    for (int iter = 0; iter < 6; ++iter) {
      double m[5][6] = {};
      for (uint32_t i = knee; i < 65536; i += 4) {
        const double u = (double)std::fmaf((float)i, p.a, p.b);
        const double t = (double)lut[i] / (u * u);
        if (!(t > 0)) continue;
        const double wt = w[i] / (t * t);
        double pw[5] = {1, u, u * u, u * u * u, u * u * u * u};
        for (int r = 0; r < 5; ++r) {
          for (int k = 0; k < 5; ++k) m[r][k] += wt * pw[r] * pw[k];
          m[r][5] += wt * pw[r] * t;
        }
      }
      if (!solve5(m)) break;
      for (int k = 0; k < 5; ++k) p.q[k] = (float)m[k][5];
      const int64_t err = pred_max_err(p, lut);
      if (err < 32768 && (!out.ok || err < 30000)) {
        p.ok = 1;
        out = p;
        if (err < 24000) return true;  // comfortable margin
      }
      for (uint32_t i = knee; i < 65536; i += 4) {  // Lawson: weight *= |relative error|
        const float v = pred_eval(p, (float)i);
        const double rel = std::fabs(((double)v - (double)lut[i]) / (double)lut[i]);
        w[i] *= std::max(rel, 1e-9);
      }
      double s = 0;
      for (uint32_t i = knee; i < 65536; i += 4) s += w[i];
      for (uint32_t i = knee; i < 65536; i += 4) w[i] /= s;
    }
    if (out.ok) return true;
  }
  return out.ok != 0;
}

}  // namespace ph
