// ph_colour.cpp - host colour maths of the product: what the reference's Loader / Saver
// constructors compute in TypeScript (src/process/colourMaths.ts, loadSave.ts:50-63,139-149)
// and Transform.getKernelParams (transform.ts:119-171).
//
// The reference stores every matrix as rows of Float32Array and does the arithmetic in JS
// doubles, so each element rounds to f32 when stored while sums/products inside one element
// are f64.  `F32Mat` models exactly that: elements are float, `mul`/`scaled` compute in
// double and round once per element.
#include <cmath>
#include <cstring>
#include <initializer_list>

#include "../../include/phaneron_hip.h"

namespace {

struct ColSpec {
  const char *name;
  double kR, kB;                          // luma coefficients
  double rx, ry, gx, gy, bx, by, wx, wy;  // primaries and white point (CIE xy)
  double alpha, beta, gamma, delta;       // OETF parameters
};

// colourMaths.ts:42-128 (ITU-R BT.601-7, BT.709-6, BT.2020-2, sRGB constants)
const ColSpec kSpecs[] = {
    {"601-625", 0.299, 0.114, 0.64, 0.33, 0.29, 0.6, 0.15, 0.06, 0.3127, 0.329, 1.099, 0.018, 0.45, 4.5},
    {"601_525", 0.299, 0.114, 0.63, 0.34, 0.31, 0.595, 0.155, 0.07, 0.3127, 0.329, 1.099, 0.018, 0.45, 4.5},
    {"709", 0.2126, 0.0722, 0.64, 0.33, 0.3, 0.6, 0.15, 0.06, 0.3127, 0.329, 1.099, 0.018, 0.45, 4.5},
    {"2020", 0.2627, 0.0593, 0.708, 0.292, 0.17, 0.797, 0.131, 0.046, 0.3127, 0.329, 1.099, 0.018, 0.45, 4.5},
    {"sRGB", 0.0, 0.0, 0.64, 0.33, 0.3, 0.6, 0.15, 0.06, 0.3127, 0.329, 1.055, 0.0031308, 1.0 / 2.4, 12.92},
};

const ColSpec &spec_of(const char *name) {
  if (name)
    for (const ColSpec &s : kSpecs)
      if (0 == std::strcmp(name, s.name)) return s;
  return kSpecs[2];  // unknown colourspace: the reference defaults to BT.709
}

template <int R, int C>
struct F32Mat {
  float e[R][C];
  F32Mat() { std::memset(e, 0, sizeof e); }
  F32Mat(std::initializer_list<double> vals) {
    int i = 0;
    for (double v : vals) {
      e[i / C][i % C] = static_cast<float>(v);
      ++i;
    }
  }
  template <int K>
  F32Mat<R, K> mul(const F32Mat<C, K> &b) const {  // colourMaths.ts:171-178
    F32Mat<R, K> out;
    for (int i = 0; i < R; ++i)
      for (int j = 0; j < K; ++j) {
        double acc = 0.0;
        for (int k = 0; k < C; ++k) acc += static_cast<double>(e[i][k]) * static_cast<double>(b.e[k][j]);
        out.e[i][j] = static_cast<float>(acc);
      }
    return out;
  }
  F32Mat scaled(double c) const {  // :180-187
    F32Mat out;
    for (int i = 0; i < R; ++i)
      for (int j = 0; j < C; ++j) out.e[i][j] = static_cast<float>(static_cast<double>(e[i][j]) * c);
    return out;
  }
  void flatten(float *dst) const { std::memcpy(dst, e, sizeof e); }  // :396-401
};
using M33 = F32Mat<3, 3>;

M33 inverse(const M33 &a) {  // :199-238: minors -> cofactors -> adjugate -> / determinant
  auto d = [&](int r, int c) { return static_cast<double>(a.e[r][c]); };
  static const int other[3][2] = {{1, 2}, {0, 2}, {0, 1}};
  M33 minors;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      const int *r = other[i], *c = other[j];
      minors.e[i][j] = static_cast<float>(d(r[0], c[0]) * d(r[1], c[1]) - d(r[0], c[1]) * d(r[1], c[0]));
    }
  M33 adjugate;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      const double sign = ((i + j) % 2) ? -1.0 : 1.0;
      adjugate.e[j][i] = static_cast<float>(static_cast<double>(minors.e[i][j]) * sign);
    }
  const double det = d(0, 0) * static_cast<double>(minors.e[0][0]) - d(0, 1) * static_cast<double>(minors.e[0][1]) +
                     d(0, 2) * static_cast<double>(minors.e[0][2]);
  return adjugate.scaled(1.0 / det);
}

M33 rgb_to_xyz(const ColSpec &s) {  // :240-266
  const F32Mat<3, 1> white{s.wx, s.wy, 1.0 - s.wx - s.wy};
  const F32Mat<3, 1> white_n = white.scaled(1.0 / static_cast<double>(white.e[1][0]));
  const M33 prim{s.rx, s.gx, s.bx, s.ry, s.gy, s.by, 1.0 - s.rx - s.ry, 1.0 - s.gx - s.gy, 1.0 - s.bx - s.by};
  const F32Mat<3, 1> f = inverse(prim).mul(white_n);
  M33 diag;
  diag.e[0][0] = f.e[0][0], diag.e[1][1] = f.e[1][0], diag.e[2][2] = f.e[2][0];
  return prim.mul(diag);
}

inline double js_or(double v, double fallback) { return (v == 0.0 || std::isnan(v)) ? fallback : v; }

}  // namespace

extern "C" {

int ph_colour_gamma2linear_lut(const char *colspec, float *lut) {  // :130-149
  if (!lut) return PH_E_INVALID;
  const ColSpec &s = spec_of(colspec);
  const double knee = s.beta * s.delta;
  for (int i = 0; i < 65536; ++i) {
    const double fi = i / 65535.0;
    lut[i] = static_cast<float>(fi < knee ? fi / s.delta : std::pow((fi + (s.alpha - 1)) / s.alpha, 1 / s.gamma));
  }
  return PH_OK;
}

int ph_colour_linear2gamma_lut(const char *colspec, float *lut) {  // :151-169
  if (!lut) return PH_E_INVALID;
  const ColSpec &s = spec_of(colspec);
  for (int i = 0; i < 65536; ++i) {
    const double fi = i / 65535.0;
    lut[i] = static_cast<float>(fi < s.beta ? fi * s.delta : s.alpha * std::pow(fi, s.gamma) - (s.alpha - 1));
  }
  return PH_OK;
}

int ph_colour_ycbcr2rgb_matrix(const char *colspec, int num_bits, int luma_black, int luma_white, int chroma_range,
                               float *m12) {  // :276-332
  if (!m12 || num_bits < 8) return PH_E_INVALID;
  const ColSpec &s = spec_of(colspec);
  const double chr_null = 128 << (num_bits - 8), luma_range = luma_white - luma_black, chr = chroma_range;
  const double kR = s.kR, kB = s.kB, kG = 1.0 - kR - kB;
  const M33 col{1.0, 0.0, 1.0 - kR, 1.0, (-(1.0 - kB) * kB) / kG, (-(1.0 - kR) * kR) / kG, 1.0, 1.0 - kB, 0.0};
  const F32Mat<3, 4> scale{1.0 / luma_range, 0.0, 0.0, -luma_black / luma_range,
                           0.0, (1.0 / chr) * 2, 0.0, -(chr_null / chr) * 2,
                           0.0, 0.0, (1.0 / chr) * 2, -(chr_null / chr) * 2};
  col.mul(scale).flatten(m12);
  return PH_OK;
}

int ph_colour_rgb2ycbcr_matrix(const char *colspec, int num_bits, int luma_black, int luma_white, int chroma_range,
                               float *m12) {  // :334-390
  if (!m12 || num_bits < 8) return PH_E_INVALID;
  const ColSpec &s = spec_of(colspec);
  const double chr_null = 128 << (num_bits - 8), luma_range = luma_white - luma_black, chr = chroma_range;
  const double kR = s.kR, kB = s.kB, kG = 1.0 - kR - kB;
  const M33 scale{luma_range, 0.0, 0.0, 0.0, chr / 2.0, 0.0, 0.0, 0.0, chr / 2.0};
  const F32Mat<3, 4> col{kR, kG, kB, luma_black / luma_range,
                         -kR / (1.0 - kB), -kG / (1.0 - kB), (1.0 - kB) / (1.0 - kB), (chr_null / chr) * 2.0,
                         (1.0 - kR) / (1.0 - kR), -kG / (1.0 - kR), -kB / (1.0 - kR), (chr_null / chr) * 2.0};
  scale.mul(col).flatten(m12);
  return PH_OK;
}

int ph_colour_rgb2rgb_matrix(const char *src, const char *dst, float *m9) {  // :392-394
  if (!m9) return PH_E_INVALID;
  inverse(rgb_to_xyz(spec_of(dst))).mul(rgb_to_xyz(spec_of(src))).flatten(m9);
  return PH_OK;
}

int ph_transform_matrix(int width, int height, int flip_h, int flip_v, double anchor_x, double anchor_y,
                        double scale_x, double scale_y, double offset_x, double offset_y, double rotate,
                        float *m9) {  // transform.ts:119-171
  if (!m9 || width <= 0 || height <= 0) return PH_E_INVALID;
  const double aspect = static_cast<double>(width) / height;
  const double ax = js_or(anchor_x, 0.0), ay = js_or(anchor_y, 0.0);
  const double sx = js_or(scale_x, 1.0) * (flip_h ? -1.0 : 1.0), sy = js_or(scale_y, 1.0) * (flip_v ? -1.0 : 1.0);
  const double ox = js_or(offset_x, 0.0), oy = js_or(offset_y, 0.0);
  const double th = js_or(rotate, 0.0) * 2 * M_PI;
  const M33 anchor_in{1.0, 0.0, ax, 0.0, 1.0, ay, 0.0, 0.0, 1.0};
  const M33 scale{1.0 / (sx * aspect), 0.0, 0.0, 0.0, 1.0 / sy, 0.0, 0.0, 0.0, 1.0};
  const M33 rot{std::cos(th), -std::sin(th), 0.0, std::sin(th), std::cos(th), 0.0, 0.0, 0.0, 1.0};
  const M33 translate{1.0, 0.0, ox * aspect, 0.0, 1.0, oy, 0.0, 0.0, 1.0};
  const M33 anchor_out{1.0, 0.0, -ax * aspect, 0.0, 1.0, -ay, 0.0, 0.0, 1.0};
  const M33 project{aspect, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 1.0};
  anchor_in.mul(scale).mul(rot).mul(translate).mul(anchor_out).mul(project).flatten(m9);
  return PH_OK;
}

}  // extern "C"
