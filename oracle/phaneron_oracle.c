/* phaneron_oracle.c - CPU restatement of the reference hot path (see phaneron_oracle.h).
 *
 * TEST INFRASTRUCTURE ONLY - the checker, never the product.
 *
 * Build with -ffp-contract=off: every rounding step below is deliberate.  `fmaf` is the
 * correctly-rounded fused multiply-add that OpenCL `fma` / AMD's `dot` lowering use
 * (SURVEY.md 7 "hard parts"; ROCm 7.2 opencl.bc: dot(float4) = fma(a3,b3,fma(a2,b2,fma(a1,b1,a0*b0)))).
 */
#include "phaneron_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static int g_threads = 0; /* 0 = OpenMP default */

int orc_num_threads(void) {
#ifdef _OPENMP
  return g_threads > 0 ? g_threads : omp_get_max_threads();
#else
  return 1;
#endif
}
void orc_set_num_threads(int n) { g_threads = n; }
#ifdef _OPENMP
#define ORC_PAR_FOR _Pragma("omp parallel for schedule(static) num_threads(orc_num_threads())")
#else
#define ORC_PAR_FOR
#endif

/* ======================================================================================== */
/* colourMaths.ts                                                                           */
/* ======================================================================================== */
typedef struct {
  const char *name;
  double kR, kB, rx, ry, gx, gy, bx, by, wx, wy, alpha, beta, gamma, delta;
} col_param;

/* the table at colourMaths.ts:42-128 (published ITU-R / sRGB constants) */
static const col_param k_col_params[] = {
    {"601-625", 0.299, 0.114, 0.64, 0.33, 0.29, 0.6, 0.15, 0.06, 0.3127, 0.329, 1.099, 0.018, 0.45, 4.5},
    {"601_525", 0.299, 0.114, 0.63, 0.34, 0.31, 0.595, 0.155, 0.07, 0.3127, 0.329, 1.099, 0.018, 0.45, 4.5},
    {"709", 0.2126, 0.0722, 0.64, 0.33, 0.3, 0.6, 0.15, 0.06, 0.3127, 0.329, 1.099, 0.018, 0.45, 4.5},
    {"2020", 0.2627, 0.0593, 0.708, 0.292, 0.17, 0.797, 0.131, 0.046, 0.3127, 0.329, 1.099, 0.018, 0.45, 4.5},
    {"sRGB", 0.0, 0.0, 0.64, 0.33, 0.3, 0.6, 0.15, 0.06, 0.3127, 0.329, 1.055, 0.0031308, 1.0 / 2.4, 12.92},
};

static const col_param *find_col(const char *spec) {
  size_t i;
  for (i = 0; i < sizeof k_col_params / sizeof k_col_params[0]; ++i)
    if (spec && 0 == strcmp(spec, k_col_params[i].name)) return &k_col_params[i];
  return &k_col_params[2]; /* "defaulting to BT.709" (colourMaths.ts:131-134) */
}

int orc_gamma2linear_lut(const char *colspec, float *lut) {
  const col_param *c = find_col(colspec);
  const double alpha = c->alpha, delta = c->delta, beta = c->beta * delta, gamma = c->gamma;
  int i;
  for (i = 0; i < 65536; ++i) {
    const double fi = (double)i / 65535.0;
    if (fi < beta)
      lut[i] = (float)(fi / delta);
    else
      lut[i] = (float)pow((fi + (alpha - 1)) / alpha, 1 / gamma);
  }
  return 0;
}

int orc_linear2gamma_lut(const char *colspec, float *lut) {
  const col_param *c = find_col(colspec);
  const double alpha = c->alpha, beta = c->beta, gamma = c->gamma, delta = c->delta;
  int i;
  for (i = 0; i < 65536; ++i) {
    const double fi = (double)i / 65535.0;
    if (fi < beta)
      lut[i] = (float)(fi * delta);
    else
      lut[i] = (float)(alpha * pow(fi, gamma) - (alpha - 1));
  }
  return 0;
}

/* Matrices are rows of f32 (the reference's Float32Array rows); products and sums are f64,
 * every store rounds to f32 (colourMaths.ts:171-238). */
typedef struct {
  int r, c;
  float v[3][4];
} mat;

static mat mat_mul(const mat *a, const mat *b) { /* :171-178 */
  mat o;
  int i, j, k;
  o.r = a->r;
  o.c = b->c;
  for (i = 0; i < a->r; ++i)
    for (j = 0; j < b->c; ++j) {
      double sum = 0.0;
      for (k = 0; k < a->c; ++k) sum = sum + (double)a->v[i][k] * (double)b->v[k][j];
      o.v[i][j] = (float)sum;
    }
  return o;
}

static mat mat_scale(const mat *a, double c) { /* :180-187 */
  mat o = *a;
  int i, j;
  for (i = 0; i < a->r; ++i)
    for (j = 0; j < a->c; ++j) o.v[i][j] = (float)((double)a->v[i][j] * c);
  return o;
}

static mat mat_invert3(const mat *a) { /* :199-238 */
  mat minors, adj;
  int i, j;
  double det;
  minors.r = minors.c = adj.r = adj.c = 3;
  for (i = 0; i < 3; ++i)
    for (j = 0; j < 3; ++j) {
      const int y0 = (1 == i) ? 0 : (i + 1) % 3, y1 = (1 == i) ? 2 : (i + 2) % 3;
      const int x0 = (1 == j) ? 0 : (j + 1) % 3, x1 = (1 == j) ? 2 : (j + 2) % 3;
      minors.v[i][j] = (float)((double)a->v[y0][x0] * (double)a->v[y1][x1] -
                               (double)a->v[y0][x1] * (double)a->v[y1][x0]);
    }
  for (i = 0; i < 3; ++i)
    for (j = 0; j < 3; ++j) {
      const float cof = (float)((double)minors.v[i][j] * (((i + j) & 1) ? -1.0 : 1.0));
      adj.v[j][i] = cof; /* transpose of the cofactor matrix */
    }
  det = (double)a->v[0][0] * (double)minors.v[0][0] - (double)a->v[0][1] * (double)minors.v[0][1] +
        (double)a->v[0][2] * (double)minors.v[0][2];
  return mat_scale(&adj, 1.0 / det);
}

static mat rgb2xyz(const col_param *c) { /* :240-266 */
  mat w, W, xyz, inv, sf, sc;
  memset(&sc, 0, sizeof sc);
  w.r = 3;
  w.c = 1;
  w.v[0][0] = (float)c->wx;
  w.v[1][0] = (float)c->wy;
  w.v[2][0] = (float)(1.0 - c->wx - c->wy);
  W = mat_scale(&w, 1.0 / (double)w.v[1][0]);
  xyz.r = xyz.c = 3;
  xyz.v[0][0] = (float)c->rx, xyz.v[0][1] = (float)c->gx, xyz.v[0][2] = (float)c->bx;
  xyz.v[1][0] = (float)c->ry, xyz.v[1][1] = (float)c->gy, xyz.v[1][2] = (float)c->by;
  xyz.v[2][0] = (float)(1.0 - c->rx - c->ry);
  xyz.v[2][1] = (float)(1.0 - c->gx - c->gy);
  xyz.v[2][2] = (float)(1.0 - c->bx - c->by);
  inv = mat_invert3(&xyz);
  sf = mat_mul(&inv, &W);
  sc.r = sc.c = 3;
  sc.v[0][0] = sf.v[0][0];
  sc.v[1][1] = sf.v[1][0];
  sc.v[2][2] = sf.v[2][0];
  return mat_mul(&xyz, &sc);
}

static void mat_flatten(const mat *m, float *out) { /* :396-401 */
  int i, j;
  for (i = 0; i < m->r; ++i)
    for (j = 0; j < m->c; ++j) out[i * m->c + j] = m->v[i][j];
}

int orc_rgb2rgb_matrix(const char *src, const char *dst, float *m9) {
  mat d = rgb2xyz(find_col(dst)), s = rgb2xyz(find_col(src));
  mat di = mat_invert3(&d); /* xyz2rgbMatrix(dst) :268-274 */
  mat o = mat_mul(&di, &s);
  mat_flatten(&o, m9);
  return 0;
}

int orc_ycbcr2rgb_matrix(const char *colspec, int num_bits, int luma_black, int luma_white,
                         int chr_range, float *m12) {
  const col_param *c = find_col(colspec);
  const double chr_null = (double)(128 << (num_bits - 8));
  const double luma_range = (double)(luma_white - luma_black);
  const double kR = c->kR, kB = c->kB, kG = 1.0 - kR - kB;
  mat cm, sm, o;
  cm.r = 3, cm.c = 3;
  cm.v[0][0] = 1.0f, cm.v[0][1] = 0.0f, cm.v[0][2] = (float)(1.0 - kR);
  cm.v[1][0] = 1.0f, cm.v[1][1] = (float)((-(1.0 - kB) * kB) / kG), cm.v[1][2] = (float)((-(1.0 - kR) * kR) / kG);
  cm.v[2][0] = 1.0f, cm.v[2][1] = (float)(1.0 - kB), cm.v[2][2] = 0.0f;
  sm.r = 3, sm.c = 4;
  memset(sm.v, 0, sizeof sm.v);
  sm.v[0][0] = (float)(1.0 / luma_range);
  sm.v[0][3] = (float)(-(double)luma_black / luma_range);
  sm.v[1][1] = (float)((1.0 / (double)chr_range) * 2);
  sm.v[1][3] = (float)(-(chr_null / (double)chr_range) * 2);
  sm.v[2][2] = (float)((1.0 / (double)chr_range) * 2);
  sm.v[2][3] = (float)(-(chr_null / (double)chr_range) * 2);
  o = mat_mul(&cm, &sm);
  mat_flatten(&o, m12);
  return 0;
}

int orc_rgb2ycbcr_matrix(const char *colspec, int num_bits, int luma_black, int luma_white,
                         int chr_range, float *m12) {
  const col_param *c = find_col(colspec);
  const double chr_null = (double)(128 << (num_bits - 8));
  const double luma_range = (double)(luma_white - luma_black);
  const double kR = c->kR, kB = c->kB, kG = 1.0 - kR - kB;
  mat sm, cm, o;
  sm.r = 3, sm.c = 3;
  memset(sm.v, 0, sizeof sm.v);
  sm.v[0][0] = (float)luma_range;
  sm.v[1][1] = (float)((double)chr_range / 2.0);
  sm.v[2][2] = (float)((double)chr_range / 2.0);
  cm.r = 3, cm.c = 4;
  cm.v[0][0] = (float)kR, cm.v[0][1] = (float)kG, cm.v[0][2] = (float)kB;
  cm.v[0][3] = (float)((double)luma_black / luma_range);
  cm.v[1][0] = (float)(-kR / (1.0 - kB)), cm.v[1][1] = (float)(-kG / (1.0 - kB));
  cm.v[1][2] = (float)((1.0 - kB) / (1.0 - kB));
  cm.v[1][3] = (float)((chr_null / (double)chr_range) * 2.0);
  cm.v[2][0] = (float)((1.0 - kR) / (1.0 - kR)), cm.v[2][1] = (float)(-kG / (1.0 - kR));
  cm.v[2][2] = (float)(-kB / (1.0 - kR));
  cm.v[2][3] = (float)((chr_null / (double)chr_range) * 2.0);
  o = mat_mul(&sm, &cm);
  mat_flatten(&o, m12);
  return 0;
}

static double js_or(double v, double dflt) { return (v == 0.0 || v != v) ? dflt : v; } /* `v || d` */

static mat mat3(double a, double b, double c, double d, double e, double f) {
  mat m;
  m.r = m.c = 3;
  m.v[0][0] = (float)a, m.v[0][1] = (float)b, m.v[0][2] = (float)c;
  m.v[1][0] = (float)d, m.v[1][1] = (float)e, m.v[1][2] = (float)f;
  m.v[2][0] = 0.0f, m.v[2][1] = 0.0f, m.v[2][2] = 1.0f;
  return m;
}

void orc_transform_matrix(int width, int height, int flip_h, int flip_v, double anchor_x,
                          double anchor_y, double scale_x, double scale_y, double offset_x,
                          double offset_y, double rotate, float *m9) { /* transform.ts:119-171 */
  const double aspect = (double)width / (double)height;
  const double flipX = flip_h ? -1.0 : 1.0, flipY = flip_v ? -1.0 : 1.0;
  const double anchorX = js_or(anchor_x, 0.0), anchorY = js_or(anchor_y, 0.0);
  const double scaleX = js_or(scale_x, 1.0) * flipX, scaleY = js_or(scale_y, 1.0) * flipY;
  const double offsetX = js_or(offset_x, 0.0), offsetY = js_or(offset_y, 0.0);
  const double rot = js_or(rotate, 0.0) * 2 * M_PI;
  mat anchor_in = mat3(1.0, 0.0, anchorX, 0.0, 1.0, anchorY);
  mat scale = mat3(1.0 / (scaleX * aspect), 0.0, 0.0, 0.0, 1.0 / scaleY, 0.0);
  mat rotm = mat3(cos(rot), -sin(rot), 0.0, sin(rot), cos(rot), 0.0);
  mat translate = mat3(1.0, 0.0, offsetX * aspect, 0.0, 1.0, offsetY);
  mat anchor_out = mat3(1.0, 0.0, -anchorX * aspect, 0.0, 1.0, -anchorY);
  mat project = mat3(aspect, 0.0, 0.0, 0.0, 1.0, 0.0);
  mat m = mat_mul(&anchor_in, &scale);
  m = mat_mul(&m, &rotm);
  m = mat_mul(&m, &translate);
  m = mat_mul(&m, &anchor_out);
  m = mat_mul(&m, &project);
  mat_flatten(&m, m9);
}

/* ======================================================================================== */
/* OpenCL built-in semantics used by the kernels                                            */
/* ======================================================================================== */
static inline float dot4(const float a[4], const float b[4]) {
  return fmaf(a[3], b[3], fmaf(a[2], b[2], fmaf(a[1], b[1], a[0] * b[0])));
}
static inline float dot3(const float a[3], const float b[3]) {
  return fmaf(a[2], b[2], fmaf(a[1], b[1], a[0] * b[0]));
}
static inline uint32_t sat_u16_trunc(float x) { /* convert_ushort_sat / _rtz */
  if (!(x > 0.0f)) return 0;
  if (x >= 65535.0f) return 65535;
  return (uint32_t)x;
}
static inline uint32_t sat_u16_rte(float x) { return sat_u16_trunc(rintf(x)); }

/* Bulk forms of the primitives above, so that tests can pin them to AMD's device-library definitions
 * (oracle/_ref's ref_builtin_* hooks; tests/test_oracle_golden.py).  `which` as ref_builtin_convert_range:
 * 0 _sat_rte, 1 _sat_rtz, 2 _sat, 4 (ushort)_sat_rtz(round(x)); 3 (uchar) lives in phaneron_oracle_formats.c. */
void orc_prim_dot4(const float *a, const float *b, float *out, size_t n) {
  size_t i;
  for (i = 0; i < n; ++i) out[i] = dot4(a + 4 * i, b + 4 * i);
}
void orc_prim_dot3(const float *a, const float *b, float *out, size_t n) {
  size_t i;
  for (i = 0; i < n; ++i) out[i] = dot3(a + 3 * i, b + 3 * i);
}
extern uint32_t orc_prim_sat_u8_rte(float x);
void orc_prim_convert_range(int which, uint32_t first_bits, uint32_t n, uint16_t *out) {
  int64_t i;
  ORC_PAR_FOR
  for (i = 0; i < (int64_t)n; ++i) {
    const uint32_t bits = first_bits + (uint32_t)i;
    float x;
    memcpy(&x, &bits, 4);
    switch (which) {
      case 0: out[i] = (uint16_t)sat_u16_rte(x); break;
      case 1:
      case 2: out[i] = (uint16_t)sat_u16_trunc(x); break;
      case 3: out[i] = (uint16_t)orc_prim_sat_u8_rte(x); break;
      default: out[i] = (uint16_t)sat_u16_trunc(roundf(x)); break;
    }
  }
}

/* ======================================================================================== */
/* v210.ts                                                                                  */
/* ======================================================================================== */
uint32_t orc_v210_pitch_pixels(uint32_t width) { return width + 47 - ((width - 1) % 48); }
uint32_t orc_v210_pitch_bytes(uint32_t width) { return orc_v210_pitch_pixels(width) * 8 / 3; }

static inline void put_le32(uint8_t *p, uint32_t v) {
  p[0] = (uint8_t)v, p[1] = (uint8_t)(v >> 8), p[2] = (uint8_t)(v >> 16), p[3] = (uint8_t)(v >> 24);
}

void orc_v210_fill_ramp(uint8_t *buf, uint32_t width, uint32_t height) { /* v210.ts:206-236 */
  const uint32_t pitch = orc_v210_pitch_bytes(width);
  const uint32_t Cb = 512, Cr = 512;
  uint32_t Y = 64, y, x;
  memset(buf, 0, (size_t)pitch * height);
  for (y = 0; y < height; ++y) {
    uint8_t *line = buf + (size_t)y * pitch;
    uint32_t off = 0;
    const uint32_t remain = width % 6;
    for (x = 0; x < (width - remain) / 6; ++x) {
      put_le32(line + off, (Cr << 20) | (Y << 10) | Cb);
      put_le32(line + off + 4, (Y << 20) | (Cb << 10) | Y);
      put_le32(line + off + 8, (Cb << 20) | (Y << 10) | Cr);
      put_le32(line + off + 12, (Y << 20) | (Cr << 10) | Y);
      off += 16;
      Y = (940 == Y) ? 64 : Y + 1;
    }
    if (remain) {
      put_le32(line + off, (Cr << 20) | (Y << 10) | Cb);
      if (2 == remain) {
        put_le32(line + off + 4, Y);
      } else if (4 == remain) {
        put_le32(line + off + 4, (Y << 20) | (Cb << 10) | Y);
        put_le32(line + off + 8, (Y << 10) | Cr);
      }
    }
  }
}

/* one pixel of the read kernel: v210.ts:65-78 (last = 1.0f) / :96-109 (tail, last = 0.0f) */
static inline void read_px(float y, float cb, float cr, float last, const float *cm,
                           const float *lut, const float *gm, float *out) {
  const float yuva[4] = {y, cb, cr, last};
  float rgb[3];
  rgb[0] = lut[sat_u16_rte(dot4(yuva, cm + 0) * 65535.0f)];
  rgb[1] = lut[sat_u16_rte(dot4(yuva, cm + 4) * 65535.0f)];
  rgb[2] = lut[sat_u16_rte(dot4(yuva, cm + 8) * 65535.0f)];
  out[0] = dot3(rgb, gm + 0);
  out[1] = dot3(rgb, gm + 3);
  out[2] = dot3(rgb, gm + 6);
  out[3] = 1.0f;
}

void orc_v210_read(const uint32_t *in, float *out, uint32_t width, uint32_t height,
                   const float *cm, const float *lut, const float *gm) {
  const uint32_t wipg = orc_v210_pitch_pixels(width) / 48; /* work items per line (:293) */
  long line;
  ORC_PAR_FOR
  for (line = 0; line < (long)height; ++line) {
    uint32_t lid;
    for (lid = 0; lid < wipg; ++lid) {
      const int last_item = (lid == wipg - 1);
      const uint32_t num_px = (last_item && (width % 48)) ? width % 48 : 48; /* :35 */
      const uint32_t loops = num_px / 6, remain = num_px % 6;
      const uint32_t *w = in + 4 * (size_t)(8 * ((size_t)line * wipg + lid)); /* inOff :39 */
      float *o = out + 4 * ((size_t)width * line + (size_t)lid * 48);        /* outOff :40 */
      uint32_t i;
      for (i = 0; i < loops; ++i, w += 4, o += 24) { /* :54-82 */
        const float y0 = (float)((w[0] >> 10) & 0x3ff), u0 = (float)(w[0] & 0x3ff), v0 = (float)((w[0] >> 20) & 0x3ff);
        const float y1 = (float)(w[1] & 0x3ff);
        const float y2 = (float)((w[1] >> 20) & 0x3ff), u2 = (float)((w[1] >> 10) & 0x3ff), v2 = (float)(w[2] & 0x3ff);
        const float y3 = (float)((w[2] >> 10) & 0x3ff);
        const float y4 = (float)(w[3] & 0x3ff), u4 = (float)((w[2] >> 20) & 0x3ff), v4 = (float)((w[3] >> 10) & 0x3ff);
        const float y5 = (float)((w[3] >> 20) & 0x3ff);
        read_px(y0, u0, v0, 1.0f, cm, lut, gm, o + 0);
        read_px(y1, u0, v0, 1.0f, cm, lut, gm, o + 4);
        read_px(y2, u2, v2, 1.0f, cm, lut, gm, o + 8);
        read_px(y3, u2, v2, 1.0f, cm, lut, gm, o + 12);
        read_px(y4, u4, v4, 1.0f, cm, lut, gm, o + 16);
        read_px(y5, u4, v4, 1.0f, cm, lut, gm, o + 20);
      }
      if (remain > 0) { /* :84-110 - 4th component is 0: the matrix offset column drops out */
        float ys[4] = {0, 0, 0, 0}, us[4] = {0, 0, 0, 0}, vs[4] = {0, 0, 0, 0};
        uint32_t p;
        ys[0] = (float)((w[0] >> 10) & 0x3ff), us[0] = (float)(w[0] & 0x3ff), vs[0] = (float)((w[0] >> 20) & 0x3ff);
        ys[1] = (float)(w[1] & 0x3ff), us[1] = us[0], vs[1] = vs[0];
        if (remain >= 3) { /* reference fills these only for remain==4 (even widths) */
          ys[2] = (float)((w[1] >> 20) & 0x3ff), us[2] = (float)((w[1] >> 10) & 0x3ff), vs[2] = (float)(w[2] & 0x3ff);
          ys[3] = (float)((w[2] >> 10) & 0x3ff), us[3] = us[2], vs[3] = vs[2];
        }
        for (p = 0; p < remain && p < 4; ++p) read_px(ys[p], us[p], vs[p], 0.0f, cm, lut, gm, o + 4 * p);
      }
    }
  }
}

void orc_v210_write(const float *in, uint32_t *out, uint32_t width, uint32_t height,
                    uint32_t interlace, const float *cm, const float *lut) {
  const uint32_t wipg = orc_v210_pitch_pixels(width) / 48;
  const uint32_t pitch_q = orc_v210_pitch_bytes(width) / 16; /* uint4s per line */
  const uint32_t groups = interlace ? height / 2 : height;   /* :323 */
  long grp;
  ORC_PAR_FOR
  for (grp = 0; grp < (long)groups; ++grp) {
    const uint32_t line = (uint32_t)grp * (interlace ? 2 : 1) + ((3 == interlace) ? 1 : 0); /* :126-127 */
    uint32_t lid;
    for (lid = 0; lid < wipg; ++lid) {
      const int last_item = (lid == wipg - 1);
      const uint32_t num_px = (last_item && (width % 48)) ? width % 48 : 48;
      const uint32_t loops = num_px / 6, remain = num_px % 6;
      const float *px = in + 4 * ((size_t)width * line + (size_t)lid * 48); /* inOff :128 */
      uint32_t *w = out + 4 * ((size_t)line * pitch_q + (size_t)lid * 8);   /* by pitch (see .h) */
      uint32_t i, p;
      if (48 != num_px) memset(w, 0, 8 * 16); /* :131-136 */
      for (i = 0; i < loops; ++i, px += 24, w += 4) { /* :142-167 */
        uint32_t y[6], u[6], v[6];
        for (p = 0; p < 6; ++p) {
          float g[4];
          g[0] = lut[sat_u16_rte(px[4 * p + 0] * 65535.0f)];
          g[1] = lut[sat_u16_rte(px[4 * p + 1] * 65535.0f)];
          g[2] = lut[sat_u16_rte(px[4 * p + 2] * 65535.0f)];
          g[3] = 1.0f;
          y[p] = sat_u16_rte(dot4(g, cm + 0));
          u[p] = sat_u16_rte(dot4(g, cm + 4));
          v[p] = sat_u16_rte(dot4(g, cm + 8));
        }
        w[0] = v[0] << 20 | y[0] << 10 | u[0];
        w[1] = y[2] << 20 | u[2] << 10 | y[1];
        w[2] = u[4] << 20 | y[3] << 10 | v[2];
        w[3] = y[5] << 20 | v[4] << 10 | y[4];
      }
      if (remain > 0) { /* :169-194 - truncating index, round-half-away output */
        uint32_t y[4] = {0, 0, 0, 0}, u[4] = {0, 0, 0, 0}, v[4] = {0, 0, 0, 0};
        for (p = 0; p < remain && p < 4; ++p) {
          float g[4];
          g[0] = lut[sat_u16_trunc(px[4 * p + 0] * 65535.0f)];
          g[1] = lut[sat_u16_trunc(px[4 * p + 1] * 65535.0f)];
          g[2] = lut[sat_u16_trunc(px[4 * p + 2] * 65535.0f)];
          g[3] = 1.0f;
          y[p] = sat_u16_trunc(roundf(dot4(g, cm + 0)));
          u[p] = sat_u16_trunc(roundf(dot4(g, cm + 4)));
          v[p] = sat_u16_trunc(roundf(dot4(g, cm + 8)));
        }
        w[0] = v[0] << 20 | y[0] << 10 | u[0];
        w[1] = w[2] = w[3] = 0;
        if (2 == remain) {
          w[1] = y[1];
        } else if (4 == remain) {
          w[1] = y[2] << 20 | u[2] << 10 | y[1];
          w[2] = y[3] << 10 | v[2];
        }
      }
    }
  }
}

/* ======================================================================================== */
/* yadifCl.ts                                                                               */
/* ======================================================================================== */
static inline const float *px_edge(const float *img, int w, int h, int x, int y) {
  x = x < 0 ? 0 : (x >= w ? w - 1 : x); /* CLK_ADDRESS_CLAMP_TO_EDGE */
  y = y < 0 ? 0 : (y >= h ? h - 1 : y);
  return img + 4 * ((size_t)y * w + x);
}

static inline float yadif_spatial(float a, float b, float c, float d, float e, float f, float g,
                                  float h, float i, float j, float k, float l, float m,
                                  float n) { /* :34-62, one component */
  float pred = (d + k) / 2.0f;
  float best = fabsf(c - j) + fabsf(d - k) + fabsf(e - l);
  float score = fabsf(b - k) + fabsf(c - l) + fabsf(d - m);
  int cmp = score < best;
  pred = cmp ? (c + l) / 2.0f : pred;
  best = cmp ? score : best;
  score = cmp ? fabsf(a - l) + fabsf(b - m) + fabsf(c - n) : score;
  cmp = cmp && (score < best);
  pred = cmp ? (b + m) / 2.0f : pred;
  best = cmp ? score : best;

  score = fabsf(d - i) + fabsf(e - j) + fabsf(f - k);
  cmp = score < best;
  pred = cmp ? (e + j) / 2.0f : pred;
  best = cmp ? score : best;
  score = cmp ? fabsf(e - h) + fabsf(f - i) + fabsf(g - j) : score;
  cmp = cmp && (score < best);
  pred = cmp ? (f + i) / 2.0f : pred;
  return pred;
}

static inline float yadif_temporal(float A, float B, float C, float D, float E, float F, float G,
                                   float H, float I, float J, float K, float L, float pred,
                                   int skip) { /* :72-103, one component */
  const float p0 = (C + H) / 2.0f, p1 = F, p2 = (D + I) / 2.0f, p3 = G, p4 = (E + J) / 2.0f;
  const float t0 = fabsf(D - I);
  const float t1 = (fabsf(A - F) + fabsf(B - G)) / 2.0f;
  const float t2 = (fabsf(K - F) + fabsf(G - L)) / 2.0f;
  float diff = fmaxf(fmaxf(t0, t1), t2);
  if (!skip) {
    const float p2mp3 = p2 - p3, p2mp1 = p2 - p1, p0mp1 = p0 - p1, p4mp3 = p4 - p3;
    const float maxi = fmaxf(fmaxf(p2mp3, p2mp1), fminf(p0mp1, p4mp3));
    const float mini = fminf(fminf(p2mp3, p2mp1), fmaxf(p0mp1, p4mp3));
    diff = fmaxf(fmaxf(diff, mini), -maxi);
  }
  pred = (pred > (p2 + diff)) ? p2 + diff : pred;
  pred = (pred < (p2 - diff)) ? p2 - diff : pred;
  return pred;
}

void orc_yadif(const float *prev, const float *cur, const float *next, int w, int h, int parity,
               int tff, int skip_spatial, float *out) {
  const int second = !(parity ^ tff); /* :143 */
  long yo;
  ORC_PAR_FOR
  for (yo = 0; yo < h; ++yo) {
    int xo, c;
    float *o = out + 4 * (size_t)yo * w;
    if ((int)(yo % 2) == parity) { /* :117-121 keep the primary field */
      memcpy(o, cur + 4 * (size_t)yo * w, (size_t)w * 16);
      continue;
    }
    for (xo = 0; xo < w; ++xo, o += 4) {
      const int y = (int)yo;
      const float *ra[7], *rb[7];
      const float *A = px_edge(prev, w, h, xo, y - 1), *B = px_edge(prev, w, h, xo, y + 1);
      const float *F = px_edge(cur, w, h, xo, y - 1), *G = px_edge(cur, w, h, xo, y + 1);
      const float *K = px_edge(next, w, h, xo, y - 1), *L = px_edge(next, w, h, xo, y + 1);
      const float *s0 = second ? cur : prev, *s1 = second ? next : cur;
      const float *C = px_edge(s0, w, h, xo, y - 2), *D = px_edge(s0, w, h, xo, y), *E = px_edge(s0, w, h, xo, y + 2);
      const float *H = px_edge(s1, w, h, xo, y - 2), *I = px_edge(s1, w, h, xo, y), *J = px_edge(s1, w, h, xo, y + 2);
      int t;
      for (t = 0; t < 7; ++t) {
        ra[t] = px_edge(cur, w, h, xo - 3 + t, y - 1); /* a..g :124-130 */
        rb[t] = px_edge(cur, w, h, xo - 3 + t, y + 1); /* h..n :132-138 */
      }
      for (c = 0; c < 4; ++c) {
        float sp = yadif_spatial(ra[0][c], ra[1][c], ra[2][c], ra[3][c], ra[4][c], ra[5][c], ra[6][c],
                                 rb[0][c], rb[1][c], rb[2][c], rb[3][c], rb[4][c], rb[5][c], rb[6][c]);
        o[c] = yadif_temporal(A[c], B[c], C[c], D[c], E[c], F[c], G[c], H[c], I[c], J[c], K[c], L[c],
                              sp, skip_spatial);
      }
      o[3] = px_edge(cur, w, h, xo, y)[3]; /* :164 reset alpha */
    }
  }
}

/* ======================================================================================== */
/* bilinear sampler: OpenCL 1.2 s8.2, NORMALIZED | CLAMP (border 0) | LINEAR                */
/* ======================================================================================== */
static inline void texel_border(const float *img, int w, int h, int x, int y, float t[4]) {
  if (x < 0 || y < 0 || x >= w || y >= h) {
    t[0] = t[1] = t[2] = t[3] = 0.0f;
  } else {
    memcpy(t, img + 4 * ((size_t)y * w + x), 16);
  }
}

static inline void orc_sample_linear(const float *img, int w, int h, float s, float t, float o[4]) {
  const float u = s * (float)w, v = t * (float)h;
  const float fu = u - 0.5f, fv = v - 0.5f;
  const float flu = floorf(fu), flv = floorf(fv);
  const int i0 = (int)flu, j0 = (int)flv;
  const float a = fu - flu, b = fv - flv;
  const float oma = 1.0f - a, omb = 1.0f - b;
  const float w00 = oma * omb, w10 = a * omb, w01 = oma * b, w11 = a * b;
  float t00[4], t10[4], t01[4], t11[4];
  int k;
  texel_border(img, w, h, i0, j0, t00);
  texel_border(img, w, h, i0 + 1, j0, t10);
  texel_border(img, w, h, i0, j0 + 1, t01);
  texel_border(img, w, h, i0 + 1, j0 + 1, t11);
  for (k = 0; k < 4; ++k) o[k] = ((w00 * t00[k] + w10 * t10[k]) + w01 * t01[k]) + w11 * t11[k];
}

void orc_transform(const float *in, int iw, int ih, const float *m, float *out, int ow, int oh) {
  const float mat0[3] = {m[0], m[1], m[2]}, mat1[3] = {m[3], m[4], m[5]}; /* :44-47 */
  long y;
  ORC_PAR_FOR
  for (y = 0; y < oh; ++y) {
    int x;
    for (x = 0; x < ow; ++x) {
      const float in_pos[3] = {(float)x / (float)ow - 0.5f, (float)(int)y / (float)oh - 0.5f, 1.0f}; /* :53 */
      const float s = dot3(mat0, in_pos) + 0.5f, t = dot3(mat1, in_pos) + 0.5f;                       /* :54 */
      orc_sample_linear(in, iw, ih, s, t, out + 4 * ((size_t)y * ow + x));
    }
  }
}

void orc_resize(const float *in, int iw, int ih, float scale, float offset_x, float offset_y,
                const float *flip, float *out, int ow, int oh) {
  const float centre_x = (-0.5f - offset_x) / scale + 0.5f; /* :49-50 */
  const float centre_y = (-0.5f - offset_y) / scale + 0.5f;
  const float off_x = fmaf(centre_x, flip[1], flip[0]), off_y = fmaf(centre_y, flip[3], flip[2]);
  const float mul_x = flip[1] / scale, mul_y = flip[3] / scale;
  long y;
  ORC_PAR_FOR
  for (y = 0; y < oh; ++y) {
    int x;
    for (x = 0; x < ow; ++x) {
      const float s = fmaf((float)x / (float)ow, mul_x, off_x);
      const float t = fmaf((float)(int)y / (float)oh, mul_y, off_y);
      orc_sample_linear(in, iw, ih, s, t, out + 4 * ((size_t)y * ow + x));
    }
  }
}

/* ======================================================================================== */
/* combine / transition / mix / wipe                                                        */
/* ======================================================================================== */
int orc_combine(int n, const float *const *layers, int w, int h, float *out) {
  const size_t npx = (size_t)w * h;
  long p;
  if (n < 2) return -1; /* Combine.getKernelParams throws (combine.ts:92-93) */
  ORC_PAR_FOR
  for (p = 0; p < (long)npx; ++p) {
    float acc[4];
    int i;
    memcpy(acc, layers[0] + 4 * p, 16);
    for (i = 1; i < n; ++i) {
      const float *l = layers[i] + 4 * p;
      const float k = 1.0f - l[3];
      acc[0] = fmaf(acc[0], k, l[0]);
      acc[1] = fmaf(acc[1], k, l[1]);
      acc[2] = fmaf(acc[2], k, l[2]);
      acc[3] = fmaf(acc[3], 0.0f, l[3]);
    }
    memcpy(out + 4 * p, acc, 16);
  }
  return 0;
}

void orc_transition_dissolve(const float *in0, const float *in1, float mix, int w, int h, float *out) {
  const size_t n = (size_t)w * h * 4;
  const float rmix = 1.0f - mix;
  long i;
  ORC_PAR_FOR
  for (i = 0; i < (long)n; ++i) out[i] = fmaf(in0[i], mix, in1[i] * rmix);
}

void orc_mixer(const float *in0, const float *in1, float mix, int w, int h, float *out) {
  orc_transition_dissolve(in0, in1, mix, w, h, out); /* identical arithmetic: mix.ts:41-42 */
}

void orc_transition_wipe(const float *in0, const float *in1, const float *mask, int w, int h,
                         float *out) {
  const size_t npx = (size_t)w * h;
  long p;
  ORC_PAR_FOR
  for (p = 0; p < (long)npx; ++p) {
    const float m = mask[4 * p], rm = 1.0f - m;
    int c;
    for (c = 0; c < 4; ++c) out[4 * p + c] = fmaf(in1[4 * p + c], m, in0[4 * p + c] * rm);
  }
}

void orc_wipe(const float *in0, const float *in1, float wipe, int w, int h, float *out) {
  const float edge = (float)w * wipe; /* wipe.ts:44: x > w * wipe (int -> float promotion) */
  long y;
  ORC_PAR_FOR
  for (y = 0; y < h; ++y) {
    int x;
    for (x = 0; x < w; ++x) {
      const size_t o = 4 * ((size_t)y * w + x);
      memcpy(out + o, ((float)x > edge ? in1 : in0) + o, 16);
    }
  }
}

/* ======================================================================================== */
/* chain: read xN -> combine_N -> write (the work the fused device kernel must reproduce)   */
/* ======================================================================================== */
int orc_pipeline_v210_combine(int n, const uint32_t *const *layers, uint32_t *out, uint32_t width,
                              uint32_t height, const float *rd_cm, const float *rd_lut,
                              const float *rd_gm, const float *wr_cm, const float *wr_lut,
                              float *scratch) {
  const size_t img = (size_t)width * height * 4;
  const float *rgba[8];
  int i;
  if (n < 1 || n > 8) return -1;
  for (i = 0; i < n; ++i) {
    orc_v210_read(layers[i], scratch + img * i, width, height, rd_cm, rd_lut, rd_gm);
    rgba[i] = scratch + img * i;
  }
  if (n == 1) { /* combiner passes a single layer straight through (combiner.ts:222-228) */
    orc_v210_write(rgba[0], out, width, height, 0, wr_cm, wr_lut);
    return 0;
  }
  orc_combine(n, rgba, (int)width, (int)height, scratch + img * n);
  orc_v210_write(scratch + img * n, out, width, height, 0, wr_cm, wr_lut);
  return 0;
}
