/* phaneron_oracle_formats.c - CPU restatement of the reference's other packed formats
 * (SURVEY.md 8f-1): yuv422p10le, yuv422p8, yuv420p, nv12, rgba8, bgra8.
 *
 * TEST INFRASTRUCTURE ONLY (see phaneron_oracle.h).  Pinned bit-for-bit to the reference's own
 * kernel text run on x86 (tests/golden/kernels.npz, cases "fmt_*").
 *
 * All six share one shape (64 pixels per work item, one line - or one line PAIR for 4:2:0 -
 * per work group); restated here per line and per group of 8 pixels.
 */
#include <math.h>
#include <string.h>

#include "phaneron_oracle.h"

static inline float dot4(const float a[4], const float b[4]) {
  return fmaf(a[3], b[3], fmaf(a[2], b[2], fmaf(a[1], b[1], a[0] * b[0])));
}
static inline float dot3(const float a[3], const float b[3]) {
  return fmaf(a[2], b[2], fmaf(a[1], b[1], a[0] * b[0]));
}
static inline uint32_t sat_u16_rte(float x) { /* convert_ushort_sat_rte */
  x = rintf(x);
  if (!(x > 0.0f)) return 0;
  if (x >= 65535.0f) return 65535;
  return (uint32_t)x;
}
static inline uint32_t sat_u8_rte(float x) { /* convert_uchar_sat_rte */
  x = rintf(x);
  if (!(x > 0.0f)) return 0;
  if (x >= 255.0f) return 255;
  return (uint32_t)x;
}

uint32_t orc_prim_sat_u8_rte(float x) { return sat_u8_rte(x); } /* test hook, see orc_prim_convert_range */

uint32_t orc_pack_pitch(int fmt, uint32_t width) { /* samples per luma line: getPitch() of each format */
  if (fmt == ORC_FMT_RGBA8 || fmt == ORC_FMT_BGRA8) return width;            /* rgba8.ts:103-105 */
  if (fmt == ORC_FMT_V210) return orc_v210_pitch_pixels(width);
  return width + 7 - ((width - 1) % 8);                                        /* yuv422p10.ts:221 */
}

int orc_pack_plane_bytes(int fmt, uint32_t width, uint32_t height, size_t bytes[3]) {
  const size_t p = orc_pack_pitch(fmt, width);
  bytes[0] = bytes[1] = bytes[2] = 0;
  switch (fmt) {
    case ORC_FMT_V210: bytes[0] = (size_t)orc_v210_pitch_bytes(width) * height; return 1;
    case ORC_FMT_YUV422P10: bytes[0] = p * 2 * height, bytes[1] = bytes[2] = bytes[0] / 2; return 3; /* :300-301 */
    case ORC_FMT_YUV422P8: bytes[0] = p * height, bytes[1] = bytes[2] = bytes[0] / 2; return 3;
    case ORC_FMT_YUV420P: bytes[0] = p * height, bytes[1] = bytes[2] = bytes[0] / 4; return 3;      /* yuv420p.ts:341 */
    case ORC_FMT_NV12: bytes[0] = p * height, bytes[1] = bytes[0] / 2; return 2;                     /* nv12.ts:329 */
    case ORC_FMT_RGBA8:
    case ORC_FMT_BGRA8: bytes[0] = p * 4 * height; return 1;
  }
  return -1;
}

/* one YCbCr pixel -> linear RGBA (identical in all four YUV readers, e.g. yuv422p10.ts:74-86) */
static inline void yuv_px(float y, float u, float v, const float *cm, const float *lut, const float *gm, float *o) {
  const float yuva[4] = {y, u, v, 1.0f};
  float rgb[3];
  rgb[0] = lut[sat_u16_rte(dot4(yuva, cm + 0) * 65535.0f)];
  rgb[1] = lut[sat_u16_rte(dot4(yuva, cm + 4) * 65535.0f)];
  rgb[2] = lut[sat_u16_rte(dot4(yuva, cm + 8) * 65535.0f)];
  o[0] = dot3(rgb, gm + 0), o[1] = dot3(rgb, gm + 3), o[2] = dot3(rgb, gm + 6), o[3] = 1.0f;
}

int orc_pack_read(int fmt, const void *p0, const void *p1, const void *p2, float *out, uint32_t width,
                  uint32_t height, const float *cm, const float *lut, const float *gm) {
  const uint32_t pitch = orc_pack_pitch(fmt, width);
  uint32_t line, x;
  if (fmt == ORC_FMT_V210) {
    orc_v210_read((const uint32_t *)p0, out, width, height, cm, lut, gm);
    return 0;
  }
  if (fmt == ORC_FMT_RGBA8 || fmt == ORC_FMT_BGRA8) { /* rgba8.ts:25-67, bgra8.ts:25-67 */
    const uint8_t *in = (const uint8_t *)p0;
    const int r_at = fmt == ORC_FMT_RGBA8 ? 0 : 2, b_at = 2 - r_at;
    for (line = 0; line < height; ++line)
      for (x = 0; x < width; ++x) {
        const uint8_t *px = in + 4 * ((size_t)line * pitch + x);
        float *o = out + 4 * ((size_t)line * width + x);
        float rgb[3];
        rgb[0] = lut[sat_u16_rte((float)px[r_at] * 65535.0f / 255.0f)];
        rgb[1] = lut[sat_u16_rte((float)px[1] * 65535.0f / 255.0f)];
        rgb[2] = lut[sat_u16_rte((float)px[b_at] * 65535.0f / 255.0f)];
        o[0] = dot3(rgb, gm + 0), o[1] = dot3(rgb, gm + 3), o[2] = dot3(rgb, gm + 6);
        o[3] = lut[sat_u16_rte((float)px[3] * 65535.0f / 255.0f)]; /* alpha goes through the LUT too */
      }
    return 0;
  }
  if (fmt < ORC_FMT_YUV422P10 || fmt > ORC_FMT_NV12) return -1;
  {
    const int v420 = (fmt == ORC_FMT_YUV420P || fmt == ORC_FMT_NV12);
    const uint32_t lines = v420 ? (height / 2) * 2 : height; /* 4:2:0 works on line pairs (yuv420p.ts:345) */
    for (line = 0; line < lines; ++line) {
      const uint32_t cl = v420 ? line / 2 : line;
      for (x = 0; x < width; ++x) {
        float y, u, v;
        if (fmt == ORC_FMT_YUV422P10) {
          y = ((const uint16_t *)p0)[(size_t)line * pitch + x];
          u = ((const uint16_t *)p1)[(size_t)cl * (pitch / 2) + x / 2];
          v = ((const uint16_t *)p2)[(size_t)cl * (pitch / 2) + x / 2];
        } else if (fmt == ORC_FMT_NV12) { /* nv12.ts:61-74: Cb, Cr interleaved */
          y = ((const uint8_t *)p0)[(size_t)line * pitch + x];
          u = ((const uint8_t *)p1)[(size_t)cl * pitch + (x / 2) * 2];
          v = ((const uint8_t *)p1)[(size_t)cl * pitch + (x / 2) * 2 + 1];
        } else {
          y = ((const uint8_t *)p0)[(size_t)line * pitch + x];
          u = ((const uint8_t *)p1)[(size_t)cl * (pitch / 2) + x / 2];
          v = ((const uint8_t *)p2)[(size_t)cl * (pitch / 2) + x / 2];
        }
        yuv_px(y, u, v, cm, lut, gm, out + 4 * ((size_t)line * width + x));
      }
    }
  }
  return 0;
}

/* linear RGB -> code values of one pixel; tail = the reference's `round()` variant (:186-188) */
static inline void px_codes(const float *px, const float *cm, const float *lut, int tail, uint32_t yuv[3]) {
  float g[4];
  int c;
  g[0] = lut[sat_u16_rte(px[0] * 65535.0f)];
  g[1] = lut[sat_u16_rte(px[1] * 65535.0f)];
  g[2] = lut[sat_u16_rte(px[2] * 65535.0f)];
  g[3] = 1.0f;
  for (c = 0; c < 3; ++c) {
    const float t = dot4(g, cm + 4 * c);
    yuv[c] = sat_u16_rte(tail ? roundf(t) : t);
  }
}

int orc_pack_write(int fmt, const float *in, void *p0, void *p1, void *p2, uint32_t width, uint32_t height,
                   uint32_t interlace, const float *cm, const float *lut) {
  const uint32_t pitch = orc_pack_pitch(fmt, width);
  if (fmt == ORC_FMT_V210) {
    orc_v210_write(in, (uint32_t *)p0, width, height, interlace, cm, lut);
    return 0;
  }
  if (fmt == ORC_FMT_RGBA8 || fmt == ORC_FMT_BGRA8) { /* rgba8.ts:69-101 */
    uint8_t *out = (uint8_t *)p0;
    const int r_at = fmt == ORC_FMT_RGBA8 ? 0 : 2, b_at = 2 - r_at;
    const uint32_t groups = interlace ? height / 2 : height;
    uint32_t g, x;
    for (g = 0; g < groups; ++g) {
      const uint32_t line = g * (interlace ? 2 : 1) + ((3 == interlace) ? 1 : 0);
      for (x = 0; x < width; ++x) {
        const float *px = in + 4 * ((size_t)line * width + x);
        uint8_t *o = out + 4 * ((size_t)line * pitch + x);
        o[r_at] = (uint8_t)sat_u8_rte(lut[sat_u16_rte(px[0] * 65535.0f)] * 255.0f);
        o[1] = (uint8_t)sat_u8_rte(lut[sat_u16_rte(px[1] * 65535.0f)] * 255.0f);
        o[b_at] = (uint8_t)sat_u8_rte(lut[sat_u16_rte(px[2] * 65535.0f)] * 255.0f);
        o[3] = 255;
      }
    }
    return 0;
  }
  if (fmt < ORC_FMT_YUV422P10 || fmt > ORC_FMT_NV12) return -1;
  {
    const int v420 = (fmt == ORC_FMT_YUV420P || fmt == ORC_FMT_NV12);
    const int wide = (fmt == ORC_FMT_YUV422P10);
    const uint32_t dflt_y = wide ? 64 : 16, dflt_c = wide ? 512 : 128; /* tail defaults (:191-193) */
    const uint32_t groups = v420 ? height / 2 : (interlace ? height / 2 : height);
    const uint32_t full = width / 8, remain = width % 8;
    uint32_t g, l, o8, p;
    for (g = 0; g < groups; ++g) {
      /* 4:2:2: line = g*(1|2)+off (:140-141).  4:2:0: line = 2g+off, 1 or 2 lines (yuv420p.ts:156-158) */
      const uint32_t first = v420 ? g * 2 + ((3 == interlace) ? 1 : 0) : g * (interlace ? 2 : 1) + ((3 == interlace) ? 1 : 0);
      const uint32_t nlines = v420 ? (interlace ? 1 : 2) : 1;
      const uint32_t crow = v420 ? g : first;
      for (l = 0; l < nlines; ++l) {
        const uint32_t line = first + l;
        for (o8 = 0; o8 < full + (remain ? 1 : 0); ++o8) {
          const int tail = (o8 == full);
          const uint32_t n = tail ? remain : 8;
          uint32_t y[8], u[4], v[4], c[6][3];
          for (p = 0; p < 8; ++p) y[p] = dflt_y;
          for (p = 0; p < 4; ++p) u[p] = v[p] = dflt_c;
          for (p = 0; p < n && p < 6 + 2 * !tail; ++p) {
            uint32_t yuv[3];
            px_codes(in + 4 * ((size_t)line * width + 8 * o8 + p), cm, lut, tail, yuv);
            if (!tail) {
              y[p] = yuv[0];
              if (!(p & 1)) u[p / 2] = yuv[1], v[p / 2] = yuv[2];
            } else {
              c[p][0] = yuv[0], c[p][1] = yuv[1], c[p][2] = yuv[2];
            }
          }
          if (tail) { /* e.g. yuv422p10.ts:198-213 */
            y[0] = c[0][0], y[1] = c[1][0], u[0] = c[0][1], v[0] = c[0][2];
            if (remain > 2) {
              y[2] = c[2][0], y[3] = c[3][0], u[1] = c[2][1], v[1] = c[2][2];
              if (remain > 4) {
                y[4] = c[4][0], y[5] = c[5][0];
                if (v420) u[2] = c[4][1], v[2] = c[4][2];   /* yuv420p.ts:243-244, nv12.ts:233-234 */
                else u[1] = c[4][1], v[1] = c[4][2];        /* the 4:2:2 writers overwrite slot 1 (:209-210) */
              }
            }
          }
          if (wide) {
            uint16_t *Y = (uint16_t *)p0 + (size_t)line * pitch + 8 * o8;
            for (p = 0; p < 8; ++p) Y[p] = (uint16_t)y[p];
          } else { /* uchar = (uchar)ushort: keeps the low 8 bits (yuv422p8.ts:166-168) */
            uint8_t *Y = (uint8_t *)p0 + (size_t)line * pitch + 8 * o8;
            for (p = 0; p < 8; ++p) Y[p] = (uint8_t)y[p];
          }
          if (l == 0) {
            if (fmt == ORC_FMT_NV12) {
              uint8_t *Cp = (uint8_t *)p1 + (size_t)crow * pitch + 8 * o8;
              for (p = 0; p < 4; ++p) Cp[2 * p] = (uint8_t)u[p], Cp[2 * p + 1] = (uint8_t)v[p];
            } else if (wide) {
              uint16_t *U = (uint16_t *)p1 + (size_t)crow * (pitch / 2) + 4 * o8;
              uint16_t *V = (uint16_t *)p2 + (size_t)crow * (pitch / 2) + 4 * o8;
              for (p = 0; p < 4; ++p) U[p] = (uint16_t)u[p], V[p] = (uint16_t)v[p];
            } else {
              uint8_t *U = (uint8_t *)p1 + (size_t)crow * (pitch / 2) + 4 * o8;
              uint8_t *V = (uint8_t *)p2 + (size_t)crow * (pitch / 2) + 4 * o8;
              for (p = 0; p < 4; ++p) U[p] = (uint8_t)u[p], V[p] = (uint8_t)v[p];
            }
          }
        }
      }
    }
  }
  return 0;
}
