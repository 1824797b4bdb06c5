// ocl_shim.cpp - host-side definitions of the OpenCL C built-ins the reference's kernel
// text calls, so that text (compiled unmodified for x86 by the same clang) can execute here.
//
// TEST INFRASTRUCTURE ONLY.  This file contains no reference code: it is the runtime the
// reference's kernels expect from an OpenCL implementation.  Definitions follow
//   * ROCm 7.2 device-lib (`/opt/rocm/amdgcn/bitcode/opencl.bc`, inspect with llvm-dis):
//       dot(float4) = fma(a.w,b.w, fma(a.z,b.z, fma(a.y,b.y, a.x*b.x)))   (@_Z3dotDv4_fS_)
//       dot(float3) = fma(a.z,b.z, fma(a.y,b.y, a.x*b.x))                 (@_Z3dotDv3_fS_)
//       convert_ushort_sat_rte(x) = (ushort)min(max(rint(x),0),65535)
//       convert_ushort_sat[_rtz](x) = truncating conversion with the same clamp
//   * OpenCL 1.2 spec section 8.2 for read_imagef (nearest / linear, clamp / clamp-to-edge).
//     LINEAR filtering evaluates the spec formula in f32, left to right, no fma:
//       T = (1-a)(1-b)*T00 + a(1-b)*T10 + (1-a)b*T01 + ab*T11
//     (CDNA GPUs have no sampler hardware, so no device result exists to compare with:
//      parity for LINEAR is pinned to this formula only.)
// Build: oracle/refbuild/build_ref.sh.  Must be compiled by the same clang as the kernels
// (vector arguments use that compiler's SysV vector ABI), with -ffp-contract=off.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

typedef float float2 __attribute__((ext_vector_type(2)));
typedef float float3 __attribute__((ext_vector_type(3)));
typedef float float4 __attribute__((ext_vector_type(4)));
typedef int int2 __attribute__((ext_vector_type(2)));
typedef unsigned short ushort4 __attribute__((ext_vector_type(4)));
typedef unsigned char uchar4 __attribute__((ext_vector_type(4)));
typedef unsigned int uint4 __attribute__((ext_vector_type(4)));

struct Img {
  float4 *p;
  int w, h;
};

// ---- work-item state, set by the drivers at the bottom --------------------------------
static thread_local unsigned g_gid[2], g_lid, g_grp, g_lsz;

unsigned long wi_global_id(unsigned d) asm("_Z13get_global_idj");
unsigned long wi_local_id(unsigned d) asm("_Z12get_local_idj");
unsigned long wi_group_id(unsigned d) asm("_Z12get_group_idj");
unsigned long wi_local_size(unsigned d) asm("_Z14get_local_sizej");
unsigned long wi_global_id(unsigned d) { return d < 2 ? g_gid[d] : 0; }
unsigned long wi_local_id(unsigned d) { return d == 0 ? g_lid : 0; }
unsigned long wi_group_id(unsigned d) { return d == 0 ? g_grp : 0; }
unsigned long wi_local_size(unsigned d) { return d == 0 ? g_lsz : 1; }

// ---- arithmetic built-ins ---------------------------------------------------------------
float b_dot3(float3 a, float3 b) asm("_Z3dotDv3_fS_");
float b_dot4(float4 a, float4 b) asm("_Z3dotDv4_fS_");
float b_fma1(float a, float b, float c) asm("_Z3fmafff");
float2 b_fma2(float2 a, float2 b, float2 c) asm("_Z3fmaDv2_fS_S_");
float4 b_fma4(float4 a, float4 b, float4 c) asm("_Z3fmaDv4_fS_S_");
float4 b_fabs4(float4 a) asm("_Z4fabsDv4_f");
float4 b_fmin4(float4 a, float4 b) asm("_Z4fminDv4_fS_");
float4 b_fmax4(float4 a, float4 b) asm("_Z4fmaxDv4_fS_");
float b_round(float a) asm("_Z5roundf");
float4 b_cvt_f4_us4(ushort4 a) asm("_Z14convert_float4Dv4_t");
float4 b_cvt_f4_uc4(uchar4 a) asm("_Z14convert_float4Dv4_h");
unsigned char b_cvt_uc_sat_rte(float a) asm("_Z21convert_uchar_sat_rtef");
unsigned short b_cvt_us_sat(float a) asm("_Z18convert_ushort_satf");
unsigned short b_cvt_us_sat_rte(float a) asm("_Z22convert_ushort_sat_rtef");
unsigned short b_cvt_us_sat_rtz(float a) asm("_Z22convert_ushort_sat_rtzf");

float b_dot3(float3 a, float3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
float b_dot4(float4 a, float4 b) {
  return fmaf(a.w, b.w, fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)));
}
float b_fma1(float a, float b, float c) { return fmaf(a, b, c); }
float2 b_fma2(float2 a, float2 b, float2 c) {
  float2 r;
  r.x = fmaf(a.x, b.x, c.x);
  r.y = fmaf(a.y, b.y, c.y);
  return r;
}
float4 b_fma4(float4 a, float4 b, float4 c) {
  float4 r;
  for (int i = 0; i < 4; ++i) r[i] = fmaf(a[i], b[i], c[i]);
  return r;
}
float4 b_fabs4(float4 a) {
  float4 r;
  for (int i = 0; i < 4; ++i) r[i] = fabsf(a[i]);
  return r;
}
float4 b_fmin4(float4 a, float4 b) {
  float4 r;
  for (int i = 0; i < 4; ++i) r[i] = fminf(a[i], b[i]);
  return r;
}
float4 b_fmax4(float4 a, float4 b) {
  float4 r;
  for (int i = 0; i < 4; ++i) r[i] = fmaxf(a[i], b[i]);
  return r;
}
float b_round(float a) { return roundf(a); }
float4 b_cvt_f4_us4(ushort4 a) {
  float4 r;
  for (int i = 0; i < 4; ++i) r[i] = (float)a[i];
  return r;
}
float4 b_cvt_f4_uc4(uchar4 a) {
  float4 r;
  for (int i = 0; i < 4; ++i) r[i] = (float)a[i];
  return r;
}
unsigned char b_cvt_uc_sat_rte(float a) {  // rint -> max 0 -> min 255 -> fptoui (device-lib form)
  float x = rintf(a);
  if (!(x > 0.0f)) return 0;
  if (x >= 255.0f) return 255;
  return (unsigned char)x;
}
static inline unsigned short clamp_us(float x) {
  // NaN -> 0 like the device's max(x,0) (fmax ignores NaN)
  if (!(x > 0.0f)) return 0;
  if (x >= 65535.0f) return 65535;
  return (unsigned short)x;  // truncation; callers pre-round when rte is wanted
}
unsigned short b_cvt_us_sat(float a) { return clamp_us(a); }
unsigned short b_cvt_us_sat_rtz(float a) { return clamp_us(a); }
unsigned short b_cvt_us_sat_rte(float a) { return clamp_us(rintf(a)); }

// ---- images and samplers ----------------------------------------------------------------
enum { S_NORM = 1, S_EDGE = 2, S_CLAMP = 4, S_NEAREST = 0x10, S_LINEAR = 0x20 };
extern "C" void *__translate_sampler_initializer(int bits) { return (void *)(intptr_t)bits; }

float4 img_read_i(Img *im, void *smp, int2 c) asm("_Z11read_imagef14ocl_image2d_ro11ocl_samplerDv2_i");
float4 img_read_f(Img *im, void *smp, float2 c) asm("_Z11read_imagef14ocl_image2d_ro11ocl_samplerDv2_f");
void img_write(Img *im, int2 c, float4 v) asm("_Z12write_imagef14ocl_image2d_woDv2_iDv4_f");
int img_width(Img *im) asm("_Z15get_image_width14ocl_image2d_wo");
int img_height(Img *im) asm("_Z16get_image_height14ocl_image2d_wo");

static inline float4 texel(const Img *im, int x, int y, bool edge) {
  if (edge) {
    x = x < 0 ? 0 : (x >= im->w ? im->w - 1 : x);
    y = y < 0 ? 0 : (y >= im->h ? im->h - 1 : y);
  } else if (x < 0 || y < 0 || x >= im->w || y >= im->h) {
    return (float4){0.f, 0.f, 0.f, 0.f};
  }
  return im->p[(size_t)y * im->w + x];
}
float4 img_read_i(Img *im, void *smp, int2 c) {
  int bits = (int)(intptr_t)smp;
  return texel(im, c.x, c.y, (bits & S_EDGE) != 0);
}
float4 img_read_f(Img *im, void *smp, float2 c) {
  int bits = (int)(intptr_t)smp;
  bool edge = (bits & S_EDGE) != 0;
  float u = c.x, v = c.y;
  if (bits & S_NORM) {
    u = u * (float)im->w;
    v = v * (float)im->h;
  }
  if (!(bits & S_LINEAR)) return texel(im, (int)floorf(u), (int)floorf(v), edge);
  float fu = u - 0.5f, fv = v - 0.5f;
  float flu = floorf(fu), flv = floorf(fv);
  int i0 = (int)flu, j0 = (int)flv;
  float a = fu - flu, b = fv - flv;
  float4 t00 = texel(im, i0, j0, edge), t10 = texel(im, i0 + 1, j0, edge);
  float4 t01 = texel(im, i0, j0 + 1, edge), t11 = texel(im, i0 + 1, j0 + 1, edge);
  float oma = 1.0f - a, omb = 1.0f - b;
  float w00 = oma * omb, w10 = a * omb, w01 = oma * b, w11 = a * b;
  float4 r;
  for (int k = 0; k < 4; ++k) r[k] = ((w00 * t00[k] + w10 * t10[k]) + w01 * t01[k]) + w11 * t11[k];
  return r;
}
void img_write(Img *im, int2 c, float4 v) {
  if (c.x < 0 || c.y < 0 || c.x >= im->w || c.y >= im->h) return;
  im->p[(size_t)c.y * im->w + c.x] = v;
}
int img_width(Img *im) { return im->w; }
int img_height(Img *im) { return im->h; }

// ---- the reference kernels (symbols renamed by build_ref.sh where they clash with libc) --
extern "C" {
void refk_v210_read(uint4 *in, float4 *out, unsigned width, float4 *colMatrix, float *gammaLut,
                    float4 *gamutMatrix);
void refk_v210_write(float4 *in, uint4 *out, unsigned width, unsigned interlace, float4 *colMatrix,
                     float *gammaLut);
// the other pack formats (planes are passed as untyped pointers: vector loads are plain loads on x86)
void refk_yuv422p10_read(void *y, void *u, void *v, float4 *out, unsigned w, float4 *cm, float *lut, float4 *gm);
void refk_yuv422p10_write(float4 *in, void *y, void *u, void *v, unsigned w, unsigned il, float4 *cm, float *lut);
void refk_yuv422p8_read(void *y, void *u, void *v, float4 *out, unsigned w, float4 *cm, float *lut, float4 *gm);
void refk_yuv422p8_write(float4 *in, void *y, void *u, void *v, unsigned w, unsigned il, float4 *cm, float *lut);
void refk_yuv420p_read(void *y, void *u, void *v, float4 *out, unsigned w, float4 *cm, float *lut, float4 *gm);
void refk_yuv420p_write(float4 *in, void *y, void *u, void *v, unsigned w, unsigned il, float4 *cm, float *lut);
void refk_nv12_read(void *y, void *c, float4 *out, unsigned w, float4 *cm, float *lut, float4 *gm);
void refk_nv12_write(float4 *in, void *y, void *c, unsigned w, unsigned il, float4 *cm, float *lut);
void refk_rgba8_read(void *in, float4 *out, unsigned w, float *lut, float4 *gm);
void refk_rgba8_write(float4 *in, void *out, unsigned w, unsigned il, float *lut);
void refk_bgra8_read(void *in, float4 *out, unsigned w, float *lut, float4 *gm);
void refk_bgra8_write(float4 *in, void *out, unsigned w, unsigned il, float *lut);
void yadif(Img *prev, Img *cur, Img *next, int parity, int tff, int skipSpatial, Img *out);
void transform(Img *in, float4 *m, Img *out);
void resize(Img *in, float scale, float offX, float offY, float *flip, Img *out);
void mixer(Img *a, Img *b, float mix, Img *out);
void wipe(Img *a, Img *b, float wipe, Img *out);
void combine_2(Img *, Img *, Img *);
void combine_3(Img *, Img *, Img *, Img *);
void combine_4(Img *, Img *, Img *, Img *, Img *);
void combine_5(Img *, Img *, Img *, Img *, Img *, Img *);
void combine_6(Img *, Img *, Img *, Img *, Img *, Img *, Img *);
void combine_7(Img *, Img *, Img *, Img *, Img *, Img *, Img *, Img *);
void combine_8(Img *, Img *, Img *, Img *, Img *, Img *, Img *, Img *, Img *);
void transition_dissolve(Img *a, Img *b, float mix, Img *out);
void transition_wipe(Img *a, Img *b, Img *mask, Img *out);
}

static inline unsigned v210_pitch_px(unsigned w) { return w + 47 - ((w - 1) % 48); }

// Work-groups / image rows are independent: the CPU-baseline leg of bench.py spreads them over
// host threads the way an OpenCL CPU device would (work-item ids are thread_local).  1 = serial.
static int g_threads = 1;
template <typename F> static void parallel_rows(unsigned n, F f) {
  unsigned t = g_threads < 1 ? 1 : (unsigned)g_threads;
  if (t > n) t = n ? n : 1;
  if (t <= 1) {
    for (unsigned i = 0; i < n; ++i) f(i);
    return;
  }
  std::vector<std::thread> pool;
  for (unsigned k = 0; k < t; ++k)
    pool.emplace_back([=] {
      for (unsigned i = (unsigned)((uint64_t)n * k / t), e = (unsigned)((uint64_t)n * (k + 1) / t); i < e; ++i) f(i);
    });
  for (auto &th : pool) th.join();
}

template <typename F> static void for_each_pixel(int w, int h, F f) {
  parallel_rows((unsigned)h, [=](unsigned y) {
    for (int x = 0; x < w; ++x) {
      g_gid[0] = x;
      g_gid[1] = y;
      f();
    }
  });
}

// ---- C entry points used by tests/golden/gen_golden.py ----------------------------------
extern "C" {

// geometry as Reader (reference v210.ts:284-295): one group per line, pitch/48 items
void ref_v210_read(const uint32_t *in, float *out, unsigned width, unsigned height,
                   const float *colMatrix12, const float *lut, const float *gamut9) {
  float cm[12], gm[12] = {0};
  memcpy(cm, colMatrix12, sizeof cm);
  memcpy(gm, gamut9, 9 * sizeof(float));  // kernel reads 3 float4s of a 36-byte buffer
  unsigned wipg = v210_pitch_px(width) / 48;
  float *cmp = cm, *gmp = gm;
  parallel_rows(height, [=](unsigned line) {
    for (unsigned lid = 0; lid < wipg; ++lid) {
      g_grp = line;
      g_lid = lid;
      g_lsz = wipg;
      g_gid[0] = line * wipg + lid;
      refk_v210_read((uint4 *)in, (float4 *)out, width, (float4 *)cmp, (float *)lut, (float4 *)gmp);
    }
  });
}

// geometry as Writer (v210.ts:312-324): groups = height (or height/2 when interlaced)
void ref_v210_write(const float *in, uint32_t *out, unsigned width, unsigned height,
                    unsigned interlace, const float *colMatrix12, const float *lut) {
  float cm[12];
  memcpy(cm, colMatrix12, sizeof cm);
  unsigned wipg = v210_pitch_px(width) / 48;
  unsigned groups = interlace ? height / 2 : height;
  float *cmp = cm;
  parallel_rows(groups, [=](unsigned grp) {
    for (unsigned lid = 0; lid < wipg; ++lid) {
      g_grp = grp;
      g_lid = lid;
      g_lsz = wipg;
      g_gid[0] = grp * wipg + lid;
      refk_v210_write((float4 *)in, (uint4 *)out, width, interlace, (float4 *)cmp, (float *)lut);
    }
  });
}

// fmt: 1 yuv422p10, 2 yuv422p8, 3 yuv420p, 4 nv12, 5 rgba8, 6 bgra8.  Geometry as the Readers /
// Writers: work items per line = ceil(pitch / 64) (yuv422p10.ts:302-303 etc.), one group per line
// (4:2:2, RGBA) or per line PAIR (4:2:0: yuv420p.ts:345, nv12.ts:332).
static unsigned pitch8(unsigned w) { return w + 7 - ((w - 1) % 8); }
static void set_item(unsigned grp, unsigned lid, unsigned wipg) {
  g_grp = grp, g_lid = lid, g_lsz = wipg, g_gid[0] = grp * wipg + lid;
}

int ref_pack_read(int fmt, void *p0, void *p1, void *p2, float *out, unsigned width, unsigned height,
                  const float *colMatrix12, const float *lut, const float *gamut9) {
  float cm[12] = {0}, gm[12] = {0};
  if (colMatrix12) memcpy(cm, colMatrix12, sizeof cm);
  memcpy(gm, gamut9, 9 * sizeof(float));
  const bool rgb = fmt >= 5;
  const unsigned wipg = rgb ? (width + 63) / 64 : (pitch8(width) + 63) / 64;
  const unsigned groups = (fmt == 3 || fmt == 4) ? height / 2 : height;
  for (unsigned grp = 0; grp < groups; ++grp)
    for (unsigned lid = 0; lid < wipg; ++lid) {
      set_item(grp, lid, wipg);
      switch (fmt) {
        case 1: refk_yuv422p10_read(p0, p1, p2, (float4 *)out, width, (float4 *)cm, (float *)lut, (float4 *)gm); break;
        case 2: refk_yuv422p8_read(p0, p1, p2, (float4 *)out, width, (float4 *)cm, (float *)lut, (float4 *)gm); break;
        case 3: refk_yuv420p_read(p0, p1, p2, (float4 *)out, width, (float4 *)cm, (float *)lut, (float4 *)gm); break;
        case 4: refk_nv12_read(p0, p1, (float4 *)out, width, (float4 *)cm, (float *)lut, (float4 *)gm); break;
        case 5: refk_rgba8_read(p0, (float4 *)out, width, (float *)lut, (float4 *)gm); break;
        case 6: refk_bgra8_read(p0, (float4 *)out, width, (float *)lut, (float4 *)gm); break;
        default: return -1;
      }
    }
  return 0;
}

int ref_pack_write(int fmt, const float *in, void *p0, void *p1, void *p2, unsigned width, unsigned height,
                   unsigned interlace, const float *colMatrix12, const float *lut) {
  float cm[12] = {0};
  if (colMatrix12) memcpy(cm, colMatrix12, sizeof cm);
  const bool rgb = fmt >= 5, v420 = (fmt == 3 || fmt == 4);
  const unsigned wipg = rgb ? (width + 63) / 64 : (pitch8(width) + 63) / 64;
  const unsigned groups = v420 ? height / 2 : (interlace ? height / 2 : height);
  for (unsigned grp = 0; grp < groups; ++grp)
    for (unsigned lid = 0; lid < wipg; ++lid) {
      set_item(grp, lid, wipg);
      switch (fmt) {
        case 1: refk_yuv422p10_write((float4 *)in, p0, p1, p2, width, interlace, (float4 *)cm, (float *)lut); break;
        case 2: refk_yuv422p8_write((float4 *)in, p0, p1, p2, width, interlace, (float4 *)cm, (float *)lut); break;
        case 3: refk_yuv420p_write((float4 *)in, p0, p1, p2, width, interlace, (float4 *)cm, (float *)lut); break;
        case 4: refk_nv12_write((float4 *)in, p0, p1, width, interlace, (float4 *)cm, (float *)lut); break;
        case 5: refk_rgba8_write((float4 *)in, p0, width, interlace, (float *)lut); break;
        case 6: refk_bgra8_write((float4 *)in, p0, width, interlace, (float *)lut); break;
        default: return -1;
      }
    }
  return 0;
}

void ref_yadif(const float *prev, const float *cur, const float *next, int w, int h, int parity,
               int tff, int skipSpatial, float *out) {
  Img p{(float4 *)prev, w, h}, c{(float4 *)cur, w, h}, n{(float4 *)next, w, h}, o{(float4 *)out, w, h};
  for_each_pixel(w, h, [&] { yadif(&p, &c, &n, parity, tff, skipSpatial, &o); });
}

void ref_transform(const float *in, int iw, int ih, const float *mat9, float *out, int ow, int oh) {
  float m[12] = {0};
  memcpy(m, mat9, 9 * sizeof(float));
  Img i{(float4 *)in, iw, ih}, o{(float4 *)out, ow, oh};
  for_each_pixel(ow, oh, [&] { transform(&i, (float4 *)m, &o); });
}

void ref_resize(const float *in, int iw, int ih, float scale, float offX, float offY,
                const float *flip4, float *out, int ow, int oh) {
  float fl[4];
  memcpy(fl, flip4, sizeof fl);
  Img i{(float4 *)in, iw, ih}, o{(float4 *)out, ow, oh};
  for_each_pixel(ow, oh, [&] { resize(&i, scale, offX, offY, fl, &o); });
}

void ref_mixer(const float *a, const float *b, float mix, int w, int h, float *out) {
  Img ia{(float4 *)a, w, h}, ib{(float4 *)b, w, h}, o{(float4 *)out, w, h};
  for_each_pixel(w, h, [&] { mixer(&ia, &ib, mix, &o); });
}

void ref_wipe(const float *a, const float *b, float wp, int w, int h, float *out) {
  Img ia{(float4 *)a, w, h}, ib{(float4 *)b, w, h}, o{(float4 *)out, w, h};
  for_each_pixel(w, h, [&] { wipe(&ia, &ib, wp, &o); });
}

void ref_transition_dissolve(const float *a, const float *b, float mix, int w, int h, float *out) {
  Img ia{(float4 *)a, w, h}, ib{(float4 *)b, w, h}, o{(float4 *)out, w, h};
  for_each_pixel(w, h, [&] { transition_dissolve(&ia, &ib, mix, &o); });
}

void ref_transition_wipe(const float *a, const float *b, const float *mask, int w, int h,
                         float *out) {
  Img ia{(float4 *)a, w, h}, ib{(float4 *)b, w, h}, im{(float4 *)mask, w, h}, o{(float4 *)out, w, h};
  for_each_pixel(w, h, [&] { transition_wipe(&ia, &ib, &im, &o); });
}

int ref_combine(int n, const float *const *layers, int w, int h, float *out) {
  if (n < 2 || n > 8) return -1;
  Img l[8], o{(float4 *)out, w, h};
  for (int i = 0; i < n; ++i) l[i] = Img{(float4 *)layers[i], w, h};
  for_each_pixel(w, h, [&] {
    switch (n) {
      case 2: combine_2(&l[0], &l[1], &o); break;
      case 3: combine_3(&l[0], &l[1], &l[2], &o); break;
      case 4: combine_4(&l[0], &l[1], &l[2], &l[3], &o); break;
      case 5: combine_5(&l[0], &l[1], &l[2], &l[3], &l[4], &o); break;
      case 6: combine_6(&l[0], &l[1], &l[2], &l[3], &l[4], &l[5], &o); break;
      case 7: combine_7(&l[0], &l[1], &l[2], &l[3], &l[4], &l[5], &l[6], &o); break;
      case 8: combine_8(&l[0], &l[1], &l[2], &l[3], &l[4], &l[5], &l[6], &l[7], &o); break;
    }
  });
  return 0;
}

void ref_set_num_threads(int n) { g_threads = n < 1 ? 1 : n; }
int ref_num_threads(void) { return g_threads; }

// The reference's own chain for one output frame, stage by stage as its job queue runs it:
// n x v210 read -> combine_n -> v210 write, float RGBA intermediates in `scratch`
// ((n+1) * width*height*4 floats).  bench.py's cpu_baseline ("reference") times this.
int ref_pipeline_v210_combine(int n, const uint32_t *const *layers, uint32_t *out, unsigned width, unsigned height,
                              const float *rd_cm12, const float *rd_lut, const float *rd_gamut9, const float *wr_cm12,
                              const float *wr_lut, float *scratch) {
  if (n < 1 || n > 8) return -1;
  const size_t img = (size_t)width * height * 4;
  const float *rgba[8];
  for (int i = 0; i < n; ++i) {
    ref_v210_read(layers[i], scratch + img * i, width, height, rd_cm12, rd_lut, rd_gamut9);
    rgba[i] = scratch + img * i;
  }
  const float *top = rgba[0];
  if (n >= 2) {  // combiner.ts:222-228 passes a single layer through
    if (ref_combine(n, rgba, (int)width, (int)height, scratch + img * n)) return -1;
    top = scratch + img * n;
  }
  ref_v210_write(top, out, width, height, 0, wr_cm12, wr_lut);
  return 0;
}

}  // extern "C"
