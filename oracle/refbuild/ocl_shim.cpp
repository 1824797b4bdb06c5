// ocl_shim.cpp - the part of an OpenCL runtime that AMD's device library does not contain, so
// that the reference's kernel text (compiled unmodified for x86 by the same clang) can execute here:
// work-item ids, image / sampler access, and the loops that play the NDRange.
//
// TEST INFRASTRUCTURE ONLY.  This file contains no reference code.
//   * The ARITHMETIC built-ins the kernels call (dot, fma, fabs, fmin, fmax, round, convert_*_sat*,
//     convert_float4) are NOT written here: they are the function bodies of ROCm 7.2's
//     /opt/rocm/amdgcn/bitcode/opencl.bc + ocml.bc, retargeted to x86-64 by devlib_builtins.py and
//     linked in by build_ref.sh.  tests/test_builtins_gpu.py runs the same bitcode on the MI355X and
//     checks that both agree with the product's and the oracle's primitives.
//   * read_imagef / write_imagef follow OpenCL 1.2 section 8.2 (nearest / linear, clamp / clamp-to-edge).
//     LINEAR filtering evaluates the spec formula in f32, left to right, no fma:
//       T = (1-a)(1-b)*T00 + a(1-b)*T10 + (1-a)b*T01 + ab*T11
//     (CDNA GPUs have no sampler hardware and ROCm's OpenCL reports no image support on them, so no
//      device result exists to compare with: parity for LINEAR is pinned to this formula only.)
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
// NOT defined here.  dot / fma / fabs / fmin / fmax / round / convert_*_sat* / convert_float4 are AMD's
// own definitions: devlib_builtins.py takes them out of /opt/rocm/amdgcn/bitcode/{opencl,ocml}.bc,
// retargets the unchanged function bodies to x86-64 and build_ref.sh links the object in.  The
// declarations below only give this file's test hooks (bottom) access to them.
float b_dot3(float3 a, float3 b) asm("_Z3dotDv3_fS_");
float b_dot4(float4 a, float4 b) asm("_Z3dotDv4_fS_");
float b_fma1(float a, float b, float c) asm("_Z3fmafff");
float b_round(float a) asm("_Z5roundf");
unsigned char b_cvt_uc_sat_rte(float a) asm("_Z21convert_uchar_sat_rtef");
unsigned short b_cvt_us_sat(float a) asm("_Z18convert_ushort_satf");
unsigned short b_cvt_us_sat_rte(float a) asm("_Z22convert_ushort_sat_rtef");
unsigned short b_cvt_us_sat_rtz(float a) asm("_Z22convert_ushort_sat_rtzf");

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


// ---- test hooks over the device-library built-ins (tests/test_oracle_golden.py, test_builtins_gpu.py):
// bulk evaluation so that Python can compare them with the oracle's and the product's primitives.
void ref_builtin_dot4(const float *a, const float *b, float *out, size_t n) {
  for (size_t i = 0; i < n; ++i) out[i] = b_dot4(*(const float4 *)(a + 4 * i), *(const float4 *)(b + 4 * i));
}
void ref_builtin_dot3(const float *a, const float *b, float *out, size_t n) {
  for (size_t i = 0; i < n; ++i) {
    float3 x = {a[3 * i], a[3 * i + 1], a[3 * i + 2]}, y = {b[3 * i], b[3 * i + 1], b[3 * i + 2]};
    out[i] = b_dot3(x, y);
  }
}
void ref_builtin_fma(const float *a, const float *b, const float *c, float *out, size_t n) {
  for (size_t i = 0; i < n; ++i) out[i] = b_fma1(a[i], b[i], c[i]);
}
// which: 0 convert_ushort_sat_rte, 1 convert_ushort_sat_rtz, 2 convert_ushort_sat, 3 convert_uchar_sat_rte,
// 4 (ushort)round(x) with the _rtz clamp (the v210 tail path, v210.ts:176-183).  Inputs are the float
// bit patterns first_bits .. first_bits + n - 1, so that a caller can sweep all 2^32 of them.
static inline uint16_t convert_one(int which, uint32_t bits) {
  float x;
  memcpy(&x, &bits, 4);
  switch (which) {
    case 0: return b_cvt_us_sat_rte(x);
    case 1: return b_cvt_us_sat_rtz(x);
    case 2: return b_cvt_us_sat(x);
    case 3: return b_cvt_uc_sat_rte(x);
    default: return b_cvt_us_sat_rtz(b_round(x));
  }
}
void ref_builtin_convert_range(int which, uint32_t first_bits, uint32_t n, uint16_t *out) {
  parallel_rows(64, [=](unsigned part) {
    for (uint64_t i = (uint64_t)n * part / 64, e = (uint64_t)n * (part + 1) / 64; i < e; ++i)
      out[i] = convert_one(which, first_bits + (uint32_t)i);
  });
}
// the same for a list of bit patterns
void ref_builtin_convert_list(int which, const uint32_t *bits, uint32_t n, uint16_t *out) {
  for (uint32_t i = 0; i < n; ++i) out[i] = convert_one(which, bits[i]);
}

}  // extern "C"
