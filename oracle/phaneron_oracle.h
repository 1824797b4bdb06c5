/* phaneron_oracle.h - CPU restatement of the reference's per-pixel hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is the checker, never the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The product
 * (libphaneron_hip.so) neither links nor calls anything in oracle/.
 *
 * Every function is an own-code scalar restatement of the algorithm in the reference
 * (Streampunk/phaneron v0.0.15) and cites the file:line it follows.  Parity pinning: the
 * restatement is checked bit-for-bit (tests/test_oracle_golden.py) against golden vectors
 * produced by RUNNING the reference itself in the build container -
 *   - its OpenCL C kernel text compiled unmodified for x86 (oracle/refbuild/build_ref.sh), and
 *   - its TypeScript host maths type-stripped and run under node 12 (oracle/refbuild/ts_strip.py)
 * - see tests/golden/gen_golden.py.  read_imagef(LINEAR) is implementation-defined in
 * OpenCL and the reference holds no test for it: for transform/resize parity is pinned to
 * the OpenCL 1.2 section 8.2 formula as evaluated in orc_sample_linear() below ("unpinned"
 * by any reference fixture; stated in DESIGN.md).
 *
 * All images are row-major float RGBA, unpadded (16 bytes per pixel), as the reference's
 * `image2d_t`-capable buffers (io.ts:69-77).  v210 buffers are little-endian 32-bit words,
 * line pitch = orc_v210_pitch_bytes(width).
 */
#ifndef PHANERON_ORACLE_H
#define PHANERON_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- colourMaths.ts -------------------------------------------------------------------- */
/* colspec in {"601-625","601_525","709","2020","sRGB"}; unknown specs fall back to "709"
 * exactly like the reference (colourMaths.ts:131-134).  Return 0 on success. */
int orc_gamma2linear_lut(const char *colspec, float *lut65536);                 /* :130-149 */
int orc_linear2gamma_lut(const char *colspec, float *lut65536);                 /* :151-169 */
int orc_ycbcr2rgb_matrix(const char *colspec, int num_bits, int luma_black, int luma_white,
                         int chr_range, float *m12);                            /* :276-332 */
int orc_rgb2ycbcr_matrix(const char *colspec, int num_bits, int luma_black, int luma_white,
                         int chr_range, float *m12);                            /* :334-390 */
int orc_rgb2rgb_matrix(const char *src_colspec, const char *dst_colspec, float *m9); /* :392-394 */

/* transform.ts:119-171 - the 3x3 matrix Transform.getKernelParams uploads. */
void orc_transform_matrix(int width, int height, int flip_h, int flip_v, double anchor_x,
                          double anchor_y, double scale_x, double scale_y, double offset_x,
                          double offset_y, double rotate, float *m9);

/* ---- v210.ts ----------------------------------------------------------------------------- */
uint32_t orc_v210_pitch_pixels(uint32_t width);                                 /* :198-200 */
uint32_t orc_v210_pitch_bytes(uint32_t width);                                  /* :202-204 */
void orc_v210_fill_ramp(uint8_t *buf, uint32_t width, uint32_t height);         /* :206-236 */
void orc_v210_read(const uint32_t *in, float *out, uint32_t width, uint32_t height,
                   const float *col_matrix12, const float *gamma_lut,
                   const float *gamut_matrix9);                                 /* :25-111  */
/* interlace: 0 progressive, 1 top field (even lines), 3 bottom field (odd lines);
 * lines not selected are left untouched.  Widths that are not a multiple of 48 address
 * the output by line pitch (the reference's `width*line/6` overlaps lines there; see
 * DESIGN.md "deviations"). */
void orc_v210_write(const float *in, uint32_t *out, uint32_t width, uint32_t height,
                    uint32_t interlace, const float *col_matrix12,
                    const float *gamma_lut);                                    /* :113-195 */

/* ---- the other pack formats (phaneron_oracle_formats.c) ------------------------ */
enum {
  ORC_FMT_V210 = 0,
  ORC_FMT_YUV422P10 = 1, /* yuv422p10.ts */
  ORC_FMT_YUV422P8 = 2,  /* yuv422p8.ts  */
  ORC_FMT_YUV420P = 3,   /* yuv420p.ts   */
  ORC_FMT_NV12 = 4,      /* nv12.ts      */
  ORC_FMT_RGBA8 = 5,     /* rgba8.ts     */
  ORC_FMT_BGRA8 = 6      /* bgra8.ts     */
};
uint32_t orc_pack_pitch(int fmt, uint32_t width);
/* returns the number of planes (1..3), fills their sizes */
int orc_pack_plane_bytes(int fmt, uint32_t width, uint32_t height, size_t bytes[3]);
int orc_pack_read(int fmt, const void *p0, const void *p1, const void *p2, float *out, uint32_t width,
                  uint32_t height, const float *col_matrix12, const float *gamma_lut,
                  const float *gamut_matrix9);
int orc_pack_write(int fmt, const float *in, void *p0, void *p1, void *p2, uint32_t width,
                   uint32_t height, uint32_t interlace, const float *col_matrix12,
                   const float *gamma_lut);

/* ---- image ops ----------------------------------------------------------------------------- */
void orc_yadif(const float *prev, const float *cur, const float *next, int width, int height,
               int parity, int tff, int skip_spatial, float *out);        /* yadifCl.ts:28-167 */
void orc_transform(const float *in, int in_w, int in_h, const float *m9, float *out, int out_w,
                   int out_h);                                            /* transform.ts:36-59 */
void orc_resize(const float *in, int in_w, int in_h, float scale, float offset_x, float offset_y,
                const float *flip4, float *out, int out_w, int out_h);    /* resize.ts:35-59   */
int orc_combine(int n, const float *const *layers, int width, int height,
                float *out);                                              /* combine.ts:24-68  */
void orc_transition_dissolve(const float *in0, const float *in1, float mix, int width, int height,
                             float *out);                                 /* transition.ts:60-65 */
void orc_transition_wipe(const float *in0, const float *in1, const float *mask, int width,
                         int height, float *out);                         /* transition.ts:66-74 */
void orc_mixer(const float *in0, const float *in1, float mix, int width, int height,
               float *out);                                               /* mix.ts:30-45      */
void orc_wipe(const float *in0, const float *in1, float wipe, int width, int height,
              float *out);                                                /* wipe.ts:30-47     */

/* ---- chains (what a fused device kernel must equal) ---------------------------------------- */
/* n v210 layers -> read -> combine_n (n>=2; n==1 passthrough as combiner.ts:222-228) -> write.
 * scratch must hold (n+1) * width*height*4 floats.  Used as the CPU baseline ("port"). */
int orc_pipeline_v210_combine(int n, const uint32_t *const *layers, uint32_t *out, uint32_t width,
                              uint32_t height, const float *rd_col_matrix12, const float *rd_lut,
                              const float *rd_gamut9, const float *wr_col_matrix12,
                              const float *wr_lut, float *scratch);

/* ---- the OpenCL built-in semantics this file assumes, in bulk (test hooks) --------------------- */
void orc_prim_dot4(const float *a4, const float *b4, float *out, size_t n);
void orc_prim_dot3(const float *a3, const float *b3, float *out, size_t n);
/* which: 0 convert_ushort_sat_rte, 1 _sat_rtz, 2 _sat, 3 convert_uchar_sat_rte, 4 _sat_rtz(round(x));
 * inputs are the float bit patterns first_bits .. first_bits + n - 1 */
void orc_prim_convert_range(int which, uint32_t first_bits, uint32_t n, uint16_t *out);

/* number of OpenMP threads the image loops will use (1 when built without OpenMP) */
int orc_num_threads(void);
void orc_set_num_threads(int n);

#ifdef __cplusplus
}
#endif
#endif
