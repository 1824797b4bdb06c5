// ph_kernels.h - host-visible launchers of the gfx950 kernels (ph_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "ph_lut.h"

namespace ph {

constexpr int kMaxLayers = 8;
// every kernel with a dynamic LDS array is allowed the CU's whole 160 KiB once and for all: the attribute is per-function state, and
// setting it to each launch's own size would let two launching threads undercut each other
constexpr int kMaxDynamicLds = 160 * 1024;

// 1 = f32 image outputs of the launches issued by this thread are streamed past the caches (ph_device.h store_image);
// set from the context's "stream_images" option before every launch (ph_api.cpp set_device)
extern thread_local uint32_t t_stream_images;  // 0 cached, 1 streamed, 2 by size
extern thread_local uint32_t t_stream_threshold_mb;
// the store policy of ONE image of `bytes` bytes written by the launch being issued
static inline uint32_t image_nt(size_t bytes) {
  return t_stream_images == 2 ? (bytes > ((size_t)t_stream_threshold_mb << 20) ? 1u : 0u) : t_stream_images;
}

struct FusedArgs {
  const void *layers[kMaxLayers];
  void *out;
  uint32_t quads_per_line_used;   // width / 6
  uint32_t quads_per_line_pitch;  // pitch bytes / 16
  uint32_t total_quads;           // quads_per_line_used * height
  const float *rd_cm, *rd_lut, *rd_gm, *wr_cm, *wr_lut;
  // widths that are not a multiple of 48 (the LDS kernel's TAIL instantiation only): total_quads counts the quad SLOTS of the
  // pitch, a line is quads_per_line_used whole quads, then - if tail_px (2 or 4) - the tail quad, then cleared slots
  uint32_t tail_px, magic_qpp;  // width % 6; ceil(2^32 / quads_per_line_pitch)
};

constexpr int kMaxBatch = 8;
struct FusedLdsArgs {
  FusedArgs f;     // job 0 (and the geometry / colour parameters of every job)
  LutView rd, wr;  // compressed reader / writer tables (ph_lut.h)
  // batched launch: `jobs` frames of identical geometry and colour parameters, each with its own
  // layers and output; workgroups [j * wg_per_job, (j + 1) * wg_per_job) work on job j
  uint32_t jobs, wg_per_job;
  const void *more_layers[kMaxBatch - 1][kMaxLayers];
  void *more_out[kMaxBatch - 1];
};

struct ComposeArgs {
  const void *layers[kMaxLayers];
  const float *matrix[kMaxLayers];  // device 3x3 (transform) or NULL = the layer is used 1:1
  // optional wipe transition on a layer (transition.ts wipe): placed layer -> mix with `wipe_with` by `wipe_mask`.x,
  // both output-size RGBA images; NULL = none (ph_compose_wipe_write_v210)
  const void *wipe_with[kMaxLayers], *wipe_mask[kMaxLayers];
  int lw[kMaxLayers], lh[kMaxLayers];
  int n;
  void *out;
  uint32_t out_w, out_h, lines, first_line, line_step;
  const float *wr_cm;
  LutView wr;
};

// ph_kernels_chan.hip: the compositor that samples v210 sources directly
// planar YCbCr sources: 4:2:2 with 16-bit samples (yuv422p10le), 4:2:2 and 4:2:0 with 8-bit samples (yuv422p8, yuv420p), 4:2:0 with
// interleaved chroma (nv12); ptr = the Y plane, ChanArgs::plane_u / plane_v the chroma planes (nv12: plane_u holds CbCr pairs)
// and the packed 8-bit RGB formats of stills and graphics (rgba8, bgra8: four bytes per pixel, alpha included)
enum : uint32_t { kChanNone = 0, kChanV210 = 1, kChanRgba = 2, kChanP10 = 3, kChanP8x422 = 4, kChanP8x420 = 5, kChanNv12 = 6, kChanRgba8 = 7, kChanBgra8 = 8 };
enum : uint32_t { kChanCut = 0, kChanDissolve = 1, kChanWipe = 2 };
struct ChanSrc {
  const void *ptr;
  uint32_t w, h, pitch;  // pixels, pixels, bytes per line
  uint32_t kind;         // kChanV210 / kChanRgba / a planar kind (kChanNone: absent)
  uint32_t sampled;      // 1 = through m (transform.ts:53-57), 0 = pixel for pixel
  uint32_t tail_from;    // v210: the first column of the line's tail, 6 * (w / 6) (== w when the width is a multiple of 6: no tail)
  float m[6];            // rows 0 and 1 of the 3x3 transform matrix
};
// The channel's frame as a flat program the kernel walks per pixel: one op per source to sample, with what to do with
// the sample.  A layer without a transition is one op (Layer); a dissolve is Hold (the layer's own source) + Dissolve (the
// incoming source); a wipe is Hold + Incoming + Wipe (the mask).  kChanActFirst marks the op that completes layer 0.
enum : uint32_t { kChanActLayer = 0, kChanActHold = 1, kChanActDissolve = 2, kChanActIncoming = 3, kChanActWipe = 4, kChanActFirst = 0x100,
                  // launcher: a v210 source shown at its own scale, unrotated - neighbouring lanes' taps are neighbouring columns, so a
                  // lane converts ONE column and takes the other from the lane beside it; bits 12..14: which halo table is the op's
                  kChanActShare = 0x200, kChanActShareShift = 12 };
constexpr int kMaxChanOps = 3 * kMaxLayers;
struct ChanOp {
  ChanSrc src;
  uint32_t action;
  float mix;
};
struct ChanArgs {
  ChanOp op[kMaxChanOps];
  int n_ops;
  uint32_t magic_cpr, magic_cpg;  // ceil(2^32 / chunks per row), ceil(2^32 / chunks per row group) (launcher)
  void *out;
  void *index;  // scratch: 8 bytes per output pixel (chan_index_bytes)
  uint32_t out_w, out_h, lines, first_line, line_step;
  const float *rd_cm, *rd_gm, *wr_cm;
  LutView rd, wr;
  // planar sources: the chroma planes of op k and, where the source has a Loader matrix of its own (8-bit code ranges), that
  // matrix (12 floats, device; NULL = rd_cm) - read by the planar instantiation of the kernel only
  const void *plane_u[kMaxChanOps], *plane_v[kMaxChanOps];
  const float *cm_op[kMaxChanOps];
  uint32_t planar;  // which instantiation the launch needs: 0 v210 / images on whole 48-pixel blocks, 1 v210 lines with tails, 2 planar / RGB sources
  // the packed frame the writer makes: 0 v210 (out), 1 yuv422p10 / 2 yuv422p8 (out = the Y plane, out_u, out_v; out_pitch = luma
  // samples per line), 5 rgba8 / 6 bgra8 (out; out_pitch = pixels per line) - PH_FMT_* numbering
  uint32_t out_fmt, out_pitch;
  void *out_u, *out_v;
  // v210 frames of any even width (1280: src/config.ts:43-54): quad slots per line by pitch, and the first column of the line's
  // tail (v210.ts:166-193; 0xFFFFFFFF when the width is a multiple of 6 or the frame is not v210)
  uint32_t out_qpitch, out_tail_from;
  // tap sharing (launcher): per sharing op and wave step of a workgroup, the converted LEFT column of the step's first lane
  // (3 rows x rgb = 36 bytes), made by a pass in front of phase 1 and kept in the LDS behind the table
  uint32_t halo_off, halo_steps;  // byte offset in the LDS; steps per op the area holds (0: no sharing in this launch)
  uint32_t any_cm;                // launcher: some op brings a Loader matrix of its own (cm_op): the whole launch takes the general dot products
  uint32_t images_only;           // launcher: every source is an f32 image (de-interlaced fields): nothing is looked up in the reader's table, it is not loaded
};

// Several channels' frames of ONE geometry and colour recipe in one launch (ph_chan_compose_batch) - what the reference runs: four
// channels of <= 1080p in one context through one queue (src/index.ts:45-71,156-160, clJobQueue.ts:114-141).  A workgroup takes its
// share of EVERY job, so the tables are loaded once and the wave steps of all jobs together are dealt to the waves in front of one
// barrier (ph_kernels_chan.hip).  Any source kind (round 6: planar / packed-RGB ones in an instantiation of their own), v210 out.
constexpr int kMaxChanJobs = 8;
constexpr int kMaxChanBatchOps = 40;  // the ops of all jobs of a launch (the argument block has to stay below 4 KiB)
struct ChanJob {
  void *out, *index;  // the v210 frame; this job's index frame (chan_index_bytes each)
  uint32_t first_op, n_ops, first_line, pad;
};
struct ChanBatchArgs {
  ChanOp op[kMaxChanBatchOps];
  ChanJob job[kMaxChanJobs];
  uint8_t op_job[kMaxChanBatchOps];  // the job an op belongs to (for the launcher)
  uint32_t share_op[8];              // launcher: the ops that share taps (ChanHalo), by halo table: op | its job << 8 | the job's first line << 16
  uint32_t jobs, n_ops, n_share;
  uint32_t magic_cpr, magic_cpg;     // as ChanArgs
  uint32_t steps;                    // wave steps per job of the workgroups that have the most chunks (launcher)
  uint32_t magic_spj, magic_qpj;     // ceil(2^32 / steps), ceil(2^32 / quads per job and workgroup = 64 * steps / 3)
  uint32_t out_w, out_h, lines, line_step;
  const float *rd_cm, *rd_gm, *wr_cm;
  LutView rd, wr;
  uint32_t tails;  // 1: lines may end in a tail (the TAILS instantiation)
  uint32_t out_qpitch, out_tail_from;
  uint32_t job_rot[kMaxChanJobs][4];  // per job and XCD (two 16-bit values per word): how far the job's share is rotated round the XCD's workgroups
  uint32_t sched_off;              // LDS byte offset of the jobs' tables behind the gamma table
  uint32_t halo_off, halo_steps;   // as ChanArgs; halo_steps = steps
  uint32_t images_only;            // as ChanArgs: every source of every job is an f32 image - the reader's table is not loaded
  // round 6: jobs with planar / packed-RGB sources (file playback with insets, graphics over clips) share launches too - the PLANAR
  // instantiation.  An op's chroma planes, and its Loader matrix as an index into a small table (0: the call's rd_cm; the argument
  // block has 4 KiB: forty pointers more would not fit)
  const void *plane_u[kMaxChanBatchOps], *plane_v[kMaxChanBatchOps];
  const float *cm_tab[8];
  uint8_t cm_idx[kMaxChanBatchOps];
  uint32_t planar;                 // 1: some op has a planar / packed-RGB source (launcher: the PLANAR instantiation)
  uint32_t any_cm;                 // some op brings a Loader matrix of its own: the general dot products throughout (as ChanArgs)
};
static_assert(sizeof(ChanBatchArgs) <= 4096, "kernel arguments are limited to 4 KiB");
// refuses (hipErrorInvalidValue) more than kMaxChanJobs jobs / kMaxChanBatchOps ops: callers split
// Route trace (ph_trace_begin / ph_trace_end, ph_api.cpp): while the calling thread traces, every launch is noted by name; returns
// true when the launch must NOT be made (a dry run).  `name` may be a launcher call's text ("ph::launch_xxx(...": noted as "xxx").
bool trace_launch(const char *name);
hipError_t launch_chan_compose_batch(hipStream_t s, const ChanBatchArgs &a, uint32_t num_cus);

// ph_kernels_up.hip: the 2 x 2-block compositor for magnifying placements
struct UpLayer {
  const void *ptr;       // f32 RGBA (16 bytes per texel) or packed f32 RGB (12)
  uint32_t w, h, pitch;  // texels, texels, bytes per row
  uint32_t pad;
  float m[6];            // rows 0 and 1 of the 3x3 transform matrix
};
constexpr int kMaxUpJobs = 4;
struct UpArgs {
  UpLayer layer[kMaxLayers];
  int n;
  void *out;
  uint32_t out_w, out_h, lines, first_line, line_step;
  const float *wr_cm;
  LutView wr;
  uint32_t magic_upr, magic_upg;  // launcher
  uint32_t shared;                // launcher: all layers have one size and one placement
  // launcher: lines by pitch - quad slots per line; the columns the wave steps cover (out_w, or the whole pitch when lines end in a
  // tail quad and cleared slots: the TAILS instantiation writes those too)
  uint32_t out_qpitch, cover_w;
  // more jobs of the same shape in the same launch (both fields of a frame; several channels under one placement): their layers' data and
  // their outputs; jobs = 1 .. kMaxUpJobs
  const void *more_ptr[kMaxUpJobs - 1][kMaxLayers];
  void *more_out[kMaxUpJobs - 1];
  uint32_t jobs;
};
// Clips straight from their wire formats: reader and 2 x 2-block compositor in ONE launch (clip_up_write_v210_kernel).  A workgroup owns a
// tile of the output; with the reader's table resident it converts the source pixels its tile's taps fall on - once each - into a
// scratch rectangle of its own (packed f32 RGB, or RGBA when a source carries alpha; it stays in the XCD's L2), swaps the writer's
// table in and composes from there.
struct ClipSrc {
  const void *p0, *p1, *p2;  // the planes (v210 / packed RGB: p0)
  const float *cm;           // the source's YCbCr -> RGB matrix (device, 12 floats; unused for packed RGB)
  uint32_t fmt;              // PH_FMT_*
  uint32_t pitch;            // line pitch in luma samples / pixels; v210: in 16-byte quads
};
struct ClipUpArgs {
  UpArgs up;                 // layer[l].w / h / m: the clip's size and placement (ptr / pitch are the kernel's: the scratch rectangle)
  ClipSrc src[kMaxLayers];
  const float *rd_gm;
  LutView rd;
  char *scratch;             // grid * wg_bytes
  uint32_t wg_bytes;
  uint32_t rect_off[kMaxLayers], rect_cap[kMaxLayers];  // a layer's rectangle inside the workgroup's scratch: offset, bytes reserved
  uint32_t tcu, trp, gx;     // a tile: wave-step columns x row pairs; tiles per row of tiles
  // several frames of ONE shape in one launch (up.jobs = 2 .. kMaxUpJobs; single-layer frames only: several channels' file playback):
  // job j's clip is src[j] (layer[0] has the shape of all), its frame up.out / up.more_out[j - 1]; gy: rows of tiles per job
  uint32_t gy;
  uint32_t info_off;         // LDS offset of the per-layer rectangle descriptors (behind the larger table)
};
// picks the tiling and fills rect_off / rect_cap / wg_bytes / tcu / trp / gx / info_off; returns the number of workgroups (0: not for this kernel)
uint32_t clip_up_plan(ClipUpArgs &a, bool rgb12, uint32_t num_cus);
hipError_t launch_clip_up_write_v210(hipStream_t s, const ClipUpArgs &a, bool rgb12, uint32_t grid);
bool compose_up_eligible(const UpArgs &a);
hipError_t launch_compose_up_write_v210(hipStream_t s, const UpArgs &a, bool rgb12, uint32_t num_cus);

struct DeintArgs {  // ph_kernels_deint.hip
  const uint4 *prev[kMaxLayers], *cur[kMaxLayers], *next[kMaxLayers];  // v210 frames, width x height
  float4 *out0[kMaxLayers], *out1[kMaxLayers];                         // RGBA f32: yadif parity 0 / parity 1
  int n, skip;
  uint32_t width, height, quads_pitch;
  uint32_t rows_per_strip, strips, col_blocks, nt;  // filled in by the launcher
  uint32_t rgb12;  // 1: the outputs are packed f32 RGB (12 bytes per pixel, the reader's alpha == 1 left out)
  const float *cm, *gm;
  LutView lut;
  // windows of planar 4:2:2 frames (pack 1: yuv422p10le, 2: yuv422p8; 0: v210): prev / cur / next are the Y planes, these the
  // chroma planes; read by the planar instantiations only
  uint32_t pack;
  const void *prev_u[kMaxLayers], *prev_v[kMaxLayers], *cur_u[kMaxLayers], *cur_v[kMaxLayers], *next_u[kMaxLayers], *next_v[kMaxLayers];
};

struct CombineArgs {
  const void *layers[kMaxLayers];
  void *out;
  size_t npx;
  uint32_t nt;  // filled in by the launcher
};

uint32_t v210_pitch_bytes(uint32_t width);

hipError_t launch_v210_read(hipStream_t s, const void *in, void *out, uint32_t width, uint32_t height,
                            const void *cm, const void *lut, const void *gm);
hipError_t launch_v210_write(hipStream_t s, const void *in, void *out, uint32_t width, uint32_t height,
                             uint32_t interlace, const void *cm, const void *lut);
hipError_t launch_fused_v210_combine(hipStream_t s, int n, const FusedArgs &a);
// LDS-LUT variants (ph_kernels_lds.hip): one 1024-lane workgroup per CU
hipError_t launch_fused_v210_combine_lds(hipStream_t s, int n, const FusedLdsArgs &a, uint32_t num_cus);
hipError_t launch_lds_base_probe(hipStream_t s, uint32_t *out_dev);  // writes the LDS address of g_lds (expected: 0)
hipError_t launch_v210_read_lds(hipStream_t s, const void *in, void *out, uint32_t width, uint32_t height,
                                const void *cm, const void *gm, const LutView &lut, uint32_t num_cus);
hipError_t launch_v210_read_lds_batch(hipStream_t s, int n, const void *const *ins, void *const *outs, uint32_t width,
                                      uint32_t height, const void *cm, const void *gm, const LutView &lut, uint32_t num_cus);
hipError_t launch_v210_write_lds(hipStream_t s, const void *in, void *out, uint32_t width, uint32_t height,
                                 uint32_t interlace, const void *cm, const LutView &lut, uint32_t num_cus);
// the other pack formats (ph_kernels_fmt.hip); lv == NULL selects the global-gather form
uint32_t pack_pitch(int fmt, uint32_t width);
int pack_plane_bytes(int fmt, uint32_t width, uint32_t height, size_t bytes[3]);
hipError_t launch_pack_read(hipStream_t s, int fmt, const void *const planes[3], void *out, uint32_t width,
                            uint32_t height, const void *cm, const void *table, const void *gm, const LutView *lv,
                            uint32_t num_cus);
hipError_t launch_pack_read_batch(hipStream_t s, int fmt, int n, const void *const (*planes)[3], void *const *outs, uint32_t width, uint32_t height,
                                  const void *cm, const void *gm, const LutView &lv, uint32_t num_cus);
hipError_t launch_pack_write(hipStream_t s, int fmt, const void *in, void *const planes[3], uint32_t width,
                             uint32_t height, uint32_t interlace, const void *cm, const void *table, const LutView *lv,
                             uint32_t num_cus);
hipError_t launch_compose_write_v210(hipStream_t s, const ComposeArgs &a, uint32_t num_cus);
size_t chan_index_bytes(uint32_t out_w, uint32_t lines);
hipError_t launch_chan_compose_v210(hipStream_t s, const ChanArgs &a, uint32_t num_cus);
bool compose_can_wipe(const ComposeArgs &a);  // the buffer-addressed compositor serves this job (needed for wipe layers)
hipError_t launch_v210_yadif_pair(hipStream_t s, DeintArgs a, int tff, uint32_t num_cus);
hipError_t launch_yadif(hipStream_t s, const void *prev, const void *cur, const void *next, int w, int h, int parity,
                        int tff, int skip, void *out);
hipError_t launch_yadif_pair(hipStream_t s, const void *prev, const void *cur, const void *next, int w, int h, int tff,
                             int skip, void *out0, void *out1);
hipError_t launch_transform(hipStream_t s, const void *in, int iw, int ih, const void *m9, void *out, int ow, int oh);
hipError_t launch_resize(hipStream_t s, const void *in, int iw, int ih, float scale, float ox, float oy,
                         const void *flip4, void *out, int ow, int oh);
hipError_t launch_combine(hipStream_t s, int n, const CombineArgs &a);
hipError_t launch_dissolve(hipStream_t s, const void *in0, const void *in1, float mix, int w, int h, void *out);
hipError_t launch_twipe(hipStream_t s, const void *in0, const void *in1, const void *mask, int w, int h, void *out);
hipError_t launch_wipe(hipStream_t s, const void *in0, const void *in1, float wipe, int w, int h, void *out);
hipError_t launch_rgb_unpack(hipStream_t s, const void *packed, void *rgba, size_t npx);

}  // namespace ph
