/* phaneron_hip.h - C ABI of libphaneron_hip.so: the MI355X (gfx950) replacement for the
 * `nodencl` binding underneath phaneron's src/process operators and src/clJobQueue.ts.
 *
 * The reference reaches its OpenCL kernels only through the nodencl JS API (npm nodencl@1.4.2,
 * not vendored).  Each entry point below replaces one piece of that surface and cites the
 * reference call sites (paths relative to the reference checkout) that define its meaning.
 * INTEGRATION.md shows the N-API stub that binds these for node (node/ph_napi.c is that stub).
 *
 * Conventions
 *   - extern "C", opaque handles, plain pointers and sizes; no C++/torch types.
 *   - every call returns 0 on success, a negative PH_E_* code otherwise; ph_last_error() gives
 *     the message (the reference surfaces failures as rejected promises / thrown Error).
 *   - one caller thread per context (the reference calls from the single node main thread);
 *     work completes asynchronously on the context's three in-order HIP streams.
 *   - device buffers are reference counted like nodencl's OpenCLBuffer (addRef/release).
 *   - there is NO CPU fallback: without a HIP device ph_ctx_create fails.
 */
#ifndef PHANERON_HIP_H
#define PHANERON_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ABI history (ph_abi_version() returns PH_ABI_VERSION; a binding compiled against another value must not load the library)
 * 2: additive over 1 - ph_program_resolve, ph_route_*, ph_yadif_pair, ph_v210_yadif_pair, ph_v210_read_batch,
 *    ph_compose_wipe_write_v210, context option "stream_images"; no signature of 1 changed
 * 3: ph_chan_compose_v210 (ph_chan_source / ph_chan_layer), ph_compose_up_write_v210, ph_v210_yadif_pair_fmt, ph_check_program,
 *    ph_route_comm_count, context option "host_pool_mb"
 * 4: BREAKING for callers of ph_v210_yadif_pair: ph_deint_source grew from 40 to 88 bytes (six chroma-plane pointers for
 *    ph_yadif_pair_packed), so an array of sources built against 2 / 3 has the wrong stride.  ph_chan_compose added
 * 5: ph_compose_up_write_v210_pair, ph_lut_layout_of
 * 6: ph_fused_field_v210 / ph_field_layer REMOVED (the slowest route of its workload by 2.7x, no caller).  The fused entry
 *    points take widths that are not a multiple of 48 (1280 x 720: the reference's third format, src/config.ts:43-54)
 * 7: additive over 6 - ph_chan_compose_batch (ph_chan_job): several channels' frames in one launch; ph_run_programs; ph_compose_up_write_v210_batch; ph_pack_read_batch;
 *    ph_event_record_timed / ph_event_elapsed_us; ph_ctx_host_pool_stats; "host_pool_mb" defaults to 4096 again and never
 *    keeps less than the working set
 * 8: additive over 7 - ph_trace_begin / ph_trace_end (which kernels made a frame; dry runs); ph_run_programs_progress; ph_buf_reuse; ph_image_unpack_rgb (program "rgb_unpack") */
#define PH_ABI_VERSION 8

enum {
  PH_OK = 0,
  PH_E_INVALID = -1,     /* bad argument / unknown kernel argument name */
  PH_E_NO_DEVICE = -2,   /* no HIP device (there is no CPU path) */
  PH_E_HIP = -3,         /* a HIP runtime call failed; see ph_last_error */
  PH_E_UNKNOWN_KERNEL = -4,
  PH_E_RANGE = -5        /* buffer too small for the geometry requested */
};

typedef struct ph_ctx ph_ctx;
typedef struct ph_buf ph_buf;
typedef struct ph_program ph_program;

/* nodencl `clContext.queue.{load,process,unload}` (src/index.ts:94-108, io.ts:79-98,
 * clJobQueue.ts:126,131): three in-order queues. */
enum { PH_QUEUE_LOAD = 0, PH_QUEUE_PROCESS = 1, PH_QUEUE_UNLOAD = 2 };

/* ---- context: `new clContext({platformIndex, deviceIndex, overlapping})` + `initialise()`
 *      (src/index.ts:94-107) ------------------------------------------------------------------ */
int ph_ctx_create(int device_index, ph_ctx **out);
int ph_ctx_destroy(ph_ctx *ctx);
/* `getPlatformInfo()` (src/index.ts:103-106): vendor / device name strings. */
int ph_ctx_info(ph_ctx *ctx, char *vendor, size_t vendor_len, char *device, size_t device_len);
/* message of the last failure on this thread (ctx may be NULL for ph_ctx_create failures). */
const char *ph_last_error(ph_ctx *ctx);

/* ---- which kernels made it (no nodencl counterpart: the reference runs one kernel per job, clJobQueue.ts:126; this library folds
 * jobs into fused launches and chooses among several routes by the shape of a frame - DESIGN.md section 5 lists them).
 * Between ph_trace_begin and ph_trace_end every kernel launch the CALLING THREAD's calls make is noted, in order; ph_trace_end
 * returns the names joined by '+' - e.g. "fused_v210_combine_lds", "chan_compose_v210<0,0>" (<phase-1 instantiation, output format>),
 * "chan_compose_batch<0>x4" (four jobs in the launch), "pack_read+compose_up_write_v210", "v210_yadif_pair+compose_up_write_v210".
 * dry_run != 0: the calls check their arguments and choose their kernels exactly as always but enqueue NOTHING (outputs untouched) -
 * "which route would this frame take".  PH_E_RANGE when `route` is too short (the trace is dropped), PH_E_INVALID without a begin. */
int ph_trace_begin(int dry_run);
int ph_trace_end(char *route, size_t len);
/* the hipStream_t behind a queue, for interop (tests wrap it as a torch external stream). */
void *ph_ctx_stream(ph_ctx *ctx, int queue);
/* `waitFinish(queue)` (clJobQueue.ts:131, io.ts callers): returns when the stream is idle. */
int ph_wait_finish(ph_ctx *ctx, int queue);
/* non-blocking: 1 = the queue is idle, 0 = work still in flight, negative = error.  Lets a caller
 * that expects the work to end within microseconds poll briefly instead of paying a thread hand-off. */
int ph_queue_query(ph_ctx *ctx, int queue);

/* ---- buffers: `createBuffer(bytes, access, svmType, imageDims?, owner?)` (19 call sites, e.g.
 *      io.ts:61-77,144-150, mixer.ts:196-205, combiner.ts:230-239, yadif.ts:76-86) -------------- */
enum { PH_ACCESS_READONLY = 0, PH_ACCESS_WRITEONLY = 1, PH_ACCESS_READWRITE = 2 };
enum { PH_SVM_NONE = 0, PH_SVM_COARSE = 1, PH_SVM_FINE = 2 };
/* width/height > 0 mark the buffer as an RGBA f32 image (nodencl `ImageDims`): row-major,
 * unpadded, 16 bytes per pixel.  Storage comes from a per-context pool (no hipMalloc per frame). */
int ph_buf_create(ph_ctx *ctx, size_t bytes, int access, int svm_type, int width, int height,
                  const char *owner, ph_buf **out);
/* adopt device memory owned by the caller (e.g. a torch tensor); never freed by the library. */
int ph_buf_wrap(ph_ctx *ctx, void *device_ptr, size_t bytes, int width, int height, ph_buf **out);
int ph_buf_addref(ph_buf *buf);  /* OpenCLBuffer.addRef()  (loadSave.ts:102-106 ...) */
int ph_buf_release(ph_buf *buf); /* OpenCLBuffer.release(): last release recycles the storage */
int ph_buf_refcount(const ph_buf *buf);
size_t ph_buf_bytes(const ph_buf *buf);
void *ph_buf_device_ptr(ph_buf *buf);
int ph_buf_dims(const ph_buf *buf, int *width, int *height);
/* `buf.hostAccess(dir, queue, src?)` (io.ts:89-94,172; loadSave.ts:76-99; transform.ts:84-89):
 *   WRITEONLY + src : copy `bytes` of host memory to the device on `queue` (async, pinned staging)
 *   WRITEONLY, no src: expose the host mirror for the caller to fill (ph_buf_host_ptr)
 *   NONE            : hand the (filled) host mirror back to the device
 *   READONLY        : make the device contents visible in the host mirror (sync on return) */
enum { PH_HOST_READONLY = 0, PH_HOST_WRITEONLY = 1, PH_HOST_NONE = 2 };
int ph_buf_host_access(ph_buf *buf, int dir, int queue, const void *src, size_t bytes);
/* For a binding that keeps released buffers itself and hands them to their next owner whole (node/index.js parks frames and images:
 * handle, device block, pinned mirror): what ph_buf_release + ph_buf_create would have seen to - waits for an asynchronous mirror copy
 * still in flight, orders the queues behind ROUTE transfers still using the device block, forgets the previous owner's pending host
 * data.  The contents are unspecified afterwards, as a fresh buffer's are. */
int ph_buf_reuse(ph_buf *b);
void *ph_buf_host_ptr(ph_buf *buf); /* pinned host mirror (allocated on first use) */
/* Staged producers / consumers (the queue.load / queue.unload roles of io.ts:79-98,
 * 166-174).  nodencl orders its three queues through host-side `waitFinish`; these two calls let a
 * caller that keeps a ring of frames in flight order them on the device instead, so uploads of frame
 * n+1 and downloads of frame n-1 overlap the kernels of frame n without a host round trip:
 *   ph_queue_wait_queue : work enqueued on `waiter` after this call starts only once everything
 *                         enqueued on `signal` before this call has finished (event record + wait).
 *   ph_buf_download_async: device -> pinned host mirror on `queue`, no host synchronisation; the
 *                         bytes at ph_buf_host_ptr() are valid after ph_wait_finish(queue) or after a
 *                         ph_event recorded behind it has been waited for.
 * (Upload without a host wait already exists: PH_HOST_WRITEONLY with no src, fill the mirror,
 * PH_HOST_NONE.) */
int ph_queue_wait_queue(ph_ctx *ctx, int waiter_queue, int signal_queue);
int ph_buf_download_async(ph_buf *buf, int queue);
/* a point in a queue the host can wait for without draining what was enqueued after it (one per
 * ring slot in a staged chain).  ph_event_wait returns when the work enqueued before the record has
 * finished; the event stays valid until ph_event_destroy. */
typedef struct ph_event ph_event;
int ph_event_record(ph_ctx *ctx, int queue, ph_event **out);
int ph_event_wait(ph_event *ev);
int ph_event_query(ph_event *ev); /* 1 = finished, 0 = still running, negative = error */
int ph_event_destroy(ph_event *ev);
/* Timed points: ph_event_record_timed records an event that ph_event_elapsed_us can measure from / to (microseconds of device
 * time between two such points of one queue, both finished).  nodencl's RunTimings (clJobQueue.ts:159-215 prints them) for a binding
 * whose launches do not coincide with its runProgram calls (node/defer.js with `profile`). */
int ph_event_record_timed(ph_ctx *ctx, int queue, ph_event **out);
int ph_event_elapsed_us(ph_event *from, ph_event *to, uint32_t *microseconds);
/* Recorded batches.  The reference submits a channel's per-frame batch (read x N, transform x N, wipe,
 * combine, write: 13 kernels at config 2) one `runProgram` at a time; at 1080p those kernels run for
 * 5-15 us each and the gaps between launches are a fifth of the frame.  A caller whose batch touches
 * the same buffers every frame (ring slots) can record it once and replay it as ONE submission:
 *   ph_graph_begin(ctx, queue)  - every launch issued on `queue` from this thread is recorded, not run
 *                                 (hipStreamBeginCapture); no waitFinish / hostAccess('readonly') inside
 *   ph_graph_end(ctx, queue, &g) - stop recording, build the executable graph
 *   ph_graph_launch(g, queue)   - replay on `queue` (asynchronous, ordered like any other launch)
 * Scalars and pointers are frozen at recording time. */
typedef struct ph_graph ph_graph;
int ph_graph_begin(ph_ctx *ctx, int queue);
int ph_graph_end(ph_ctx *ctx, int queue, ph_graph **out);
int ph_graph_launch(ph_graph *graph, int queue);
int ph_graph_destroy(ph_graph *graph);
/* ---- ROUTE across GPUs (src/producer/routeProducer.ts:63-126, src/channel.ts:290-300).  In the reference a
 *      route producer taps another channel's combiner output by taking a reference on the same OpenCL buffer:
 *      one process, one device.  Here channels are partitioned one process per GPU, so a route whose
 *      two ends live on different GPUs becomes ONE point-to-point message per frame: RCCL send / recv over xGMI
 *      on a communication stream of its own, ordered against the compute queues on the device:
 *        source rank:  combine ... ; ph_route_after_queue(r, PROCESS); ph_route_send(r, frame, peer)
 *        sink rank:    ph_route_recv(r, frame, peer); ph_queue_after_route(r, PROCESS); combine(... frame ...)
 *      so the transfer overlaps whatever else the sink's process queue runs before that combine (its own layers'
 *      v210 reads), and no host thread waits.  Sends / receives of one frame period go between
 *      ph_route_group_begin / _end (ncclGroupStart / End): every rank may then post them in any order.
 *      librccl is loaded on first use (dlopen); a process that never routes never needs it.
 *      ph_route_unique_id: 128 bytes made by ONE rank and handed to the others out of band (the caller's own
 *      control channel - torch.distributed in this repo's tools, a socket in a node deployment). ------------- */
typedef struct ph_route ph_route;
#define PH_ROUTE_ID_BYTES 128
int ph_route_unique_id(void *id128);
int ph_route_init(ph_ctx *ctx, const void *id128, int rank, int world, ph_route **out);
int ph_route_destroy(ph_route *route);
int ph_route_group_begin(ph_route *route);
int ph_route_group_end(ph_route *route);
/* device pointers (ph_buf_device_ptr / a wrapped tensor); bytes % 4 == 0; peer == own rank is allowed inside a group */
int ph_route_send(ph_route *route, const void *device_src, size_t bytes, int peer);
int ph_route_recv(ph_route *route, void *device_dst, size_t bytes, int peer);
/* the communication stream waits for everything enqueued so far on `queue` (the frame being sent is complete) */
int ph_route_after_queue(ph_route *route, int queue);
/* `queue` waits for everything enqueued so far on the communication stream (the received frame has landed) */
int ph_queue_after_route(ph_route *route, int queue);
int ph_route_wait(ph_route *route); /* host wait for the communication stream (tests, shutdown) */
void *ph_route_stream(ph_route *route);
/* ranks in the route's RCCL communicator (ncclCommCount): what RCCL itself saw, not what the caller passed in */
int ph_route_comm_count(ph_route *route, int *count);
/* Buffer lifetime: a frame handed to ph_route_send may be RELEASED right after the call - the pool orders the three
 * queues behind the transfers in flight before it hands a recycled block out again.  REWRITING the same buffer (a
 * source that composes the next frame into it) needs ph_queue_after_route(route, queue) first, like any other
 * cross-stream reuse. */

/* the `logBuffers()` debug hook (src/index.ts:184): live buffers / pooled bytes */
int ph_ctx_buffer_stats(ph_ctx *ctx, size_t *live_buffers, size_t *live_bytes, size_t *pooled_bytes);
/* the pinned host mirrors (an OpenCLBuffer of the node binding IS its mirror): bytes attached to live buffers, bytes kept for reuse,
 * the most that was ever attached at once, and how many blocks have been pinned (hipHostMalloc) since the context was made - a
 * count that keeps growing in steady state means the pool is smaller than the working set (context option "host_pool_mb") */
int ph_ctx_host_pool_stats(ph_ctx *ctx, size_t *in_use_bytes, size_t *pooled_bytes, size_t *peak_in_use_bytes, uint64_t *pins);

/* ---- programs: `createProgram(kernelSrc, {name, globalWorkItems, workItemsPerGroup})`
 *      (imageProcess.ts:69-72, packer.ts:97-103).  The OpenCL C text is NOT compiled: it (or a
 *      "phaneron:<op>" tag) only selects a precompiled gfx950 kernel by kernel name + argument
 *      list.  Names: read/write (v210, yuv422p10le, yuv422p8, yuv420p, nv12, rgba8, bgra8 - told
 *      apart by their argument lists), yadif, transform, resize, combine_N, transition_dissolve,
 *      transition_wipe, mixer, wipe. ------------------------------------------------------------ */
int ph_program_create(ph_ctx *ctx, const char *kernel_src, const char *name,
                      const uint32_t *global_work_items, int n_dims,
                      uint32_t work_items_per_group, ph_program **out);
/* The selection alone, without a context or a device (same rules, same errors): writes the resolved
 * kernel id ("v210_read", "yuv420p_write", "combine_4", ...), the PH_FMT_* of a read / write program
 * (-1 otherwise) and HOW it was selected - by "phaneron:" tag, by unique kernel name, by the exact text of
 * one of the reference's seven pack-format sources, or (any other text) by the kernel's argument list. */
enum { PH_RESOLVED_BY_TAG = 0, PH_RESOLVED_BY_NAME = 1, PH_RESOLVED_BY_TEXT = 2, PH_RESOLVED_BY_SIGNATURE = 3 };
int ph_program_resolve(const char *kernel_src, const char *name, char *kernel_id, size_t kernel_id_len,
                       int *format, int *how);
int ph_program_destroy(ph_program *prog);
const char *ph_program_kernel(const ph_program *prog); /* resolved kernel id, e.g. "v210_read" */

/* ---- `runProgram(program, params, queue)` (clJobQueue.ts:126): params keyed by the OpenCL
 *      kernel ARGUMENT NAME (input, output, width, colMatrix, gammaLut, gamutMatrix, interlace,
 *      prev, cur, next, parity, tff, skipSpatial, transformMatrix, scale, offsetX, offsetY, flip,
 *      l<i>In, input0, input1, mix, wipe, maskIn) ------------------------------------------------- */
enum { PH_ARG_BUF = 0, PH_ARG_U32 = 1, PH_ARG_I32 = 2, PH_ARG_F32 = 3 };
typedef struct ph_arg {
  const char *name;
  int kind;
  union {
    ph_buf *buf;
    uint32_t u32;
    int32_t i32;
    float f32;
  } v;
} ph_arg;
/* nodencl `RunTimings` in microseconds (clJobQueue.ts:159-215 prints them) */
typedef struct ph_run_timings {
  uint32_t data_to_kernel;
  uint32_t kernel_exec;
  uint32_t total_time;
} ph_run_timings;
/* timings may be NULL (no events recorded, fully asynchronous).  With timings the call
 * returns after the kernel has finished (hipEvent pair on the queue's stream). */
int ph_run_program(ph_ctx *ctx, ph_program *prog, const ph_arg *args, int n_args, int queue,
                   ph_run_timings *timings);
/* Everything ph_run_program checks before it launches - argument names and kinds, buffer sizes against the frame geometry,
 * image-ness - with nothing enqueued and no buffer touched: the same return codes and ph_last_error() texts.  For a binding
 * that records jobs and launches them later (node/defer.js): a bad job is reported where the reference posts it
 * (clJobQueue.ts:126 awaits runProgram), not where it is finally run. */
int ph_check_program(ph_ctx *ctx, ph_program *prog, const ph_arg *args, int n_args, int queue);
/* Several jobs in one call, for a binding that records jobs and launches them later (node/defer.js): exactly ph_run_program(progs[j],
 * args[j], n_args[j]) for j = 0 .. n_jobs - 1 in that order (a job that reads or writes what an earlier job of the call writes, or writes what one
 * reads, is kept out of that job's launch).  Every job is
 * checked before anything is launched (a bad job refuses the whole call).  Channel frames among the jobs - chan_compose_v210_<n>
 * programs of one geometry that name the SAME Loader / Saver buffers and make v210 frames - go to the device together
 * (ph_chan_compose_batch: the reference's channels share one context and one queue, src/index.ts:45-71,156-160); likewise consecutive
 * fused_v210_combine_<n> programs of one geometry, layer count and recipe (ph_fused_v210_combine_batch, up to eight per launch; a frame
 * that touches an earlier one's output starts the next launch), and consecutive compose_up_write_v210_<n> jobs of one shape - layer count,
 * image format and sizes, placements, frame size, field mode, Saver - each one frame or a frame's two fields (ph_compose_up_write_v210_batch, up to four frames
 * per launch). */
int ph_run_programs(ph_ctx *ctx, int n_jobs, ph_program *const *progs, const ph_arg *const *args, const int *n_args, int queue);
/* How far the calling thread's LAST ph_run_programs call got: the launches of jobs 0 .. *jobs_done - 1 were made (n_jobs after a call
 * that returned PH_OK, 0 after one refused by its checks).  A call that fails at a launch has enqueued the jobs before the failing
 * group; a binding that falls back to ph_run_program for the rest starts at *jobs_done instead of rendering those frames twice. */
int ph_run_programs_progress(int *jobs_done);

/* ---- typed entry points: the same kernels on raw device pointers, launched on `queue`.
 *      Images are row-major float RGBA (16 B/pixel); v210 is LE 32-bit words with line pitch
 *      ph_v210_pitch_bytes(width).  Matrices / LUTs are DEVICE pointers (the reference passes
 *      them as OpenCLBuffers, loadSave.ts:114-127). --------------------------------------------- */
uint32_t ph_v210_pitch_bytes(uint32_t width); /* v210.ts:198-204 */
/* v210.ts:25-111 */
int ph_v210_read(ph_ctx *ctx, int queue, const void *in, void *out, uint32_t width, uint32_t height,
                 const void *col_matrix12, const void *gamma_lut, const void *gamut_matrix9);
/* n frames of one size and one colour recipe (the layers of a channel) unpacked in one launch; outs[i] receives exactly
 * what ph_v210_read(ins[i]) writes.  The launch and the per-CU table load are paid once per batch. */
int ph_v210_read_batch(ph_ctx *ctx, int queue, int n, const void *const *ins, void *const *outs, uint32_t width,
                       uint32_t height, const void *col_matrix12, const void *gamma_lut, const void *gamut_matrix9);
/* v210.ts:113-195; interlace 0 / 1 (top, even lines) / 3 (bottom, odd lines) (packer.ts:24-28) */
int ph_v210_write(ph_ctx *ctx, int queue, const void *in, void *out, uint32_t width,
                  uint32_t height, uint32_t interlace, const void *col_matrix12,
                  const void *gamma_lut);
/* yadifCl.ts:105-167 */
int ph_yadif(ph_ctx *ctx, int queue, const void *prev, const void *cur, const void *next, int width,
             int height, int parity, int tff, int skip_spatial, void *out);
/* Both output fields of one frame in one pass: out_parity0 / out_parity1 receive exactly what ph_yadif writes with
 * parity 0 / parity 1 (the Yadif wrapper's send_field mode runs the filter twice per frame over the same window,
 * parity 1 ^ tff then parity tff: yadif.ts:100-145).  Each source row is read once instead of up to twice. */
int ph_yadif_pair(ph_ctx *ctx, int queue, const void *prev, const void *cur, const void *next, int width,
                  int height, int tff, int skip_spatial, void *out_parity0, void *out_parity1);
/* ---- fused de-interlacing reader (no single reference equivalent): per layer, ToRGBA on the frames of the Yadif
 *      window (v210.ts:25-111) and both send_field outputs of Yadif (yadif.ts:100-145) as ONE kernel.  The window
 *      stays in v210; its rows are unpacked, matrixed and gamma-corrected on the fly, and only the two de-interlaced
 *      RGBA frames are written.  out_parity0 / out_parity1 are bit-identical to ph_v210_read on prev, cur and next
 *      followed by ph_yadif with parity 0 / 1.  n sources of one size and one colour recipe per call (the layers of a
 *      channel).  Needs width % 6 == 0 and the reader LUT registered (PH_E_INVALID otherwise - run the separate kernels). */
/* (ABI 4 and later: 88 bytes - the six chroma-plane pointers were added behind the original five; an array of the 40-byte
 * struct of ABI 2 / 3 has the wrong stride, see the ABI history at the top) */
typedef struct ph_deint_source {
  const void *prev, *cur, *next;      /* device, v210, width x height (ph_yadif_pair_packed: or the Y planes of planar frames) */
  void *out_parity0, *out_parity1;    /* device, float RGBA, width x height */
  const void *prev_u, *prev_v, *cur_u, *cur_v, *next_u, *next_v; /* ph_yadif_pair_packed with a planar packing: the chroma planes */
} ph_deint_source;
int ph_v210_yadif_pair(ph_ctx *ctx, int queue, int n, const ph_deint_source *sources, uint32_t width, uint32_t height,
                       int tff, int skip_spatial, const void *rd_col_matrix12, const void *rd_gamma_lut,
                       const void *rd_gamut_matrix9);
/* the same with a choice of output layout.  PH_IMG_RGB_F32: three floats per pixel (12 bytes, rows unpadded) - what a
 * v210 reader computes, without the alpha it sets to 1 (v210.ts:73-77, yadifCl.ts:164).  Only ph_compose_up_write_v210
 * reads that layout: a quarter less to write here and to fetch there. */
#define PH_IMG_RGBA_F32 0
#define PH_IMG_RGB_F32 1
int ph_v210_yadif_pair_fmt(ph_ctx *ctx, int queue, int n, const ph_deint_source *sources, uint32_t width, uint32_t height,
                           int tff, int skip_spatial, int out_format, const void *rd_col_matrix12, const void *rd_gamma_lut,
                           const void *rd_gamut_matrix9);
/* the same over windows of interlaced FILE frames: packing = PH_FMT_YUV422P10 or PH_FMT_YUV422P8 (planar 4:2:2, what decoders of
 * XDCAM / ProRes / DNxHD material hand over), PH_FMT_YUV420P or PH_FMT_NV12 (4:2:0: interlaced H.264 / MPEG-2; nv12: *_u are the
 * interleaved CbCr planes, *_v unused; even heights) - every source of the call in that packing, unpacked with the call's Loader recipe
 * (for the 8-bit packings the 8-bit Loader's matrix) - or PH_FMT_V210 (= the call above) */
int ph_yadif_pair_packed(ph_ctx *ctx, int queue, int n, const ph_deint_source *sources, int packing, uint32_t width, uint32_t height,
                         int tff, int skip_spatial, int out_format, const void *rd_col_matrix12, const void *rd_gamma_lut,
                         const void *rd_gamut_matrix9);

/* Extension (no reference kernel): a packed f32 RGB image (12 bytes per pixel; PH_IMG_RGB_F32, what ph_v210_yadif_pair_fmt can write) expanded IN
 * PLACE into the f32 RGBA image (alpha 1) its buffer is declared as - through the queue's scratch area.  Program name "rgb_unpack"
 * (argument `image`).  For a binding that keeps de-interlaced fields packed while only ph_compose_up_write_v210 reads them. */
int ph_image_unpack_rgb(ph_ctx *ctx, int queue, void *image, int w, int h);
/* transform.ts:36-59 (matrix9: device pointer to the 3x3 row-major matrix) */
int ph_transform(ph_ctx *ctx, int queue, const void *in, int in_w, int in_h, const void *matrix9,
                 void *out, int out_w, int out_h);
/* resize.ts:35-59 (flip4: device pointer) */
int ph_resize(ph_ctx *ctx, int queue, const void *in, int in_w, int in_h, float scale,
              float offset_x, float offset_y, const void *flip4, void *out, int out_w, int out_h);
/* combine.ts:24-68, 2 <= n <= 8 */
int ph_combine(ph_ctx *ctx, int queue, int n, const void *const *layers, int width, int height,
               void *out);
/* transition.ts:60-65 / :66-74, mix.ts:30-45, wipe.ts:30-47 */
int ph_transition_dissolve(ph_ctx *ctx, int queue, const void *in0, const void *in1, float mix,
                           int width, int height, void *out);
int ph_transition_wipe(ph_ctx *ctx, int queue, const void *in0, const void *in1, const void *mask,
                       int width, int height, void *out);
int ph_mixer(ph_ctx *ctx, int queue, const void *in0, const void *in1, float mix, int width,
             int height, void *out);
int ph_wipe(ph_ctx *ctx, int queue, const void *in0, const void *in1, float wipe, int width,
            int height, void *out);

/* ---- the other pack formats (src/process/{yuv422p10,yuv422p8,yuv420p,nv12,rgba8,bgra8}.ts):
 *      planes as the reference's Readers/Writers lay them out (numBytes[] of each format).  v210
 *      is accepted too (format 0, one plane).  col_matrix12 is NULL for the RGB formats (their
 *      Loader/Saver skip it: loadSave.ts:52-61,141-149).  Returns the number of planes. ---------- */
enum {
  PH_FMT_V210 = 0,
  PH_FMT_YUV422P10 = 1,
  PH_FMT_YUV422P8 = 2,
  PH_FMT_YUV420P = 3,
  PH_FMT_NV12 = 4,
  PH_FMT_RGBA8 = 5,
  PH_FMT_BGRA8 = 6
};
int ph_pack_plane_bytes(int format, uint32_t width, uint32_t height, size_t bytes[3]);
int ph_pack_read(ph_ctx *ctx, int queue, int format, const void *const planes[3], void *out,
                 uint32_t width, uint32_t height, const void *col_matrix12, const void *gamma_lut,
                 const void *gamut_matrix9);
/* n frames (1 .. 8) of ONE format, size and Loader recipe in one launch - several channels' clips of a tick (the reference's channels share a
 * context and a queue: src/index.ts:45-71,156-160): planes[i] are frame i's planes as ph_pack_read takes them, outs[i] its f32 RGBA image.
 * Exactly n ph_pack_read calls; PH_FMT_V210 goes to ph_v210_read_batch. */
int ph_pack_read_batch(ph_ctx *ctx, int queue, int format, int n, const void *const (*planes)[3], void *const *outs, uint32_t width,
                       uint32_t height, const void *col_matrix12, const void *gamma_lut, const void *gamut_matrix9);
int ph_pack_write(ph_ctx *ctx, int queue, int format, const void *in, void *const planes[3],
                  uint32_t width, uint32_t height, uint32_t interlace, const void *col_matrix12,
                  const void *gamma_lut);

/* ---- fused channel pipeline (no reference equivalent: it is the reference's job batch
 *      [v210 read] x n -> combine_n -> v210 write executed as ONE kernel with the
 *      f32 RGBA intermediates kept in registers).  Bit-identical to running the separate
 *      kernels above.  1 <= n <= 8 (n == 1: combiner passthrough, combiner.ts:222-228).
 *      All layers share one reader colourspec (matrices/LUT are device pointers). ------------- */
int ph_fused_v210_combine(ph_ctx *ctx, int queue, int n, const void *const *layers, void *out,
                          uint32_t width, uint32_t height, const void *rd_col_matrix12,
                          const void *rd_gamma_lut, const void *rd_gamut_matrix9,
                          const void *wr_col_matrix12, const void *wr_gamma_lut);
/* The same for `jobs` frames in ONE launch (channels of identical geometry and colour parameters, e.g.
 * the 1080p channels of BASELINE config 4 / 5 sharing a GPU): layers[j * n + l] is layer l of job j,
 * outs[j] its output.  The CUs are divided between the jobs, so launch, table loads and the
 * partially filled last slice are paid once per batch instead of once per frame: four 1080p frames
 * cost what one 2160p frame costs.  Up to 8 jobs; results identical to `jobs` separate calls. */
int ph_fused_v210_combine_batch(ph_ctx *ctx, int queue, int jobs, int n, const void *const *layers,
                                void *const *outs, uint32_t width, uint32_t height, const void *rd_col_matrix12,
                                const void *rd_gamma_lut, const void *rd_gamut_matrix9, const void *wr_col_matrix12,
                                const void *wr_gamma_lut);

/* ---- fused layer compositor (no single reference equivalent): the reference's job batches
 *      [transform] per layer (producer/mixer.ts:189-228) -> combine_N (combiner.ts:219-254) ->
 *      v210 write (io.ts:152-164) as ONE kernel, so the N + 1 consumer-size f32 RGBA frames between
 *      them never reach HBM.  Bit-identical to ph_transform x N + ph_combine + ph_v210_write.
 *      A layer with matrix9 == NULL is used 1:1 and must have the output size.  n == 1: passthrough
 *      of the single (transformed) layer as the combiner does.  Any even out_width (one that is not a
 *      multiple of 48 - 1280 - is served by the quad-per-lane kernel with the reference's tail
 *      arithmetic); the writer LUT must be registered (LDS form).  interlace as ph_v210_write. -- */
typedef struct ph_layer {
  const void *rgba;     /* device, float RGBA, width x height */
  int width, height;
  const void *matrix9;  /* device 3x3 transform matrix (ph_transform_matrix), or NULL */
} ph_layer;
int ph_compose_write_v210(ph_ctx *ctx, int queue, int n, const ph_layer *layers, void *out,
                          uint32_t out_width, uint32_t out_height, uint32_t interlace,
                          const void *wr_col_matrix12, const void *wr_gamma_lut);

/* The same compositor with wipe transitions inside: layer i's placed image is mixed with wipes[i].incoming_rgba by
 * the red channel of wipes[i].mask_rgba (transition.ts wipe, as the Transitioner runs it: transitioner.ts:165-176)
 * before the combine - [transform] -> transition_wipe -> combine_N -> write of a channel in mid-wipe as one kernel,
 * bit-identical to ph_transform + ph_transition_wipe + ph_combine + ph_v210_write.  Both images have the output
 * size; an entry with two NULLs leaves its layer alone.  Needs out_width % 192 == 0 (PH_E_INVALID otherwise). */
typedef struct ph_layer_wipe {
  const void *incoming_rgba; /* device, float RGBA, output size: the source the wipe reveals */
  const void *mask_rgba;     /* device, float RGBA, output size: mix factor in .x */
} ph_layer_wipe;
int ph_compose_wipe_write_v210(ph_ctx *ctx, int queue, int n, const ph_layer *layers, const ph_layer_wipe *wipes,
                               void *out, uint32_t out_width, uint32_t out_height, uint32_t interlace,
                               const void *wr_col_matrix12, const void *wr_gamma_lut);

/* ---- the compositor for magnified layers (no single reference equivalent): [transform] x N -> combine_N -> v210 write
 *      as ph_compose_write_v210, for placements that enlarge every layer (by 1 % or more) without rotation or mirroring -
 *      Mixer's default fill of HD sources on a UHD channel (producer/mixer.ts:209-223).  A lane produces a 2 x 2 block
 *      of output pixels from ONE 3 x 3 patch of each source (neighbouring output pixels share their taps): 2.25 texel
 *      loads per layer and pixel instead of 4.  Sources are f32 RGBA images or packed f32 RGB (PH_IMG_RGB_F32, alpha
 *      == 1 implied: ph_v210_yadif_pair_fmt), all layers of a call in the same layout.  Bit-identical to ph_transform +
 *      ph_combine + ph_v210_write.  A field write (interlace 1 / 3) needs more than 2x vertically: its rows are two lines apart.
 *      One placement at scale one qualifies too: an image of the frame's size under the Mixer's default fill (the identity matrix, a
 *      frame write) - de-interlaced 1080i fields on a 1080 channel.
 *      PH_E_INVALID when a placement does not qualify (use ph_compose_write_v210),
 *      out_width % 48 != 0 or the writer LUT is not registered.  interlace as ph_v210_write. ------------------- */
typedef struct ph_image_layer {
  const void *data;          /* device: width x height texels, rows unpadded */
  int format;                /* PH_IMG_RGBA_F32 | PH_IMG_RGB_F32 */
  int width, height;
  const float *matrix9_host; /* HOST: the nine values of ph_transform_matrix */
} ph_image_layer;
int ph_compose_up_write_v210(ph_ctx *ctx, int queue, int n, const ph_image_layer *layers, void *out, uint32_t out_width,
                             uint32_t out_height, uint32_t interlace, const void *wr_col_matrix12, const void *wr_gamma_lut);
/* Both fields of a de-interlaced frame in ONE launch: exactly ph_compose_up_write_v210(layers_a -> out_a) followed by
 * ph_compose_up_write_v210(layers_b -> out_b), for two sets of layers that differ in their data only (same count, formats, sizes and
 * placements: the parity-0 and parity-1 outputs of ph_v210_yadif_pair_fmt under one Mixer setting).  One launch, one table load and one
 * partly filled last round of wave steps instead of two. */
int ph_compose_up_write_v210_pair(ph_ctx *ctx, int queue, int n, const ph_image_layer *layers_a, const ph_image_layer *layers_b, void *out_a,
                                  void *out_b, uint32_t out_width, uint32_t out_height, uint32_t interlace, const void *wr_col_matrix12,
                                  const void *wr_gamma_lut);
/* The same for 1 .. 4 sets of layers that differ in their data only - several channels showing clips of one size under one placement
 * (the reference's channels share a context and a queue: src/index.ts:45-71,156-160): layer_sets[j][l] is layer l of job j, outs[j] its
 * output (all different).  ph_chan_compose_batch uses it for the frames of enlarged clips among its jobs. */
int ph_compose_up_write_v210_batch(ph_ctx *ctx, int queue, int jobs, int n, const ph_image_layer *const *layer_sets, void *const *outs,
                                   uint32_t out_width, uint32_t out_height, uint32_t interlace, const void *wr_col_matrix12,
                                   const void *wr_gamma_lut);

/* ---- the channel compositor straight from the wire format (no single reference equivalent): a channel's whole
 *      per-frame job batch - per layer ToRGBA (io.ts:79-98, v210.ts:25-111) -> Mixer transform (producer/mixer.ts:
 *      209-223, transform.ts:36-59) -> optionally the Transitioner's dissolve / wipe against a second source
 *      (transitioner.ts:165-176, transition.ts:54-79) -> combine_N (combiner.ts:219-254) -> FromRGBA (io.ts:152-164,
 *      v210.ts:113-195) - as ONE kernel that samples the v210 words themselves: each bilinear tap is unpacked,
 *      matrixed, gamma-looked-up and gamut-converted on the fly, so no f32 frame reaches HBM at all.  Bit-identical
 *      to ph_v210_read + ph_transform (+ ph_transition_dissolve / ph_transition_wipe) + ph_combine + ph_v210_write.
 *      A source is a v210 frame or (e.g. a routed frame, a generated mask) an f32 RGBA image; with matrix9_host ==
 *      NULL it is taken 1:1 and must have the output size.  All v210 sources share the reader's colour recipe.
 *      Limits (PH_E_INVALID otherwise - run the separate kernels): even widths for v210 frames, input or output (a width that
 *      is not a multiple of 6 / of 48 - 1280 x 720 - takes the reference's tail arithmetic, v210.ts:84-110,166-193, lines by
 *      pitch), planar output widths % 8 == 0, frames below 1 GiB, both gamma LUTs registered (LDS form).  interlace as
 *      ph_v210_write. ---------------------------------------------------------------------------------------------- */
#define PH_SRC_NONE 0
#define PH_SRC_V210 1
#define PH_SRC_RGBA_F32 2
/* planar YCbCr frames, as file decoders hand them over (ffmpegProducer.ts:398-412): data = the Y plane, data_u / data_v the chroma
 * planes (nv12: data_u = the interleaved CbCr plane, data_v unused), plane sizes as ph_pack_plane_bytes(PH_FMT_*, ...).  A 10-bit 4:2:2
 * Loader's matrix does not depend on the packing, so PH_SRC_YUV422P10 sources are unpacked with the call's Loader recipe like the
 * v210 ones; the 8-bit formats have code ranges of their own: col_matrix12 = that source's Loader matrix (its LUT and gamut
 * matrix are the call's: one colour space per call) */
#define PH_SRC_YUV422P10 3 /* yuv422p10.ts: 4:2:2, 16-bit little-endian samples */
#define PH_SRC_YUV422P8 4  /* yuv422p8.ts:  4:2:2, 8-bit */
#define PH_SRC_YUV420P 5   /* yuv420p.ts:   4:2:0, 8-bit */
#define PH_SRC_NV12 6      /* nv12.ts:      4:2:0, 8-bit, Cb and Cr interleaved */
/* packed 8-bit RGB (stills, graphics with alpha): four bytes per pixel, every byte - alpha too - through the call's gamma table,
 * r g b through its gamut matrix (rgba8.ts:49-62); no YCbCr matrix, no planes */
#define PH_SRC_RGBA8 7
#define PH_SRC_BGRA8 8
typedef struct ph_chan_source {
  const void *data;          /* device: v210 words (pitch ph_v210_pitch_bytes(width)), float RGBA, or the Y plane; width x height */
  int format;                /* PH_SRC_V210 | PH_SRC_RGBA_F32 | a planar PH_SRC_* | PH_SRC_RGBA8 | PH_SRC_BGRA8 (PH_SRC_NONE: absent) */
  int width, height;
  const float *matrix9_host; /* HOST: the nine values of ph_transform_matrix, or NULL = 1:1 */
  const void *data_u, *data_v; /* planar formats: the chroma plane(s) (ignored otherwise) */
  const void *col_matrix12;  /* planar formats: DEVICE, this source's YCbCr -> RGB matrix, or NULL = the call's */
} ph_chan_source;
#define PH_TRANSITION_CUT 0
#define PH_TRANSITION_DISSOLVE 1 /* fma(src, mix, incoming * (1 - mix))            transition.ts:58-64 */
#define PH_TRANSITION_WIPE 2     /* fma(incoming, mask.r, src * (1 - mask.r))      transition.ts:66-77 */
typedef struct ph_chan_layer {
  ph_chan_source src;
  int transition;
  float mix;                     /* dissolve */
  ph_chan_source incoming, mask; /* the transition's second source and (wipe) its mask */
} ph_chan_layer;
int ph_chan_compose_v210(ph_ctx *ctx, int queue, int n, const ph_chan_layer *layers, void *out, uint32_t out_width,
                         uint32_t out_height, uint32_t interlace, const void *rd_col_matrix12, const void *rd_gamma_lut,
                         const void *rd_gamut9, const void *wr_col_matrix12, const void *wr_gamma_lut);
/* The same with the packed frame in another wire format - FromRGBA with the Writers of the reference's other consumers: rgba8 / bgra8
 * (the screen, screenConsumer.ts:131; alpha 255, no writer matrix: wr_col_matrix12 may be NULL), yuv422p8 (an encoder,
 * ffmpegConsumer.ts:144), yuv422p10, and the 4:2:0 formats yuv420p / nv12 (yuv420p.ts:150-216, nv12.ts:139-196: chroma from the upper
 * line of a line pair; even heights; nv12: out_planes[1] is the interleaved CbCr plane); out_planes as ph_pack_plane_bytes(out_format,
 * ...) sizes them.  PH_FMT_V210 = the call above. */
int ph_chan_compose(ph_ctx *ctx, int queue, int n, const ph_chan_layer *layers, int out_format, void *const out_planes[3],
                    uint32_t out_width, uint32_t out_height, uint32_t interlace, const void *rd_col_matrix12, const void *rd_gamma_lut,
                    const void *rd_gamut9, const void *wr_col_matrix12, const void *wr_gamma_lut);
/* Several channels' frames in ONE launch.  The reference runs its channels - four of them, 1080p50 or smaller - in one context through
 * one queue (src/index.ts:45-71,156-160; src/clJobQueue.ts:114-141); a frame of that size does not fill the chip on its own.  Here the
 * jobs (frames of one geometry, one Loader / Saver colour recipe; each with its own layers, placements, transitions and interlace:
 * two jobs may be the two fields of one frame) share the workgroups of one launch: the gamma tables are loaded once, and the wave
 * steps of all jobs together are handed to the waves as they come free, dearest first.  Exactly `n_jobs` calls of ph_chan_compose_v210
 * in the order given, PROVIDED no job reads what another job of the call writes (jobs of one launch run side by side).  Jobs the
 * batch kernel does not take (planar / packed-RGB sources; more jobs, ops or wave steps than one launch holds) are split off into
 * launches of their own inside the call.  Limits and errors as ph_chan_compose_v210. */
typedef struct ph_chan_job {
  int n;                       /* layers */
  const ph_chan_layer *layers;
  void *out;                   /* device: the v210 frame, out_width x out_height */
  uint32_t interlace;          /* as ph_v210_write: 0 frame, 1 / 3 one field */
} ph_chan_job;
int ph_chan_compose_batch(ph_ctx *ctx, int queue, int n_jobs, const ph_chan_job *jobs, uint32_t out_width, uint32_t out_height,
                          const void *rd_col_matrix12, const void *rd_gamma_lut, const void *rd_gamut9, const void *wr_col_matrix12,
                          const void *wr_gamma_lut);

/* ---- gamma LUT placement.  The reference hands its kernels a 65536-entry f32 `gammaLut` buffer
 *      (loadSave.ts:65-73,152-160) and gathers from it 3x per pixel.  Registering the table's
 *      host contents lets the library keep an exact compressed copy for the CU's LDS
 *      (DESIGN.md "LUT placement"); kernels given a registered device pointer then use the LDS
 *      kernels, any other pointer uses the global-gather kernels.  Results are bit-identical.
 *      Buffers filled through ph_buf_host_access with 262144 bytes are registered automatically
 *      when first used as `gammaLut`.  Returns 1 if the table is LDS-capable, 0 if it is kept
 *      plain (not exactly compressible into 160 KiB), negative on error. ----------------------- */
int ph_lut_register(ph_ctx *ctx, const void *device_lut_f32, const float *host_lut65536);
int ph_lut_unregister(ph_ctx *ctx, const void *device_lut_f32);
/* lds_bytes = 0 when the pointer is unknown or plain; index_bias and blocks_per_octave_log2
 * describe the logarithmic block layout chosen for the table (ph_lut.h) */
int ph_lut_query(ph_ctx *ctx, const void *device_lut_f32, uint32_t *lds_bytes, uint32_t *index_bias,
                 uint32_t *blocks_per_octave_log2);
/* Device-free: compress a table exactly as ph_lut_register would and hand back the IMAGE of the LDS the table kernels
 * see - [0, hole) unused, then the anchors (u32 per logarithmic block), then the deltas (u16 per entry) - with the constants
 * of the lookup (phaneron_amd/csrc/ph_ldslut.h): for y = (float)idx + 1.5 * 2^23
 *     anchor byte address = (bits(fma(y, a_scale, -(1.5 * 2^23 - index_bias) * a_scale)) >> (shift - 2)) & ~3
 *     delta  byte address = delta_off + 2 * idx            (on the device: the bit pattern of a denormal fma)
 *     table[idx]          = the float whose bits are  u32 at the anchor address + u16 at the delta address.
 * `lds_image` may be NULL (layout only) or hold `capacity` >= lds_bytes bytes.  Returns the LDS footprint in bytes,
 * 0 when the table is not exactly compressible into 160 KiB (it stays plain: global-gather kernels), negative on error.
 * Replaces nothing in the reference: its kernels read the 256 KiB f32 table from global memory (v210.ts:68-70). */
typedef struct ph_lut_layout {
  uint32_t lds_bytes, hole, delta_off, shift, index_bias;
  float a_scale;
} ph_lut_layout;
int ph_lut_layout_of(const float *host_lut65536, ph_lut_layout *layout, void *lds_image, size_t capacity);
/* options: "lds_lut" (default 1): 0 forces the global-gather kernels (A/B tests, profiles);
 *          "stream_images" (default 0): 1 stores f32 image outputs (ToRGBA, Yadif, Transform, Combine ...) past the
 *          caches, for a caller that knows nothing on the device reads the image soon; 2 does so only for images
 *          larger than "stream_threshold_mb" MiB (default 64; measured neutral on the reference-shaped chains).  By
 *          default an image is treated as what it is in a channel, an intermediate the next operator reads back;
 *          wire-format outputs always stream;
 *          "host_pool_mb" (default 4096): how much pinned host memory released buffers' mirrors may keep for the next
 *          buffer of the same size (the reference creates its destinations per job and frame: io.ts:64-72) - this much,
 *          or as much as was ever attached to live buffers at once if that is more (a pool smaller than the working set
 *          pins a block per buffer again, ~40 ms each); over the budget the oldest blocks are freed first; 0: no pool;
 *          setting the option trims the pool to the new size and restarts the working-set measurement (after a 2160p
 *          period a caller sheds the memory by setting the option again; node/index.js parks its frames under a budget
 *          of its own - parkMb - and trim() hands them back);
 *          "fail_launches" (default 0) - TEST ONLY, a fault injection for the error paths of a binding (node/test/soak_run.js): > 0:
 *          every launch made through ph_run_program / ph_run_programs fails with PH_E_HIP; -k: the next k launches (groups of
 *          ph_run_programs) go through, then every one fails; checks still pass.  Nothing in a deployment sets it;
 *          "chan_enlarged" (default 1; 0 when PH_CHAN_ENLARGED=0 is in the environment): ph_chan_compose* make a frame all of whose
 *          layers are ENLARGED v210 clips (a 720p or SD clip filling a 1080 channel: the reference uploads clips at their own size,
 *          ffmpegProducer.ts:395-442) by ph_v210_read into scratch images + ph_compose_up_write_v210 - one conversion per source pixel
 *          instead of four per output pixel, the same bits; 0: the channel kernel for those frames too. */
int ph_ctx_set_option(ph_ctx *ctx, const char *name, int value);

/* ---- host colour maths (src/process/colourMaths.ts, run by Loader/Saver constructors
 *      loadSave.ts:50-63,139-149): outputs are HOST arrays the caller uploads. ------------------ */
int ph_colour_gamma2linear_lut(const char *colspec, float *lut65536);     /* :130-149 */
int ph_colour_linear2gamma_lut(const char *colspec, float *lut65536);     /* :151-169 */
int ph_colour_ycbcr2rgb_matrix(const char *colspec, int num_bits, int luma_black, int luma_white,
                               int chroma_range, float *m12);             /* :276-332 */
int ph_colour_rgb2ycbcr_matrix(const char *colspec, int num_bits, int luma_black, int luma_white,
                               int chroma_range, float *m12);             /* :334-390 */
int ph_colour_rgb2rgb_matrix(const char *src_colspec, const char *dst_colspec, float *m9); /* :392 */
/* transform.ts:119-171 */
int ph_transform_matrix(int width, int height, int flip_h, int flip_v, double anchor_x,
                        double anchor_y, double scale_x, double scale_y, double offset_x,
                        double offset_y, double rotate, float *m9);

int ph_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif
