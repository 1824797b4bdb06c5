// ph_api.cpp - the C ABI of libphaneron_hip.so (include/phaneron_hip.h): context, ref-counted
// pooled device buffers, program lookup by kernel name, named-argument dispatch (the nodencl
// `runProgram` contract the reference's clJobQueue drives) and the typed entry points.
//
// No CPU path exists here: every entry point that does work needs a HIP device.
#include <atomic>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <dlfcn.h>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/phaneron_hip.h"
#include "ph_kernels.h"
#include "ph_lut_host.h"
#include "ph_program.h"

using ph::KernelId;
using namespace ph;  // K_* kernel ids

namespace {

thread_local std::string g_err;

int fail(int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

#define PH_HIP(call)                                                                          \
  do {                                                                                        \
    hipError_t e_ = (call);                                                                   \
    if (e_ != hipSuccess) return fail(PH_E_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); \
  } while (0)

}  // namespace

// ---- route trace: which kernels made it ------------------------------------------------------------------------------------------
// Which of the library's routes makes a frame is decided here and in the launchers (headline kernel / channel kernel and its
// instantiations / batch kernel / read + 2 x 2-block compositor / pair launch ...), so a caller - and a test - can ASK: between
// ph_trace_begin and ph_trace_end every launch of the calling thread is noted by kernel name, and in a dry run that is all that happens.
namespace {
thread_local bool g_trace_on = false, g_trace_dry = false;
thread_local std::string g_trace;
}  // namespace
namespace ph {
bool trace_launch(const char *name) {
  if (!g_trace_on) return false;
  if (const char *p = strstr(name, "launch_")) name = p + 7;
  if (!g_trace.empty()) g_trace += '+';
  g_trace.append(name, strcspn(name, "("));
  return g_trace_dry;
}
}  // namespace ph

struct LutEntry {
  ph::LutView view{};  // bytes == 0: a plain table
  void *blob_dev = nullptr;
};

// Lifetime: buffers, programs, events and graphs each hold a reference on their context, so the storage a
// handle points into outlives the handle whatever order a garbage collector finalises them in.
// ph_ctx_destroy drains the queues, marks the context closed (new work is refused) and drops the creator's
// reference; the last handle released tears the device state down.
// Threading: `mu` guards the pool, the LUT registry and the counters (the node addon calls hostAccess and
// timed runProgram from libuv pool threads while the JS thread creates and releases buffers).
struct ph_ctx {
  int device = 0;
  std::map<const void *, LutEntry> luts;  // device f32 table -> compressed LDS form
  bool use_lds_lut = true;
  int stream_images = 0;        // f32 image outputs: 0 through the caches, 1 streamed past them, 2 by size (ph_device.h store_image)
  int stream_threshold_mb = 64;  // policy 2: images larger than this stream
  hipStream_t streams[3] = {nullptr, nullptr, nullptr};
  hipDeviceProp_t props;
  std::multimap<size_t, void *> pool;  // free device blocks by exact size
  size_t pooled_bytes = 0, live_buffers = 0, live_bytes = 0;
  // Pinned host mirrors by exact size.  The reference makes a fresh destination per job and frame (io.ts:64-72, mixer.ts:196,
  // combiner.ts:230) and the node binding gives every buffer its mirror at once (an OpenCLBuffer IS a node Buffer): a
  // hipHostMalloc / hipHostFree pair of a 2160p image is ~40 ms, a pool hit nothing
  // (a block carries the event recorded behind the last asynchronous copy that touched it: the next owner waits for it -
  // normally long complete - before it writes the mirror; blocks are evicted oldest first when the budget is exceeded)
  struct HostBlock {
    void *p;
    hipEvent_t busy;  // may be null
    uint64_t seq;
  };
  std::multimap<size_t, HostBlock> host_pool;
  size_t host_pooled_bytes = 0;
  uint64_t host_pool_seq = 0;
  // what the pool may keep pinned: this many MiB, or - if that is more - as much as was ever in use at once (host_peak_bytes): a
  // pool smaller than the working set frees and pins a block per buffer again, 40 ms each (four 1080p channels create 36 images of
  // 33 MB per tick: round 5 measured 40 ms per tick under the fixed 1 GiB of round 4)
  // option "chan_enlarged" (default 1; PH_CHAN_ENLARGED=0 in the environment makes it 0): frames of enlarged clips by read + 2 x 2-block compositor
  int chan_enlarged = !(getenv("PH_CHAN_ENLARGED") && getenv("PH_CHAN_ENLARGED")[0] == '0');
  std::atomic<int> fail_launches{0};  // option "fail_launches" (a TEST hook): > 0 every launch through ph_run_program(s) fails; -k: the next k go through, then every one fails
  int host_pool_mb = 4096;
  size_t host_live_bytes = 0, host_peak_bytes = 0;  // mirrors attached to buffers now / at most
  uint64_t host_pins = 0;                           // hipHostMalloc calls so far (ph_ctx_host_pool_stats)
  void *chan_index[3] = {nullptr, nullptr, nullptr};  // index frame of the channel compositor, one per queue (ph_chan_compose_v210)
  size_t chan_index_bytes[3] = {0, 0, 0};
  // who is between taking a piece of a queue's area and enqueueing the last launch that uses it (callers on several threads: the launches of
  // two calls on one queue must not interleave around the shared scratch); taken before ctx->mu, never the other way round
  std::mutex chan_scratch_mu[3];
  unsigned chan_scratch_turn[3] = {0, 0, 0};  // which third of the area the next frame of enlarged clips puts its images in (chan_compose_enlarged)
  std::vector<struct ph_route *> routes;  // open ROUTEs: a recycled block must not be handed out under a transfer in flight
  std::mutex mu;
  std::atomic<int> refs{1};
  std::atomic<bool> closed{false};
  std::atomic<bool> lds_base_checked{false};  // ph_lut_register: the kernels' dynamic shared array starts at LDS address 0
};

struct ph_buf {
  ph_ctx *ctx;
  void *dptr;
  void *hptr;  // pinned host mirror, lazily allocated
  size_t bytes;
  int width, height;
  std::atomic<int> refs;
  bool owned;
  // written by libuv pool threads (hostAccess) and read by the launching thread (flush_dirty_args)
  std::atomic<bool> host_dirty;
  std::atomic<bool> lut_dirty;  // host data went into a table-sized buffer since its LDS form was last built
  std::string owner;
  hipEvent_t mirror_busy = nullptr;  // recorded behind the last asynchronous copy into or out of the mirror (travels with it into the pool)
  hipStream_t mirror_stream = nullptr;  // the stream that copy was enqueued on
};


struct ph_program {
  ph_ctx *ctx;
  KernelId id;
  int n_layers;  // combine_N
  int format;    // PH_FMT_* of a read/write program
  std::string kernel;
  uint32_t global[2];
  uint32_t local;
};

namespace {

int set_device(ph_ctx *ctx) {
  if (ctx->closed.load()) return fail(PH_E_INVALID, "the context has been destroyed");
  PH_HIP(hipSetDevice(ctx->device));
  ph::t_stream_images = (uint32_t)ctx->stream_images;  // the launch that follows on this thread takes the context's store policy
  ph::t_stream_threshold_mb = (uint32_t)ctx->stream_threshold_mb;
  return PH_OK;
}

bool queue_ok(int queue) { return queue >= 0 && queue < 3; }
int bad_queue(const char *fn, int queue) {
  return fail(PH_E_INVALID, "%s: queue %d is not PH_QUEUE_LOAD (0), PH_QUEUE_PROCESS (1) or PH_QUEUE_UNLOAD (2)", fn, queue);
}
// The stream behind a queue index.  There is no unchecked accessor: an out-of-range index makes the ENCLOSING entry point
// return PH_E_INVALID here, whether or not it remembered PH_QUEUE() at its top - never a silent alias of the process queue.
#define stream_of(ctx, queue)                                  \
  ({                                                           \
    const int ph_q_ = (queue);                                 \
    if (!queue_ok(ph_q_)) return bad_queue(__func__, ph_q_);   \
    (ctx)->streams[ph_q_];                                     \
  })
#define PH_QUEUE(fn, queue)                        \
  do {                                             \
    if (!queue_ok(queue)) return bad_queue(fn, queue); \
  } while (0)

void order_queues_after_routes(ph_ctx *ctx);  // below, with ph_route

int pool_alloc(ph_ctx *ctx, size_t bytes, void **out) {
  {
    std::lock_guard<std::mutex> lock(ctx->mu);
    auto it = ctx->pool.find(bytes);
    if (it != ctx->pool.end()) {
      *out = it->second;
      ctx->pool.erase(it);
      ctx->pooled_bytes -= bytes;
      // the block's previous owner may have released it right after ph_route_send: RCCL could still be reading it on
      // the communication stream, which the three queues know nothing about
      order_queues_after_routes(ctx);
      return PH_OK;
    }
  }
  PH_HIP(hipMalloc(out, bytes ? bytes : 1));
  return PH_OK;
}

void pool_free(ph_ctx *ctx, size_t bytes, void *p) {
  // Recycling is stream-safe because all kernels touching a buffer were enqueued (in order)
  // before its last release, and the next user enqueues after it on the same in-order queues;
  // cross-queue users call ph_wait_finish first, exactly as the reference does (io.ts, clJobQueue.ts:131).
  // ROUTE transfers run on a fourth stream: pool_alloc orders the queues behind them before a block is reused.
  std::lock_guard<std::mutex> lock(ctx->mu);
  ctx->pool.emplace(bytes, p);
  ctx->pooled_bytes += bytes;
}

void ctx_ref(ph_ctx *ctx) { ctx->refs.fetch_add(1); }

// contexts that exist: ph_ctx_destroy on a pointer that is no longer (or never was) a context is refused instead of
// touching freed storage (a second destroy after the last handle went)
std::mutex g_live_mu;
std::vector<ph_ctx *> g_live;
bool ctx_is_live(ph_ctx *ctx) {
  std::lock_guard<std::mutex> lock(g_live_mu);
  for (ph_ctx *c : g_live)
    if (c == ctx) return true;
  return false;
}

// drops one reference; the last one frees the device state (streams, pool, LUT blobs)
void ctx_unref(ph_ctx *ctx) {
  if (ctx->refs.fetch_sub(1) != 1) return;
  {
    std::lock_guard<std::mutex> lock(g_live_mu);
    for (size_t i = 0; i < g_live.size(); ++i)
      if (g_live[i] == ctx) {
        g_live.erase(g_live.begin() + (long)i);
        break;
      }
  }
  hipSetDevice(ctx->device);
  for (int i = 0; i < 3; ++i)
    if (ctx->streams[i]) {
      hipStreamSynchronize(ctx->streams[i]);
      hipStreamDestroy(ctx->streams[i]);
    }
  for (auto &kv : ctx->pool) hipFree(kv.second);
  for (auto &kv : ctx->host_pool) {
    hipHostFree(kv.second.p);
    if (kv.second.busy) hipEventDestroy(kv.second.busy);
  }
  for (auto &kv : ctx->luts)
    if (kv.second.blob_dev) hipFree(kv.second.blob_dev);
  for (int i = 0; i < 3; ++i)
    if (ctx->chan_index[i]) hipFree(ctx->chan_index[i]);
  delete ctx;
}

int closed_error(const char *fn) { return fail(PH_E_INVALID, "%s: the context has been destroyed", fn); }

const ph_arg *find_arg(const ph_arg *args, int n, const char *name) {
  for (int i = 0; i < n; ++i)
    if (args[i].name && 0 == strcmp(args[i].name, name)) return &args[i];
  return nullptr;
}

int need_buf(const ph_arg *args, int n, const char *name, size_t min_bytes, ph_buf **out) {
  const ph_arg *a = find_arg(args, n, name);
  if (!a || a->kind != PH_ARG_BUF || !a->v.buf) return fail(PH_E_INVALID, "kernel argument '%s' (buffer) missing", name);
  if (a->v.buf->bytes < min_bytes)
    return fail(PH_E_RANGE, "kernel argument '%s': buffer of %zu bytes, %zu needed", name, a->v.buf->bytes, min_bytes);
  *out = a->v.buf;
  return PH_OK;
}

int need_num(const ph_arg *args, int n, const char *name, double *out) {
  const ph_arg *a = find_arg(args, n, name);
  if (!a || a->kind == PH_ARG_BUF) return fail(PH_E_INVALID, "kernel argument '%s' (number) missing", name);
  *out = a->kind == PH_ARG_F32 ? (double)a->v.f32 : a->kind == PH_ARG_I32 ? (double)a->v.i32 : (double)a->v.u32;
  return PH_OK;
}

int need_image(ph_buf *b, const char *name, int *w, int *h) {
  if (b->width <= 0 || b->height <= 0) return fail(PH_E_INVALID, "kernel argument '%s' is not an image buffer", name);
  if (b->bytes < (size_t)b->width * b->height * 16) return fail(PH_E_RANGE, "image '%s' smaller than its dims", name);
  *w = b->width;
  *h = b->height;
  return PH_OK;
}

}  // namespace

extern "C" {

int ph_abi_version(void) { return PH_ABI_VERSION; }

const char *ph_last_error(ph_ctx *) { return g_err.c_str(); }

int ph_trace_begin(int dry_run) {
  g_trace_on = true, g_trace_dry = dry_run != 0;
  g_trace.clear();
  return PH_OK;
}
int ph_trace_end(char *route, size_t len) {
  const bool was_on = g_trace_on;
  g_trace_on = g_trace_dry = false;
  if (!was_on) return fail(PH_E_INVALID, "ph_trace_end: no ph_trace_begin on this thread");
  if (!route || len <= g_trace.size()) return fail(PH_E_RANGE, "ph_trace_end: the route takes %zu bytes", g_trace.size() + 1);
  memcpy(route, g_trace.c_str(), g_trace.size() + 1);
  return PH_OK;
}

int ph_ctx_create(int device_index, ph_ctx **out) {
  if (!out) return fail(PH_E_INVALID, "ph_ctx_create: out is NULL");
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0)
    return fail(PH_E_NO_DEVICE, "no HIP device available (%s); libphaneron_hip has no CPU path",
                e != hipSuccess ? hipGetErrorString(e) : "device count 0");
  if (device_index < 0 || device_index >= count) return fail(PH_E_INVALID, "device index %d out of range (%d devices)", device_index, count);
  ph_ctx *ctx = new ph_ctx();
  ctx->device = device_index;
  PH_HIP(hipSetDevice(device_index));
  PH_HIP(hipGetDeviceProperties(&ctx->props, device_index));
  for (int i = 0; i < 3; ++i) PH_HIP(hipStreamCreateWithFlags(&ctx->streams[i], hipStreamNonBlocking));
  {
    std::lock_guard<std::mutex> lock(g_live_mu);
    g_live.push_back(ctx);
  }
  *out = ctx;
  return PH_OK;
}

int ph_ctx_destroy(ph_ctx *ctx) {
  if (!ctx) return PH_OK;
  if (!ctx_is_live(ctx)) return PH_OK;  // destroyed before and every handle gone since: the storage no longer exists
  if (ctx->closed.exchange(true)) return PH_OK;  // second destroy: the creator's reference is already gone
  hipSetDevice(ctx->device);
  for (int i = 0; i < 3; ++i)
    if (ctx->streams[i]) hipStreamSynchronize(ctx->streams[i]);
  ctx_unref(ctx);  // handles still alive keep the storage they point into; the last one tears down
  return PH_OK;
}

int ph_ctx_info(ph_ctx *ctx, char *vendor, size_t vlen, char *device, size_t dlen) {
  if (!ctx) return fail(PH_E_INVALID, "ph_ctx_info: ctx is NULL");
  if (vendor && vlen) snprintf(vendor, vlen, "Advanced Micro Devices, Inc.");
  if (device && dlen) snprintf(device, dlen, "%s (%s)", ctx->props.name, ctx->props.gcnArchName);
  return PH_OK;
}

void *ph_ctx_stream(ph_ctx *ctx, int queue) { return ctx && queue_ok(queue) ? (void *)ctx->streams[queue] : nullptr; }

int ph_wait_finish(ph_ctx *ctx, int queue) {
  if (!ctx) return fail(PH_E_INVALID, "ph_wait_finish: ctx is NULL");
  if (ctx->closed.load()) return closed_error("ph_wait_finish");
  PH_QUEUE("ph_wait_finish", queue);
  PH_HIP(hipStreamSynchronize(stream_of(ctx, queue)));
  return PH_OK;
}

// ---- buffers -------------------------------------------------------------------------------
int ph_buf_create(ph_ctx *ctx, size_t bytes, int access, int svm_type, int width, int height, const char *owner,
                  ph_buf **out) {
  (void)access;
  (void)svm_type;  // OpenCL SVM hints have no HIP counterpart: device memory + pinned mirror
  if (!ctx || !out) return fail(PH_E_INVALID, "ph_buf_create: NULL argument");
  if (width > 0 && height > 0 && bytes < (size_t)width * height * 16)
    return fail(PH_E_RANGE, "ph_buf_create: %zu bytes cannot hold a %dx%d RGBA f32 image", bytes, width, height);
  int rc = set_device(ctx);
  if (rc) return rc;
  void *d = nullptr;
  rc = pool_alloc(ctx, bytes, &d);
  if (rc) return rc;
  ph_buf *b = new ph_buf{ctx, d, nullptr, bytes, width > 0 ? width : 0, height > 0 ? height : 0, {1}, true, false,
                         false, owner ? owner : ""};
  ctx_ref(ctx);
  {
    std::lock_guard<std::mutex> lock(ctx->mu);
    ctx->live_buffers++;
    ctx->live_bytes += bytes;
  }
  *out = b;
  return PH_OK;
}

int ph_buf_wrap(ph_ctx *ctx, void *device_ptr, size_t bytes, int width, int height, ph_buf **out) {
  if (!ctx || !out || !device_ptr) return fail(PH_E_INVALID, "ph_buf_wrap: NULL argument");
  if (ctx->closed.load()) return closed_error("ph_buf_wrap");
  *out = new ph_buf{ctx, device_ptr, nullptr, bytes, width > 0 ? width : 0, height > 0 ? height : 0, {1}, false, false,
                    false, "wrapped"};
  ctx_ref(ctx);
  {
    std::lock_guard<std::mutex> lock(ctx->mu);
    ctx->live_buffers++;
  }
  return PH_OK;
}

int ph_buf_addref(ph_buf *b) {
  if (!b) return fail(PH_E_INVALID, "ph_buf_addref: NULL buffer");
  return b->refs.fetch_add(1), PH_OK;
}

int ph_buf_release(ph_buf *b) {
  if (!b) return fail(PH_E_INVALID, "ph_buf_release: NULL buffer");
  if (b->refs.fetch_sub(1) > 1) return PH_OK;
  ph_ctx *ctx = b->ctx;
  hipSetDevice(ctx->device);
  ph_lut_unregister(ctx, b->dptr);  // the storage goes back to the pool: forget any LUT form of it
  if (b->owned) pool_free(ctx, b->bytes, b->dptr);
  if (b->hptr) {
    // The mirror goes to the pool WITH the event behind its last asynchronous copy (the next owner waits for it).  Over budget the
    // OLDEST blocks make room - after a format change the pool would otherwise fill with sizes nobody asks for any more and every
    // new size pay hipHostMalloc / hipHostFree again (~40 ms for a 2160p image); a block larger than the whole budget is freed.
    std::vector<ph_ctx::HostBlock> victims;
    {
      std::lock_guard<std::mutex> lock(ctx->mu);
      const size_t fixed = (size_t)ctx->host_pool_mb << 20;
      const size_t budget = ctx->host_pool_mb && ctx->host_peak_bytes > fixed ? ctx->host_peak_bytes : fixed;  // (0: no pool at all)
      ctx->host_live_bytes -= b->bytes;
      if (b->bytes > budget) {
        victims.push_back(ph_ctx::HostBlock{b->hptr, b->mirror_busy, 0});
      } else {
        while (ctx->host_pooled_bytes + b->bytes > budget && !ctx->host_pool.empty()) {
          auto oldest = ctx->host_pool.begin();
          for (auto it = ctx->host_pool.begin(); it != ctx->host_pool.end(); ++it)
            if (it->second.seq < oldest->second.seq) oldest = it;
          victims.push_back(oldest->second);
          ctx->host_pooled_bytes -= oldest->first;
          ctx->host_pool.erase(oldest);
        }
        ctx->host_pool.emplace(b->bytes, ph_ctx::HostBlock{b->hptr, b->mirror_busy, ctx->host_pool_seq++});
        ctx->host_pooled_bytes += b->bytes;
      }
    }
    for (auto &v : victims) {
      hipHostFree(v.p);  // (waits for copies in flight)
      if (v.busy) hipEventDestroy(v.busy);
    }
  } else if (b->mirror_busy) {
    hipEventDestroy(b->mirror_busy);
  }
  {
    std::lock_guard<std::mutex> lock(ctx->mu);
    if (b->owned) ctx->live_bytes -= b->bytes;
    ctx->live_buffers--;
  }
  delete b;
  ctx_unref(ctx);
  return PH_OK;
}

int ph_buf_refcount(const ph_buf *b) { return b ? b->refs.load() : 0; }
size_t ph_buf_bytes(const ph_buf *b) { return b ? b->bytes : 0; }
void *ph_buf_device_ptr(ph_buf *b) { return b ? b->dptr : nullptr; }
int ph_buf_dims(const ph_buf *b, int *w, int *h) {
  if (!b) return fail(PH_E_INVALID, "ph_buf_dims: NULL buffer");
  if (w) *w = b->width;
  if (h) *h = b->height;
  return PH_OK;
}

// an asynchronous copy into or out of b's mirror has just been enqueued on `s`
static void mirror_mark(ph_buf *b, hipStream_t s) {
  if (!b->mirror_busy && hipEventCreateWithFlags(&b->mirror_busy, hipEventDisableTiming) != hipSuccess) {
    b->mirror_busy = nullptr;
    hipStreamSynchronize(s);  // no event to carry: wait here rather than let the mirror go back to the pool under the copy
    return;
  }
  // ONE event stands for every copy of this mirror still in flight: a copy on another queue than the last one's (an upload on LOAD,
  // then a downloadAsync on UNLOAD) is ordered behind it first, so that the event recorded now covers both
  if (b->mirror_stream && b->mirror_stream != s) hipStreamWaitEvent(s, b->mirror_busy, 0);
  b->mirror_stream = s;
  hipEventRecord(b->mirror_busy, s);
}

void *ph_buf_host_ptr(ph_buf *b) {
  if (!b) return nullptr;
  if (!b->hptr) {
    {
      std::unique_lock<std::mutex> lock(b->ctx->mu);
      auto it = b->ctx->host_pool.find(b->bytes);
      if (it != b->ctx->host_pool.end()) {
        b->hptr = it->second.p;
        b->mirror_busy = it->second.busy;
        b->ctx->host_pool.erase(it);
        b->ctx->host_pooled_bytes -= b->bytes;
        b->ctx->host_live_bytes += b->bytes;
        if (b->ctx->host_live_bytes > b->ctx->host_peak_bytes) b->ctx->host_peak_bytes = b->ctx->host_live_bytes;
        lock.unlock();
        // the previous owner may have released the buffer with a download or an upload of this block still in flight
        // (release after downloadAsync, before its waitFinish: ADVICE r3); normally the event completed long ago
        if (b->mirror_busy) hipEventSynchronize(b->mirror_busy);
        return b->hptr;
      }
    }
    hipSetDevice(b->ctx->device);
    if (hipHostMalloc(&b->hptr, b->bytes ? b->bytes : 1, hipHostMallocDefault) != hipSuccess) {
      fail(PH_E_HIP, "hipHostMalloc(%zu) failed", b->bytes);
      b->hptr = nullptr;
    } else {
      std::lock_guard<std::mutex> lock(b->ctx->mu);
      b->ctx->host_pins++;
      b->ctx->host_live_bytes += b->bytes;
      if (b->ctx->host_live_bytes > b->ctx->host_peak_bytes) b->ctx->host_peak_bytes = b->ctx->host_live_bytes;
    }
  }
  return b->hptr;
}

/* A binding that keeps released buffers of its own (node/index.js parks frames and images whole: handle, device block, pinned mirror)
 * calls this when one of them gets its next owner: what ph_buf_release + ph_buf_create would have seen to - an asynchronous copy into
 * or out of the mirror still in flight (release after downloadAsync, before its waitFinish), a ROUTE transfer still reading or writing
 * the device block on the communication stream, and the previous owner's pending host data (a mirror filled but never handed back). */
int ph_buf_reuse(ph_buf *b) {
  if (!b || b->refs.load() <= 0) return fail(PH_E_INVALID, "ph_buf_reuse: NULL or released buffer");
  if (b->mirror_busy && hipEventSynchronize(b->mirror_busy) != hipSuccess) (void)hipGetLastError();
  {
    std::lock_guard<std::mutex> lock(b->ctx->mu);
    order_queues_after_routes(b->ctx);
  }
  b->host_dirty = false;
  b->lut_dirty = false;
  return PH_OK;
}

int ph_buf_host_access(ph_buf *b, int dir, int queue, const void *src, size_t bytes) {
  if (!b) return fail(PH_E_INVALID, "ph_buf_host_access: NULL buffer");
  PH_QUEUE("ph_buf_host_access", queue);
  int rc = set_device(b->ctx);
  if (rc) return rc;
  hipStream_t s = stream_of(b->ctx, queue);
  if (!ph_buf_host_ptr(b)) return PH_E_HIP;
  switch (dir) {
    case PH_HOST_WRITEONLY:
      if (src) {
        if (bytes > b->bytes) return fail(PH_E_RANGE, "hostAccess: %zu source bytes into a %zu byte buffer", bytes, b->bytes);
        // previous async upload from the mirror must have drained before it is overwritten
        PH_HIP(hipStreamSynchronize(s));
        memcpy(b->hptr, src, bytes);
        PH_HIP(hipMemcpyAsync(b->dptr, b->hptr, bytes, hipMemcpyHostToDevice, s));
        mirror_mark(b, s);
        b->host_dirty = false;
        b->lut_dirty = b->lut_dirty || b->bytes >= 65536 * 4;
      } else {
        b->host_dirty = true;  // caller fills the mirror, then calls hostAccess('none')
      }
      return PH_OK;
    case PH_HOST_NONE:
      if (b->host_dirty) {
        PH_HIP(hipMemcpyAsync(b->dptr, b->hptr, b->bytes, hipMemcpyHostToDevice, s));
        mirror_mark(b, s);
        b->host_dirty = false;
        b->lut_dirty = b->lut_dirty || b->bytes >= 65536 * 4;
      }
      return PH_OK;
    case PH_HOST_READONLY:
      PH_HIP(hipMemcpyAsync(b->hptr, b->dptr, b->bytes, hipMemcpyDeviceToHost, s));
      PH_HIP(hipStreamSynchronize(s));
      return PH_OK;
    default:
      return fail(PH_E_INVALID, "hostAccess: unknown direction %d", dir);
  }
}

int ph_queue_query(ph_ctx *ctx, int queue) {
  if (!ctx) return fail(PH_E_INVALID, "ph_queue_query: ctx is NULL");
  PH_QUEUE("ph_queue_query", queue);
  int rc = set_device(ctx);
  if (rc) return rc;
  hipError_t e = hipStreamQuery(stream_of(ctx, queue));
  if (e == hipSuccess) return 1;
  if (e == hipErrorNotReady) return 0;
  return fail(PH_E_HIP, "ph_queue_query: %s", hipGetErrorString(e));
}

int ph_queue_wait_queue(ph_ctx *ctx, int waiter_queue, int signal_queue) {
  if (!ctx) return fail(PH_E_INVALID, "ph_queue_wait_queue: ctx is NULL");
  if (waiter_queue < 0 || waiter_queue > 2 || signal_queue < 0 || signal_queue > 2)
    return fail(PH_E_INVALID, "ph_queue_wait_queue: queues are 0..2");
  if (waiter_queue == signal_queue) return PH_OK;  // in-order already
  int rc = set_device(ctx);
  if (rc) return rc;
  hipEvent_t ev;
  PH_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  hipError_t e = hipEventRecord(ev, ctx->streams[signal_queue]);
  if (e == hipSuccess) e = hipStreamWaitEvent(ctx->streams[waiter_queue], ev, 0);
  hipEventDestroy(ev);  // released by the runtime once the recorded work has completed
  if (e != hipSuccess) return fail(PH_E_HIP, "ph_queue_wait_queue: %s", hipGetErrorString(e));
  return PH_OK;
}

int ph_buf_download_async(ph_buf *b, int queue) {
  if (!b) return fail(PH_E_INVALID, "ph_buf_download_async: NULL buffer");
  PH_QUEUE("ph_buf_download_async", queue);
  int rc = set_device(b->ctx);
  if (rc) return rc;
  if (!ph_buf_host_ptr(b)) return PH_E_HIP;
  hipStream_t s = stream_of(b->ctx, queue);
  PH_HIP(hipMemcpyAsync(b->hptr, b->dptr, b->bytes, hipMemcpyDeviceToHost, s));
  mirror_mark(b, s);
  return PH_OK;
}

struct ph_event {
  ph_ctx *ctx;
  hipEvent_t ev;
};

int ph_event_record(ph_ctx *ctx, int queue, ph_event **out) {
  if (!ctx || !out) return fail(PH_E_INVALID, "ph_event_record: NULL argument");
  PH_QUEUE("ph_event_record", queue);
  int rc = set_device(ctx);
  if (rc) return rc;
  hipEvent_t ev;
  PH_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  hipError_t e = hipEventRecord(ev, stream_of(ctx, queue));
  if (e != hipSuccess) {
    hipEventDestroy(ev);
    return fail(PH_E_HIP, "ph_event_record: %s", hipGetErrorString(e));
  }
  *out = new ph_event{ctx, ev};
  ctx_ref(ctx);
  return PH_OK;
}

/* a point in `queue` that can be timed against another (ph_event_elapsed_us): RunTimings of a recording binding's fused launches */
int ph_event_record_timed(ph_ctx *ctx, int queue, ph_event **out) {
  if (!ctx || !out) return fail(PH_E_INVALID, "ph_event_record_timed: NULL argument");
  PH_QUEUE("ph_event_record_timed", queue);
  int rc = set_device(ctx);
  if (rc) return rc;
  hipEvent_t ev;
  PH_HIP(hipEventCreate(&ev));
  hipError_t e = hipEventRecord(ev, stream_of(ctx, queue));
  if (e != hipSuccess) {
    hipEventDestroy(ev);
    return fail(PH_E_HIP, "ph_event_record_timed: %s", hipGetErrorString(e));
  }
  *out = new ph_event{ctx, ev};
  ctx_ref(ctx);
  return PH_OK;
}

int ph_event_elapsed_us(ph_event *from, ph_event *to, uint32_t *us) {
  if (!from || !to || !us) return fail(PH_E_INVALID, "ph_event_elapsed_us: NULL argument");
  float ms = 0.f;
  hipError_t e = hipEventElapsedTime(&ms, from->ev, to->ev);  // both finished, both recorded with timing (ph_event_record_timed)
  if (e != hipSuccess) return fail(PH_E_HIP, "ph_event_elapsed_us: %s", hipGetErrorString(e));
  *us = (uint32_t)(ms * 1000.0f + 0.5f);
  return PH_OK;
}

int ph_event_wait(ph_event *ev) {
  if (!ev) return fail(PH_E_INVALID, "ph_event_wait: NULL event");
  int rc = set_device(ev->ctx);
  if (rc) return rc;
  PH_HIP(hipEventSynchronize(ev->ev));
  return PH_OK;
}

int ph_event_query(ph_event *ev) {
  if (!ev) return fail(PH_E_INVALID, "ph_event_query: NULL event");
  hipError_t e = hipEventQuery(ev->ev);
  if (e == hipSuccess) return 1;
  if (e == hipErrorNotReady) return 0;
  return fail(PH_E_HIP, "ph_event_query: %s", hipGetErrorString(e));
}

int ph_event_destroy(ph_event *ev) {
  if (!ev) return PH_OK;
  ph_ctx *ctx = ev->ctx;
  hipEventDestroy(ev->ev);
  delete ev;
  ctx_unref(ctx);
  return PH_OK;
}

struct ph_graph {
  ph_ctx *ctx;
  hipGraph_t graph;
  hipGraphExec_t exec;
};

int ph_graph_begin(ph_ctx *ctx, int queue) {
  if (!ctx) return fail(PH_E_INVALID, "ph_graph_begin: ctx is NULL");
  PH_QUEUE("ph_graph_begin", queue);
  int rc = set_device(ctx);
  if (rc) return rc;
  // relaxed: other threads (the libuv pool of the node addon) may keep calling into HIP meanwhile
  PH_HIP(hipStreamBeginCapture(stream_of(ctx, queue), hipStreamCaptureModeRelaxed));
  return PH_OK;
}

int ph_graph_end(ph_ctx *ctx, int queue, ph_graph **out) {
  if (!ctx || !out) return fail(PH_E_INVALID, "ph_graph_end: NULL argument");
  PH_QUEUE("ph_graph_end", queue);
  int rc = set_device(ctx);
  if (rc) return rc;
  hipGraph_t g = nullptr;
  PH_HIP(hipStreamEndCapture(stream_of(ctx, queue), &g));
  if (!g) return fail(PH_E_HIP, "ph_graph_end: nothing was recorded");
  hipGraphExec_t exec = nullptr;
  hipError_t e = hipGraphInstantiate(&exec, g, nullptr, nullptr, 0);
  if (e != hipSuccess) {
    hipGraphDestroy(g);
    return fail(PH_E_HIP, "ph_graph_end: hipGraphInstantiate: %s", hipGetErrorString(e));
  }
  *out = new ph_graph{ctx, g, exec};
  ctx_ref(ctx);
  return PH_OK;
}

int ph_graph_launch(ph_graph *g, int queue) {
  if (!g) return fail(PH_E_INVALID, "ph_graph_launch: NULL graph");
  PH_QUEUE("ph_graph_launch", queue);
  int rc = set_device(g->ctx);
  if (rc) return rc;
  PH_HIP(hipGraphLaunch(g->exec, stream_of(g->ctx, queue)));
  return PH_OK;
}

int ph_graph_destroy(ph_graph *g) {
  if (!g) return PH_OK;
  ph_ctx *ctx = g->ctx;
  hipGraphExecDestroy(g->exec);
  hipGraphDestroy(g->graph);
  delete g;
  ctx_unref(ctx);
  return PH_OK;
}

// ---- ROUTE: RCCL point-to-point on a communication stream of its own ----------------------------------
// RCCL is loaded with dlopen(RTLD_LOCAL): the library has no link-time dependency on it, and a host process
// that already carries another copy (PyTorch ships one) keeps the two apart.
struct Id128 {  // ncclUniqueId, passed to ncclCommInitRank by value
  char bytes[128];
};
namespace {
struct Rccl {
  void *handle = nullptr;
  int (*GetUniqueId)(void *) = nullptr;
  int (*CommInitRank)(void **, int, Id128, int) = nullptr;
  int (*CommDestroy)(void *) = nullptr;
  int (*Send)(const void *, size_t, int, int, void *, hipStream_t) = nullptr;
  int (*Recv)(void *, size_t, int, int, void *, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  int (*CommCount)(void *, int *) = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
};
Rccl g_rccl;
std::mutex g_rccl_mu;
const int kNcclUint32 = 3;  // ncclUint32 (rccl.h ncclDataType_t)

int load_rccl() {
  std::lock_guard<std::mutex> lock(g_rccl_mu);
  if (g_rccl.handle) return PH_OK;
  const char *names[] = {getenv("PH_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void *h = nullptr;
  for (const char *n : names)
    if (n && *n && (h = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
  if (!h) return fail(PH_E_HIP, "ROUTE needs RCCL and librccl.so could not be loaded (%s)", dlerror());
#define PH_SYM(field, name)                                                              \
  if (!(*(void **)(&g_rccl.field) = dlsym(h, name))) {                                    \
    dlclose(h);                                                                          \
    return fail(PH_E_HIP, "librccl.so lacks %s", name);                                  \
  }
  PH_SYM(GetUniqueId, "ncclGetUniqueId")
  PH_SYM(CommInitRank, "ncclCommInitRank")
  PH_SYM(CommDestroy, "ncclCommDestroy")
  PH_SYM(Send, "ncclSend")
  PH_SYM(Recv, "ncclRecv")
  PH_SYM(GroupStart, "ncclGroupStart")
  PH_SYM(GroupEnd, "ncclGroupEnd")
  PH_SYM(CommCount, "ncclCommCount")
  PH_SYM(GetErrorString, "ncclGetErrorString")
#undef PH_SYM
  g_rccl.handle = h;
  return PH_OK;
}
#define PH_NCCL(call)                                                                               \
  do {                                                                                              \
    int r_ = (call);                                                                                \
    if (r_ != 0) return fail(PH_E_HIP, "%s failed: %s", #call, g_rccl.GetErrorString(r_));          \
  } while (0)
}  // namespace

struct ph_route {
  ph_ctx *ctx;
  void *comm;
  hipStream_t stream;
  int rank, world;
  hipEvent_t last = nullptr;         // recorded behind the latest transfer enqueued on `stream`
  bool in_group = false;
  std::atomic<bool> busy{false};     // `last` may not have completed yet
};

extern "C++" {
namespace {
// Buffer lifetime against ROUTE (the source rank's `send(frame); frame.release()`): the pool hands a recycled block out
// only after the three queues have been ordered behind every transfer enqueued so far.  Device-side waits only, and
// only while a transfer is really in flight (hipEventQuery).  Called with ctx->mu held.
// `busy` and `last` change only under ctx->mu (route_mark takes it too): a pool thread that found the old record complete
// cannot clear the flag after another thread has recorded a new transfer.  A queue that is being captured
// (ph_graph_begin) is left alone - an event recorded outside the capture cannot be waited for inside it; if a wait cannot
// be enqueued the host waits for the transfer instead, so a recycled block is never handed out under RCCL's read.
void order_queues_after_routes(ph_ctx *ctx) {
  for (ph_route *r : ctx->routes) {
    if (!r->busy.load() || !r->last) continue;
    if (hipEventQuery(r->last) == hipSuccess) {
      r->busy.store(false);
      continue;
    }
    for (int q = 0; q < 3; ++q) {
      hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
      if (hipStreamIsCapturing(ctx->streams[q], &capturing) == hipSuccess && capturing != hipStreamCaptureStatusNone) continue;
      if (hipStreamWaitEvent(ctx->streams[q], r->last, 0) != hipSuccess) {
        (void)hipGetLastError();
        hipEventSynchronize(r->last);
        r->busy.store(false);
        break;
      }
    }
  }
}
int route_mark(ph_route *r) {  // a transfer has just been enqueued on the communication stream
  std::lock_guard<std::mutex> lock(r->ctx->mu);
  PH_HIP(hipEventRecord(r->last, r->stream));
  r->busy.store(true);
  return PH_OK;
}
}  // namespace
}  // extern "C++"

int ph_route_unique_id(void *id128) {
  if (!id128) return fail(PH_E_INVALID, "ph_route_unique_id: NULL argument");
  int rc = load_rccl();
  if (rc) return rc;
  static_assert(PH_ROUTE_ID_BYTES == sizeof(Id128), "ncclUniqueId is 128 bytes");
  PH_NCCL(g_rccl.GetUniqueId(id128));
  return PH_OK;
}

int ph_route_init(ph_ctx *ctx, const void *id128, int rank, int world, ph_route **out) {
  if (!ctx || !id128 || !out) return fail(PH_E_INVALID, "ph_route_init: NULL argument");
  if (world < 1 || rank < 0 || rank >= world) return fail(PH_E_INVALID, "ph_route_init: rank %d of %d", rank, world);
  int rc = set_device(ctx);
  if (rc) return rc;
  rc = load_rccl();
  if (rc) return rc;
  Id128 id;
  memcpy(id.bytes, id128, sizeof id.bytes);
  void *comm = nullptr;
  PH_NCCL(g_rccl.CommInitRank(&comm, world, id, rank));
  hipStream_t s = nullptr;
  hipError_t e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  if (e != hipSuccess) {
    g_rccl.CommDestroy(comm);
    return fail(PH_E_HIP, "ph_route_init: hipStreamCreate: %s", hipGetErrorString(e));
  }
  hipEvent_t ev = nullptr;
  e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
  if (e != hipSuccess) {
    g_rccl.CommDestroy(comm);
    hipStreamDestroy(s);
    return fail(PH_E_HIP, "ph_route_init: hipEventCreate: %s", hipGetErrorString(e));
  }
  ph_route *r = new ph_route{ctx, comm, s, rank, world};
  r->last = ev;
  {
    std::lock_guard<std::mutex> lock(ctx->mu);
    ctx->routes.push_back(r);
  }
  *out = r;
  ctx_ref(ctx);
  return PH_OK;
}

int ph_route_destroy(ph_route *r) {
  if (!r) return PH_OK;
  hipSetDevice(r->ctx->device);
  hipStreamSynchronize(r->stream);
  ph_ctx *ctx = r->ctx;
  {
    std::lock_guard<std::mutex> lock(ctx->mu);
    for (size_t i = 0; i < ctx->routes.size(); ++i)
      if (ctx->routes[i] == r) {
        ctx->routes.erase(ctx->routes.begin() + (long)i);
        break;
      }
  }
  g_rccl.CommDestroy(r->comm);
  hipStreamDestroy(r->stream);
  if (r->last) hipEventDestroy(r->last);
  delete r;
  ctx_unref(ctx);
  return PH_OK;
}

int ph_route_group_begin(ph_route *r) {
  if (!r) return fail(PH_E_INVALID, "ph_route_group_begin: NULL route");
  PH_NCCL(g_rccl.GroupStart());
  r->in_group = true;
  return PH_OK;
}
int ph_route_group_end(ph_route *r) {
  if (!r) return fail(PH_E_INVALID, "ph_route_group_end: NULL route");
  int rc = set_device(r->ctx);
  if (rc) return rc;
  PH_NCCL(g_rccl.GroupEnd());
  r->in_group = false;
  return route_mark(r);  // the group's transfers are on the stream now
}

static int route_args(ph_route *r, const void *p, size_t bytes, int peer, const char *fn) {
  if (!r || !p) return fail(PH_E_INVALID, "%s: NULL argument", fn);
  if (bytes % 4) return fail(PH_E_INVALID, "%s: %zu bytes is not a multiple of 4", fn, bytes);
  if (peer < 0 || peer >= r->world) return fail(PH_E_INVALID, "%s: peer %d of %d ranks", fn, peer, r->world);
  return set_device(r->ctx);
}
int ph_route_send(ph_route *r, const void *src, size_t bytes, int peer) {
  int rc = route_args(r, src, bytes, peer, "ph_route_send");
  if (rc) return rc;
  PH_NCCL(g_rccl.Send(src, bytes / 4, kNcclUint32, peer, r->comm, r->stream));
  return r->in_group ? PH_OK : route_mark(r);
}
int ph_route_recv(ph_route *r, void *dst, size_t bytes, int peer) {
  int rc = route_args(r, dst, bytes, peer, "ph_route_recv");
  if (rc) return rc;
  PH_NCCL(g_rccl.Recv(dst, bytes / 4, kNcclUint32, peer, r->comm, r->stream));
  return r->in_group ? PH_OK : route_mark(r);
}

static int order_streams(ph_ctx *ctx, hipStream_t waiter, hipStream_t signal, const char *fn) {
  int rc = set_device(ctx);
  if (rc) return rc;
  hipEvent_t ev;
  PH_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  hipError_t e = hipEventRecord(ev, signal);
  if (e == hipSuccess) e = hipStreamWaitEvent(waiter, ev, 0);
  hipEventDestroy(ev);
  if (e != hipSuccess) return fail(PH_E_HIP, "%s: %s", fn, hipGetErrorString(e));
  return PH_OK;
}
int ph_route_after_queue(ph_route *r, int queue) {
  if (!r || queue < 0 || queue > 2) return fail(PH_E_INVALID, "ph_route_after_queue: bad argument");
  return order_streams(r->ctx, r->stream, r->ctx->streams[queue], "ph_route_after_queue");
}
int ph_queue_after_route(ph_route *r, int queue) {
  if (!r || queue < 0 || queue > 2) return fail(PH_E_INVALID, "ph_queue_after_route: bad argument");
  return order_streams(r->ctx, r->ctx->streams[queue], r->stream, "ph_queue_after_route");
}
int ph_route_wait(ph_route *r) {
  if (!r) return fail(PH_E_INVALID, "ph_route_wait: NULL route");
  int rc = set_device(r->ctx);
  if (rc) return rc;
  PH_HIP(hipStreamSynchronize(r->stream));
  return PH_OK;
}
void *ph_route_stream(ph_route *r) { return r ? (void *)r->stream : nullptr; }
int ph_route_comm_count(ph_route *r, int *count) {
  if (!r || !count) return fail(PH_E_INVALID, "ph_route_comm_count: NULL argument");
  PH_NCCL(g_rccl.CommCount(r->comm, count));
  return PH_OK;
}

int ph_ctx_buffer_stats(ph_ctx *ctx, size_t *live_buffers, size_t *live_bytes, size_t *pooled_bytes) {
  if (!ctx) return fail(PH_E_INVALID, "ph_ctx_buffer_stats: ctx is NULL");
  std::lock_guard<std::mutex> lock(ctx->mu);
  if (live_buffers) *live_buffers = ctx->live_buffers;
  if (live_bytes) *live_bytes = ctx->live_bytes;
  if (pooled_bytes) *pooled_bytes = ctx->pooled_bytes;
  return PH_OK;
}

int ph_ctx_host_pool_stats(ph_ctx *ctx, size_t *in_use_bytes, size_t *pooled_bytes, size_t *peak_in_use_bytes, uint64_t *pins) {
  if (!ctx) return fail(PH_E_INVALID, "ph_ctx_host_pool_stats: ctx is NULL");
  std::lock_guard<std::mutex> lock(ctx->mu);
  if (in_use_bytes) *in_use_bytes = ctx->host_live_bytes;
  if (pooled_bytes) *pooled_bytes = ctx->host_pooled_bytes;
  if (peak_in_use_bytes) *peak_in_use_bytes = ctx->host_peak_bytes;
  if (pins) *pins = ctx->host_pins;
  return PH_OK;
}

// ---- gamma LUT registry ------------------------------------------------------------------------
int ph_lut_unregister(ph_ctx *ctx, const void *dev) {
  if (!ctx) return fail(PH_E_INVALID, "ph_lut_unregister: ctx is NULL");
  void *blob = nullptr;
  {
    std::lock_guard<std::mutex> lock(ctx->mu);
    auto it = ctx->luts.find(dev);
    if (it == ctx->luts.end()) return PH_OK;
    blob = it->second.blob_dev;
    ctx->luts.erase(it);
  }
  if (blob) {
    // kernels already enqueued may still read the blob: drain before freeing
    for (int q = 0; q < 3; ++q) hipStreamSynchronize(ctx->streams[q]);
    hipFree(blob);
  }
  return PH_OK;
}

int ph_lut_register(ph_ctx *ctx, const void *dev, const float *host) {
  if (!ctx || !dev || !host) return fail(PH_E_INVALID, "ph_lut_register: NULL argument");
  int rc = set_device(ctx);
  if (rc) return rc;
  ph_lut_unregister(ctx, dev);
  std::vector<uint32_t> blob;
  ph::LutHostInfo info;
  LutEntry e;
  if (ph::lut_compress(host, ph::kLutMaxLdsBytes, blob, info)) {
    if (!ctx->lds_base_checked.load()) {  // the lookups address the table absolutely: the kernels' dynamic shared array must start at LDS 0
      uint32_t *probe = nullptr, base = ~0u;
      PH_HIP(hipMalloc(&probe, 4));
      hipError_t pe = ph::launch_lds_base_probe(ctx->streams[0], probe);
      if (pe == hipSuccess) pe = hipStreamSynchronize(ctx->streams[0]);
      if (pe == hipSuccess) pe = hipMemcpy(&base, probe, 4, hipMemcpyDeviceToHost);
      hipFree(probe);
      if (pe != hipSuccess) return fail(PH_E_HIP, "ph_lut_register: LDS base probe: %s", hipGetErrorString(pe));
      if (base != 0) return fail(PH_E_HIP, "ph_lut_register: dynamic shared memory starts at LDS address %u, not 0: the table kernels cannot run", base);
      ctx->lds_base_checked.store(true);
    }
    const size_t blob_bytes = blob.size() * sizeof(uint32_t);  // info.bytes is the LDS footprint: the hole in front of the blob included
    PH_HIP(hipMalloc(&e.blob_dev, blob_bytes));
    PH_HIP(hipMemcpy(e.blob_dev, blob.data(), blob_bytes, hipMemcpyHostToDevice));
    e.view = ph::lut_view(info, e.blob_dev);
  }
  {
    std::lock_guard<std::mutex> lock(ctx->mu);
    ctx->luts[dev] = e;
  }
  return e.view.bytes ? 1 : 0;
}

int ph_lut_layout_of(const float *host, ph_lut_layout *layout, void *lds_image, size_t capacity) {
  if (!host || !layout) return fail(PH_E_INVALID, "ph_lut_layout_of: NULL argument");
  std::vector<uint32_t> blob;
  ph::LutHostInfo info;
  memset(layout, 0, sizeof(*layout));
  if (!ph::lut_compress(host, ph::kLutMaxLdsBytes, blob, info)) return 0;
  const ph::LutView v = ph::lut_view(info, nullptr);
  layout->lds_bytes = info.bytes, layout->hole = info.hole, layout->delta_off = info.delta_off, layout->shift = info.shift;
  layout->index_bias = (uint32_t)info.bias, layout->a_scale = v.a_scale;
  if (lds_image) {
    if (capacity < info.bytes) return fail(PH_E_INVALID, "ph_lut_layout_of: the image needs %u bytes, %zu given", info.bytes, capacity);
    memset(lds_image, 0, info.hole);
    memcpy(static_cast<char *>(lds_image) + info.hole, blob.data(), info.bytes - info.hole);
  }
  return (int)info.bytes;
}

int ph_lut_query(ph_ctx *ctx, const void *dev, uint32_t *lds_bytes, uint32_t *toe, uint32_t *shift) {
  if (!ctx) return fail(PH_E_INVALID, "ph_lut_query: ctx is NULL");
  std::lock_guard<std::mutex> lock(ctx->mu);
  auto it = ctx->luts.find(dev);
  const bool have = it != ctx->luts.end();
  if (lds_bytes) *lds_bytes = have ? it->second.view.bytes : 0;
  if (toe) *toe = have ? (uint32_t)it->second.view.bias : 0;
  if (shift) *shift = have ? 23 - it->second.view.shift : 0;
  return PH_OK;
}

int ph_ctx_set_option(ph_ctx *ctx, const char *name, int value) {
  if (!ctx || !name) return fail(PH_E_INVALID, "ph_ctx_set_option: NULL argument");
  if (0 == strcmp(name, "lds_lut")) return ctx->use_lds_lut = (value != 0), PH_OK;
  if (0 == strcmp(name, "stream_images")) {
    if (value < 0 || value > 2) return fail(PH_E_INVALID, "stream_images: 0 (cached), 1 (streamed) or 2 (by size)");
    return ctx->stream_images = value, PH_OK;
  }
  if (0 == strcmp(name, "stream_threshold_mb")) {
    if (value < 0) return fail(PH_E_INVALID, "stream_threshold_mb: a size in MiB");
    return ctx->stream_threshold_mb = value, PH_OK;
  }
  if (0 == strcmp(name, "fail_launches")) return ctx->fail_launches.store(value), PH_OK;
  if (0 == strcmp(name, "chan_enlarged")) return ctx->chan_enlarged = (value < 0 || value > 2 ? 1 : value), PH_OK;  // 2: never the one-launch form (A/B, tests)
  if (0 == strcmp(name, "host_pool_mb")) {
    if (value < 0) return fail(PH_E_INVALID, "host_pool_mb: a size in MiB");
    std::vector<ph_ctx::HostBlock> drop;
    {
      std::lock_guard<std::mutex> lock(ctx->mu);
      ctx->host_pool_mb = value;
      ctx->host_peak_bytes = ctx->host_live_bytes;  // the working set is measured afresh from here (a burst long ago no longer sizes the pool)
      while (ctx->host_pooled_bytes > (size_t)value << 20 && !ctx->host_pool.empty()) {
        auto it = std::prev(ctx->host_pool.end());
        ctx->host_pooled_bytes -= it->first;
        drop.push_back(it->second);
        ctx->host_pool.erase(it);
      }
    }
    for (auto &v : drop) {
      hipHostFree(v.p);
      if (v.busy) hipEventDestroy(v.busy);
    }
    return PH_OK;
  }
  return fail(PH_E_INVALID, "unknown option '%s'", name);
}

// The compressed form of a device LUT pointer (unknown / plain / LDS path switched off: none).  A COPY, by value: the registry entry
// may be replaced by another thread once the lock is dropped, and a caller may hold its copy across any number of further look-ups
// (until round 6 this handed out slots of a 16-entry per-thread ring; a later launch once got the writer's table as its reader's).
struct LutRef {
  ph::LutView view{};
  bool found = false;
  const ph::LutView *get() const { return found ? &view : nullptr; }
};
static LutRef lds_view(ph_ctx *ctx, const void *dev) {
  LutRef r;
  if (!ctx || !ctx->use_lds_lut) return r;
  std::lock_guard<std::mutex> lock(ctx->mu);
  auto it = ctx->luts.find(dev);
  if (it == ctx->luts.end() || !it->second.view.bytes) return r;
  r.view = it->second.view, r.found = true;
  return r;
}

// a ph_buf used as `gammaLut`: (re)compress from its host mirror if new data went in
static void refresh_buf_lut(ph_ctx *ctx, ph_buf *b) {
  if (b->lut_dirty && b->hptr && b->bytes >= 65536 * 4) {
    hipStreamSynchronize(ctx->streams[PH_QUEUE_LOAD]);  // the mirror must be stable
    ph_lut_register(ctx, b->dptr, (const float *)b->hptr);
    b->lut_dirty = false;
  }
}

// ---- programs ---------------------------------------------------------------------------------
int ph_program_resolve(const char *src, const char *name, char *kernel_id, size_t kernel_id_len, int *format, int *how) {
  ph::ProgramChoice c;
  std::string err;
  const int rc = ph::resolve_program(src, name, c, err);
  if (rc != PH_OK) return fail(rc, "%s", err.c_str());
  if (kernel_id && kernel_id_len) snprintf(kernel_id, kernel_id_len, "%s", c.kernel.c_str());
  if (format) *format = (c.id <= K_V210_WRITE) ? c.format : -1;
  if (how) *how = c.how;
  return PH_OK;
}

int ph_program_create(ph_ctx *ctx, const char *src, const char *name, const uint32_t *gwi, int n_dims, uint32_t wipg,
                      ph_program **out) {
  if (!ctx || !name || !out) return fail(PH_E_INVALID, "ph_program_create: NULL argument");
  ph::ProgramChoice c;
  std::string err;
  const int rc = ph::resolve_program(src, name, c, err);
  if (rc != PH_OK) return fail(rc, "%s", err.c_str());
  ph_program p{ctx, c.id, c.n_layers, c.format, c.kernel, {0, 0}, wipg};
  for (int i = 0; i < n_dims && i < 2; ++i) p.global[i] = gwi ? gwi[i] : 0;
  if (ctx->closed.load()) return closed_error("ph_program_create");
  *out = new ph_program(p);
  ctx_ref(ctx);
  return PH_OK;
}

int ph_program_destroy(ph_program *p) {
  if (!p) return PH_OK;
  ph_ctx *ctx = p->ctx;
  delete p;
  ctx_unref(ctx);
  return PH_OK;
}

const char *ph_program_kernel(const ph_program *p) { return p ? p->kernel.c_str() : ""; }

// nodencl lets a caller map a buffer for writing (hostAccess('writeonly')), fill it and launch
// without an explicit unmap (loadSave.ts:76-99): flush such mirrors on the launch queue first.
static int flush_dirty_args(ph_ctx *ctx, const ph_arg *args, int n, int queue) {
  for (int i = 0; i < n; ++i) {
    if (args[i].kind != PH_ARG_BUF || !args[i].v.buf) continue;
    ph_buf *b = args[i].v.buf;
    if (b->host_dirty && b->hptr) {
      hipStream_t s = stream_of(ctx, queue);
      PH_HIP(hipMemcpyAsync(b->dptr, b->hptr, b->bytes, hipMemcpyHostToDevice, s));
      mirror_mark(b, s);
      b->host_dirty = false;
      b->lut_dirty = b->lut_dirty || b->bytes >= 65536 * 4;
    }
  }
  return PH_OK;
}

// A channel's frame as the by-name program chan_compose_v210_<n> describes it (dispatch K_CHAN_COMPOSE; ph_run_programs puts several
// such calls into one launch): the arguments checked and turned into ph_chan_compose's own.
struct ChanCall {
  ph_chan_layer layers[ph::kMaxLayers];
  int n_layers, out_format;
  uint32_t width, height, interlace;
  void *out_planes[3];
  ph_buf *rd_cm, *rd_lut, *rd_gm, *wr_cm, *wr_lut;
};
static int chan_call_parse(ph_ctx *ctx, ph_program *prog, const ph_arg *args, int n, bool check_only, ChanCall *call) {
  ph_buf *b = nullptr, *c = nullptr, *d = nullptr, *o = nullptr;
  int rc;
#define TRY(x) \
  if ((rc = (x)) != PH_OK) return rc
  // l<i>In: a layer's source - a v210 frame (l<i>Width / l<i>Height: its size, default the output's) or an RGBA image buffer;
  // l<i>Matrix (optional): its placement, a buffer whose host mirror holds the nine floats (Transform writes it through
  // hostAccess: transform.ts:84-89), absent = 1:1; l<i>Transition: 0 cut / 1 dissolve / 2 wipe; l<i>Mix; l<i>Incoming(In|Matrix|
  // Width|Height) and l<i>Mask(...): the transition's other sources; output: v210; colMatrix / gammaLut / gamutMatrix: the
  // Loader's, outColMatrix / outGammaLut: the Saver's; interlace as 'write'
  {
      const uint32_t width = prog->global[0], height = prog->global[1];
      if (!width || !height) return fail(PH_E_INVALID, "%s: globalWorkItems must be [width, height]", prog->kernel.c_str());
      ph_chan_layer layers[ph::kMaxLayers];
      memset(layers, 0, sizeof layers);
      auto source = [&](int i, const char *role, ph_chan_source *s) -> int {
        char nm[40];
        ph_buf *x = nullptr;
        snprintf(nm, sizeof nm, "l%d%sIn", i, role);
        TRY(need_buf(args, n, nm, 0, &x));
        double sw = width, sh = height;
        s->data = x->dptr;
        if (x->width > 0 && x->height > 0) {  // an image buffer (createBuffer with imageDims): f32 RGBA
          s->format = PH_SRC_RGBA_F32, sw = x->width, sh = x->height;
        } else {
          s->format = PH_SRC_V210;
          s->data_u = s->data_v = nullptr, s->col_matrix12 = nullptr;
          // another wire format: l<i>Packing = its PH_FMT_* (1 yuv422p10, 2 yuv422p8, 3 yuv420p, 4 nv12: l<i>In the Y plane, l<i>InU /
          // l<i>InV the chroma planes (nv12: l<i>InU the CbCr plane), l<i>ColMatrix (optional) its own Loader matrix; 5 rgba8, 6 bgra8: l<i>In the frame)
          double packing = 0;
          snprintf(nm, sizeof nm, "l%d%sPacking", i, role);
          if (find_arg(args, n, nm)) TRY(need_num(args, n, nm, &packing));
          if (packing != 0) {
            if (packing < PH_FMT_YUV422P10 || packing > PH_FMT_BGRA8) return fail(PH_E_INVALID, "kernel argument '%s': %g is not a pack format other than v210", nm, packing);
            s->format = PH_SRC_YUV422P10 + ((int)packing - PH_FMT_YUV422P10);  // PH_SRC_* follow PH_FMT_* from here on
          }
          snprintf(nm, sizeof nm, "l%d%sWidth", i, role);
          if (find_arg(args, n, nm)) TRY(need_num(args, n, nm, &sw));
          snprintf(nm, sizeof nm, "l%d%sHeight", i, role);
          if (find_arg(args, n, nm)) TRY(need_num(args, n, nm, &sh));
          if (s->format == PH_SRC_RGBA8 || s->format == PH_SRC_BGRA8) {
            if (sw > 0 && sh > 0 && x->bytes < (size_t)sw * (size_t)sh * 4) return fail(PH_E_RANGE, "kernel argument 'l%d%sIn': buffer of %zu bytes is smaller than a %gx%g frame of 4 bytes per pixel", i, role, x->bytes, sw, sh);
          } else if (s->format != PH_SRC_V210) {
            const int fmt = PH_FMT_YUV422P10 + (s->format - PH_SRC_YUV422P10);
            size_t pb[3] = {0, 0, 0};
            if (sw > 0 && sh > 0) ph_pack_plane_bytes(fmt, (uint32_t)sw, (uint32_t)sh, pb);
            char nu[40];
            ph_buf *pu = nullptr, *pv = nullptr, *pm = nullptr;
            snprintf(nu, sizeof nu, "l%d%sInU", i, role);
            TRY(need_buf(args, n, nu, pb[1], &pu));
            s->data_u = pu->dptr;
            if (fmt != PH_FMT_NV12) {
              snprintf(nu, sizeof nu, "l%d%sInV", i, role);
              TRY(need_buf(args, n, nu, pb[2], &pv));
              s->data_v = pv->dptr;
            }
            snprintf(nu, sizeof nu, "l%d%sColMatrix", i, role);
            if (find_arg(args, n, nu)) {
              TRY(need_buf(args, n, nu, 48, &pm));
              s->col_matrix12 = pm->dptr;
            }
            if (x->bytes < pb[0]) return fail(PH_E_RANGE, "kernel argument 'l%d%sIn': buffer of %zu bytes is smaller than the Y plane of a %gx%g frame", i, role, x->bytes, sw, sh);
          } else if (sw > 0 && sh > 0 && x->bytes < (size_t)ph_v210_pitch_bytes((uint32_t)sw) * (size_t)sh)
            return fail(PH_E_RANGE, "kernel argument 'l%d%sIn': buffer of %zu bytes is smaller than a %gx%g v210 frame", i, role, x->bytes, sw, sh);
        }
        s->width = (int)sw, s->height = (int)sh, s->matrix9_host = nullptr;
        snprintf(nm, sizeof nm, "l%d%sMatrix", i, role);
        if (find_arg(args, n, nm)) {
          ph_buf *m = nullptr;
          TRY(need_buf(args, n, nm, 36, &m));
          if (!m->hptr) return fail(PH_E_INVALID, "kernel argument '%s': the matrix must have been written through hostAccess (its host copy is what the launch reads)", nm);
          s->matrix9_host = (const float *)m->hptr;
        }
        return PH_OK;
      };
      for (int i = 0; i < prog->n_layers; ++i) {
        char nm[40];
        TRY(source(i, "", &layers[i].src));
        double tr = 0, mix = 0;
        snprintf(nm, sizeof nm, "l%dTransition", i);
        if (find_arg(args, n, nm)) TRY(need_num(args, n, nm, &tr));
        snprintf(nm, sizeof nm, "l%dMix", i);
        if (find_arg(args, n, nm)) TRY(need_num(args, n, nm, &mix));
        layers[i].transition = (int)tr, layers[i].mix = (float)mix;
        if (layers[i].transition != PH_TRANSITION_CUT) TRY(source(i, "Incoming", &layers[i].incoming));
        if (layers[i].transition == PH_TRANSITION_WIPE) TRY(source(i, "Mask", &layers[i].mask));
      }
      // output: the packed frame - v210, or with outPacking = PH_FMT_* another wire format: 1 yuv422p10 / 2 yuv422p8 / 3 yuv420p (output =
      // the Y plane, outputU, outputV), 4 nv12 (output, outputC), 5 rgba8 / 6 bgra8 (no outColMatrix)
      double interlace = 0, out_packing = 0;
      ph_buf *wcm = nullptr, *wl = nullptr, *ou = nullptr, *ov = nullptr;
      if (find_arg(args, n, "outPacking")) TRY(need_num(args, n, "outPacking", &out_packing));
      const int ofmt = (int)out_packing;
      size_t opb[3] = {0, 0, 0};
      if (ph_pack_plane_bytes(ofmt, width, height, opb) < 0) return fail(PH_E_INVALID, "kernel argument 'outPacking': %g is not a pack format", out_packing);
      TRY(need_buf(args, n, "output", opb[0], &o));
      if (ofmt == PH_FMT_YUV422P10 || ofmt == PH_FMT_YUV422P8 || ofmt == PH_FMT_YUV420P) {
        TRY(need_buf(args, n, "outputU", opb[1], &ou));
        TRY(need_buf(args, n, "outputV", opb[2], &ov));
      } else if (ofmt == PH_FMT_NV12) {  // nv12.ts:374: the interleaved CbCr plane is `outputC`
        TRY(need_buf(args, n, "outputC", opb[1], &ou));
      }
      TRY(need_buf(args, n, "colMatrix", 48, &b));
      TRY(need_buf(args, n, "gammaLut", 65536 * 4, &c));
      TRY(need_buf(args, n, "gamutMatrix", 36, &d));
      if (ofmt != PH_FMT_RGBA8 && ofmt != PH_FMT_BGRA8) TRY(need_buf(args, n, "outColMatrix", 48, &wcm));
      TRY(need_buf(args, n, "outGammaLut", 65536 * 4, &wl));
      if (find_arg(args, n, "interlace")) TRY(need_num(args, n, "interlace", &interlace));
      if (!check_only) refresh_buf_lut(ctx, c);
      if (!check_only) refresh_buf_lut(ctx, wl);
      ph_chan_layer *dst = call->layers;
      memcpy(dst, layers, sizeof layers);
      call->n_layers = prog->n_layers, call->out_format = ofmt, call->width = width, call->height = height, call->interlace = (uint32_t)interlace;
      call->out_planes[0] = o->dptr, call->out_planes[1] = ou ? ou->dptr : nullptr, call->out_planes[2] = ov ? ov->dptr : nullptr;
      call->rd_cm = b, call->rd_lut = c, call->rd_gm = d, call->wr_cm = wcm, call->wr_lut = wl;
  }
#undef TRY
  return PH_OK;
}

// check_only: everything up to the launch - argument names, kinds, buffer sizes, geometry - and nothing on the device
// (ph_check_program: a recording binding reports a bad job where it is posted, not where it is run)
// fused_v210_combine_<n> (dispatch K_FUSED_V210; ph_run_programs puts several such calls of one shape into one launch):
// l<i>In: v210 sources; colMatrix / gammaLut / gamutMatrix: the Loader's; outColMatrix / outGammaLut: the Saver's
struct FusedCall {
  const void *layers[ph::kMaxLayers];
  int n;
  void *out;
  uint32_t width, height;
  size_t frame_bytes;
  ph_buf *rd_cm, *rd_lut, *rd_gm, *wr_cm, *wr_lut;
};
static int fused_call_parse(ph_ctx *ctx, ph_program *prog, const ph_arg *args, int n, bool check_only, FusedCall *call) {
  int rc;
#define TRY(x) \
  if ((rc = (x)) != PH_OK) return rc
  call->width = prog->global[0], call->height = prog->global[1], call->n = prog->n_layers;
  if (!call->width || !call->height) return fail(PH_E_INVALID, "%s: globalWorkItems must be [width, height]", prog->kernel.c_str());
  call->frame_bytes = (size_t)ph_v210_pitch_bytes(call->width) * call->height;
  ph_buf *a = nullptr, *o = nullptr;
  for (int i = 0; i < prog->n_layers; ++i) {
    char nm[16];
    snprintf(nm, sizeof nm, "l%dIn", i);
    TRY(need_buf(args, n, nm, call->frame_bytes, &a));
    call->layers[i] = a->dptr;
  }
  TRY(need_buf(args, n, "output", call->frame_bytes, &o));
  call->out = o->dptr;
  TRY(need_buf(args, n, "colMatrix", 48, &call->rd_cm));
  TRY(need_buf(args, n, "gammaLut", 65536 * 4, &call->rd_lut));
  TRY(need_buf(args, n, "gamutMatrix", 36, &call->rd_gm));
  TRY(need_buf(args, n, "outColMatrix", 48, &call->wr_cm));
  TRY(need_buf(args, n, "outGammaLut", 65536 * 4, &call->wr_lut));
  if (!check_only) refresh_buf_lut(ctx, call->rd_lut);
  if (!check_only) refresh_buf_lut(ctx, call->wr_lut);
  return PH_OK;
#undef TRY
}

// a compose_up_write_v210_<n> job's arguments (ph_run_program and ph_run_programs, which puts like jobs into one launch)
struct UpCall {
  int n;
  bool rgb, pair;
  ph_image_layer layers[ph::kMaxLayers], layers2[ph::kMaxLayers];
  ph_buf *o, *o2, *wcm, *wl;
  uint32_t width, height, interlace;
};
static int up_call_parse(ph_ctx *ctx, ph_program *prog, const ph_arg *args, int n, bool check_only, UpCall *u) {
  int rc;
#define TRY(x) \
  if ((rc = (x)) != PH_OK) return rc
  // l<i>In: the layer's image - an RGBA image buffer, or with packedRgb = 1 a buffer of packed f32 RGB (l<i>Width / l<i>Height:
  // its size); l<i>Matrix: its placement (host mirror, as above); output: v210; outColMatrix / outGammaLut; interlace
  const uint32_t width = prog->global[0], height = prog->global[1];
  if (!width || !height) return fail(PH_E_INVALID, "%s: globalWorkItems must be [width, height]", prog->kernel.c_str());
  double rgb = 0, interlace = 0;
  if (find_arg(args, n, "packedRgb")) TRY(need_num(args, n, "packedRgb", &rgb));
  // output2 + l<i>In2 (optional): a second job of the same shape in the same launch - the other field of a de-interlaced frame
  // (ph_compose_up_write_v210_pair): same sizes, formats and placements, other data
  const bool pair = find_arg(args, n, "output2") != nullptr;
  for (int i = 0; i < prog->n_layers; ++i) {
    char nm[24];
    ph_buf *x = nullptr, *m = nullptr, *x2 = nullptr;
    double lw = 0, lh = 0;
    snprintf(nm, sizeof nm, "l%dIn", i);
    TRY(need_buf(args, n, nm, 0, &x));
    if (pair) {
      snprintf(nm, sizeof nm, "l%dIn2", i);
      TRY(need_buf(args, n, nm, x->bytes, &x2));
      if (rgb == 0) {
        int w2, h2, w1, h1;
        TRY(need_image(x2, nm, &w2, &h2));
        TRY(need_image(x, nm, &w1, &h1));
        if (w1 != w2 || h1 != h2) return fail(PH_E_INVALID, "kernel argument '%s': the second job's image is %dx%d, the first's %dx%d", nm, w2, h2, w1, h1);
      }
      snprintf(nm, sizeof nm, "l%dIn", i);
    }
    if (rgb != 0) {
      snprintf(nm, sizeof nm, "l%dWidth", i);
      TRY(need_num(args, n, nm, &lw));
      snprintf(nm, sizeof nm, "l%dHeight", i);
      TRY(need_num(args, n, nm, &lh));
      if (lw <= 0 || lh <= 0 || x->bytes < (size_t)lw * (size_t)lh * 12) return fail(PH_E_RANGE, "kernel argument 'l%dIn': smaller than its %gx%g packed-RGB image", i, lw, lh);
    } else {
      int iw, ih;
      TRY(need_image(x, nm, &iw, &ih));
      lw = iw, lh = ih;
    }
    snprintf(nm, sizeof nm, "l%dMatrix", i);
    TRY(need_buf(args, n, nm, 36, &m));
    if (!m->hptr) return fail(PH_E_INVALID, "kernel argument '%s': the matrix must have been written through hostAccess (its host copy is what the launch reads)", nm);
    u->layers[i].data = x->dptr, u->layers[i].format = rgb != 0 ? PH_IMG_RGB_F32 : PH_IMG_RGBA_F32;
    u->layers[i].width = (int)lw, u->layers[i].height = (int)lh, u->layers[i].matrix9_host = (const float *)m->hptr;
    u->layers2[i] = u->layers[i];
    if (pair) u->layers2[i].data = x2->dptr;
  }
  u->o = u->o2 = u->wcm = u->wl = nullptr;
  TRY(need_buf(args, n, "output", (size_t)ph_v210_pitch_bytes(width) * height, &u->o));
  if (pair) TRY(need_buf(args, n, "output2", (size_t)ph_v210_pitch_bytes(width) * height, &u->o2));
  TRY(need_buf(args, n, "outColMatrix", 48, &u->wcm));
  TRY(need_buf(args, n, "outGammaLut", 65536 * 4, &u->wl));
  if (find_arg(args, n, "interlace")) TRY(need_num(args, n, "interlace", &interlace));
  if (!check_only) refresh_buf_lut(ctx, u->wl);
  u->n = prog->n_layers, u->rgb = rgb != 0, u->pair = pair, u->width = width, u->height = height, u->interlace = (uint32_t)interlace;
#undef TRY
  return PH_OK;
}

// the "fail_launches" fault injection (tests of a binding's error paths: node/test/soak_run.js, tests/test_boundary_gpu.py): does this launch fail?
static bool inject_failure(ph_ctx *ctx) {
  const int v = ctx->fail_launches.load();
  if (v > 0) return true;
  if (v < 0 && ctx->fail_launches.fetch_add(1) == -1) ctx->fail_launches.store(1);  // the last one that goes through
  return false;
}

// ph_run_program's argument marshalling, one function per kernel family: each checks the job's named arguments against the frame geometry
// (everything ph_check_program reports) and, unless check_only, makes the typed call.
#define TRY(x) \
  if ((rc = (x)) != PH_OK) return rc
// the wire formats: v210 / pack readers and writers (v210.ts, yuv422p10.ts ... bgra8.ts; Reader / Writer geometry packer.ts:30-83)
static int dispatch_wire(ph_ctx *ctx, ph_program *prog, const ph_arg *args, int n, int queue, bool check_only) {
  ph_buf *a = nullptr, *b = nullptr, *c = nullptr, *d = nullptr, *o = nullptr;
  double num = 0;
  int rc, w, h;
  (void)a, (void)b, (void)c, (void)d, (void)o, (void)num, (void)w, (void)h;
  switch (prog->id) {
    case K_PACK_READ:
    case K_PACK_WRITE: {
      const bool rd = prog->id == K_PACK_READ;
      const int fmt = prog->format;
      const bool rgb = fmt >= PH_FMT_RGBA8, v420 = (fmt == PH_FMT_YUV420P || fmt == PH_FMT_NV12);
      double il = 0;
      TRY(need_num(args, n, "width", &num));
      if (!rd) TRY(need_num(args, n, "interlace", &il));
      const uint32_t width = (uint32_t)num, interlace = (uint32_t)il;
      if (!prog->local || !width) return fail(PH_E_INVALID, "%s: width / workItemsPerGroup not set", prog->kernel.c_str());
      // Readers: global = wipg*height (4:2:0: /2).  Writers: /2 when interlaced, 4:2:0 always /2
      uint32_t height = prog->global[0] / prog->local;
      if (v420) height *= 2;
      else if (!rd && interlace) height *= 2;
      size_t pb[3];
      const int np = ph_pack_plane_bytes(fmt, width, height, pb);
      static const char *in_names[3][3] = {{"input", "", ""}, {"inputY", "inputC", ""}, {"inputY", "inputU", "inputV"}};
      static const char *out_names[3][3] = {{"output", "", ""}, {"outputY", "outputC", ""}, {"outputY", "outputU", "outputV"}};
      const void *planes[3] = {nullptr, nullptr, nullptr};
      ph_buf *pl = nullptr;
      for (int i = 0; i < np; ++i) {
        TRY(need_buf(args, n, (rd ? in_names : out_names)[np - 1][i], pb[i], &pl));
        planes[i] = pl->dptr;
      }
      TRY(need_buf(args, n, rd ? "output" : "input", (size_t)width * height * 16, &o));
      if (!rgb) TRY(need_buf(args, n, "colMatrix", 48, &b));
      TRY(need_buf(args, n, "gammaLut", 65536 * 4, &c));
      if (!check_only) refresh_buf_lut(ctx, c);
      if (rd) {
        TRY(need_buf(args, n, "gamutMatrix", 36, &d));
        return check_only ? PH_OK : ph_pack_read(ctx, queue, fmt, planes, o->dptr, width, height, rgb ? nullptr : b->dptr, c->dptr, d->dptr);
      }
      return check_only ? PH_OK : ph_pack_write(ctx, queue, fmt, o->dptr, const_cast<void *const *>(planes), width, height, interlace,
                           rgb ? nullptr : b->dptr, c->dptr);
    }
    case K_V210_READ: {
      TRY(need_num(args, n, "width", &num));
      const uint32_t width = (uint32_t)num;
      if (!prog->local || !width) return fail(PH_E_INVALID, "v210 read: width / workItemsPerGroup not set");
      const uint32_t height = prog->global[0] / prog->local;  // Reader: global = wipg * height (v210.ts:293-294)
      TRY(need_buf(args, n, "input", (size_t)ph_v210_pitch_bytes(width) * height, &a));
      TRY(need_buf(args, n, "output", (size_t)width * height * 16, &o));
      TRY(need_buf(args, n, "colMatrix", 48, &b));
      TRY(need_buf(args, n, "gammaLut", 65536 * 4, &c));
      TRY(need_buf(args, n, "gamutMatrix", 36, &d));
      if (!check_only) refresh_buf_lut(ctx, c);
      return check_only ? PH_OK : ph_v210_read(ctx, queue, a->dptr, o->dptr, width, height, b->dptr, c->dptr, d->dptr);
    }
    case K_V210_WRITE: {
      double il = 0;
      TRY(need_num(args, n, "width", &num));
      TRY(need_num(args, n, "interlace", &il));
      const uint32_t width = (uint32_t)num, interlace = (uint32_t)il;
      if (!prog->local || !width) return fail(PH_E_INVALID, "v210 write: width / workItemsPerGroup not set");
      // Writer: global = wipg * height / (interlaced ? 2 : 1) (v210.ts:322-323)
      const uint32_t height = prog->global[0] / prog->local * (interlace ? 2 : 1);
      TRY(need_buf(args, n, "input", (size_t)width * height * 16, &a));
      TRY(need_buf(args, n, "output", (size_t)ph_v210_pitch_bytes(width) * height, &o));
      TRY(need_buf(args, n, "colMatrix", 48, &b));
      TRY(need_buf(args, n, "gammaLut", 65536 * 4, &c));
      if (!check_only) refresh_buf_lut(ctx, c);
      return check_only ? PH_OK : ph_v210_write(ctx, queue, a->dptr, o->dptr, width, height, interlace, b->dptr, c->dptr);
    }
    case K_V210_READ_BATCH: {  // l<i>In: v210 frames; l<i>Out: RGBA images; colMatrix / gammaLut / gamutMatrix: the Loader's
      const uint32_t width = prog->global[0], height = prog->global[1];
      if (!width || !height) return fail(PH_E_INVALID, "%s: globalWorkItems must be [width, height]", prog->kernel.c_str());
      const size_t vb = (size_t)ph_v210_pitch_bytes(width) * height, img = (size_t)width * height * 16;
      const void *ins[ph::kMaxLayers];
      void *outs[ph::kMaxLayers];
      for (int i = 0; i < prog->n_layers; ++i) {
        char nm[16];
        ph_buf *x = nullptr;
        snprintf(nm, sizeof nm, "l%dIn", i);
        TRY(need_buf(args, n, nm, vb, &x));
        ins[i] = x->dptr;
        snprintf(nm, sizeof nm, "l%dOut", i);
        TRY(need_buf(args, n, nm, img, &x));
        outs[i] = x->dptr;
      }
      TRY(need_buf(args, n, "colMatrix", 48, &b));
      TRY(need_buf(args, n, "gammaLut", 65536 * 4, &c));
      TRY(need_buf(args, n, "gamutMatrix", 36, &d));
      if (!check_only) refresh_buf_lut(ctx, c);
      return check_only ? PH_OK : ph_v210_read_batch(ctx, queue, prog->n_layers, ins, outs, width, height, b->dptr, c->dptr, d->dptr);
    }
    default: break;
  }
  return fail(PH_E_UNKNOWN_KERNEL, "unhandled kernel id");
}
// de-interlacing: yadif, both parities in one launch, and the reader fused with it (yadifCl.ts, yadif.ts:88-145)
static int dispatch_deint(ph_ctx *ctx, ph_program *prog, const ph_arg *args, int n, int queue, bool check_only) {
  ph_buf *a = nullptr, *b = nullptr, *c = nullptr, *d = nullptr, *o = nullptr;
  double num = 0;
  int rc, w, h;
  (void)a, (void)b, (void)c, (void)d, (void)o, (void)num, (void)w, (void)h;
  switch (prog->id) {
    case K_YADIF: {
      double parity, tff, skip;
      TRY(need_buf(args, n, "output", 0, &o));
      TRY(need_image(o, "output", &w, &h));
      const size_t img = (size_t)w * h * 16;
      TRY(need_buf(args, n, "prev", img, &a));
      TRY(need_buf(args, n, "cur", img, &b));
      TRY(need_buf(args, n, "next", img, &c));
      TRY(need_num(args, n, "parity", &parity));
      TRY(need_num(args, n, "tff", &tff));
      TRY(need_num(args, n, "skipSpatial", &skip));
      return check_only ? PH_OK : ph_yadif(ctx, queue, a->dptr, b->dptr, c->dptr, w, h, (int)parity, (int)tff, (int)skip, o->dptr);
    }
    case K_YADIF_PAIR: {  // output0 / output1: what 'yadif' writes with parity 0 / 1
      double tff, skip;
      ph_buf *o1 = nullptr;
      TRY(need_buf(args, n, "output0", 0, &o));
      TRY(need_image(o, "output0", &w, &h));
      const size_t img = (size_t)w * h * 16;
      TRY(need_buf(args, n, "output1", img, &o1));
      TRY(need_buf(args, n, "prev", img, &a));
      TRY(need_buf(args, n, "cur", img, &b));
      TRY(need_buf(args, n, "next", img, &c));
      TRY(need_num(args, n, "tff", &tff));
      TRY(need_num(args, n, "skipSpatial", &skip));
      return check_only ? PH_OK : ph_yadif_pair(ctx, queue, a->dptr, b->dptr, c->dptr, w, h, (int)tff, (int)skip, o->dptr, o1->dptr);
    }
    case K_V210_YADIF_PAIR: {
      // l<i>Prev / l<i>Cur / l<i>Next: v210 window; l<i>Out0 / l<i>Out1: RGBA; colMatrix / gammaLut / gamutMatrix: the Loader's
      const uint32_t width = prog->global[0], height = prog->global[1];
      if (!width || !height) return fail(PH_E_INVALID, "%s: globalWorkItems must be [width, height]", prog->kernel.c_str());
      double tff, skip, rgb = 0, packing = 0;  // packedRgb (optional): 1 = the outputs are packed f32 RGB (12 bytes per pixel) for compose_up_write_v210_<n>
      if (find_arg(args, n, "packedRgb")) TRY(need_num(args, n, "packedRgb", &rgb));
      // packing (optional): 1 yuv422p10 / 2 yuv422p8 / 3 yuv420p / 4 nv12 - the windows are planar frames: l<i>Prev / Cur / Next their Y planes,
      // l<i>PrevU, l<i>PrevV, l<i>CurU ... their chroma planes
      if (find_arg(args, n, "packing")) TRY(need_num(args, n, "packing", &packing));
      const int pfmt = (int)packing;
      size_t pb[3] = {0, 0, 0};
      if (pfmt != PH_FMT_V210 && pfmt != PH_FMT_YUV422P10 && pfmt != PH_FMT_YUV422P8 && pfmt != PH_FMT_YUV420P && pfmt != PH_FMT_NV12)
        return fail(PH_E_INVALID, "kernel argument 'packing': %g (0 v210, 1 yuv422p10, 2 yuv422p8, 3 yuv420p, 4 nv12)", packing);
      ph_pack_plane_bytes(pfmt, width, height, pb);
      const size_t vb = pb[0], img = (size_t)width * height * (rgb != 0 ? 12 : 16);
      ph_deint_source src[ph::kMaxLayers];
      memset(src, 0, sizeof src);
      for (int i = 0; i < prog->n_layers; ++i) {
        char nm[16];
        ph_buf *x = nullptr;
        if (pfmt != PH_FMT_V210) {
          static const char *const which[3] = {"Prev", "Cur", "Next"};
          const void **slots[3][2] = {{&src[i].prev_u, &src[i].prev_v}, {&src[i].cur_u, &src[i].cur_v}, {&src[i].next_u, &src[i].next_v}};
          for (int f = 0; f < 3; ++f)
            for (int c = 0; c < (pfmt == PH_FMT_NV12 ? 1 : 2); ++c) {  // (nv12: l<i>PrevU ... are the interleaved CbCr planes)
              snprintf(nm, sizeof nm, "l%d%s%c", i, which[f], c ? 'V' : 'U');
              TRY(need_buf(args, n, nm, pb[1 + c], &x));
              *slots[f][c] = x->dptr;
            }
        }
        snprintf(nm, sizeof nm, "l%dPrev", i);
        TRY(need_buf(args, n, nm, vb, &x));
        src[i].prev = x->dptr;
        snprintf(nm, sizeof nm, "l%dCur", i);
        TRY(need_buf(args, n, nm, vb, &x));
        src[i].cur = x->dptr;
        snprintf(nm, sizeof nm, "l%dNext", i);
        TRY(need_buf(args, n, nm, vb, &x));
        src[i].next = x->dptr;
        snprintf(nm, sizeof nm, "l%dOut0", i);
        TRY(need_buf(args, n, nm, img, &x));
        src[i].out_parity0 = x->dptr;
        snprintf(nm, sizeof nm, "l%dOut1", i);
        TRY(need_buf(args, n, nm, img, &x));
        src[i].out_parity1 = x->dptr;
      }
      TRY(need_buf(args, n, "colMatrix", 48, &b));
      TRY(need_buf(args, n, "gammaLut", 65536 * 4, &c));
      TRY(need_buf(args, n, "gamutMatrix", 36, &d));
      TRY(need_num(args, n, "tff", &tff));
      TRY(need_num(args, n, "skipSpatial", &skip));
      if (!check_only) refresh_buf_lut(ctx, c);
      return check_only ? PH_OK : ph_yadif_pair_packed(ctx, queue, prog->n_layers, src, pfmt, width, height, (int)tff, (int)skip, rgb != 0 ? PH_IMG_RGB_F32 : PH_IMG_RGBA_F32,
                                    b->dptr, c->dptr, d->dptr);
    }
    default: break;
  }
  return fail(PH_E_UNKNOWN_KERNEL, "unhandled kernel id");
}
// the compositors: a channel's frame, enlarged layers, the buffer-addressed compositor, the headline kernel, combine_N (combine.ts, mixer.ts:189-228)
static int dispatch_compose(ph_ctx *ctx, ph_program *prog, const ph_arg *args, int n, int queue, bool check_only) {
  ph_buf *a = nullptr, *b = nullptr, *c = nullptr, *d = nullptr, *o = nullptr;
  double num = 0;
  int rc, w, h;
  (void)a, (void)b, (void)c, (void)d, (void)o, (void)num, (void)w, (void)h;
  switch (prog->id) {
    case K_CHAN_COMPOSE: {
      ChanCall call;
      TRY(chan_call_parse(ctx, prog, args, n, check_only, &call));
      return check_only ? PH_OK : ph_chan_compose(ctx, queue, call.n_layers, call.layers, call.out_format, call.out_planes, call.width, call.height, call.interlace,
                                  call.rd_cm->dptr, call.rd_lut->dptr, call.rd_gm->dptr, call.wr_cm ? call.wr_cm->dptr : nullptr, call.wr_lut->dptr);
    }
    case K_COMPOSE_UP: {
      UpCall u;
      TRY(up_call_parse(ctx, prog, args, n, check_only, &u));
      if (check_only) return PH_OK;
      if (u.pair)
        return ph_compose_up_write_v210_pair(ctx, queue, u.n, u.layers, u.layers2, u.o->dptr, u.o2->dptr, u.width, u.height, u.interlace, u.wcm->dptr, u.wl->dptr);
      return ph_compose_up_write_v210(ctx, queue, u.n, u.layers, u.o->dptr, u.width, u.height, u.interlace, u.wcm->dptr, u.wl->dptr);
    }
    case K_COMPOSE_V210: {
      // l<i>In: RGBA image; l<i>Matrix (optional): its 3x3 placement, absent = taken 1:1; l<i>WipeIn + l<i>WipeMask (optional):
      // a wipe transition on the placed layer; output: v210; outColMatrix / outGammaLut: the Saver's; interlace as 'write'
      const uint32_t width = prog->global[0], height = prog->global[1];
      if (!width || !height) return fail(PH_E_INVALID, "%s: globalWorkItems must be [width, height]", prog->kernel.c_str());
      ph_layer layers[ph::kMaxLayers];
      ph_layer_wipe wipes[ph::kMaxLayers];
      bool any_wipe = false;
      for (int i = 0; i < prog->n_layers; ++i) {
        char nm[24];
        ph_buf *x = nullptr;
        int lw, lh;
        snprintf(nm, sizeof nm, "l%dIn", i);
        TRY(need_buf(args, n, nm, 0, &x));
        TRY(need_image(x, nm, &lw, &lh));
        layers[i].rgba = x->dptr, layers[i].width = lw, layers[i].height = lh, layers[i].matrix9 = nullptr;
        snprintf(nm, sizeof nm, "l%dMatrix", i);
        if (find_arg(args, n, nm)) {
          TRY(need_buf(args, n, nm, 36, &x));
          layers[i].matrix9 = x->dptr;
        }
        wipes[i].incoming_rgba = wipes[i].mask_rgba = nullptr;
        snprintf(nm, sizeof nm, "l%dWipeIn", i);
        if (find_arg(args, n, nm)) {
          TRY(need_buf(args, n, nm, (size_t)width * height * 16, &x));
          wipes[i].incoming_rgba = x->dptr;
          snprintf(nm, sizeof nm, "l%dWipeMask", i);
          TRY(need_buf(args, n, nm, (size_t)width * height * 16, &x));
          wipes[i].mask_rgba = x->dptr;
          any_wipe = true;
        }
      }
      double interlace = 0;
      ph_buf *wcm = nullptr, *wl = nullptr;
      TRY(need_buf(args, n, "output", (size_t)ph_v210_pitch_bytes(width) * height, &o));
      TRY(need_buf(args, n, "outColMatrix", 48, &wcm));
      TRY(need_buf(args, n, "outGammaLut", 65536 * 4, &wl));
      if (find_arg(args, n, "interlace")) TRY(need_num(args, n, "interlace", &interlace));
      if (!check_only) refresh_buf_lut(ctx, wl);
      if (any_wipe)
        return check_only ? PH_OK : ph_compose_wipe_write_v210(ctx, queue, prog->n_layers, layers, wipes, o->dptr, width, height, (uint32_t)interlace,
                                          wcm->dptr, wl->dptr);
      return check_only ? PH_OK : ph_compose_write_v210(ctx, queue, prog->n_layers, layers, o->dptr, width, height, (uint32_t)interlace, wcm->dptr, wl->dptr);
    }
    case K_COMBINE: {
      const void *layers[ph::kMaxLayers];
      TRY(need_buf(args, n, "output", 0, &o));
      TRY(need_image(o, "output", &w, &h));
      for (int i = 0; i < prog->n_layers; ++i) {
        char nm[16];
        snprintf(nm, sizeof nm, "l%dIn", i);
        TRY(need_buf(args, n, nm, (size_t)w * h * 16, &a));
        layers[i] = a->dptr;
      }
      return check_only ? PH_OK : ph_combine(ctx, queue, prog->n_layers, layers, w, h, o->dptr);
    }
    case K_FUSED_V210: {
      FusedCall f;
      TRY(fused_call_parse(ctx, prog, args, n, check_only, &f));
      return check_only ? PH_OK : ph_fused_v210_combine(ctx, queue, f.n, f.layers, f.out, f.width, f.height, f.rd_cm->dptr, f.rd_lut->dptr, f.rd_gm->dptr,
                                   f.wr_cm->dptr, f.wr_lut->dptr);
    }
    default: break;
  }
  return fail(PH_E_UNKNOWN_KERNEL, "unhandled kernel id");
}
// the f32 image operators: transform, resize, dissolve / mixer / wipe, transition_wipe (transform.ts, resize.ts, transition.ts, mix.ts, wipe.ts)
static int dispatch_image(ph_ctx *ctx, ph_program *prog, const ph_arg *args, int n, int queue, bool check_only) {
  ph_buf *a = nullptr, *b = nullptr, *c = nullptr, *d = nullptr, *o = nullptr;
  double num = 0;
  int rc, w, h;
  (void)a, (void)b, (void)c, (void)d, (void)o, (void)num, (void)w, (void)h;
  switch (prog->id) {
    case K_TRANSFORM: {
      int iw, ih;
      TRY(need_buf(args, n, "input", 0, &a));
      TRY(need_image(a, "input", &iw, &ih));
      TRY(need_buf(args, n, "output", 0, &o));
      TRY(need_image(o, "output", &w, &h));
      TRY(need_buf(args, n, "transformMatrix", 32, &b));
      return check_only ? PH_OK : ph_transform(ctx, queue, a->dptr, iw, ih, b->dptr, o->dptr, w, h);
    }
    case K_RESIZE: {
      int iw, ih;
      double scale, ox, oy;
      TRY(need_buf(args, n, "input", 0, &a));
      TRY(need_image(a, "input", &iw, &ih));
      TRY(need_buf(args, n, "output", 0, &o));
      TRY(need_image(o, "output", &w, &h));
      TRY(need_buf(args, n, "flip", 16, &b));
      TRY(need_num(args, n, "scale", &scale));
      TRY(need_num(args, n, "offsetX", &ox));
      TRY(need_num(args, n, "offsetY", &oy));
      return check_only ? PH_OK : ph_resize(ctx, queue, a->dptr, iw, ih, (float)scale, (float)ox, (float)oy, b->dptr, o->dptr, w, h);
    }
    case K_DISSOLVE:
    case K_MIXER:
    case K_WIPE: {
      TRY(need_buf(args, n, "output", 0, &o));
      TRY(need_image(o, "output", &w, &h));
      TRY(need_buf(args, n, "input0", (size_t)w * h * 16, &a));
      TRY(need_buf(args, n, "input1", (size_t)w * h * 16, &b));
      TRY(need_num(args, n, prog->id == K_WIPE ? "wipe" : "mix", &num));
      if (prog->id == K_WIPE) return check_only ? PH_OK : ph_wipe(ctx, queue, a->dptr, b->dptr, (float)num, w, h, o->dptr);
      if (prog->id == K_MIXER) return check_only ? PH_OK : ph_mixer(ctx, queue, a->dptr, b->dptr, (float)num, w, h, o->dptr);
      return check_only ? PH_OK : ph_transition_dissolve(ctx, queue, a->dptr, b->dptr, (float)num, w, h, o->dptr);
    }
    case K_RGB_UNPACK: {  // image: an f32 RGBA image buffer whose first width * height * 12 bytes hold packed f32 RGB - expanded in place
      TRY(need_buf(args, n, "image", 0, &o));
      TRY(need_image(o, "image", &w, &h));
      return check_only ? PH_OK : ph_image_unpack_rgb(ctx, queue, o->dptr, w, h);
    }
    case K_TWIPE: {
      TRY(need_buf(args, n, "output", 0, &o));
      TRY(need_image(o, "output", &w, &h));
      TRY(need_buf(args, n, "input0", (size_t)w * h * 16, &a));
      TRY(need_buf(args, n, "input1", (size_t)w * h * 16, &b));
      TRY(need_buf(args, n, "maskIn", (size_t)w * h * 16, &c));
      return check_only ? PH_OK : ph_transition_wipe(ctx, queue, a->dptr, b->dptr, c->dptr, w, h, o->dptr);
    }
    default: break;
  }
  return fail(PH_E_UNKNOWN_KERNEL, "unhandled kernel id");
}
#undef TRY
static int dispatch(ph_ctx *ctx, ph_program *prog, const ph_arg *args, int n, int queue, bool check_only = false) {
  if (!check_only && inject_failure(ctx)) return fail(PH_E_HIP, "%s: launch failed: injected (context option fail_launches)", prog->kernel.c_str());
  switch (prog->id) {
    case K_PACK_READ:
    case K_PACK_WRITE:
    case K_V210_READ:
    case K_V210_WRITE:
    case K_V210_READ_BATCH:
      return dispatch_wire(ctx, prog, args, n, queue, check_only);
    case K_YADIF:
    case K_YADIF_PAIR:
    case K_V210_YADIF_PAIR:
      return dispatch_deint(ctx, prog, args, n, queue, check_only);
    case K_CHAN_COMPOSE:
    case K_COMPOSE_UP:
    case K_COMPOSE_V210:
    case K_COMBINE:
    case K_FUSED_V210:
      return dispatch_compose(ctx, prog, args, n, queue, check_only);
    case K_TRANSFORM:
    case K_RESIZE:
    case K_DISSOLVE:
    case K_MIXER:
    case K_WIPE:
    case K_TWIPE:
    case K_RGB_UNPACK:
      return dispatch_image(ctx, prog, args, n, queue, check_only);
  }
  return fail(PH_E_UNKNOWN_KERNEL, "unhandled kernel id");
}

int ph_run_program(ph_ctx *ctx, ph_program *prog, const ph_arg *args, int n_args, int queue, ph_run_timings *t) {
  if (!ctx || !prog || (n_args > 0 && !args)) return fail(PH_E_INVALID, "ph_run_program: NULL argument");
  PH_QUEUE("ph_run_program", queue);
  int rc = set_device(ctx);
  if (rc) return rc;
  rc = flush_dirty_args(ctx, args, n_args, queue);
  if (rc) return rc;
  if (!t) return dispatch(ctx, prog, args, n_args, queue);
  hipStream_t s = stream_of(ctx, queue);
  const auto t0 = std::chrono::steady_clock::now();
  hipEvent_t ev0 = nullptr, ev1 = nullptr;  // per call: timed runs may come from several threads
  PH_HIP(hipEventCreate(&ev0));
  if (hipEventCreate(&ev1) != hipSuccess) {
    hipEventDestroy(ev0);
    return fail(PH_E_HIP, "ph_run_program: hipEventCreate failed");
  }
  hipError_t te = hipEventRecord(ev0, s);
  rc = te == hipSuccess ? dispatch(ctx, prog, args, n_args, queue) : fail(PH_E_HIP, "hipEventRecord: %s", hipGetErrorString(te));
  float ms = 0.f;
  if (rc == PH_OK) {
    te = hipEventRecord(ev1, s);
    if (te == hipSuccess) te = hipEventSynchronize(ev1);
    if (te == hipSuccess) te = hipEventElapsedTime(&ms, ev0, ev1);
    if (te != hipSuccess) rc = fail(PH_E_HIP, "ph_run_program: timing failed: %s", hipGetErrorString(te));
  }
  hipEventDestroy(ev0);
  hipEventDestroy(ev1);
  if (rc) return rc;
  const auto t1 = std::chrono::steady_clock::now();
  t->data_to_kernel = 0;  // arguments are device-resident: nothing moves at launch
  t->kernel_exec = (uint32_t)(ms * 1000.0f + 0.5f);
  t->total_time = (uint32_t)std::chrono::duration_cast<std::chrono::microseconds>(t1 - t0).count();
  return PH_OK;
}

int ph_check_program(ph_ctx *ctx, ph_program *prog, const ph_arg *args, int n_args, int queue) {
  if (!ctx || !prog || (n_args > 0 && !args)) return fail(PH_E_INVALID, "ph_check_program: NULL argument");
  PH_QUEUE("ph_check_program", queue);
  return dispatch(ctx, prog, args, n_args, queue, true);
}

/* Several recorded jobs handed over in one call (a binding that records jobs and launches them later: node/defer.js).  Exactly the
 * ph_run_program calls in the order given - with the channel frames among them (chan_compose_v210_<n> programs of one geometry that name
 * the SAME Loader / Saver buffers and make v210 frames) put into launches together (ph_chan_compose_batch), likewise consecutive
 * fused_v210_combine_<n> frames; a job that reads or writes what an earlier job of its group writes (or writes what one reads) is
 * detected here and starts the next launch, so call order holds (include/phaneron_hip.h).  A call that fails after its checks
 * (a launch refused) has made the launches of the jobs before the failing group: ph_run_programs_progress says how many. */
namespace {
thread_local int g_programs_done = 0;  // jobs of the calling thread's last ph_run_programs call whose launches were made
}
int ph_run_programs_progress(int *jobs_done) {
  if (!jobs_done) return fail(PH_E_INVALID, "ph_run_programs_progress: NULL argument");
  *jobs_done = g_programs_done;
  return PH_OK;
}
int ph_run_programs(ph_ctx *ctx, int n_jobs, ph_program *const *progs, const ph_arg *const *args, const int *n_args, int queue) {
  if (!ctx || n_jobs < 1 || !progs || !args || !n_args) return fail(PH_E_INVALID, "ph_run_programs: NULL argument");
  PH_QUEUE("ph_run_programs", queue);
  g_programs_done = 0;
  int rc = set_device(ctx);
  if (rc) return rc;
  if (ctx->fail_launches.load() > 0) return fail(PH_E_HIP, "ph_run_programs: launch failed: injected (context option fail_launches)");
  std::vector<ChanCall> calls((size_t)n_jobs);
  std::vector<FusedCall> fused((size_t)n_jobs);
  std::vector<UpCall> ups;  // (sized when the first compose_up job shows up: most calls have none)
  std::vector<char> kind((size_t)n_jobs, 0);  // 1: a v210 frame from the channel kernel, 2: fused_v210_combine, 3: compose_up_write_v210, 0: whatever else, launched as it is
  for (int j = 0; j < n_jobs; ++j) {  // every job is checked before anything is launched: a bad one refuses the call as a whole
    if (!progs[j] || (n_args[j] > 0 && !args[j])) return fail(PH_E_INVALID, "ph_run_programs: job %d: NULL argument", j);
    if ((rc = flush_dirty_args(ctx, args[j], n_args[j], queue))) return rc;
    if (progs[j]->id == K_CHAN_COMPOSE) {
      if ((rc = chan_call_parse(ctx, progs[j], args[j], n_args[j], false, &calls[(size_t)j]))) return rc;
      kind[(size_t)j] = calls[(size_t)j].out_format == PH_FMT_V210;
    } else if (progs[j]->id == K_FUSED_V210) {
      if ((rc = fused_call_parse(ctx, progs[j], args[j], n_args[j], false, &fused[(size_t)j]))) return rc;
      kind[(size_t)j] = 2;
    } else if (progs[j]->id == K_COMPOSE_UP) {
      if (ups.empty()) ups.resize((size_t)n_jobs);
      if ((rc = up_call_parse(ctx, progs[j], args[j], n_args[j], false, &ups[(size_t)j]))) return rc;
      kind[(size_t)j] = 3;
    } else if ((rc = dispatch(ctx, progs[j], args[j], n_args[j], queue, true))) {
      return rc;
    }
  }
  for (int j = 0; j < n_jobs;) {
    if (!kind[(size_t)j]) {
      if ((rc = dispatch(ctx, progs[j], args[j], n_args[j], queue))) return rc;
      g_programs_done = ++j;
      continue;
    }
    int k = j;
    if (inject_failure(ctx)) return fail(PH_E_HIP, "ph_run_programs: launch failed: injected (context option fail_launches)");
    if (kind[(size_t)j] == 2) {
      // frames of one size, layer count and recipe: one launch of the headline kernel (ph_fused_v210_combine_batch).  A frame that reads
      // what an earlier frame of the run writes (or writes what one reads or writes) starts the next launch: call order is kept.
      const FusedCall &f0 = fused[(size_t)j];
      std::vector<const void *> layers;
      std::vector<void *> outs;
      auto overlap = [&](const void *p, const void *q) {
        const char *a0 = (const char *)p, *b0 = (const char *)q;
        return a0 < b0 + f0.frame_bytes && b0 < a0 + f0.frame_bytes;
      };
      for (; k < n_jobs && kind[(size_t)k] == 2 && k - j < ph::kMaxBatch; ++k) {
        const FusedCall &f = fused[(size_t)k];
        if (f.n != f0.n || f.width != f0.width || f.height != f0.height || f.rd_cm != f0.rd_cm || f.rd_lut != f0.rd_lut || f.rd_gm != f0.rd_gm ||
            f.wr_cm != f0.wr_cm || f.wr_lut != f0.wr_lut)
          break;
        bool clash = false;
        for (size_t e = 0; e < outs.size() && !clash; ++e) {
          clash = overlap(f.out, outs[e]);
          for (int l = 0; l < f.n && !clash; ++l) clash = overlap(f.layers[l], outs[e]) || overlap(f.out, layers[e * (size_t)f0.n + (size_t)l]);
        }
        if (clash) break;
        layers.insert(layers.end(), f.layers, f.layers + f.n);
        outs.push_back(f.out);
      }
      rc = ph_fused_v210_combine_batch(ctx, queue, (int)outs.size(), f0.n, layers.data(), outs.data(), f0.width, f0.height, f0.rd_cm->dptr, f0.rd_lut->dptr,
                                       f0.rd_gm->dptr, f0.wr_cm->dptr, f0.wr_lut->dptr);
      if (rc) return rc;
      g_programs_done = j = k;
      continue;
    }
    if (kind[(size_t)j] == 3) {
      // frames of the 2 x 2-block compositor of ONE shape (layer count, image format and sizes, placements, output size, field mode, Saver) -
      // several channels' frames from de-interlaced fields, each job one frame or a frame's two fields - in one launch of up to
      // kMaxUpJobs frames (ph_compose_up_write_v210_batch); a job that writes a frame an earlier one of the group writes starts the next
      const UpCall &u0 = ups[(size_t)j];
      const ph_image_layer *sets[ph::kMaxUpJobs];
      void *outs[ph::kMaxUpJobs];
      int frames = 0;
      for (; k < n_jobs && kind[(size_t)k] == 3; ++k) {
        const UpCall &u = ups[(size_t)k];
        bool same = u.n == u0.n && u.rgb == u0.rgb && u.width == u0.width && u.height == u0.height && u.interlace == u0.interlace && u.wcm == u0.wcm && u.wl == u0.wl;
        for (int l = 0; l < u.n && same; ++l) {
          same = u.layers[l].width == u0.layers[l].width && u.layers[l].height == u0.layers[l].height;
          for (int e = 0; e < 9 && same; ++e) same = u.layers[l].matrix9_host[e] == u0.layers[l].matrix9_host[e];
        }
        if (!same || frames + (u.pair ? 2 : 1) > ph::kMaxUpJobs) break;
        bool clash = u.pair && u.o->dptr == u.o2->dptr;
        for (int f = 0; f < frames && !clash; ++f) clash = outs[f] == u.o->dptr || (u.pair && outs[f] == u.o2->dptr);
        if (clash) break;
        sets[frames] = u.layers, outs[frames++] = u.o->dptr;
        if (u.pair) sets[frames] = u.layers2, outs[frames++] = u.o2->dptr;
      }
      if (k == j) {  // (the first job does not fit a group of its own making - a pair writing one buffer twice: as it is, for its own error)
        if ((rc = dispatch(ctx, progs[j], args[j], n_args[j], queue))) return rc;
        g_programs_done = ++j;
        continue;
      }
      rc = ph_compose_up_write_v210_batch(ctx, queue, frames, u0.n, sets, outs, u0.width, u0.height, u0.interlace, u0.wcm->dptr, u0.wl->dptr);
      if (rc) return rc;
      g_programs_done = j = k;
      continue;
    }
    const ChanCall &c0 = calls[(size_t)j];
    std::vector<ph_chan_job> batch;
    // a frame that reads what an earlier frame of the group writes (a channel routed into another), or writes what one reads, starts the next
    // call of ph_chan_compose_batch: the jobs of one such call may share launches, and call order has to hold
    const size_t out_bytes = (size_t)ph_v210_pitch_bytes(c0.width) * c0.height;
    auto touches = [&](const void *p, size_t bytes, const void *out) {
      const char *a0 = (const char *)p, *b0 = (const char *)out;
      return p && a0 < b0 + out_bytes && b0 < a0 + (bytes ? bytes : 1);
    };
    auto reads = [&](const ChanCall &c, const void *out) {  // does a source of c overlap the frame at `out`?
      for (int l = 0; l < c.n_layers; ++l)
        for (const ph_chan_source *s2 : {&c.layers[l].src, &c.layers[l].incoming, &c.layers[l].mask})
          if (s2->data && (touches(s2->data, (size_t)s2->width * s2->height * 16u, out) || touches(s2->data_u, (size_t)s2->width * s2->height * 2u, out) ||
                           touches(s2->data_v, (size_t)s2->width * s2->height * 2u, out)))
            return true;
      return false;
    };
    for (; k < n_jobs && kind[(size_t)k] == 1; ++k) {
      const ChanCall &c = calls[(size_t)k];
      if (c.width != c0.width || c.height != c0.height || c.rd_cm != c0.rd_cm || c.rd_lut != c0.rd_lut || c.rd_gm != c0.rd_gm || c.wr_cm != c0.wr_cm ||
          c.wr_lut != c0.wr_lut)
        break;
      bool clash = false;
      for (int e = j; e < k && !clash; ++e) clash = reads(c, calls[(size_t)e].out_planes[0]) || reads(calls[(size_t)e], c.out_planes[0]);
      if (clash) break;
      batch.push_back(ph_chan_job{c.n_layers, c.layers, c.out_planes[0], c.interlace});
    }
    rc = ph_chan_compose_batch(ctx, queue, (int)batch.size(), batch.data(), c0.width, c0.height, c0.rd_cm->dptr, c0.rd_lut->dptr, c0.rd_gm->dptr,
                               c0.wr_cm->dptr, c0.wr_lut->dptr);
    if (rc) return rc;
    g_programs_done = j = k;
  }
  return PH_OK;
}

// ---- typed entry points ------------------------------------------------------------------------
uint32_t ph_v210_pitch_bytes(uint32_t width) { return width ? ph::v210_pitch_bytes(width) : 0; }

#define PH_LAUNCH(expr)                                                                         \
  do {                                                                                          \
    if (!ctx) return fail(PH_E_INVALID, "%s: ctx is NULL", __func__);                           \
    PH_QUEUE(__func__, queue);                                                                  \
    int rc_ = set_device(ctx);                                                                  \
    if (rc_) return rc_;                                                                        \
    hipError_t e_ = ph::trace_launch(#expr) ? hipSuccess : (expr);                              \
    if (e_ != hipSuccess) return fail(PH_E_HIP, "%s: launch failed: %s", __func__, hipGetErrorString(e_)); \
    return PH_OK;                                                                               \
  } while (0)

int ph_v210_read(ph_ctx *ctx, int queue, const void *in, void *out, uint32_t width, uint32_t height, const void *cm,
                 const void *lut, const void *gm) {
  if (!in || !out || !cm || !lut || !gm || !width) return fail(PH_E_INVALID, "ph_v210_read: NULL/zero argument");
  if (!height) return PH_OK;
  if (ctx && width % 2 == 0) {  // (a tail of 2 or 4 pixels is the reference's: v210.ts:84-110; other widths: the general kernel)
    const LutRef lref = lds_view(ctx, lut);
    if (const ph::LutView *v = lref.get())
      PH_LAUNCH(ph::launch_v210_read_lds(stream_of(ctx, queue), in, out, width, height, cm, gm, *v,
                                         (uint32_t)ctx->props.multiProcessorCount));
  }
  PH_LAUNCH(ph::launch_v210_read(stream_of(ctx, queue), in, out, width, height, cm, lut, gm));
}

int ph_v210_read_batch(ph_ctx *ctx, int queue, int n, const void *const *ins, void *const *outs, uint32_t width,
                       uint32_t height, const void *cm, const void *lut, const void *gm) {
  if (n < 1 || n > ph::kMaxLayers) return fail(PH_E_INVALID, "ph_v210_read_batch: 1..%d frames", ph::kMaxLayers);
  if (!ins || !outs || !cm || !lut || !gm || !width) return fail(PH_E_INVALID, "ph_v210_read_batch: NULL/zero argument");
  for (int i = 0; i < n; ++i)
    if (!ins[i] || !outs[i]) return fail(PH_E_INVALID, "ph_v210_read_batch: frame %d is NULL", i);
  if (!height) return PH_OK;
  if (ctx && width % 2 == 0) {
    const LutRef lref = lds_view(ctx, lut);
    if (const ph::LutView *v = lref.get())
      PH_LAUNCH(ph::launch_v210_read_lds_batch(stream_of(ctx, queue), n, ins, outs, width, height, cm, gm, *v,
                                               (uint32_t)ctx->props.multiProcessorCount));
  }
  for (int i = 0; i < n; ++i) {  // table not LDS-resident / ragged width: one gather-kernel launch per frame
    const int rc = ph_v210_read(ctx, queue, ins[i], outs[i], width, height, cm, lut, gm);
    if (rc) return rc;
  }
  return PH_OK;
}

int ph_v210_write(ph_ctx *ctx, int queue, const void *in, void *out, uint32_t width, uint32_t height,
                  uint32_t interlace, const void *cm, const void *lut) {
  if (!in || !out || !cm || !lut || !width) return fail(PH_E_INVALID, "ph_v210_write: NULL/zero argument");
  if (interlace != 0 && interlace != 1 && interlace != 3) return fail(PH_E_INVALID, "ph_v210_write: interlace must be 0, 1 or 3");
  if (!height) return PH_OK;
  if (ctx && width % 2 == 0) {
    const LutRef lref = lds_view(ctx, lut);
    if (const ph::LutView *v = lref.get())
      PH_LAUNCH(ph::launch_v210_write_lds(stream_of(ctx, queue), in, out, width, height, interlace, cm, *v,
                                          (uint32_t)ctx->props.multiProcessorCount));
  }
  PH_LAUNCH(ph::launch_v210_write(stream_of(ctx, queue), in, out, width, height, interlace, cm, lut));
}

int ph_fused_v210_combine(ph_ctx *ctx, int queue, int n, const void *const *layers, void *out, uint32_t width,
                          uint32_t height, const void *rd_cm, const void *rd_lut, const void *rd_gm,
                          const void *wr_cm, const void *wr_lut) {
  if (n < 1 || n > ph::kMaxLayers) return fail(PH_E_INVALID, "ph_fused_v210_combine: 1..%d layers", ph::kMaxLayers);
  if (!layers || !out || !rd_cm || !rd_lut || !rd_gm || !wr_cm || !wr_lut || !width)
    return fail(PH_E_INVALID, "ph_fused_v210_combine: NULL/zero argument");
  // (a width that is not a multiple of 48 - 1280 x 720, src/config.ts:43-54 - has lines that end in a tail quad and cleared slots:
  // the LDS kernel's TAIL instantiation; the reference's kernels serve tails of 2 or 4 pixels, v210.ts:84-110,166-193)
  const bool ragged = width % 48 != 0;
  if (width & 1) return fail(PH_E_INVALID, "ph_fused_v210_combine: width %u is odd; run the separate kernels", width);
  if (!height) return PH_OK;
  ph::FusedArgs a{};
  for (int i = 0; i < n; ++i) {
    if (!layers[i]) return fail(PH_E_INVALID, "ph_fused_v210_combine: layer %d is NULL", i);
    a.layers[i] = layers[i];
  }
  a.out = out;
  a.quads_per_line_used = width / 6;
  a.quads_per_line_pitch = ph::v210_pitch_bytes(width) / 16;
  a.total_quads = (ragged ? a.quads_per_line_pitch : a.quads_per_line_used) * height;
  a.tail_px = width % 6, a.magic_qpp = (uint32_t)(((1ull << 32) + a.quads_per_line_pitch - 1) / a.quads_per_line_pitch);
  if (ragged && (uint64_t)a.total_quads * a.quads_per_line_pitch >= (1ull << 32))  // the kernel's slot -> column division by reciprocal
    return fail(PH_E_INVALID, "ph_fused_v210_combine: a %u x %u frame with ragged lines is too large; run the separate kernels", width, height);
  a.rd_cm = (const float *)rd_cm, a.rd_lut = (const float *)rd_lut, a.rd_gm = (const float *)rd_gm;
  a.wr_cm = (const float *)wr_cm, a.wr_lut = (const float *)wr_lut;
  if (ctx) {
    const LutRef rref = lds_view(ctx, rd_lut), wref = lds_view(ctx, wr_lut);
    const ph::LutView *rv = rref.get(), *wv = wref.get();
    if (rv && wv) {
      ph::FusedLdsArgs la{};
      la.f = a, la.rd = *rv, la.wr = *wv, la.jobs = 1;
      PH_LAUNCH(ph::launch_fused_v210_combine_lds(stream_of(ctx, queue), n, la, (uint32_t)ctx->props.multiProcessorCount));
    }
  }
  PH_LAUNCH(ph::launch_fused_v210_combine(stream_of(ctx, queue), n, a));
}

int ph_pack_plane_bytes(int format, uint32_t width, uint32_t height, size_t bytes[3]) {
  if (!bytes || !width) return fail(PH_E_INVALID, "ph_pack_plane_bytes: NULL/zero argument");
  const int n = ph::pack_plane_bytes(format, width, height, bytes);
  return n < 0 ? fail(PH_E_INVALID, "ph_pack_plane_bytes: unknown format %d", format) : n;
}

int ph_fused_v210_combine_batch(ph_ctx *ctx, int queue, int jobs, int n, const void *const *layers, void *const *outs,
                                uint32_t width, uint32_t height, const void *rd_cm, const void *rd_lut, const void *rd_gm,
                                const void *wr_cm, const void *wr_lut) {
  if (jobs < 1 || jobs > ph::kMaxBatch) return fail(PH_E_INVALID, "ph_fused_v210_combine_batch: 1..%d jobs", ph::kMaxBatch);
  if (n < 1 || n > ph::kMaxLayers) return fail(PH_E_INVALID, "ph_fused_v210_combine_batch: 1..%d layers", ph::kMaxLayers);
  if (!layers || !outs || !rd_cm || !rd_lut || !rd_gm || !wr_cm || !wr_lut)
    return fail(PH_E_INVALID, "ph_fused_v210_combine_batch: NULL argument");
  for (int j = 0; j < jobs; ++j) {
    if (!outs[j]) return fail(PH_E_INVALID, "ph_fused_v210_combine_batch: output %d is NULL", j);
    for (int l = 0; l < n; ++l)
      if (!layers[j * n + l]) return fail(PH_E_INVALID, "ph_fused_v210_combine_batch: job %d layer %d is NULL", j, l);
  }
  const LutRef rref = lds_view(ctx, rd_lut), wref = lds_view(ctx, wr_lut);  // (a NULL context has no tables)
  const ph::LutView *rv = rref.get(), *wv = wref.get();
  if (jobs == 1 || !rv || !wv || !width || width % 48 || !height) {
    // one job, or no LDS form of the tables: the plain entry point per job (same results)
    for (int j = 0; j < jobs; ++j) {
      int rc = ph_fused_v210_combine(ctx, queue, n, layers + (size_t)j * n, outs[j], width, height, rd_cm, rd_lut, rd_gm,
                                     wr_cm, wr_lut);
      if (rc != PH_OK) return rc;
    }
    return PH_OK;
  }
  ph::FusedLdsArgs la{};
  for (int l = 0; l < n; ++l) la.f.layers[l] = layers[l];
  la.f.out = outs[0];
  la.f.quads_per_line_used = width / 6;
  la.f.quads_per_line_pitch = ph::v210_pitch_bytes(width) / 16;
  la.f.total_quads = la.f.quads_per_line_used * height;
  la.f.rd_cm = (const float *)rd_cm, la.f.rd_lut = (const float *)rd_lut, la.f.rd_gm = (const float *)rd_gm;
  la.f.wr_cm = (const float *)wr_cm, la.f.wr_lut = (const float *)wr_lut;
  la.rd = *rv, la.wr = *wv, la.jobs = (uint32_t)jobs;
  for (int j = 1; j < jobs; ++j) {
    for (int l = 0; l < n; ++l) la.more_layers[j - 1][l] = layers[(size_t)j * n + l];
    la.more_out[j - 1] = outs[j];
  }
  PH_LAUNCH(ph::launch_fused_v210_combine_lds(stream_of(ctx, queue), n, la, (uint32_t)ctx->props.multiProcessorCount));
}

int ph_pack_read(ph_ctx *ctx, int queue, int format, const void *const planes[3], void *out, uint32_t width,
                 uint32_t height, const void *cm, const void *lut, const void *gm) {
  if (!ctx || !planes || !out || !lut || !gm || !width) return fail(PH_E_INVALID, "ph_pack_read: NULL/zero argument");
  if (format == PH_FMT_V210) return ph_v210_read(ctx, queue, planes[0], out, width, height, cm, lut, gm);
  size_t pb[3];
  const int np = ph::pack_plane_bytes(format, width, height, pb);
  if (np < 0) return fail(PH_E_INVALID, "ph_pack_read: unknown format %d", format);
  for (int i = 0; i < np; ++i)
    if (!planes[i]) return fail(PH_E_INVALID, "ph_pack_read: plane %d is NULL", i);
  if (format < PH_FMT_RGBA8 && !cm) return fail(PH_E_INVALID, "ph_pack_read: YCbCr formats need a colMatrix");
  if (!height) return PH_OK;
  PH_LAUNCH(ph::launch_pack_read(stream_of(ctx, queue), format, planes, out, width, height, cm, lut, gm, lds_view(ctx, lut).get(),
                                 (uint32_t)ctx->props.multiProcessorCount));
}

int ph_pack_read_batch(ph_ctx *ctx, int queue, int format, int n, const void *const (*planes)[3], void *const *outs, uint32_t width, uint32_t height,
                       const void *cm, const void *lut, const void *gm) {
  if (!ctx || !planes || !outs || !lut || !gm || !width) return fail(PH_E_INVALID, "ph_pack_read_batch: NULL/zero argument");
  if (n < 1 || n > ph::kMaxLayers) return fail(PH_E_INVALID, "ph_pack_read_batch: 1..%d frames", ph::kMaxLayers);
  const LutRef lref = lds_view(ctx, lut);
  const ph::LutView *v = format == PH_FMT_V210 ? nullptr : lref.get();
  if (n == 1 || !v) {  // one frame, v210 (its own batch call), or no LDS form of the table: frame by frame
    if (format == PH_FMT_V210) {
      const void *ins[ph::kMaxLayers];
      for (int i = 0; i < n; ++i) ins[i] = planes[i][0];
      return ph_v210_read_batch(ctx, queue, n, ins, outs, width, height, cm, lut, gm);
    }
    for (int i = 0; i < n; ++i) {
      const int rc = ph_pack_read(ctx, queue, format, planes[i], outs[i], width, height, cm, lut, gm);
      if (rc) return rc;
    }
    return PH_OK;
  }
  size_t pb[3];
  const int np = ph::pack_plane_bytes(format, width, height, pb);
  if (np < 0) return fail(PH_E_INVALID, "ph_pack_read_batch: unknown format %d", format);
  for (int i = 0; i < n; ++i) {
    if (!outs[i]) return fail(PH_E_INVALID, "ph_pack_read_batch: output %d is NULL", i);
    for (int k = 0; k < np; ++k)
      if (!planes[i][k]) return fail(PH_E_INVALID, "ph_pack_read_batch: frame %d: plane %d is NULL", i, k);
  }
  if (format < PH_FMT_RGBA8 && !cm) return fail(PH_E_INVALID, "ph_pack_read_batch: YCbCr formats need a colMatrix");
  if (!height) return PH_OK;
  PH_LAUNCH(ph::launch_pack_read_batch(stream_of(ctx, queue), format, n, planes, outs, width, height, cm, gm, *v, (uint32_t)ctx->props.multiProcessorCount));
}

int ph_pack_write(ph_ctx *ctx, int queue, int format, const void *in, void *const planes[3], uint32_t width,
                  uint32_t height, uint32_t interlace, const void *cm, const void *lut) {
  if (!ctx || !planes || !in || !lut || !width) return fail(PH_E_INVALID, "ph_pack_write: NULL/zero argument");
  if (interlace != 0 && interlace != 1 && interlace != 3) return fail(PH_E_INVALID, "ph_pack_write: interlace must be 0, 1 or 3");
  if (format == PH_FMT_V210) return ph_v210_write(ctx, queue, in, planes[0], width, height, interlace, cm, lut);
  size_t pb[3];
  const int np = ph::pack_plane_bytes(format, width, height, pb);
  if (np < 0) return fail(PH_E_INVALID, "ph_pack_write: unknown format %d", format);
  for (int i = 0; i < np; ++i)
    if (!planes[i]) return fail(PH_E_INVALID, "ph_pack_write: plane %d is NULL", i);
  if (format < PH_FMT_RGBA8 && !cm) return fail(PH_E_INVALID, "ph_pack_write: YCbCr formats need a colMatrix");
  if (!height) return PH_OK;
  PH_LAUNCH(ph::launch_pack_write(stream_of(ctx, queue), format, in, planes, width, height, interlace, cm, lut,
                                  lds_view(ctx, lut).get(), (uint32_t)ctx->props.multiProcessorCount));
}

static int compose_write(ph_ctx *ctx, int queue, int n, const ph_layer *layers, const ph_layer_wipe *wipes, void *out, uint32_t out_w,
                         uint32_t out_h, uint32_t interlace, const void *wr_cm, const void *wr_lut);

int ph_compose_write_v210(ph_ctx *ctx, int queue, int n, const ph_layer *layers, void *out, uint32_t out_w,
                          uint32_t out_h, uint32_t interlace, const void *wr_cm, const void *wr_lut) {
  return compose_write(ctx, queue, n, layers, nullptr, out, out_w, out_h, interlace, wr_cm, wr_lut);
}

int ph_compose_wipe_write_v210(ph_ctx *ctx, int queue, int n, const ph_layer *layers, const ph_layer_wipe *wipes, void *out,
                               uint32_t out_w, uint32_t out_h, uint32_t interlace, const void *wr_cm, const void *wr_lut) {
  if (!wipes) return fail(PH_E_INVALID, "ph_compose_wipe_write_v210: NULL argument");
  return compose_write(ctx, queue, n, layers, wipes, out, out_w, out_h, interlace, wr_cm, wr_lut);
}

static int compose_write(ph_ctx *ctx, int queue, int n, const ph_layer *layers, const ph_layer_wipe *wipes, void *out, uint32_t out_w,
                         uint32_t out_h, uint32_t interlace, const void *wr_cm, const void *wr_lut) {
  if (!ctx || !layers || !out || !wr_cm || !wr_lut) return fail(PH_E_INVALID, "ph_compose_write_v210: NULL argument");
  if (n < 1 || n > ph::kMaxLayers) return fail(PH_E_INVALID, "ph_compose_write_v210: 1..%d layers", ph::kMaxLayers);
  // (a width that is not a multiple of 48 - 1280 x 720 - goes to the quad-per-lane kernel, which writes the reference's tail quad
  // and cleared slots, v210.ts:131-136,166-193; the reference's writer serves tails of 2 or 4 pixels)
  if (!out_w || (out_w & 1)) return fail(PH_E_INVALID, "ph_compose_write_v210: width %u is zero or odd; run the separate kernels", out_w);
  if (interlace != 0 && interlace != 1 && interlace != 3) return fail(PH_E_INVALID, "ph_compose_write_v210: interlace must be 0, 1 or 3");
  const LutRef wref = lds_view(ctx, wr_lut);
  const ph::LutView *wv = wref.get();
  if (!wv) return fail(PH_E_INVALID, "ph_compose_write_v210: the writer gamma LUT has no LDS form (ph_lut_register it, or run the separate kernels)");
  ph::ComposeArgs a{};
  a.n = n;
  for (int i = 0; i < n; ++i) {
    if (!layers[i].rgba || layers[i].width <= 0 || layers[i].height <= 0)
      return fail(PH_E_INVALID, "ph_compose_write_v210: layer %d is empty", i);
    if (!layers[i].matrix9 && ((uint32_t)layers[i].width != out_w || (uint32_t)layers[i].height != out_h))
      return fail(PH_E_INVALID, "ph_compose_write_v210: layer %d has no transform but is %dx%d, not the output size", i,
                  layers[i].width, layers[i].height);
    a.layers[i] = layers[i].rgba, a.matrix[i] = (const float *)layers[i].matrix9;
    a.lw[i] = layers[i].width, a.lh[i] = layers[i].height;
    if (wipes && (wipes[i].incoming_rgba || wipes[i].mask_rgba)) {
      if (!wipes[i].incoming_rgba || !wipes[i].mask_rgba)
        return fail(PH_E_INVALID, "ph_compose_wipe_write_v210: layer %d needs both the incoming image and the mask", i);
      a.wipe_with[i] = wipes[i].incoming_rgba, a.wipe_mask[i] = wipes[i].mask_rgba;
    }
  }
  a.out = out, a.out_w = out_w, a.out_h = out_h;
  a.wr = *wv;  // the path choice depends on the table's size
  bool any_wipe = false;  // a wipes array without a single entry set is the plain compositor (the header: such entries leave the layer alone)
  for (int i = 0; i < n; ++i) any_wipe = any_wipe || a.wipe_with[i] != nullptr;
  if (any_wipe && !ph::compose_can_wipe(a))
    return fail(PH_E_INVALID, "ph_compose_wipe_write_v210: needs out_width %% 192 == 0, source images below 2 GiB and a writer table that leaves "
                              "the staging area free in LDS; run the separate kernels");
  a.line_step = interlace ? 2 : 1, a.first_line = (interlace == 3) ? 1 : 0;
  a.lines = interlace ? out_h / 2 : out_h;
  a.wr_cm = (const float *)wr_cm, a.wr = *wv;
  if (!a.lines) return PH_OK;
  PH_LAUNCH(ph::launch_compose_write_v210(stream_of(ctx, queue), a, (uint32_t)ctx->props.multiProcessorCount));
}

static int chan_source(const ph_chan_source &s, const char *what, int layer, uint32_t out_w, uint32_t out_h, ph::ChanSrc *o, const void **pu,
                       const void **pv, const float **cm, uint32_t *planar) {
  if (!s.data || s.width <= 0 || s.height <= 0)
    return fail(PH_E_INVALID, "ph_chan_compose_v210: layer %d: the %s is empty", layer, what);
  const bool is_planar = s.format >= PH_SRC_YUV422P10 && s.format <= PH_SRC_NV12, is_rgb8 = s.format == PH_SRC_RGBA8 || s.format == PH_SRC_BGRA8;
  if (s.format != PH_SRC_V210 && s.format != PH_SRC_RGBA_F32 && !is_planar && !is_rgb8)
    return fail(PH_E_INVALID, "ph_chan_compose_v210: layer %d: the %s has format %d (not a PH_SRC_*)", layer, what, s.format);
  *pu = *pv = nullptr, *cm = nullptr;
  if (is_rgb8) *planar = 2;  // (served by the kernel's wire-format instantiation)
  if (is_planar) {
    if (!s.data_u || (s.format != PH_SRC_NV12 && !s.data_v) || (s.width & 1) || ((s.format == PH_SRC_YUV420P || s.format == PH_SRC_NV12) && (s.height & 1)))
      return fail(PH_E_INVALID, "ph_chan_compose_v210: layer %d: the %s is planar: it needs its chroma plane(s), an even width and, for 4:2:0, an even height", layer, what);
    *pu = s.data_u, *pv = s.data_v, *cm = (const float *)s.col_matrix12, *planar = 2;
  }
  // (the reference's v210 reader serves tails of 2 or 4 pixels: v210.ts:84-110)
  if (s.format == PH_SRC_V210 && (s.width & 1))
    return fail(PH_E_INVALID, "ph_chan_compose_v210: layer %d: the %s is a v210 frame %d wide, an odd width; run the separate kernels", layer, what, s.width);
  if (s.format == PH_SRC_V210 && s.width % 6 && *planar < 1) *planar = 1;  // a line with a tail: the kernel's instantiation for those
  if (!s.matrix9_host && ((uint32_t)s.width != out_w || (uint32_t)s.height != out_h))
    return fail(PH_E_INVALID, "ph_chan_compose_v210: layer %d: the %s has no transform but is %dx%d, not the output size", layer, what, s.width, s.height);
  o->ptr = s.data, o->w = (uint32_t)s.width, o->h = (uint32_t)s.height;
  static const uint32_t kinds[] = {ph::kChanNone, ph::kChanV210, ph::kChanRgba, ph::kChanP10, ph::kChanP8x422, ph::kChanP8x420, ph::kChanNv12, ph::kChanRgba8, ph::kChanBgra8};
  o->kind = kinds[s.format];
  // planar: the luma line pitch in samples is the width rounded up to 8 (yuv422p10.ts:221), one or two bytes each
  o->pitch = s.format == PH_SRC_V210 ? ph_v210_pitch_bytes((uint32_t)s.width)
             : is_planar ? (((uint32_t)s.width + 7u) & ~7u) * (s.format == PH_SRC_YUV422P10 ? 2u : 1u)
             : is_rgb8 ? (uint32_t)s.width * 4u : (uint32_t)s.width * 16u;  // rgba8.ts:103-105: no line padding
  if ((uint64_t)o->pitch * o->h >= (1ull << 30))
    return fail(PH_E_INVALID, "ph_chan_compose_v210: layer %d: the %s is 1 GiB or larger; run the separate kernels", layer, what);
  o->sampled = s.matrix9_host ? 1u : 0u;
  o->tail_from = (uint32_t)s.width - (uint32_t)s.width % 6u;
  for (int i = 0; i < 6; ++i) o->m[i] = s.matrix9_host ? s.matrix9_host[i] : 0.0f;
  return PH_OK;
}

// PH_CHAN_BATCH=0: every job of a batch call through the one-job kernel (A/B runs, tools/chan_bench.py)
static bool chan_batch_on() {
  static const bool on = !(getenv("PH_CHAN_BATCH") && getenv("PH_CHAN_BATCH")[0] == '0');
  return on;
}
// the index frames between the phases: one area per queue (launches on one queue are in order), grown on demand (ctx->mu held)
static int chan_index_reserve(ph_ctx *ctx, int queue, size_t need) {
  if (ctx->chan_index_bytes[queue] >= need) return PH_OK;
  if (ctx->chan_index[queue]) {
    hipStreamSynchronize(ctx->streams[queue]);
    hipFree(ctx->chan_index[queue]);
    ctx->chan_index[queue] = nullptr, ctx->chan_index_bytes[queue] = 0;
  }
  PH_HIP(hipMalloc(&ctx->chan_index[queue], need));
  ctx->chan_index_bytes[queue] = need;
  return PH_OK;
}
// one launch of the batch kernel: b holds the jobs and their ops, g the geometry, colour recipe and tables (set_device done)
static int chan_batch_launch(ph_ctx *ctx, int queue, ph::ChanBatchArgs &b, const ph::ChanArgs &g) {
  b.out_w = g.out_w, b.out_h = g.out_h, b.lines = g.lines, b.line_step = g.line_step;
  b.rd_cm = g.rd_cm, b.rd_gm = g.rd_gm, b.wr_cm = g.wr_cm, b.rd = g.rd, b.wr = g.wr;
  b.tails = g.planar >= 1 ? 1u : 0u, b.planar = g.planar == 2 ? 1u : 0u, b.out_qpitch = g.out_qpitch, b.out_tail_from = g.out_tail_from;
  const size_t each = (ph::chan_index_bytes(g.out_w, g.lines) + 255u) & ~(size_t)255u;
  std::lock_guard<std::mutex> scratch(ctx->chan_scratch_mu[queue]);
  {
    std::lock_guard<std::mutex> lock(ctx->mu);
    int rc = chan_index_reserve(ctx, queue, each * b.jobs);
    if (rc) return rc;
    for (uint32_t j = 0; j < b.jobs; ++j) b.job[j].index = (char *)ctx->chan_index[queue] + each * j;
  }
  hipError_t e = ph::launch_chan_compose_batch(stream_of(ctx, queue), b, (uint32_t)ctx->props.multiProcessorCount);
  if (e != hipSuccess) return fail(PH_E_HIP, "ph_chan_compose_batch: launch failed: %s", hipGetErrorString(e));
  return PH_OK;
}

// a channel's layers as the kernel's flat program: one op per source to sample (ph_kernels.h ChanOp)
static int chan_ops(int n, const ph_chan_layer *layers, uint32_t out_w, uint32_t out_h, ph::ChanOp *op, const void **plane_u, const void **plane_v,
                    const float **cm_op, uint32_t *planar, int *n_ops) {
  int k = 0;
  for (int i = 0; i < n; ++i) {
    const ph_chan_layer &L = layers[i];
    const uint32_t first = i == 0 ? ph::kChanActFirst : 0u;
    int rc = chan_source(L.src, "source", i, out_w, out_h, &op[k].src, &plane_u[k], &plane_v[k], &cm_op[k], planar);
    if (rc) return rc;
    if (L.transition == PH_TRANSITION_CUT) {
      op[k++].action = ph::kChanActLayer | first;
    } else if (L.transition == PH_TRANSITION_DISSOLVE || L.transition == PH_TRANSITION_WIPE) {
      op[k++].action = ph::kChanActHold;
      rc = chan_source(L.incoming, "transition's incoming source", i, out_w, out_h, &op[k].src, &plane_u[k], &plane_v[k], &cm_op[k], planar);
      if (rc) return rc;
      if (L.transition == PH_TRANSITION_DISSOLVE) {
        op[k].mix = L.mix;
        op[k++].action = ph::kChanActDissolve | first;
      } else {
        op[k++].action = ph::kChanActIncoming;
        rc = chan_source(L.mask, "wipe's mask", i, out_w, out_h, &op[k].src, &plane_u[k], &plane_v[k], &cm_op[k], planar);
        if (rc) return rc;
        op[k++].action = ph::kChanActWipe | first;
      }
    } else {
      return fail(PH_E_INVALID, "ph_chan_compose_v210: layer %d: transition %d", i, L.transition);
    }
  }
  *n_ops = k;
  return PH_OK;
}

// ---- a frame of ENLARGED clips: read once, then the 2 x 2-block compositor --------------------------------------------------------------
// The reference uploads a clip at its own size and lets the Mixer's transform fill the channel (src/producer/ffmpegProducer.ts:395-442,
// mixer.ts:189-228): a 720p or SD clip on a 1080 channel, an HD clip on a 2160p one.  The channel kernel converts every TAP - four
// conversions per output pixel - where an enlarged clip has fewer pixels than the frame it fills.  When every layer of a frame is such a
// clip (v210 or a decoder's planar / packed-RGB frame, no transition, placed without rotation or mirroring, under 0.99 source texels per output pixel and written row: what
// ph_compose_up_write_v210 takes), the frame is made by the kernels that exist for exactly this: ph_v210_read(_batch) / ph_pack_read into scratch images,
// one conversion per SOURCE pixel, then ph_compose_up_write_v210 on them.  Same arithmetic, same bits (tests/test_chan_gpu.py checks both
// routes against the chain of the reference's operators); 1280 x 720 -> 1920 x 1080: 26.4 -> 17.7 us, 1080p -> 2160p: 82 -> 38 us (tools/enlarge_bench.py).
// Context option "chan_enlarged" = 0 (or PH_CHAN_ENLARGED=0): such frames through the channel kernel like any other (A/B runs, tests of that path).
// v210_fill: v210 frames under the default fill count too - only worth it when several channels' frames of one shape share the launches
// (one alone: 22.3 us by the channel kernel against 23.5; four to a call 18.7 against 15.6): ph_chan_compose_batch asks that way
static bool chan_layers_enlarged(int n, const ph_chan_layer *layers, uint32_t out_w, uint32_t out_h, uint32_t interlace, bool v210_fill = false) {
  if (out_w % 2u) return false;
  for (int i = 0; i < n; ++i) {
    const ph_chan_layer &L = layers[i];
    const float *m = L.src.matrix9_host;
    // (v210 frames, a file decoder's planar frames, packed 8-bit RGB: whatever has a reader of its own - ph_v210_read, ph_pack_read)
    // - and finished f32 images, which the compositor takes as they are
    if (L.transition != PH_TRANSITION_CUT || L.src.format < PH_SRC_V210 || L.src.format > PH_SRC_BGRA8 || !m || L.src.width <= 0 ||
        L.src.height <= 0 || ((L.src.width & 1) && L.src.format != PH_SRC_RGBA8 && L.src.format != PH_SRC_BGRA8 && L.src.format != PH_SRC_RGBA_F32))
      return false;
    if (m[1] != 0.0f || m[3] != 0.0f || !(m[0] > 0.0f) || !(m[4] > 0.0f)) return false;                      // (ph_kernels_up.hip compose_up_eligible)
    // ... or a decoder's frame of the channel's size under the Mixer's default fill (the compositor takes exactly that placement beside the
    // enlargements: compose_up_eligible): a 1080p yuv420p clip 26.1 -> 22.6 us, under a bgra8 graphic 45.2 -> 40.2.  Not v210 frames: the channel
    // kernel's shared taps serve those better (22.3 against 23.2 us)
    const bool fill = (v210_fill || L.src.format != PH_SRC_V210) && !interlace && (uint32_t)L.src.width == out_w && (uint32_t)L.src.height == out_h && m[0] == 1.0f && m[4] == 1.0f &&
                      m[2] == 0.0f && m[5] == 0.0f;
    if (!fill && ((double)m[0] * L.src.width > 0.99 * out_w || (double)m[4] * L.src.height * (interlace ? 2 : 1) > 0.99 * out_h)) return false;
    if ((uint64_t)L.src.width * 16u * (uint64_t)L.src.height >= (1ull << 30) || L.src.width >= (1 << 22)) return false;
  }
  return true;
}
// two such frames of one shape (layer count, clip sizes, placements): one launch of the compositor can make both
static bool chan_enlarged_same_shape(int n, const ph_chan_layer *a, const ph_chan_layer *b) {
  for (int i = 0; i < n; ++i) {
    if (a[i].src.width != b[i].src.width || a[i].src.height != b[i].src.height) return false;
    for (int k = 0; k < 9; ++k)
      if (a[i].src.matrix9_host[k] != b[i].src.matrix9_host[k]) return false;
  }
  return true;
}
// `jobs` frames of one shape and field mode (1 .. kMaxUpJobs, different outputs): the clips read - as few launches as their sizes allow -
// and ONE compositor launch (ph_compose_up_write_v210_batch)
static int chan_compose_enlarged(ph_ctx *ctx, int queue, int jobs, int n, const ph_chan_layer *const *layers, void *const *outs, uint32_t out_w, uint32_t out_h,
                                 uint32_t interlace, const void *rd_cm, const void *rd_lut, const void *rd_gm, const void *wr_cm, const void *wr_lut) {
  // ONE frame whose layers are all clips in their wire formats: reader and compositor in one launch (ph_kernels_up.hip clip_up_write_v210_kernel) -
  // no image leaves the chip.  (Several frames of one shape: the batched reads + one compositor launch below.)
  // Measured (tools/enlarge_bench.py, profiles/r06_clip_up.txt): one decoder's frame 22.5 -> 19.3 us (1080p yuv420p under the default fill), 18.0 -> 17.3
  // (720p filling 1080p); a v210 clip and frames of several layers are not faster this way (two layers 23.7 -> 24.1 us, four 1080 layers on
  // 2160p 95 -> 110: the conversions of a tile's rim are done twice) and keep the two launches.
  // (several frames of one shape - several channels' file playback - share the launch: a tile belongs to one frame)
  if (jobs <= ph::kMaxUpJobs && n == 1 && ctx->chan_enlarged == 1) {
    bool wire = true, alpha = false;
    for (int j = 0; j < jobs && wire; ++j) {
      const ph_chan_source &S = layers[j][0].src;
      wire = S.format != PH_SRC_RGBA_F32 && S.format != PH_SRC_V210 && !(((S.format == PH_SRC_YUV420P || S.format == PH_SRC_NV12) && (S.height & 1)));
      alpha = alpha || S.format == PH_SRC_RGBA8 || S.format == PH_SRC_BGRA8;
    }
    const LutRef rref = lds_view(ctx, rd_lut), wref = lds_view(ctx, wr_lut);
    if (wire && rref.found && wref.found) {
      ph::ClipUpArgs c{};
      ph::UpArgs &a = c.up;
      a.n = 1, a.jobs = (uint32_t)jobs, a.out = outs[0], a.out_w = out_w, a.out_h = out_h;
      for (int j = 1; j < jobs; ++j) a.more_out[j - 1] = outs[j];
      a.line_step = interlace ? 2 : 1, a.first_line = (interlace == 3) ? 1 : 0, a.lines = interlace ? out_h / 2 : out_h;
      a.wr_cm = (const float *)wr_cm, a.wr = wref.view;
      c.rd = rref.view, c.rd_gm = (const float *)rd_gm;
      {
        const ph_chan_source &S = layers[0][0].src;  // (every job's clip has this size and placement: chan_enlarged_same_shape)
        a.layer[0].w = (uint32_t)S.width, a.layer[0].h = (uint32_t)S.height, a.layer[0].pitch = (uint32_t)S.width * 16u;
        for (int k = 0; k < 6; ++k) a.layer[0].m[k] = S.matrix9_host[k];
      }
      for (int j = 0; j < jobs; ++j) {
        const ph_chan_source &S = layers[j][0].src;
        const int fmt = PH_FMT_YUV422P10 + (S.format - PH_SRC_YUV422P10);
        c.src[j] = ph::ClipSrc{S.data, S.data_u, S.data_v, (const float *)(S.col_matrix12 ? S.col_matrix12 : rd_cm), (uint32_t)fmt, ph::pack_pitch(fmt, (uint32_t)S.width)};
      }
      if (a.lines && ph::compose_up_eligible(a)) {
        const uint32_t grid = ph::clip_up_plan(c, !alpha, (uint32_t)ctx->props.multiProcessorCount);
        if (grid) {
          std::lock_guard<std::mutex> scratch(ctx->chan_scratch_mu[queue]);  // until the launch is enqueued
          {
            std::lock_guard<std::mutex> lock(ctx->mu);
            int rc = chan_index_reserve(ctx, queue, (size_t)c.wg_bytes * grid + 4096u);
            if (rc) return rc;
            c.scratch = (char *)ctx->chan_index[queue];
          }
          hipError_t e = ph::launch_clip_up_write_v210(stream_of(ctx, queue), c, !alpha, grid);
          if (e != hipSuccess) return fail(PH_E_HIP, "ph_chan_compose_v210: launch failed: %s", hipGetErrorString(e));
          return PH_OK;
        }
      }
    }
  }
  // the images live in the channel compositor's scratch area of the queue (launches on one queue are in order)
  size_t off[ph::kMaxLayers], per_job = 0;
  bool one_size = true;
  for (int i = 0; i < n; ++i) {
    off[i] = per_job;
    per_job += ((size_t)layers[0][i].src.width * layers[0][i].src.height * 16u + 255u) & ~(size_t)255u;
    one_size = one_size && layers[0][i].src.width == layers[0][0].src.width && layers[0][i].src.height == layers[0][0].src.height;
  }
  const size_t total = per_job * (size_t)jobs;
  // Three sets of images taken in turn: a frame's read kernel overwriting the very lines the previous frame's compositor has just read
  // (still cached, in several XCDs' L2s) measured 11.7 us against 7.9 us into lines nobody holds (tools/enlarge_bench.py, PH_ENLARGE_RING)
  // - while the sets together stay inside the 256 MB of last-level cache: four 1080p images (132 MB) in three sets measured 109 us per
  // 2160p frame against 94 in one set that stays cached
  const unsigned kTurns = total <= ((size_t)64 << 20) ? 3u : 1u;
  std::lock_guard<std::mutex> scratch(ctx->chan_scratch_mu[queue]);  // until the compositor's launch is enqueued
  char *base;
  {
    std::lock_guard<std::mutex> lock(ctx->mu);
    int rc = chan_index_reserve(ctx, queue, total * kTurns);
    if (rc) return rc;
    base = (char *)ctx->chan_index[queue] + total * (ctx->chan_scratch_turn[queue]++ % kTurns);
  }
  const void *ins[ph::kMaxUpJobs * ph::kMaxLayers];
  void *imgs[ph::kMaxUpJobs * ph::kMaxLayers];
  ph_image_layer il[ph::kMaxUpJobs][ph::kMaxLayers];
  const ph_image_layer *sets[ph::kMaxUpJobs];
  for (int j = 0; j < jobs; ++j) {
    for (int i = 0; i < n; ++i) {
      const ph_chan_source &S = layers[j][i].src;
      ins[j * n + i] = S.data, imgs[j * n + i] = S.format == PH_SRC_RGBA_F32 ? const_cast<void *>(S.data) : base + per_job * (size_t)j + off[i];
      il[j][i] = ph_image_layer{imgs[j * n + i], PH_IMG_RGBA_F32, S.width, S.height, S.matrix9_host};
    }
    sets[j] = il[j];
  }
  int rc = PH_OK;
  const int frames = jobs * n;
  bool all_v210 = true;
  for (int f = 0; f < frames; ++f) all_v210 = all_v210 && layers[f / n][f % n].src.format == PH_SRC_V210;
  bool one_reader = !all_v210 && one_size && frames > 1;  // every frame a decoder's frame of ONE format, size and Loader matrix: one launch reads them all
  for (int f = 0; f < frames && one_reader; ++f) {
    const ph_chan_source &S = layers[f / n][f % n].src, &S0 = layers[0][0].src;
    one_reader = S.format == S0.format && S.format > PH_SRC_RGBA_F32 && S.col_matrix12 == S0.col_matrix12;
  }
  if (one_reader) {
    const void *planes[ph::kMaxUpJobs * ph::kMaxLayers][3];
    for (int f = 0; f < frames; ++f) {
      const ph_chan_source &S = layers[f / n][f % n].src;
      planes[f][0] = S.data, planes[f][1] = S.data_u, planes[f][2] = S.data_v;
    }
    const ph_chan_source &S0 = layers[0][0].src;
    for (int f = 0; f < frames && rc == PH_OK; f += ph::kMaxLayers)
      rc = ph_pack_read_batch(ctx, queue, PH_FMT_YUV422P10 + (S0.format - PH_SRC_YUV422P10), frames - f < ph::kMaxLayers ? frames - f : ph::kMaxLayers, planes + f, imgs + f,
                              (uint32_t)S0.width, (uint32_t)S0.height, S0.col_matrix12 ? S0.col_matrix12 : rd_cm, rd_lut, rd_gm);
  } else if (!all_v210) {  // each clip through the reader of its format (a source with code ranges of its own brings its Loader matrix)
    for (int f = 0; f < frames && rc == PH_OK; ++f) {
      const ph_chan_source &S = layers[f / n][f % n].src;
      if (S.format == PH_SRC_RGBA_F32) continue;  // an image: nothing to read
      if (S.format == PH_SRC_V210) {
        rc = ph_v210_read(ctx, queue, S.data, imgs[f], (uint32_t)S.width, (uint32_t)S.height, rd_cm, rd_lut, rd_gm);
      } else {
        const void *planes[3] = {S.data, S.data_u, S.data_v};
        rc = ph_pack_read(ctx, queue, PH_FMT_YUV422P10 + (S.format - PH_SRC_YUV422P10), planes, imgs[f], (uint32_t)S.width, (uint32_t)S.height,
                          S.col_matrix12 ? S.col_matrix12 : rd_cm, rd_lut, rd_gm);
      }
    }
  } else if (one_size && frames > 1) {
    for (int f = 0; f < frames && rc == PH_OK; f += ph::kMaxLayers)
      rc = ph_v210_read_batch(ctx, queue, frames - f < ph::kMaxLayers ? frames - f : ph::kMaxLayers, ins + f, imgs + f, (uint32_t)layers[0][0].src.width,
                              (uint32_t)layers[0][0].src.height, rd_cm, rd_lut, rd_gm);
  } else {
    for (int f = 0; f < frames && rc == PH_OK; ++f)
      rc = ph_v210_read(ctx, queue, ins[f], imgs[f], (uint32_t)layers[f / n][f % n].src.width, (uint32_t)layers[f / n][f % n].src.height, rd_cm, rd_lut, rd_gm);
  }
  if (rc) return rc;
  return ph_compose_up_write_v210_batch(ctx, queue, jobs, n, sets, outs, out_w, out_h, interlace, wr_cm, wr_lut);
}

int ph_chan_compose_v210(ph_ctx *ctx, int queue, int n, const ph_chan_layer *layers, void *out, uint32_t out_w, uint32_t out_h,
                         uint32_t interlace, const void *rd_cm, const void *rd_lut, const void *rd_gm, const void *wr_cm,
                         const void *wr_lut) {
  void *planes[3] = {out, nullptr, nullptr};
  return ph_chan_compose(ctx, queue, n, layers, PH_FMT_V210, planes, out_w, out_h, interlace, rd_cm, rd_lut, rd_gm, wr_cm, wr_lut);
}

int ph_chan_compose(ph_ctx *ctx, int queue, int n, const ph_chan_layer *layers, int out_format, void *const out_planes[3], uint32_t out_w,
                    uint32_t out_h, uint32_t interlace, const void *rd_cm, const void *rd_lut, const void *rd_gm, const void *wr_cm,
                    const void *wr_lut) {
  if (!ctx || !layers || !out_planes || !out_planes[0] || !rd_cm || !rd_lut || !rd_gm || !wr_lut) return fail(PH_E_INVALID, "ph_chan_compose_v210: NULL argument");
  void *const out = out_planes[0];
  const bool out_rgb8 = out_format == PH_FMT_RGBA8 || out_format == PH_FMT_BGRA8, out_420 = out_format == PH_FMT_YUV420P || out_format == PH_FMT_NV12;
  const bool out_planar = out_format == PH_FMT_YUV422P10 || out_format == PH_FMT_YUV422P8 || out_420;
  if (out_format != PH_FMT_V210 && !out_rgb8 && !out_planar)
    return fail(PH_E_INVALID, "ph_chan_compose_v210: output format %d is not a PH_FMT_*", out_format);
  if (!out_rgb8 && !wr_cm) return fail(PH_E_INVALID, "ph_chan_compose_v210: the writer's RGB -> YCbCr matrix is missing");
  if (out_planar && (!out_planes[1] || (out_format != PH_FMT_NV12 && !out_planes[2])))
    return fail(PH_E_INVALID, "ph_chan_compose_v210: a planar output needs its three planes (nv12: Y and the interleaved CbCr plane)");
  if (out_420 && (out_h & 1)) return fail(PH_E_INVALID, "ph_chan_compose_v210: a 4:2:0 frame needs an even height (%u)", out_h);
  PH_QUEUE("ph_chan_compose_v210", queue);
  if (n < 1 || n > ph::kMaxLayers) return fail(PH_E_INVALID, "ph_chan_compose_v210: 1..%d layers", ph::kMaxLayers);
  // A v210 line of a width that is not a multiple of 48 (1280 x 720, src/config.ts:43-54) ends in a padded block: whole quads, the
  // tail quad with the reference's tail arithmetic (v210.ts:166-193), cleared slots - lines addressed by pitch.
  // The planar writers here take whole groups of eight pixels; rgba8 / bgra8 any width.
  if (!out_w || (out_format == PH_FMT_V210 && (out_w & 1)) || (out_planar && out_w % 8))
    return fail(PH_E_INVALID, "ph_chan_compose_v210: width %u (a v210 frame needs an even width, a planar one a multiple of 8); run the separate kernels", out_w);
  if (interlace != 0 && interlace != 1 && interlace != 3) return fail(PH_E_INVALID, "ph_chan_compose_v210: interlace must be 0, 1 or 3");
  const LutRef rref = lds_view(ctx, rd_lut), wref = lds_view(ctx, wr_lut);
  const ph::LutView *rv = rref.get(), *wv = wref.get();
  if (!rv || !wv)
    return fail(PH_E_INVALID, "ph_chan_compose_v210: the %s gamma LUT has no LDS form (ph_lut_register it, or run the separate kernels)", rv ? "writer" : "reader");
  ph::ChanArgs a{};
  int k = 0;
  int rc_ops = chan_ops(n, layers, out_w, out_h, a.op, a.plane_u, a.plane_v, a.cm_op, &a.planar, &k);
  if (rc_ops) return rc_ops;
  a.n_ops = k;
  a.out = out, a.out_w = out_w, a.out_h = out_h;
  a.line_step = interlace ? 2 : 1, a.first_line = (interlace == 3) ? 1 : 0;
  a.lines = interlace ? out_h / 2 : out_h;
  a.rd_cm = (const float *)rd_cm, a.rd_gm = (const float *)rd_gm, a.wr_cm = (const float *)(wr_cm ? wr_cm : rd_cm), a.rd = *rv, a.wr = *wv;
  a.out_fmt = (uint32_t)out_format, a.out_u = out_planes[1], a.out_v = out_planes[2];
  a.out_pitch = out_w;  // planar: width rounded up to 8 samples (yuv422p10.ts:221) - widths here are multiples of 8; rgba8: no padding
  a.out_qpitch = ph_v210_pitch_bytes(out_w) / 16u;
  a.out_tail_from = 0xFFFFFFFFu;
  if (out_format == PH_FMT_V210 && out_w % 48) {
    if (a.planar < 1) a.planar = 1;  // the tail instantiation writes lines that end in a tail / cleared slots
    if (out_w % 6) a.out_tail_from = out_w - out_w % 6u;
  }
  if (!a.lines) return PH_OK;
  int rc = set_device(ctx);
  if (rc) return rc;
  if (out_format == PH_FMT_V210 && wr_cm && ctx->chan_enlarged && chan_layers_enlarged(n, layers, out_w, out_h, interlace))
    return chan_compose_enlarged(ctx, queue, 1, n, &layers, &out, out_w, out_h, interlace, rd_cm, rd_lut, rd_gm, wr_cm, wr_lut);
  // the index frame between the phases: one per queue (launches on one queue are in order), grown on demand
  std::lock_guard<std::mutex> scratch(ctx->chan_scratch_mu[queue]);
  {
    std::lock_guard<std::mutex> lock(ctx->mu);
    rc = chan_index_reserve(ctx, queue, ph::chan_index_bytes(out_w, a.lines));
    if (rc) return rc;
    a.index = ctx->chan_index[queue];
  }
  hipError_t e = ph::launch_chan_compose_v210(stream_of(ctx, queue), a, (uint32_t)ctx->props.multiProcessorCount);
  if (e != hipSuccess) return fail(PH_E_HIP, "ph_chan_compose_v210: launch failed: %s", hipGetErrorString(e));
  return PH_OK;
}

/* Several channels' frames in one launch: see include/phaneron_hip.h.  Jobs the batch kernel does not take (planar / packed-RGB sources,
 * more ops or wave steps than one launch holds) run through ph_chan_compose_v210 in their turn, so the call as a whole is always the
 * `n_jobs` separate calls it stands for. */
int ph_chan_compose_batch(ph_ctx *ctx, int queue, int n_jobs, const ph_chan_job *jobs, uint32_t out_w, uint32_t out_h, const void *rd_cm,
                          const void *rd_lut, const void *rd_gm, const void *wr_cm, const void *wr_lut) {
  if (!ctx || !jobs || !rd_cm || !rd_lut || !rd_gm || !wr_cm || !wr_lut) return fail(PH_E_INVALID, "ph_chan_compose_batch: NULL argument");
  PH_QUEUE("ph_chan_compose_batch", queue);
  if (n_jobs < 1) return fail(PH_E_INVALID, "ph_chan_compose_batch: no jobs");
  if (!out_w || (out_w & 1)) return fail(PH_E_INVALID, "ph_chan_compose_batch: width %u (a v210 frame needs an even width)", out_w);
  const LutRef rref = lds_view(ctx, rd_lut), wref = lds_view(ctx, wr_lut);
  const ph::LutView *rv = rref.get(), *wv = wref.get();
  if (!rv || !wv)
    return fail(PH_E_INVALID, "ph_chan_compose_batch: the %s gamma LUT has no LDS form (ph_lut_register it, or run the separate kernels)", rv ? "writer" : "reader");
  for (int j = 0; j < n_jobs; ++j) {
    const ph_chan_job &J = jobs[j];
    if (!J.layers || !J.out || J.n < 1 || J.n > ph::kMaxLayers) return fail(PH_E_INVALID, "ph_chan_compose_batch: job %d: 1..%d layers and an output", j, ph::kMaxLayers);
    if (J.interlace != 0 && J.interlace != 1 && J.interlace != 3) return fail(PH_E_INVALID, "ph_chan_compose_batch: job %d: interlace must be 0, 1 or 3", j);
  }
  const ph::LutView &rview = rref.view, &wview = wref.view;
  int rc = set_device(ctx);
  if (rc) return rc;
  // the common part of a launch's arguments (ChanArgs carries it to chan_batch_launch)
  auto common = [&](uint32_t interlace) {
    ph::ChanArgs a{};
    a.out_w = out_w, a.out_h = out_h;
    a.line_step = interlace ? 2 : 1, a.lines = interlace ? out_h / 2 : out_h;
    a.rd_cm = (const float *)rd_cm, a.rd_gm = (const float *)rd_gm, a.wr_cm = (const float *)wr_cm, a.rd = rview, a.wr = wview;
    a.out_qpitch = ph_v210_pitch_bytes(out_w) / 16u;
    a.out_tail_from = out_w % 6 ? out_w - out_w % 6u : 0xFFFFFFFFu;
    a.planar = out_w % 48 ? 1u : 0u;
    return a;
  };
  // How many jobs a launch should take so that the launches come out even (eight jobs of six ops are 4 + 4, not 6 + 2: a launch of
  // two is a poor one): by the jobs and ops of the call as a whole - a plan, the limits below still hold for every launch
  uint32_t plan_jobs = 0, plan_ops = 0;
  for (int j = 0; j < n_jobs; ++j) {
    uint32_t ops = 0;  // (planar / packed-RGB sources share launches too since round 6: the batch kernel's PLANAR instantiation)
    for (int l = 0; l < jobs[j].n; ++l) {
      const ph_chan_layer &L = jobs[j].layers[l];
      ops += L.transition == PH_TRANSITION_WIPE ? 3u : L.transition == PH_TRANSITION_DISSOLVE ? 2u : 1u;
    }
    if (ops <= (uint32_t)ph::kMaxChanBatchOps && !(ctx->chan_enlarged && chan_layers_enlarged(jobs[j].n, jobs[j].layers, out_w, out_h, jobs[j].interlace)))
      ++plan_jobs, plan_ops += ops;
  }
  const uint32_t by_jobs = (plan_jobs + (uint32_t)ph::kMaxChanJobs - 1u) / (uint32_t)ph::kMaxChanJobs;
  const uint32_t by_ops = (plan_ops + (uint32_t)ph::kMaxChanBatchOps - 1u) / (uint32_t)ph::kMaxChanBatchOps;
  const uint32_t launches = by_jobs > by_ops ? by_jobs : by_ops;
  const uint32_t jobs_per_launch = launches ? (plan_jobs + launches - 1u) / launches : 1u;
  ph::ChanBatchArgs b{};
  ph::ChanArgs a = common(0);
  bool fields = false;
  int first_job = 0;  // of the launch being collected
  // frames of enlarged clips among the jobs (chan_compose_enlarged): those of one shape and field mode, one after the other, are made together
  const ph_chan_layer *enl_layers[ph::kMaxUpJobs];
  void *enl_outs[ph::kMaxUpJobs];
  int enl = 0, enl_n = 0;
  uint32_t enl_interlace = 0;
  auto flush_enlarged = [&]() -> int {
    if (!enl) return PH_OK;
    const int r = chan_compose_enlarged(ctx, queue, enl, enl_n, enl_layers, enl_outs, out_w, out_h, enl_interlace, rd_cm, rd_lut, rd_gm, wr_cm, wr_lut);
    enl = 0;
    return r;
  };
  auto flush = [&]() -> int {
    if (!b.jobs) return PH_OK;
    int r;
    if (b.jobs == 1) {  // nothing to share: the one-job kernel
      const ph_chan_job &J = jobs[first_job];
      r = ph_chan_compose_v210(ctx, queue, J.n, J.layers, J.out, out_w, out_h, J.interlace, rd_cm, rd_lut, rd_gm, wr_cm, wr_lut);
    } else {
      r = chan_batch_launch(ctx, queue, b, a);
    }
    b = ph::ChanBatchArgs{};
    return r;
  };
  for (int j = 0; j < n_jobs; ++j) {
    const ph_chan_job &J = jobs[j];
    ph::ChanArgs one{};  // this job's program, checked as the single call checks it
    int k = 0;
    rc = chan_ops(J.n, J.layers, out_w, out_h, one.op, one.plane_u, one.plane_v, one.cm_op, &one.planar, &k);
    if (rc) return rc;
    const bool is_field = J.interlace != 0;
    const uint32_t lines = is_field ? out_h / 2 : out_h;
    const uint32_t fit = chan_batch_on() ? jobs_per_launch : 0u;
    bool routed = lines && ctx->chan_enlarged && chan_layers_enlarged(J.n, J.layers, out_w, out_h, J.interlace);  // read + 2 x 2-block compositor, in its turn
    bool fits = false;
    auto fits_group = [&]() {
      bool f = enl > 0 && enl < ph::kMaxUpJobs && J.n == enl_n && J.interlace == enl_interlace && chan_enlarged_same_shape(J.n, enl_layers[0], J.layers);
      for (int e = 0; e < enl && f; ++e) f = enl_outs[e] != J.out;
      return f;
    };
    // (frames of v210 clips ONLY: with a packed-RGB graphic on top the images carry alpha and the route loses to the channel kernel - 42 against 34 us)
    bool all_v210 = true;
    for (int l = 0; l < J.n; ++l) all_v210 = all_v210 && J.layers[l].src.format == PH_SRC_V210;
    if (!routed && all_v210 && lines && ctx->chan_enlarged && chan_layers_enlarged(J.n, J.layers, out_w, out_h, J.interlace, true)) {
      // v210 frames under the default fill: by the route if they share its launches with a neighbour of their shape, else the batch kernel's
      const bool with_next = j + 1 < n_jobs && jobs[j + 1].n == J.n && jobs[j + 1].interlace == J.interlace && jobs[j + 1].out != J.out && jobs[j + 1].layers &&
                             chan_layers_enlarged(jobs[j + 1].n, jobs[j + 1].layers, out_w, out_h, jobs[j + 1].interlace, true) &&
                             chan_enlarged_same_shape(J.n, J.layers, jobs[j + 1].layers);
      routed = fits_group() || with_next;
    }
    if (routed) {
      if ((rc = flush())) return rc;
      fits = fits_group();
      if (!fits && (rc = flush_enlarged())) return rc;
      enl_layers[enl] = J.layers, enl_outs[enl] = J.out, enl_n = J.n, enl_interlace = J.interlace, ++enl;
      continue;
    }
    if ((rc = flush_enlarged())) return rc;
    // a planar job's Loader matrices as indices into the launch's small table (those already there are found again)
    uint8_t cm_of[ph::kMaxChanOps];
    const float *cm_new[8];
    int n_cm_new = 0;
    bool cm_fits = true;
    auto cm_known = [&](const float *cm) -> int {  // 1 .. 8, or 0
      for (uint32_t t = 0; t < 8 && b.cm_tab[t]; ++t)
        if (b.cm_tab[t] == cm) return (int)t + 1;
      return 0;
    };
    uint32_t tab_used = 0;
    while (tab_used < 8 && b.cm_tab[tab_used]) ++tab_used;
    for (int i = 0; i < k && one.planar == 2; ++i) {
      cm_of[i] = 0;
      const float *cm = one.cm_op[i];
      if (!cm || cm == (const float *)rd_cm) continue;
      int at = cm_known(cm);
      for (int t = 0; t < n_cm_new && !at; ++t)
        if (cm_new[t] == cm) at = (int)tab_used + t + 1;
      if (!at) {
        if (tab_used + (uint32_t)n_cm_new >= 8u) { cm_fits = false; break; }
        cm_new[n_cm_new] = cm, at = (int)tab_used + (++n_cm_new);
      }
      cm_of[i] = (uint8_t)at;
    }
    if ((one.planar == 2 && !cm_fits) || k > ph::kMaxChanBatchOps || fit < 1 || !lines) {  // not for the batch kernel: in its turn, on its own
      if ((rc = flush())) return rc;
      if ((rc = ph_chan_compose_v210(ctx, queue, J.n, J.layers, J.out, out_w, out_h, J.interlace, rd_cm, rd_lut, rd_gm, wr_cm, wr_lut))) return rc;
      continue;
    }
    // a launch holds jobs of one kind (frames or fields: the same lines), no two of which write the same lines of one frame
    bool clash = false;
    for (uint32_t i = 0; i < b.jobs; ++i)
      clash = clash || (b.job[i].out == J.out && (!is_field || b.job[i].first_line == (J.interlace == 3 ? 1u : 0u)));
    if (b.jobs && (fields != is_field || clash || b.jobs >= fit || b.n_ops + (uint32_t)k > (uint32_t)ph::kMaxChanBatchOps))
      if ((rc = flush())) return rc;
    if (!b.jobs) a = common(J.interlace), fields = is_field, first_job = j;
    if (one.planar > a.planar) a.planar = one.planar;  // a source whose lines end in a tail
    ph::ChanJob &jb = b.job[b.jobs];
    jb.out = J.out, jb.first_op = b.n_ops, jb.n_ops = (uint32_t)k, jb.first_line = J.interlace == 3 ? 1u : 0u;
    for (int i = 0; i < k; ++i) b.op[b.n_ops + i] = one.op[i], b.op_job[b.n_ops + i] = (uint8_t)b.jobs;
    if (one.planar == 2) {
      // (a flush above has emptied the launch: the table then starts again - indices made against the old one are made again)
      if (!b.cm_tab[0] && tab_used) {
        int renum = 0;
        for (int i = 0; i < k; ++i)
          if (cm_of[i]) {
            const float *cm = one.cm_op[i];
            int at = 0;
            for (int t = 0; t < renum && !at; ++t)
              if (b.cm_tab[t] == cm) at = t + 1;
            if (!at) b.cm_tab[renum] = cm, at = ++renum;
            cm_of[i] = (uint8_t)at;
          }
      } else {
        for (int t = 0; t < n_cm_new; ++t) b.cm_tab[tab_used + (uint32_t)t] = cm_new[t];
      }
      for (int i = 0; i < k; ++i) {
        b.plane_u[b.n_ops + i] = one.plane_u[i], b.plane_v[b.n_ops + i] = one.plane_v[i], b.cm_idx[b.n_ops + i] = cm_of[i];
        b.any_cm |= cm_of[i] ? 1u : 0u;
      }
    }
    b.n_ops += (uint32_t)k, ++b.jobs;
  }
  if ((rc = flush_enlarged())) return rc;
  return flush();
}

int ph_yadif(ph_ctx *ctx, int queue, const void *prev, const void *cur, const void *next, int w, int h, int parity,
             int tff, int skip, void *out) {
  if (!prev || !cur || !next || !out || w <= 0 || h <= 0) return fail(PH_E_INVALID, "ph_yadif: NULL/zero argument");
  PH_LAUNCH(ph::launch_yadif(stream_of(ctx, queue), prev, cur, next, w, h, parity, tff ? 1 : 0, skip ? 1 : 0, out));
}

int ph_yadif_pair(ph_ctx *ctx, int queue, const void *prev, const void *cur, const void *next, int w, int h, int tff,
                  int skip, void *out_parity0, void *out_parity1) {
  if (!prev || !cur || !next || !out_parity0 || !out_parity1 || w <= 0 || h <= 0) return fail(PH_E_INVALID, "ph_yadif_pair: NULL/zero argument");
  if (out_parity0 == out_parity1) return fail(PH_E_INVALID, "ph_yadif_pair: the two outputs are the same buffer");
  PH_LAUNCH(ph::launch_yadif_pair(stream_of(ctx, queue), prev, cur, next, w, h, tff ? 1 : 0, skip ? 1 : 0, out_parity0, out_parity1));
}

int ph_v210_yadif_pair(ph_ctx *ctx, int queue, int n, const ph_deint_source *src, uint32_t width, uint32_t height, int tff,
                       int skip, const void *cm, const void *lut, const void *gm) {
  return ph_v210_yadif_pair_fmt(ctx, queue, n, src, width, height, tff, skip, PH_IMG_RGBA_F32, cm, lut, gm);
}

// `jobs` sets of layers that differ in their data only (same count, formats, sizes and placements), each into its own output: one launch
static int compose_up_common(const char *fn, ph_ctx *ctx, int queue, int jobs, int n, const ph_image_layer *const *sets, void *const *outs,
                             uint32_t out_w, uint32_t out_h, uint32_t interlace, const void *wr_cm, const void *wr_lut) {
  if (!ctx || !sets || !outs || !wr_cm || !wr_lut) return fail(PH_E_INVALID, "%s: NULL argument", fn);
  if (jobs < 1 || jobs > ph::kMaxUpJobs) return fail(PH_E_INVALID, "%s: 1..%d jobs", fn, ph::kMaxUpJobs);
  for (int j = 0; j < jobs; ++j)
    if (!sets[j] || !outs[j]) return fail(PH_E_INVALID, "%s: NULL argument", fn);
  const ph_image_layer *layers = sets[0];
  if (n < 1 || n > ph::kMaxLayers) return fail(PH_E_INVALID, "%s: 1..%d layers", fn, ph::kMaxLayers);
  // (a width that is not a multiple of 48 - 1280 - ends its lines in a tail quad and cleared slots: the kernel's TAILS instantiation)
  if (!out_w || out_w % 2) return fail(PH_E_INVALID, "%s: width %u is odd (a v210 frame needs an even width); run the separate kernels", fn, out_w);
  if (interlace != 0 && interlace != 1 && interlace != 3) return fail(PH_E_INVALID, "%s: interlace must be 0, 1 or 3", fn);
  for (int j = 1; j < jobs; ++j)
    for (int k = 0; k < j; ++k)
      if (outs[j] == outs[k]) return fail(PH_E_INVALID, jobs == 2 ? "%s: the two outputs are the same buffer" : "%s: two jobs have the same output buffer", fn);
  const LutRef wref = lds_view(ctx, wr_lut);
  const ph::LutView *wv = wref.get();
  if (!wv) return fail(PH_E_INVALID, "%s: the writer gamma LUT has no LDS form (ph_lut_register it, or run the separate kernels)", fn);
  ph::UpArgs a{};
  a.n = n;
  const int fmt = layers[0].format;
  if (fmt != PH_IMG_RGBA_F32 && fmt != PH_IMG_RGB_F32) return fail(PH_E_INVALID, "%s: image format %d", fn, fmt);
  for (int i = 0; i < n; ++i) {
    const ph_image_layer &L = layers[i];
    if (!L.data || L.width <= 0 || L.height <= 0 || !L.matrix9_host) return fail(PH_E_INVALID, "%s: layer %d is incomplete", fn, i);
    if (L.format != fmt) return fail(PH_E_INVALID, "%s: layer %d has another image format than layer 0", fn, i);
    a.layer[i].ptr = L.data, a.layer[i].w = (uint32_t)L.width, a.layer[i].h = (uint32_t)L.height;
    a.layer[i].pitch = (uint32_t)L.width * (fmt == PH_IMG_RGB_F32 ? 12u : 16u);
    for (int k = 0; k < 6; ++k) a.layer[i].m[k] = L.matrix9_host[k];
    for (int j = 1; j < jobs; ++j) {
      const ph_image_layer &B = sets[j][i];
      if (!B.data || !B.matrix9_host || B.format != L.format || B.width != L.width || B.height != L.height)
        return fail(PH_E_INVALID, jobs == 2 ? "%s: layer %d of the second set differs from the first in more than its data" : "%s: layer %d of a further set differs from the first in more than its data", fn, i);
      for (int k = 0; k < 9; ++k)
        if (B.matrix9_host[k] != L.matrix9_host[k])
          return fail(PH_E_INVALID, jobs == 2 ? "%s: layer %d of the second set is placed differently" : "%s: layer %d of a further set is placed differently", fn, i);
      a.more_ptr[j - 1][i] = B.data;
    }
  }
  a.out = outs[0], a.jobs = (uint32_t)jobs, a.out_w = out_w, a.out_h = out_h;
  for (int j = 1; j < jobs; ++j) a.more_out[j - 1] = outs[j];
  a.line_step = interlace ? 2 : 1, a.first_line = (interlace == 3) ? 1 : 0;
  a.lines = interlace ? out_h / 2 : out_h;
  a.wr_cm = (const float *)wr_cm, a.wr = *wv;
  if (!ph::compose_up_eligible(a))
    return fail(PH_E_INVALID, "%s: every layer must be enlarged (by 1 %% or more in both directions, per written row) without "
                              "rotation or mirroring and be below 1 GiB; use ph_compose_write_v210", fn);
  if (!a.lines) return PH_OK;
  PH_LAUNCH(ph::launch_compose_up_write_v210(stream_of(ctx, queue), a, fmt == PH_IMG_RGB_F32, (uint32_t)ctx->props.multiProcessorCount));
}

int ph_compose_up_write_v210(ph_ctx *ctx, int queue, int n, const ph_image_layer *layers, void *out, uint32_t out_w, uint32_t out_h,
                             uint32_t interlace, const void *wr_cm, const void *wr_lut) {
  if (!layers || !out) return fail(PH_E_INVALID, "ph_compose_up_write_v210: NULL argument");
  return compose_up_common("ph_compose_up_write_v210", ctx, queue, 1, n, &layers, &out, out_w, out_h, interlace, wr_cm, wr_lut);
}

int ph_compose_up_write_v210_pair(ph_ctx *ctx, int queue, int n, const ph_image_layer *layers_a, const ph_image_layer *layers_b, void *out_a, void *out_b,
                                  uint32_t out_w, uint32_t out_h, uint32_t interlace, const void *wr_cm, const void *wr_lut) {
  if (!layers_a || !layers_b || !out_a || !out_b) return fail(PH_E_INVALID, "ph_compose_up_write_v210_pair: NULL argument");
  const ph_image_layer *sets[2] = {layers_a, layers_b};
  void *outs[2] = {out_a, out_b};
  return compose_up_common("ph_compose_up_write_v210_pair", ctx, queue, 2, n, sets, outs, out_w, out_h, interlace, wr_cm, wr_lut);
}

int ph_compose_up_write_v210_batch(ph_ctx *ctx, int queue, int jobs, int n, const ph_image_layer *const *layer_sets, void *const *outs, uint32_t out_w,
                                   uint32_t out_h, uint32_t interlace, const void *wr_cm, const void *wr_lut) {
  return compose_up_common("ph_compose_up_write_v210_batch", ctx, queue, jobs, n, layer_sets, outs, out_w, out_h, interlace, wr_cm, wr_lut);
}

int ph_v210_yadif_pair_fmt(ph_ctx *ctx, int queue, int n, const ph_deint_source *src, uint32_t width, uint32_t height, int tff,
                           int skip, int out_format, const void *cm, const void *lut, const void *gm) {
  return ph_yadif_pair_packed(ctx, queue, n, src, PH_FMT_V210, width, height, tff, skip, out_format, cm, lut, gm);
}

int ph_yadif_pair_packed(ph_ctx *ctx, int queue, int n, const ph_deint_source *src, int packing, uint32_t width, uint32_t height, int tff,
                         int skip, int out_format, const void *cm, const void *lut, const void *gm) {
  if (!ctx || !src || !cm || !lut || !gm) return fail(PH_E_INVALID, "ph_v210_yadif_pair: NULL argument");
  if (packing != PH_FMT_V210 && packing != PH_FMT_YUV422P10 && packing != PH_FMT_YUV422P8 && packing != PH_FMT_YUV420P && packing != PH_FMT_NV12)
    return fail(PH_E_INVALID, "ph_v210_yadif_pair: packing %d (v210 or a planar YCbCr format; run the separate kernels for the others)", packing);
  if ((packing == PH_FMT_YUV420P || packing == PH_FMT_NV12) && (height & 1))
    return fail(PH_E_INVALID, "ph_v210_yadif_pair: a 4:2:0 frame needs an even height (%u)", height);
  const bool planar = packing != PH_FMT_V210;
  if (out_format != PH_IMG_RGBA_F32 && out_format != PH_IMG_RGB_F32) return fail(PH_E_INVALID, "ph_v210_yadif_pair: output format %d", out_format);
  if (n < 1 || n > ph::kMaxLayers) return fail(PH_E_INVALID, "ph_v210_yadif_pair: 1..%d sources", ph::kMaxLayers);
  if (!width || (planar ? width % 2 : width % 6))
    return fail(PH_E_INVALID, "ph_v210_yadif_pair: width %u is not a multiple of %d; run the separate kernels", width, planar ? 2 : 6);
  const LutRef lref = lds_view(ctx, lut);
  const ph::LutView *v = lref.get();
  if (!v) return fail(PH_E_INVALID, "ph_v210_yadif_pair: the reader gamma LUT has no LDS form (ph_lut_register it, or run the separate kernels)");
  ph::DeintArgs a{};
  for (int i = 0; i < n; ++i) {
    const ph_deint_source &s = src[i];
    if (!s.prev || !s.cur || !s.next || !s.out_parity0 || !s.out_parity1)
      return fail(PH_E_INVALID, "ph_v210_yadif_pair: source %d is incomplete", i);
    if (s.out_parity0 == s.out_parity1) return fail(PH_E_INVALID, "ph_v210_yadif_pair: source %d: the two outputs are the same buffer", i);
    a.prev[i] = (const uint4 *)s.prev, a.cur[i] = (const uint4 *)s.cur, a.next[i] = (const uint4 *)s.next;
    a.out0[i] = (float4 *)s.out_parity0, a.out1[i] = (float4 *)s.out_parity1;
    if (planar) {
      if (!s.prev_u || !s.cur_u || !s.next_u || (packing != PH_FMT_NV12 && (!s.prev_v || !s.cur_v || !s.next_v)))
        return fail(PH_E_INVALID, "ph_v210_yadif_pair: source %d: a planar window needs its chroma planes", i);
      a.prev_u[i] = s.prev_u, a.prev_v[i] = s.prev_v, a.cur_u[i] = s.cur_u, a.cur_v[i] = s.cur_v, a.next_u[i] = s.next_u, a.next_v[i] = s.next_v;
    }
  }
  if (!height) return PH_OK;
  a.pack = packing == PH_FMT_V210 ? 0u : packing == PH_FMT_YUV422P10 ? 1u : packing == PH_FMT_YUV422P8 ? 2u : packing == PH_FMT_YUV420P ? 3u : 4u;
  // quads_pitch: a v210 line in 16-byte quads, or (planar) the luma samples per line: the width rounded up to 8 (yuv422p10.ts:221)
  a.n = n, a.skip = skip ? 1 : 0, a.width = width, a.height = height, a.quads_pitch = planar ? ((width + 7u) & ~7u) : ph_v210_pitch_bytes(width) / 16;
  a.rgb12 = out_format == PH_IMG_RGB_F32 ? 1u : 0u;
  a.cm = (const float *)cm, a.gm = (const float *)gm, a.lut = *v;
  PH_LAUNCH(ph::launch_v210_yadif_pair(stream_of(ctx, queue), a, tff ? 1 : 0, (uint32_t)ctx->props.multiProcessorCount));
}

/* A packed f32 RGB image (12 bytes per pixel, alpha == 1 implied: what ph_v210_yadif_pair_fmt writes with PH_IMG_RGB_F32) made the f32 RGBA
 * image of the same pixels, IN PLACE: the packed pixels are copied to the queue's scratch area, then expanded back into the buffer.
 * For a binding that lets the de-interlacing reader write packed fields into the application's RGBA image buffers while only the 2 x 2-block
 * compositor reads them, and has to hand over a real image when anybody else asks (node/defer.js). */
int ph_image_unpack_rgb(ph_ctx *ctx, int queue, void *image, int w, int h) {
  if (!ctx || !image || w <= 0 || h <= 0) return fail(PH_E_INVALID, "ph_image_unpack_rgb: NULL/zero argument");
  PH_QUEUE("ph_image_unpack_rgb", queue);
  int rc = set_device(ctx);
  if (rc) return rc;
  const size_t npx = (size_t)w * h;
  if (ph::trace_launch("rgb_unpack")) return PH_OK;
  std::lock_guard<std::mutex> scratch(ctx->chan_scratch_mu[queue]);  // until both operations are enqueued (launches on one queue are in order)
  void *tmp;
  {
    std::lock_guard<std::mutex> lock(ctx->mu);
    rc = chan_index_reserve(ctx, queue, npx * 12);
    if (rc) return rc;
    tmp = ctx->chan_index[queue];
  }
  PH_HIP(hipMemcpyAsync(tmp, image, npx * 12, hipMemcpyDeviceToDevice, stream_of(ctx, queue)));
  hipError_t e = ph::launch_rgb_unpack(stream_of(ctx, queue), tmp, image, npx);
  if (e != hipSuccess) return fail(PH_E_HIP, "ph_image_unpack_rgb: launch failed: %s", hipGetErrorString(e));
  return PH_OK;
}

int ph_transform(ph_ctx *ctx, int queue, const void *in, int iw, int ih, const void *m9, void *out, int ow, int oh) {
  if (!in || !m9 || !out || iw <= 0 || ih <= 0 || ow <= 0 || oh <= 0) return fail(PH_E_INVALID, "ph_transform: NULL/zero argument");
  PH_LAUNCH(ph::launch_transform(stream_of(ctx, queue), in, iw, ih, m9, out, ow, oh));
}

int ph_resize(ph_ctx *ctx, int queue, const void *in, int iw, int ih, float scale, float ox, float oy,
              const void *flip4, void *out, int ow, int oh) {
  if (!in || !flip4 || !out || iw <= 0 || ih <= 0 || ow <= 0 || oh <= 0) return fail(PH_E_INVALID, "ph_resize: NULL/zero argument");
  PH_LAUNCH(ph::launch_resize(stream_of(ctx, queue), in, iw, ih, scale, ox, oy, flip4, out, ow, oh));
}

int ph_combine(ph_ctx *ctx, int queue, int n, const void *const *layers, int w, int h, void *out) {
  // the reference's Combine throws below 2 inputs (combine.ts:92-93)
  if (n < 2 || n > ph::kMaxLayers) return fail(PH_E_INVALID, "Combine requires between 2 and %d input buffers, got %d", ph::kMaxLayers, n);
  if (!layers || !out || w <= 0 || h <= 0) return fail(PH_E_INVALID, "ph_combine: NULL/zero argument");
  ph::CombineArgs a{};
  for (int i = 0; i < n; ++i) {
    if (!layers[i]) return fail(PH_E_INVALID, "ph_combine: layer %d is NULL", i);
    a.layers[i] = layers[i];
  }
  a.out = out;
  a.npx = (size_t)w * h;
  PH_LAUNCH(ph::launch_combine(stream_of(ctx, queue), n, a));
}

int ph_transition_dissolve(ph_ctx *ctx, int queue, const void *in0, const void *in1, float mix, int w, int h, void *out) {
  if (!in0 || !in1 || !out || w <= 0 || h <= 0) return fail(PH_E_INVALID, "ph_transition_dissolve: NULL/zero argument");
  PH_LAUNCH(ph::launch_dissolve(stream_of(ctx, queue), in0, in1, mix, w, h, out));
}

int ph_mixer(ph_ctx *ctx, int queue, const void *in0, const void *in1, float mix, int w, int h, void *out) {
  if (!in0 || !in1 || !out || w <= 0 || h <= 0) return fail(PH_E_INVALID, "ph_mixer: NULL/zero argument");
  PH_LAUNCH(ph::launch_dissolve(stream_of(ctx, queue), in0, in1, mix, w, h, out));  // same arithmetic (mix.ts:41-42)
}

int ph_transition_wipe(ph_ctx *ctx, int queue, const void *in0, const void *in1, const void *mask, int w, int h,
                       void *out) {
  if (!in0 || !in1 || !mask || !out || w <= 0 || h <= 0) return fail(PH_E_INVALID, "ph_transition_wipe: NULL/zero argument");
  PH_LAUNCH(ph::launch_twipe(stream_of(ctx, queue), in0, in1, mask, w, h, out));
}

int ph_wipe(ph_ctx *ctx, int queue, const void *in0, const void *in1, float wipe, int w, int h, void *out) {
  if (!in0 || !in1 || !out || w <= 0 || h <= 0) return fail(PH_E_INVALID, "ph_wipe: NULL/zero argument");
  PH_LAUNCH(ph::launch_wipe(stream_of(ctx, queue), in0, in1, wipe, w, h, out));
}

}  // extern "C"
