/* ph_napi.c - raw N-API glue between node and libphaneron_hip.so (include/phaneron_hip.h).
 *
 * This is the thin addon the reference would load INSTEAD of `nodencl`: node/index.js wraps
 * these functions into the nodencl-shaped `clContext` / `OpenCLBuffer` objects that
 * src/process and src/clJobQueue.ts call.  Plain C, no node-addon-api; nothing here computes
 * pixels - every entry forwards to the C ABI.
 *
 * Blocking calls (waitFinish, hostAccess, timed runProgram) run on the libuv thread pool via
 * napi_async_work and settle a promise, so the JS thread is never blocked - the reference
 * awaits these (clJobQueue.ts:126-131, io.ts:79-98).
 */
#include <node_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "../include/phaneron_hip.h"

#define NAPI_OK(call)                                                         \
  do {                                                                        \
    if ((call) != napi_ok) {                                                  \
      napi_throw_error(env, NULL, "N-API call failed: " #call);               \
      return NULL;                                                            \
    }                                                                         \
  } while (0)

typedef struct {
  ph_ctx *ctx;
} ctx_box;

typedef struct {
  ph_buf *buf; /* NULL once the last JS reference has been released */
  int js_refs; /* addRef / release calls made from JS (async jobs hold library references of their own) */
} buf_box;

typedef struct {
  ph_program *prog;
} prog_box;

static napi_value throw_ph(napi_env env, const char *what) {
  char msg[640];
  snprintf(msg, sizeof msg, "%s: %s", what, ph_last_error(NULL));
  napi_throw_error(env, NULL, msg);
  return NULL;
}

static void ctx_finalize(napi_env env, void *data, void *hint) {
  (void)env, (void)hint;
  ctx_box *b = (ctx_box *)data;
  if (b->ctx) ph_ctx_destroy(b->ctx);
  free(b);
}
static void buf_finalize(napi_env env, void *data, void *hint) {
  (void)env, (void)hint;
  buf_box *b = (buf_box *)data;
  /* references leaked by JS die with the JS object; the library keeps the context alive until the last
   * buffer is gone (ph_api.cpp "Lifetime"), so the order in which the collector finalises things is free */
  for (; b->buf && b->js_refs > 0; --b->js_refs) ph_buf_release(b->buf);
  free(b);
}
static void prog_finalize(napi_env env, void *data, void *hint) {
  (void)env, (void)hint;
  prog_box *b = (prog_box *)data;
  if (b->prog) ph_program_destroy(b->prog);
  free(b);
}
static void noop_finalize(napi_env env, void *data, void *hint) { (void)env, (void)data, (void)hint; }

static int get_box(napi_env env, napi_value v, void **out) { return napi_get_value_external(env, v, out) == napi_ok && *out; }

static int get_i32(napi_env env, napi_value v, int32_t *out) { return napi_get_value_int32(env, v, out) == napi_ok; }

/* createContext(deviceIndex) -> external */
static napi_value CreateContext(napi_env env, napi_callback_info info) {
  size_t argc = 1;
  napi_value argv[1], out;
  int32_t dev = 0;
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  if (argc >= 1) get_i32(env, argv[0], &dev);
  ctx_box *box = (ctx_box *)calloc(1, sizeof *box);
  if (ph_ctx_create(dev, &box->ctx) != PH_OK) {
    free(box);
    return throw_ph(env, "createContext");
  }
  NAPI_OK(napi_create_external(env, box, ctx_finalize, NULL, &out));
  return out;
}

/* contextInfo(ctx) -> { vendor, device } */
static napi_value ContextInfo(napi_env env, napi_callback_info info) {
  size_t argc = 1;
  napi_value argv[1], out, s;
  ctx_box *c;
  char vendor[128], device[256];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  if (argc < 1 || !get_box(env, argv[0], (void **)&c)) return throw_ph(env, "contextInfo: bad context");
  if (ph_ctx_info(c->ctx, vendor, sizeof vendor, device, sizeof device) != PH_OK) return throw_ph(env, "contextInfo");
  NAPI_OK(napi_create_object(env, &out));
  NAPI_OK(napi_create_string_utf8(env, vendor, NAPI_AUTO_LENGTH, &s));
  NAPI_OK(napi_set_named_property(env, out, "vendor", s));
  NAPI_OK(napi_create_string_utf8(env, device, NAPI_AUTO_LENGTH, &s));
  NAPI_OK(napi_set_named_property(env, out, "device", s));
  return out;
}

/* createBuffer(ctx, bytes, access, svm, width, height, owner) -> { handle, buffer }
 * `buffer` is a node Buffer over the pinned host mirror: the reference treats OpenCLBuffer as a
 * Buffer (Buffer.copy / readFloatLE / concat, displayFrame). */
static napi_value CreateBuffer(napi_env env, napi_callback_info info) {
  size_t argc = 7;
  napi_value argv[7], out, handle, nodebuf;
  ctx_box *c;
  double bytes = 0;
  int32_t access = 0, svm = 0, w = 0, h = 0;
  char owner[256] = "";
  size_t len = 0;
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  if (argc < 2 || !get_box(env, argv[0], (void **)&c)) return throw_ph(env, "createBuffer: bad context");
  if (napi_get_value_double(env, argv[1], &bytes) != napi_ok || bytes < 0) return throw_ph(env, "createBuffer: bad size");
  if (argc > 2) get_i32(env, argv[2], &access);
  if (argc > 3) get_i32(env, argv[3], &svm);
  if (argc > 4) get_i32(env, argv[4], &w);
  if (argc > 5) get_i32(env, argv[5], &h);
  if (argc > 6) napi_get_value_string_utf8(env, argv[6], owner, sizeof owner, &len);
  buf_box *box = (buf_box *)calloc(1, sizeof *box);
  box->js_refs = 1;
  if (ph_buf_create(c->ctx, (size_t)bytes, access, svm, w, h, owner, &box->buf) != PH_OK) {
    free(box);
    return throw_ph(env, "createBuffer");
  }
  void *host = ph_buf_host_ptr(box->buf);
  if (!host) {
    ph_buf_release(box->buf);
    free(box);
    return throw_ph(env, "createBuffer (host mirror)");
  }
  NAPI_OK(napi_create_external(env, box, buf_finalize, NULL, &handle));
  NAPI_OK(napi_create_external_buffer(env, (size_t)bytes, host, noop_finalize, NULL, &nodebuf));
  NAPI_OK(napi_create_object(env, &out));
  NAPI_OK(napi_set_named_property(env, out, "handle", handle));
  NAPI_OK(napi_set_named_property(env, out, "buffer", nodebuf));
  return out;
}

static napi_value BufAddRef(napi_env env, napi_callback_info info) {
  size_t argc = 1;
  napi_value argv[1], out;
  buf_box *b;
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  if (argc < 1 || !get_box(env, argv[0], (void **)&b) || !b->buf) {
    napi_throw_error(env, NULL, "addRef on a released buffer");
    return NULL;
  }
  ph_buf_addref(b->buf);
  NAPI_OK(napi_create_int32(env, ++b->js_refs, &out));
  return out;
}

/* bufReuse(handle): a parked buffer gets its next owner (ph_buf_reuse) */
static napi_value BufReuse(napi_env env, napi_callback_info info) {
  size_t argc = 1;
  napi_value argv[1];
  buf_box *b;
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  if (argc < 1 || !get_box(env, argv[0], (void **)&b) || !b->buf) {
    napi_throw_error(env, NULL, "reuse of a released buffer");
    return NULL;
  }
  if (ph_buf_reuse(b->buf) != PH_OK) return throw_ph(env, "bufReuse");
  return NULL;
}

static napi_value BufRelease(napi_env env, napi_callback_info info) {
  size_t argc = 1;
  napi_value argv[1], out;
  buf_box *b;
  int left;
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  if (argc < 1 || !get_box(env, argv[0], (void **)&b) || !b->buf) {
    napi_throw_error(env, NULL, "release on a released buffer");
    return NULL;
  }
  left = --b->js_refs;
  ph_buf_release(b->buf); /* an async job still using the buffer holds its own reference */
  if (left <= 0) b->buf = NULL;
  NAPI_OK(napi_create_int32(env, left, &out));
  return out;
}

static napi_value BufRefCount(napi_env env, napi_callback_info info) {
  size_t argc = 1;
  napi_value argv[1], out;
  buf_box *b;
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  if (argc < 1 || !get_box(env, argv[0], (void **)&b)) return throw_ph(env, "refCount: bad buffer");
  NAPI_OK(napi_create_int32(env, b->buf ? b->js_refs : 0, &out));
  return out;
}

/* ---- async plumbing ----------------------------------------------------------------------- */
typedef enum { JOB_WAIT, JOB_HOST_ACCESS, JOB_RUN_TIMED, JOB_EVENT_WAIT } job_kind;

typedef struct {
  job_kind kind;
  napi_async_work work;
  napi_deferred deferred;
  ph_ctx *ctx;
  int queue;
  ph_event *event; /* JOB_EVENT_WAIT */
  /* host access */
  ph_buf *buf;
  int dir;
  const void *src;
  size_t src_bytes;
  napi_ref src_ref; /* keeps the source Buffer alive while the copy runs */
  napi_ref pin[2];  /* context / program externals of a job running on the pool */
  ph_buf **held;    /* buffers the job uses: one library reference each until it completes */
  int n_held;
  /* timed run */
  ph_program *prog;
  ph_arg *args;
  char *names; /* storage for argument names */
  int n_args;
  ph_run_timings timings;
  /* result */
  int rc;
  char err[512];
} job;

static void job_execute(napi_env env, void *data) {
  (void)env;
  job *j = (job *)data;
  switch (j->kind) {
    case JOB_WAIT: j->rc = ph_wait_finish(j->ctx, j->queue); break;
    case JOB_HOST_ACCESS: j->rc = ph_buf_host_access(j->buf, j->dir, j->queue, j->src, j->src_bytes); break;
    case JOB_RUN_TIMED: j->rc = ph_run_program(j->ctx, j->prog, j->args, j->n_args, j->queue, &j->timings); break;
    case JOB_EVENT_WAIT: j->rc = ph_event_wait(j->event); break;
  }
  if (j->rc != PH_OK) snprintf(j->err, sizeof j->err, "%s", ph_last_error(NULL)); /* thread-local: copy here */
}

static napi_value timings_object(napi_env env, const ph_run_timings *t) {
  napi_value o, v;
  napi_create_object(env, &o);
  napi_create_uint32(env, t->data_to_kernel, &v);
  napi_set_named_property(env, o, "dataToKernel", v);
  napi_create_uint32(env, t->kernel_exec, &v);
  napi_set_named_property(env, o, "kernelExec", v);
  napi_create_uint32(env, t->total_time, &v);
  napi_set_named_property(env, o, "totalTime", v);
  return o;
}

static void job_complete(napi_env env, napi_status status, void *data) {
  job *j = (job *)data;
  napi_value v;
  (void)status;
  if (j->rc == PH_OK) {
    if (j->kind == JOB_RUN_TIMED)
      v = timings_object(env, &j->timings);
    else
      napi_get_undefined(env, &v);
    napi_resolve_deferred(env, j->deferred, v);
  } else {
    napi_value msg;
    napi_create_string_utf8(env, j->err, NAPI_AUTO_LENGTH, &msg);
    napi_create_error(env, NULL, msg, &v);
    napi_reject_deferred(env, j->deferred, v);
  }
  if (j->src_ref) napi_delete_reference(env, j->src_ref);
  for (int i = 0; i < 2; ++i)
    if (j->pin[i]) napi_delete_reference(env, j->pin[i]);
  for (int i = 0; i < j->n_held; ++i) ph_buf_release(j->held[i]);
  free(j->held);
  napi_delete_async_work(env, j->work);
  free(j->args);
  free(j->names);
  free(j);
}

static napi_value start_job(napi_env env, job *j, const char *name) {
  napi_value promise, resname;
  if (napi_create_promise(env, &j->deferred, &promise) != napi_ok ||
      napi_create_string_utf8(env, name, NAPI_AUTO_LENGTH, &resname) != napi_ok ||
      napi_create_async_work(env, NULL, resname, job_execute, job_complete, j, &j->work) != napi_ok ||
      napi_queue_async_work(env, j->work) != napi_ok) {
    for (int i = 0; i < j->n_held; ++i) ph_buf_release(j->held[i]);
    free(j->held);
    free(j->args);
    free(j->names);
    free(j);
    napi_throw_error(env, NULL, "could not start async work");
    return NULL;
  }
  return promise;
}

/* waitFinish(ctx, queue) -> Promise<void> */
static napi_value WaitFinish(napi_env env, napi_callback_info info) {
  size_t argc = 2;
  napi_value argv[2];
  ctx_box *c;
  int32_t q = PH_QUEUE_PROCESS;
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  if (argc < 1 || !get_box(env, argv[0], (void **)&c)) return throw_ph(env, "waitFinish: bad context");
  if (argc > 1) get_i32(env, argv[1], &q);
  job *j = (job *)calloc(1, sizeof *j);
  j->kind = JOB_WAIT, j->ctx = c->ctx, j->queue = q;
  napi_create_reference(env, argv[0], 1, &j->pin[0]);
  return start_job(env, j, "phaneron.waitFinish");
}

/* waitFinishSpin(ctx, queue, micros) -> boolean: polls the queue on the calling thread for at most
 * `micros`; true = it went idle (no thread hand-off needed), false = still busy (use waitFinish). */
static napi_value WaitFinishSpin(napi_env env, napi_callback_info info) {
  size_t argc = 3;
  napi_value argv[3], out;
  ctx_box *c;
  int32_t q = PH_QUEUE_PROCESS, micros = 0;
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  if (argc < 1 || !get_box(env, argv[0], (void **)&c)) return throw_ph(env, "waitFinishSpin: bad context");
  if (argc > 1) get_i32(env, argv[1], &q);
  if (argc > 2) get_i32(env, argv[2], &micros);
  struct timespec t0, t;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  int idle = 0;
  for (;;) {
    idle = ph_queue_query(c->ctx, q);
    if (idle != 0) break;
    clock_gettime(CLOCK_MONOTONIC, &t);
    if ((t.tv_sec - t0.tv_sec) * 1000000L + (t.tv_nsec - t0.tv_nsec) / 1000L >= micros) break;
  }
  if (idle < 0) return throw_ph(env, "waitFinishSpin");
  NAPI_OK(napi_get_boolean(env, idle == 1, &out));
  return out;
}

/* hostAccess(buf, dir, queue, srcBuffer?) -> Promise<void> */
static napi_value HostAccess(napi_env env, napi_callback_info info) {
  size_t argc = 4;
  napi_value argv[4];
  buf_box *b;
  int32_t dir = PH_HOST_NONE, q = 0;
  bool is_buf = false;
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  if (argc < 2 || !get_box(env, argv[0], (void **)&b) || !b->buf) {
    napi_throw_error(env, NULL, "hostAccess on a released buffer");
    return NULL;
  }
  get_i32(env, argv[1], &dir);
  if (argc > 2) get_i32(env, argv[2], &q);
  job *j = (job *)calloc(1, sizeof *j);
  j->kind = JOB_HOST_ACCESS, j->buf = b->buf, j->dir = dir, j->queue = q;
  j->held = (ph_buf **)malloc(sizeof *j->held);
  j->held[0] = b->buf, j->n_held = 1;
  ph_buf_addref(b->buf); /* release() before the promise settles must not free it under the copy */
  if (argc > 3 && napi_is_buffer(env, argv[3], &is_buf) == napi_ok && is_buf) {
    void *p;
    napi_get_buffer_info(env, argv[3], &p, &j->src_bytes);
    j->src = p;
    napi_create_reference(env, argv[3], 1, &j->src_ref);
  }
  return start_job(env, j, "phaneron.hostAccess");
}

/* ---- staging primitives (include/phaneron_hip.h "Staged producers / consumers") ---------------- */
/* queueWaitQueue(ctx, waiterQueue, signalQueue) */
static napi_value QueueWaitQueue(napi_env env, napi_callback_info info) {
  size_t argc = 3;
  napi_value argv[3], out;
  ctx_box *c;
  int32_t waiter = 0, signal = 0;
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  if (argc < 3 || !get_box(env, argv[0], (void **)&c) || !get_i32(env, argv[1], &waiter) || !get_i32(env, argv[2], &signal))
    return throw_ph(env, "queueWaitQueue(ctx, waiter, signal)");
  if (ph_queue_wait_queue(c->ctx, waiter, signal) != PH_OK) return throw_ph(env, "queueWaitQueue");
  NAPI_OK(napi_get_undefined(env, &out));
  return out;
}

/* downloadAsync(buf, queue): device -> mirror, no host wait */
static napi_value DownloadAsync(napi_env env, napi_callback_info info) {
  size_t argc = 2;
  napi_value argv[2], out;
  buf_box *b;
  int32_t q = PH_QUEUE_UNLOAD;
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  if (argc < 1 || !get_box(env, argv[0], (void **)&b) || !b->buf) return throw_ph(env, "downloadAsync on a released buffer");
  if (argc > 1) get_i32(env, argv[1], &q);
  if (ph_buf_download_async(b->buf, q) != PH_OK) return throw_ph(env, "downloadAsync");
  NAPI_OK(napi_get_undefined(env, &out));
  return out;
}

static void event_finalize(napi_env env, void *data, void *hint) {
  (void)env, (void)hint;
  ph_event_destroy((ph_event *)data);
}

/* eventRecord(ctx, queue) -> external (destroyed with the JS object) */
static napi_value EventRecord(napi_env env, napi_callback_info info) {
  size_t argc = 3;
  napi_value argv[3], out;
  ctx_box *c;
  int32_t q = PH_QUEUE_PROCESS;
  ph_event *ev = NULL;
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  if (argc < 1 || !get_box(env, argv[0], (void **)&c)) return throw_ph(env, "eventRecord: bad context");
  bool timed = false; /* eventRecord(ctx, queue, timed): a point eventElapsed can measure from / to */
  if (argc > 1) get_i32(env, argv[1], &q);
  if (argc > 2) napi_get_value_bool(env, argv[2], &timed);
  if ((timed ? ph_event_record_timed(c->ctx, q, &ev) : ph_event_record(c->ctx, q, &ev)) != PH_OK) return throw_ph(env, "eventRecord");
  NAPI_OK(napi_create_external(env, ev, event_finalize, NULL, &out));
  return out;
}

/* eventWait(event) -> Promise<void> (on the libuv pool) */
static napi_value EventWait(napi_env env, napi_callback_info info) {
  size_t argc = 1;
  napi_value argv[1];
  ph_event *ev;
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  if (argc < 1 || !get_box(env, argv[0], (void **)&ev)) return throw_ph(env, "eventWait: bad event");
  job *j = (job *)calloc(1, sizeof *j);
  j->kind = JOB_EVENT_WAIT, j->event = ev;
  napi_create_reference(env, argv[0], 1, &j->src_ref); /* the event outlives the wait */
  return start_job(env, j, "phaneron.eventWait");
}

/* eventDone(event) -> boolean */
static napi_value EventDone(napi_env env, napi_callback_info info) {
  size_t argc = 1;
  napi_value argv[1], out;
  ph_event *ev;
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  if (argc < 1 || !get_box(env, argv[0], (void **)&ev)) return throw_ph(env, "eventDone: bad event");
  int r = ph_event_query(ev);
  if (r < 0) return throw_ph(env, "eventDone");
  NAPI_OK(napi_get_boolean(env, r == 1, &out));
  return out;
}

/* eventElapsed(from, to) -> microseconds of device time between two timed, finished events of one queue */
static napi_value EventElapsed(napi_env env, napi_callback_info info) {
  size_t argc = 2;
  napi_value argv[2], out;
  ph_event *a, *b;
  uint32_t us = 0;
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  if (argc < 2 || !get_box(env, argv[0], (void **)&a) || !get_box(env, argv[1], (void **)&b)) return throw_ph(env, "eventElapsed: bad event");
  if (ph_event_elapsed_us(a, b, &us) != PH_OK) return throw_ph(env, "eventElapsed");
  NAPI_OK(napi_create_uint32(env, us, &out));
  return out;
}

/* createProgram(ctx, kernelSrc, name, globalWorkItems[], workItemsPerGroup) -> external */
static napi_value CreateProgram(napi_env env, napi_callback_info info) {
  size_t argc = 5, len = 0, srclen = 0;
  napi_value argv[5], out;
  ctx_box *c;
  char name[128] = "";
  char *src = NULL;
  uint32_t gwi[2] = {0, 0}, n_dims = 0, wipg = 0;
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  if (argc < 3 || !get_box(env, argv[0], (void **)&c)) return throw_ph(env, "createProgram: bad context");
  if (napi_get_value_string_utf8(env, argv[1], NULL, 0, &srclen) == napi_ok) {
    src = (char *)malloc(srclen + 1);
    napi_get_value_string_utf8(env, argv[1], src, srclen + 1, &srclen);
  }
  napi_get_value_string_utf8(env, argv[2], name, sizeof name, &len);
  if (argc > 3) {
    bool is_arr = false;
    napi_is_array(env, argv[3], &is_arr);
    if (is_arr) {
      napi_get_array_length(env, argv[3], &n_dims);
      for (uint32_t i = 0; i < n_dims && i < 2; ++i) {
        napi_value e;
        napi_get_element(env, argv[3], i, &e);
        napi_get_value_uint32(env, e, &gwi[i]);
      }
      if (n_dims > 2) n_dims = 2;
    } else if (napi_get_value_uint32(env, argv[3], &gwi[0]) == napi_ok) {
      n_dims = 1;
    }
  }
  if (argc > 4) napi_get_value_uint32(env, argv[4], &wipg);
  prog_box *box = (prog_box *)calloc(1, sizeof *box);
  int rc = ph_program_create(c->ctx, src, name, gwi, (int)n_dims, wipg, &box->prog);
  free(src);
  if (rc != PH_OK) {
    free(box);
    return throw_ph(env, "createProgram");
  }
  NAPI_OK(napi_create_external(env, box, prog_finalize, NULL, &out));
  return out;
}

/* runProgram(ctx, prog, names[], values[], queue, timed[, checkOnly]) -> RunTimings | Promise<RunTimings>
 * values[i] is a buffer handle (external) or a number; float-valued kernel arguments are the
 * ones named in FLOAT_ARGS and any value that is not a whole number (per-layer arguments such as l<i>Mix of
 * chan_compose_v210_<n>); everything else numeric is passed as a 32-bit integer.  The library reads a
 * numeric argument of either kind as a number (ph_api.cpp need_num). */
static const char *FLOAT_ARGS[] = {"scale", "offsetX", "offsetY", "mix", "wipe", NULL};

/* names[] / values[] of one job -> ph_arg array (caller frees *args and *names); 0 = a JS exception is pending */
static int marshal_args(napi_env env, napi_value names_arr, napi_value values_arr, ph_arg **args_out, char **names_out, uint32_t *n_out) {
  uint32_t n = 0;
  napi_get_array_length(env, names_arr, &n);
  ph_arg *args = (ph_arg *)calloc(n ? n : 1, sizeof *args);
  char *names = (char *)calloc(n ? n : 1, 64);
  for (uint32_t i = 0; i < n; ++i) {
    napi_value nm, val;
    napi_valuetype t;
    size_t len;
    napi_get_element(env, names_arr, i, &nm);
    napi_get_element(env, values_arr, i, &val);
    napi_get_value_string_utf8(env, nm, names + 64 * i, 64, &len);
    args[i].name = names + 64 * i;
    napi_typeof(env, val, &t);
    if (t == napi_external) {
      buf_box *b;
      napi_get_value_external(env, val, (void **)&b);
      if (!b || !b->buf) {
        free(args), free(names);
        napi_throw_error(env, NULL, "runProgram: a buffer argument has already been released");
        return 0;
      }
      args[i].kind = PH_ARG_BUF, args[i].v.buf = b->buf;
    } else {
      double d = 0;
      napi_coerce_to_number(env, val, &val);
      napi_get_value_double(env, val, &d);
      int is_float = !(d >= -2147483648.0 && d <= 2147483647.0 && d == (double)(int32_t)d);
      for (const char **f = FLOAT_ARGS; *f; ++f) is_float |= (0 == strcmp(*f, args[i].name));
      if (is_float)
        args[i].kind = PH_ARG_F32, args[i].v.f32 = (float)d;
      else
        args[i].kind = PH_ARG_I32, args[i].v.i32 = (int32_t)d;
    }
  }
  *args_out = args, *names_out = names, *n_out = n;
  return 1;
}

/* runPrograms(ctx, progs[], names[][], values[][], queue): several recorded jobs in one call (ph_run_programs) - the jobs in their
 * order, channel frames of one shape among them in one launch.  Asynchronous like an untimed runProgram; throws what the library refuses. */
static napi_value RunPrograms(napi_env env, napi_callback_info info) {
  size_t argc = 5;
  napi_value argv[5];
  ctx_box *c;
  uint32_t jobs = 0;
  int32_t q = PH_QUEUE_PROCESS;
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  if (argc < 4 || !get_box(env, argv[0], (void **)&c)) return throw_ph(env, "runPrograms: bad context");
  napi_get_array_length(env, argv[1], &jobs);
  if (argc > 4) get_i32(env, argv[4], &q);
  if (!jobs) return NULL;
  ph_program **progs = (ph_program **)calloc(jobs, sizeof *progs);
  ph_arg **args = (ph_arg **)calloc(jobs, sizeof *args);
  char **names = (char **)calloc(jobs, sizeof *names);
  int *counts = (int *)calloc(jobs, sizeof *counts);
  int ok = 1;
  for (uint32_t j = 0; j < jobs && ok; ++j) {
    napi_value pv, nv, vv;
    prog_box *p = NULL;
    uint32_t n = 0;
    napi_get_element(env, argv[1], j, &pv);
    napi_get_element(env, argv[2], j, &nv);
    napi_get_element(env, argv[3], j, &vv);
    if (!get_box(env, pv, (void **)&p)) {
      napi_throw_error(env, NULL, "runPrograms: bad program");
      ok = 0;
      break;
    }
    progs[j] = p->prog;
    ok = marshal_args(env, nv, vv, &args[j], &names[j], &n);
    counts[j] = (int)n;
  }
  int rc = ok ? ph_run_programs(c->ctx, (int)jobs, progs, (const ph_arg *const *)args, counts, q) : PH_OK;
  for (uint32_t j = 0; j < jobs; ++j) free(args[j]), free(names[j]);
  free(progs), free(args), free(names), free(counts);
  if (!ok) return NULL;
  if (rc != PH_OK) return throw_ph(env, "runPrograms");
  return NULL;
}

/* runProgramsProgress(): how many jobs of this thread's last runPrograms call had their launches made (ph_run_programs_progress) -
 * after a call that threw at a launch, the jobs before the failing group are on the device already. */
static napi_value RunProgramsProgress(napi_env env, napi_callback_info info) {
  napi_value v;
  int done = 0;
  (void)info;
  if (ph_run_programs_progress(&done) != PH_OK) return throw_ph(env, "runProgramsProgress");
  NAPI_OK(napi_create_int32(env, done, &v));
  return v;
}

/* traceBegin(dryRun) / traceEnd() -> string: which kernels the calls in between launched, joined by '+' (ph_trace_begin / ph_trace_end);
 * dryRun: the calls choose their kernels but enqueue nothing. */
static napi_value TraceBegin(napi_env env, napi_callback_info info) {
  size_t argc = 1;
  napi_value argv[1];
  bool dry = false;
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  if (argc > 0) napi_get_value_bool(env, argv[0], &dry);
  if (ph_trace_begin(dry ? 1 : 0) != PH_OK) return throw_ph(env, "traceBegin");
  return NULL;
}
static napi_value TraceEnd(napi_env env, napi_callback_info info) {
  static char route[8192];
  napi_value v;
  (void)info;
  if (ph_trace_end(route, sizeof route) != PH_OK) return throw_ph(env, "traceEnd");
  NAPI_OK(napi_create_string_utf8(env, route, NAPI_AUTO_LENGTH, &v));
  return v;
}

static napi_value RunProgram(napi_env env, napi_callback_info info) {
  size_t argc = 7;
  napi_value argv[7];
  ctx_box *c;
  prog_box *p;
  uint32_t n = 0;
  int32_t q = PH_QUEUE_PROCESS;
  bool timed = false, check_only = false; /* check_only: ph_check_program - the job's arguments are examined, nothing is launched */
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  if (argc < 4 || !get_box(env, argv[0], (void **)&c) || !get_box(env, argv[1], (void **)&p))
    return throw_ph(env, "runProgram: bad context/program");
  if (argc > 6) napi_get_value_bool(env, argv[6], &check_only);
  if (argc > 4) get_i32(env, argv[4], &q);
  if (argc > 5) napi_get_value_bool(env, argv[5], &timed);
  ph_arg *args = NULL;
  char *names = NULL;
  if (!marshal_args(env, argv[2], argv[3], &args, &names, &n)) return NULL;
  if (timed && !check_only) {
    job *j = (job *)calloc(1, sizeof *j);
    j->kind = JOB_RUN_TIMED, j->ctx = c->ctx, j->prog = p->prog, j->args = args, j->names = names;
    j->n_args = (int)n, j->queue = q;
    napi_create_reference(env, argv[0], 1, &j->pin[0]);
    napi_create_reference(env, argv[1], 1, &j->pin[1]);
    j->held = (ph_buf **)calloc(n ? n : 1, sizeof *j->held);
    for (uint32_t i = 0; i < n; ++i)
      if (args[i].kind == PH_ARG_BUF) ph_buf_addref(j->held[j->n_held++] = args[i].v.buf);
    return start_job(env, j, "phaneron.runProgram");
  }
  int rc = check_only ? ph_check_program(c->ctx, p->prog, args, (int)n, q) : ph_run_program(c->ctx, p->prog, args, (int)n, q, NULL);
  free(args), free(names);
  if (rc != PH_OK) return throw_ph(env, "runProgram");
  ph_run_timings zero = {0, 0, 0};
  return timings_object(env, &zero);
}

/* bufferStats(ctx) -> { liveBuffers, liveBytes, pooledBytes } (nodencl logBuffers) */
static napi_value BufferStats(napi_env env, napi_callback_info info) {
  size_t argc = 1, a = 0, b = 0, p = 0;
  napi_value argv[1], out, v;
  ctx_box *c;
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  if (argc < 1 || !get_box(env, argv[0], (void **)&c)) return throw_ph(env, "bufferStats: bad context");
  ph_ctx_buffer_stats(c->ctx, &a, &b, &p);
  NAPI_OK(napi_create_object(env, &out));
  napi_create_double(env, (double)a, &v), napi_set_named_property(env, out, "liveBuffers", v);
  napi_create_double(env, (double)b, &v), napi_set_named_property(env, out, "liveBytes", v);
  napi_create_double(env, (double)p, &v), napi_set_named_property(env, out, "pooledBytes", v);
  { /* the pinned host mirrors (ph_ctx_host_pool_stats): a `pins` count that keeps growing = the pool is smaller than the working set */
    size_t in_use = 0, pooled = 0, peak = 0;
    uint64_t pins = 0;
    ph_ctx_host_pool_stats(c->ctx, &in_use, &pooled, &peak, &pins);
    napi_create_double(env, (double)in_use, &v), napi_set_named_property(env, out, "pinnedInUse", v);
    napi_create_double(env, (double)pooled, &v), napi_set_named_property(env, out, "pinnedPooled", v);
    napi_create_double(env, (double)peak, &v), napi_set_named_property(env, out, "pinnedPeak", v);
    napi_create_double(env, (double)pins, &v), napi_set_named_property(env, out, "pins", v);
  }
  return out;
}

/* setOption(ctx, name, value): ph_ctx_set_option ("lds_lut", "stream_images", "stream_threshold_mb") */
static napi_value SetOption(napi_env env, napi_callback_info info) {
  size_t argc = 3;
  napi_value argv[3];
  ctx_box *c;
  char name[64];
  int32_t value = 0;
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  if (argc < 3 || !get_box(env, argv[0], (void **)&c)) return throw_ph(env, "setOption: bad context");
  size_t len = 0;
  if (napi_get_value_string_utf8(env, argv[1], name, sizeof name, &len) != napi_ok || napi_get_value_int32(env, argv[2], &value) != napi_ok)
    return throw_ph(env, "setOption(ctx, name, value): a string and an integer");
  if (ph_ctx_set_option(c->ctx, name, value)) return throw_ph(env, "setOption");
  return NULL;
}

/* ---- device-free helpers: kernel selection and the host colour maths (src/process/colourMaths.ts is run
 *      by the reference's Loader / Saver; a caller that builds its own parameter buffers gets the same
 *      numbers from the library) --------------------------------------------------------------------- */
static int get_str(napi_env env, napi_value v, char *buf, size_t n) {
  size_t len = 0;
  return napi_get_value_string_utf8(env, v, buf, n, &len) == napi_ok;
}
static napi_value f32_array(napi_env env, const float *data, size_t n) {
  napi_value ab, ta;
  void *p = NULL;
  if (napi_create_arraybuffer(env, n * 4, &p, &ab) != napi_ok) return NULL;
  memcpy(p, data, n * 4);
  if (napi_create_typedarray(env, napi_float32_array, n, ab, 0, &ta) != napi_ok) return NULL;
  return ta;
}

/* resolveProgram(kernelSrc, name) -> { kernel, format, how } */
static napi_value ResolveProgram(napi_env env, napi_callback_info info) {
  size_t argc = 2, srclen = 0;
  napi_value argv[2], out, v;
  char name[128] = "", kernel[64] = "";
  char *src = NULL;
  int fmt = -1, how = 0;
  static const char *hows[] = {"tag", "name", "text", "signature"};
  static const char *fmts[] = {"v210", "yuv422p10", "yuv422p8", "yuv420p", "nv12", "rgba8", "bgra8"};
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  if (argc < 2 || !get_str(env, argv[1], name, sizeof name)) return throw_ph(env, "resolveProgram(kernelSrc, name)");
  if (napi_get_value_string_utf8(env, argv[0], NULL, 0, &srclen) == napi_ok) {
    src = (char *)malloc(srclen + 1);
    napi_get_value_string_utf8(env, argv[0], src, srclen + 1, &srclen);
  }
  int rc = ph_program_resolve(src, name, kernel, sizeof kernel, &fmt, &how);
  free(src);
  if (rc != PH_OK) return throw_ph(env, "resolveProgram");
  NAPI_OK(napi_create_object(env, &out));
  napi_create_string_utf8(env, kernel, NAPI_AUTO_LENGTH, &v), napi_set_named_property(env, out, "kernel", v);
  if (fmt >= 0 && fmt < 7) napi_create_string_utf8(env, fmts[fmt], NAPI_AUTO_LENGTH, &v);
  else napi_get_null(env, &v);
  napi_set_named_property(env, out, "format", v);
  napi_create_string_utf8(env, hows[how & 3], NAPI_AUTO_LENGTH, &v), napi_set_named_property(env, out, "how", v);
  return out;
}

/* gammaLut(kind: 'gamma2linear' | 'linear2gamma', colSpec) -> Float32Array(65536) */
static napi_value GammaLut(napi_env env, napi_callback_info info) {
  size_t argc = 2;
  napi_value argv[2];
  char kind[32] = "", spec[32] = "";
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  if (argc < 2 || !get_str(env, argv[0], kind, sizeof kind) || !get_str(env, argv[1], spec, sizeof spec))
    return throw_ph(env, "gammaLut(kind, colSpec)");
  float *lut = (float *)malloc(65536 * 4);
  int rc = 0 == strcmp(kind, "linear2gamma") ? ph_colour_linear2gamma_lut(spec, lut) : ph_colour_gamma2linear_lut(spec, lut);
  napi_value out = rc == PH_OK ? f32_array(env, lut, 65536) : NULL;
  free(lut);
  return out ? out : throw_ph(env, "gammaLut");
}

/* colourMatrix(kind: 'ycbcr2rgb' | 'rgb2ycbcr', colSpec, numBits, lumaBlack, lumaWhite, chromaRange) -> Float32Array(12)
 * colourMatrix('rgb2rgb', srcSpec, dstSpec) -> Float32Array(9) */
static napi_value ColourMatrix(napi_env env, napi_callback_info info) {
  size_t argc = 6;
  napi_value argv[6];
  char kind[32] = "", a[32] = "", b[32] = "";
  int32_t n[4] = {10, 64, 940, 896};
  float m[12];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  if (argc < 2 || !get_str(env, argv[0], kind, sizeof kind) || !get_str(env, argv[1], a, sizeof a))
    return throw_ph(env, "colourMatrix(kind, colSpec, ...)");
  if (0 == strcmp(kind, "rgb2rgb")) {
    if (argc < 3 || !get_str(env, argv[2], b, sizeof b)) return throw_ph(env, "colourMatrix('rgb2rgb', src, dst)");
    if (ph_colour_rgb2rgb_matrix(a, b, m) != PH_OK) return throw_ph(env, "colourMatrix");
    return f32_array(env, m, 9);
  }
  for (size_t i = 0; i < 4 && i + 2 < argc; ++i) get_i32(env, argv[i + 2], &n[i]);
  int rc = 0 == strcmp(kind, "rgb2ycbcr") ? ph_colour_rgb2ycbcr_matrix(a, n[0], n[1], n[2], n[3], m)
                                          : ph_colour_ycbcr2rgb_matrix(a, n[0], n[1], n[2], n[3], m);
  if (rc != PH_OK) return throw_ph(env, "colourMatrix");
  return f32_array(env, m, 12);
}

/* transformMatrix(width, height, flipH, flipV, anchorX, anchorY, scaleX, scaleY, offsetX, offsetY, rotate) -> Float32Array(9)
 * (transform.ts:119-171) */
static napi_value TransformMatrix(napi_env env, napi_callback_info info) {
  size_t argc = 11;
  napi_value argv[11];
  double d[11] = {0, 0, 0, 0, 0, 0, 1, 1, 0, 0, 0};
  float m[9];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  for (size_t i = 0; i < argc && i < 11; ++i) {
    napi_value num;
    if (napi_coerce_to_number(env, argv[i], &num) == napi_ok) napi_get_value_double(env, num, &d[i]);
  }
  if (ph_transform_matrix((int)d[0], (int)d[1], d[2] != 0, d[3] != 0, d[4], d[5], d[6], d[7], d[8], d[9], d[10], m) != PH_OK)
    return throw_ph(env, "transformMatrix");
  return f32_array(env, m, 9);
}

/* planeBytes(format 0..6, width, height) -> [bytes per plane] (the Readers' / Writers' numBytes) */
static napi_value PlaneBytes(napi_env env, napi_callback_info info) {
  size_t argc = 3, pb[3] = {0, 0, 0};
  napi_value argv[3], out, v;
  int32_t fmt = 0, w = 0, h = 0;
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  if (argc < 3 || !get_i32(env, argv[0], &fmt) || !get_i32(env, argv[1], &w) || !get_i32(env, argv[2], &h))
    return throw_ph(env, "planeBytes(format, width, height)");
  int n = fmt == 0 ? (pb[0] = (size_t)ph_v210_pitch_bytes((uint32_t)w) * (size_t)h, 1) : ph_pack_plane_bytes(fmt, (uint32_t)w, (uint32_t)h, pb);
  if (n < 0) return throw_ph(env, "planeBytes");
  NAPI_OK(napi_create_array_with_length(env, (size_t)n, &out));
  for (int i = 0; i < n; ++i) napi_create_double(env, (double)pb[i], &v), napi_set_element(env, out, (uint32_t)i, v);
  return out;
}

/* ---- ROUTE across GPUs (include/phaneron_hip.h "ROUTE"): RCCL point-to-point on its own stream --------------- */
static void route_finalize(napi_env env, void *data, void *hint) {
  (void)env, (void)hint;
  ph_route_destroy((ph_route *)data);
}
/* routeUniqueId() -> Buffer(128): made on ONE rank, handed to the others by the caller */
static napi_value RouteUniqueId(napi_env env, napi_callback_info info) {
  napi_value out;
  void *p = NULL;
  (void)info;
  NAPI_OK(napi_create_buffer(env, PH_ROUTE_ID_BYTES, &p, &out));
  if (ph_route_unique_id(p) != PH_OK) return throw_ph(env, "routeUniqueId");
  return out;
}
/* routeInit(ctx, idBuffer, rank, world) -> external */
static napi_value RouteInit(napi_env env, napi_callback_info info) {
  size_t argc = 4, len = 0;
  napi_value argv[4], out;
  ctx_box *c;
  void *id = NULL;
  int32_t rank = 0, world = 1;
  ph_route *r = NULL;
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  if (argc < 4 || !get_box(env, argv[0], (void **)&c) || napi_get_buffer_info(env, argv[1], &id, &len) != napi_ok ||
      len != PH_ROUTE_ID_BYTES || !get_i32(env, argv[2], &rank) || !get_i32(env, argv[3], &world))
    return throw_ph(env, "routeInit(ctx, id128, rank, world)");
  if (ph_route_init(c->ctx, id, rank, world, &r) != PH_OK) return throw_ph(env, "routeInit");
  NAPI_OK(napi_create_external(env, r, route_finalize, NULL, &out));
  return out;
}
/* routeOp(route, op, a, b): op 0 groupBegin, 1 groupEnd, 2 afterQueue(a = queue), 3 queueAfterRoute(a = queue), 4 wait
 *                          5 send(a = buffer handle, b = peer), 6 recv(a = buffer handle, b = peer) */
static napi_value RouteOp(napi_env env, napi_callback_info info) {
  size_t argc = 4;
  napi_value argv[4], out;
  ph_route *r;
  int32_t op = -1, a = 0, b = 0;
  int rc = PH_E_INVALID;
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  if (argc < 2 || !get_box(env, argv[0], (void **)&r) || !get_i32(env, argv[1], &op)) return throw_ph(env, "routeOp(route, op, ...)");
  if (op == 5 || op == 6) {
    buf_box *bb;
    if (argc < 4 || !get_box(env, argv[2], (void **)&bb) || !bb->buf || !get_i32(env, argv[3], &b)) {
      napi_throw_error(env, NULL, "route send / recv needs a live buffer and a peer rank");
      return NULL;
    }
    rc = op == 5 ? ph_route_send(r, ph_buf_device_ptr(bb->buf), ph_buf_bytes(bb->buf), b)
                 : ph_route_recv(r, ph_buf_device_ptr(bb->buf), ph_buf_bytes(bb->buf), b);
  } else {
    if (argc > 2) get_i32(env, argv[2], &a);
    switch (op) {
      case 0: rc = ph_route_group_begin(r); break;
      case 1: rc = ph_route_group_end(r); break;
      case 2: rc = ph_route_after_queue(r, a); break;
      case 3: rc = ph_queue_after_route(r, a); break;
      case 4: rc = ph_route_wait(r); break;
      default: break;
    }
  }
  if (rc != PH_OK) return throw_ph(env, "route");
  NAPI_OK(napi_get_undefined(env, &out));
  return out;
}

static napi_value AbiVersion(napi_env env, napi_callback_info info) {
  napi_value v;
  (void)info;
  napi_create_int32(env, ph_abi_version(), &v);
  return v;
}

NAPI_MODULE_INIT() {
  static const struct {
    const char *name;
    napi_callback fn;
  } fns[] = {
      {"abiVersion", AbiVersion},   {"setOption", SetOption},   {"createContext", CreateContext}, {"contextInfo", ContextInfo},
      {"createBuffer", CreateBuffer}, {"bufAddRef", BufAddRef},       {"bufRelease", BufRelease},
      {"bufRefCount", BufRefCount}, {"hostAccess", HostAccess},       {"waitFinish", WaitFinish},
      {"createProgram", CreateProgram}, {"runProgram", RunProgram}, {"runPrograms", RunPrograms}, {"eventElapsed", EventElapsed},   {"bufferStats", BufferStats},
      {"queueWaitQueue", QueueWaitQueue}, {"downloadAsync", DownloadAsync}, {"eventRecord", EventRecord},
      {"eventWait", EventWait},     {"eventDone", EventDone},       {"waitFinishSpin", WaitFinishSpin},
      {"resolveProgram", ResolveProgram}, {"gammaLut", GammaLut},   {"colourMatrix", ColourMatrix},
      {"transformMatrix", TransformMatrix}, {"planeBytes", PlaneBytes},
      {"routeUniqueId", RouteUniqueId}, {"routeInit", RouteInit},   {"routeOp", RouteOp},
      {"runProgramsProgress", RunProgramsProgress}, {"traceBegin", TraceBegin}, {"traceEnd", TraceEnd}, {"bufReuse", BufReuse},
  };
  for (size_t i = 0; i < sizeof fns / sizeof fns[0]; ++i) {
    napi_value f;
    if (napi_create_function(env, fns[i].name, NAPI_AUTO_LENGTH, fns[i].fn, NULL, &f) != napi_ok ||
        napi_set_named_property(env, exports, fns[i].name, f) != napi_ok)
      return NULL;
  }
  return exports;
}
