"""ctypes binding of libphaneron_hip.so (include/phaneron_hip.h).

Device memory, streams and process groups are torch's job (plumbing); every pixel is
computed by the library's HIP kernels.  Nothing here falls back to the CPU: a missing
library or a failing call raises PhaneronError.
"""
import ctypes as C
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# PHANERON_HIP_LIB selects an alternative build of the same library (A/B kernel experiments)
LIB_PATH = os.environ.get("PHANERON_HIP_LIB") or os.path.join(_HERE, "lib", "libphaneron_hip.so")

QUEUE_LOAD, QUEUE_PROCESS, QUEUE_UNLOAD = 0, 1, 2
HOST_READONLY, HOST_WRITEONLY, HOST_NONE = 0, 1, 2
ARG_BUF, ARG_U32, ARG_I32, ARG_F32 = 0, 1, 2, 3

EXPORTS = [
    "ph_abi_version", "ph_last_error", "ph_ctx_create", "ph_ctx_destroy", "ph_ctx_info", "ph_ctx_stream",
    "ph_wait_finish", "ph_buf_create", "ph_buf_wrap", "ph_buf_addref", "ph_buf_release", "ph_buf_refcount",
    "ph_buf_bytes", "ph_buf_device_ptr", "ph_buf_dims", "ph_buf_host_access", "ph_buf_host_ptr",
    "ph_ctx_buffer_stats", "ph_program_create", "ph_program_destroy", "ph_program_kernel", "ph_run_program", "ph_check_program", "ph_chan_compose", "ph_yadif_pair_packed",
    "ph_v210_pitch_bytes", "ph_v210_read", "ph_v210_read_batch", "ph_v210_write", "ph_yadif", "ph_yadif_pair", "ph_v210_yadif_pair", "ph_transform", "ph_resize", "ph_combine",
    "ph_transition_dissolve", "ph_transition_wipe", "ph_mixer", "ph_wipe", "ph_fused_v210_combine",
    "ph_colour_gamma2linear_lut", "ph_colour_linear2gamma_lut", "ph_colour_ycbcr2rgb_matrix",
    "ph_colour_rgb2ycbcr_matrix", "ph_colour_rgb2rgb_matrix", "ph_transform_matrix",
    "ph_lut_register", "ph_lut_unregister", "ph_lut_query", "ph_lut_layout_of", "ph_compose_up_write_v210_pair", "ph_compose_up_write_v210_batch", "ph_pack_read_batch", "ph_ctx_set_option", "ph_compose_write_v210", "ph_compose_wipe_write_v210",
    "ph_pack_plane_bytes", "ph_pack_read", "ph_pack_write", "ph_queue_wait_queue", "ph_buf_download_async",
    "ph_event_record", "ph_event_wait", "ph_event_query", "ph_event_destroy", "ph_queue_query",
    "ph_graph_begin", "ph_graph_end", "ph_graph_launch", "ph_graph_destroy", "ph_fused_v210_combine_batch",
    "ph_program_resolve", "ph_route_unique_id", "ph_route_init", "ph_route_destroy", "ph_route_group_begin",
    "ph_route_group_end", "ph_route_send", "ph_route_recv", "ph_route_after_queue", "ph_queue_after_route",
    "ph_route_wait", "ph_route_stream", "ph_route_comm_count", "ph_chan_compose_v210", "ph_chan_compose_batch", "ph_run_programs", "ph_event_record_timed", "ph_event_elapsed_us", "ph_ctx_host_pool_stats",
    "ph_v210_yadif_pair_fmt", "ph_compose_up_write_v210", "ph_trace_begin", "ph_trace_end", "ph_run_programs_progress", "ph_buf_reuse", "ph_image_unpack_rgb",
]


class PhaneronError(RuntimeError):
    pass


class _ArgVal(C.Union):
    _fields_ = [("buf", C.c_void_p), ("u32", C.c_uint32), ("i32", C.c_int32), ("f32", C.c_float)]


class PhArg(C.Structure):
    _fields_ = [("name", C.c_char_p), ("kind", C.c_int), ("v", _ArgVal)]


class PhLayer(C.Structure):
    _fields_ = [("rgba", C.c_void_p), ("width", C.c_int), ("height", C.c_int), ("matrix9", C.c_void_p)]


class PhLayerWipe(C.Structure):
    _fields_ = [("incoming_rgba", C.c_void_p), ("mask_rgba", C.c_void_p)]


class PhDeintSource(C.Structure):
    _fields_ = [("prev", C.c_void_p), ("cur", C.c_void_p), ("next", C.c_void_p), ("out_parity0", C.c_void_p), ("out_parity1", C.c_void_p),
                ("prev_u", C.c_void_p), ("prev_v", C.c_void_p), ("cur_u", C.c_void_p), ("cur_v", C.c_void_p), ("next_u", C.c_void_p), ("next_v", C.c_void_p)]


class PhChanSource(C.Structure):
    _fields_ = [("data", C.c_void_p), ("format", C.c_int), ("width", C.c_int), ("height", C.c_int),
                ("matrix9_host", C.POINTER(C.c_float)), ("data_u", C.c_void_p), ("data_v", C.c_void_p), ("col_matrix12", C.c_void_p)]


class PhChanLayer(C.Structure):
    _fields_ = [("src", PhChanSource), ("transition", C.c_int), ("mix", C.c_float), ("incoming", PhChanSource), ("mask", PhChanSource)]


class PhChanJob(C.Structure):
    _fields_ = [("n", C.c_int), ("layers", C.POINTER(PhChanLayer)), ("out", C.c_void_p), ("interlace", C.c_uint32)]


class PhImageLayer(C.Structure):
    _fields_ = [("data", C.c_void_p), ("format", C.c_int), ("width", C.c_int), ("height", C.c_int), ("matrix9_host", C.POINTER(C.c_float))]


IMG_RGBA_F32, IMG_RGB_F32 = 0, 1
SRC_V210, SRC_RGBA_F32, SRC_YUV422P10, SRC_YUV422P8, SRC_YUV420P, SRC_NV12, SRC_RGBA8, SRC_BGRA8 = 1, 2, 3, 4, 5, 6, 7, 8
SRC_PLANAR = {"yuv422p10": SRC_YUV422P10, "yuv422p8": SRC_YUV422P8, "yuv420p": SRC_YUV420P, "nv12": SRC_NV12}
TRANSITION_CUT, TRANSITION_DISSOLVE, TRANSITION_WIPE = 0, 1, 2


class LutLayout(C.Structure):  # ph_lut_layout
    _fields_ = [("lds_bytes", C.c_uint32), ("hole", C.c_uint32), ("delta_off", C.c_uint32), ("shift", C.c_uint32), ("index_bias", C.c_uint32),
                ("a_scale", C.c_float)]


class RunTimings(C.Structure):
    _fields_ = [("data_to_kernel", C.c_uint32), ("kernel_exec", C.c_uint32), ("total_time", C.c_uint32)]


_lib = None


def lib():
    """The loaded library.  Raises if it has not been built (python -m phaneron_amd.build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PhaneronError("%s is missing: build it with `python -m phaneron_amd.build` "
                            "(there is no CPU fallback)" % LIB_PATH)
    # A process that also uses PyTorch must bring torch's bundled HIP runtime in FIRST: loaded the other way round
    # (this library's /opt/rocm libamdhip64, then torch on top of it) the second runtime finds no device
    # ("no ROCm-capable device is detected" - seen on the GPU box with a test that called a host-maths function
    # before importing torch).  PyTorch is only the tests' and bench's device-memory plumbing; a process without it (the
    # node addon, a C caller) never takes this branch.  PH_NO_TORCH_PRELOAD=1 skips it.
    if "torch" not in sys.modules and os.environ.get("PH_NO_TORCH_PRELOAD") != "1":
        import importlib.util
        if importlib.util.find_spec("torch") is not None:
            import torch  # noqa: F401
    l = C.CDLL(LIB_PATH)
    vp, ci, cu, cf, cd, cs = C.c_void_p, C.c_int, C.c_uint32, C.c_float, C.c_double, C.c_size_t
    f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
    sig = {
        "ph_abi_version": (ci, []),
        "ph_last_error": (C.c_char_p, [vp]),
        "ph_ctx_create": (ci, [ci, C.POINTER(vp)]),
        "ph_ctx_destroy": (ci, [vp]),
        "ph_ctx_info": (ci, [vp, C.c_char_p, cs, C.c_char_p, cs]),
        "ph_ctx_stream": (vp, [vp, ci]),
        "ph_wait_finish": (ci, [vp, ci]),
        "ph_buf_create": (ci, [vp, cs, ci, ci, ci, ci, C.c_char_p, C.POINTER(vp)]),
        "ph_buf_wrap": (ci, [vp, vp, cs, ci, ci, C.POINTER(vp)]),
        "ph_buf_addref": (ci, [vp]),
        "ph_buf_release": (ci, [vp]),
        "ph_buf_refcount": (ci, [vp]),
        "ph_buf_bytes": (cs, [vp]),
        "ph_buf_device_ptr": (vp, [vp]),
        "ph_buf_dims": (ci, [vp, C.POINTER(ci), C.POINTER(ci)]),
        "ph_buf_host_access": (ci, [vp, ci, ci, vp, cs]),
        "ph_buf_host_ptr": (vp, [vp]),
        "ph_queue_wait_queue": (ci, [vp, ci, ci]),
        "ph_buf_download_async": (ci, [vp, ci]),
        "ph_queue_query": (ci, [vp, ci]),
        "ph_graph_begin": (ci, [vp, ci]),
        "ph_graph_end": (ci, [vp, ci, C.POINTER(vp)]),
        "ph_graph_launch": (ci, [vp, ci]),
        "ph_graph_destroy": (ci, [vp]),
        "ph_event_record": (ci, [vp, ci, C.POINTER(vp)]),
        "ph_event_wait": (ci, [vp]),
        "ph_event_query": (ci, [vp]),
        "ph_event_destroy": (ci, [vp]),
        "ph_ctx_buffer_stats": (ci, [vp, C.POINTER(cs), C.POINTER(cs), C.POINTER(cs)]),
        "ph_program_create": (ci, [vp, C.c_char_p, C.c_char_p, C.POINTER(cu), ci, cu, C.POINTER(vp)]),
        "ph_program_resolve": (ci, [C.c_char_p, C.c_char_p, C.c_char_p, cs, C.POINTER(ci), C.POINTER(ci)]),
        "ph_program_destroy": (ci, [vp]),
        "ph_program_kernel": (C.c_char_p, [vp]),
        "ph_run_program": (ci, [vp, vp, C.POINTER(PhArg), ci, ci, C.POINTER(RunTimings)]),
        "ph_check_program": (ci, [vp, vp, C.POINTER(PhArg), ci, ci]),
        "ph_v210_pitch_bytes": (cu, [cu]),
        "ph_v210_read": (ci, [vp, ci, vp, vp, cu, cu, vp, vp, vp]),
        "ph_v210_read_batch": (ci, [vp, ci, ci, vp, vp, cu, cu, vp, vp, vp]),
        "ph_v210_write": (ci, [vp, ci, vp, vp, cu, cu, cu, vp, vp]),
        "ph_yadif": (ci, [vp, ci, vp, vp, vp, ci, ci, ci, ci, ci, vp]),
        "ph_yadif_pair": (ci, [vp, ci, vp, vp, vp, ci, ci, ci, ci, vp, vp]),
        "ph_v210_yadif_pair": (ci, [vp, ci, ci, vp, cu, cu, ci, ci, vp, vp, vp]),
        "ph_v210_yadif_pair_fmt": (ci, [vp, ci, ci, vp, cu, cu, ci, ci, ci, vp, vp, vp]),
        "ph_yadif_pair_packed": (ci, [vp, ci, ci, vp, ci, cu, cu, ci, ci, ci, vp, vp, vp]),
        "ph_transform": (ci, [vp, ci, vp, ci, ci, vp, vp, ci, ci]),
        "ph_resize": (ci, [vp, ci, vp, ci, ci, cf, cf, cf, vp, vp, ci, ci]),
        "ph_combine": (ci, [vp, ci, ci, C.POINTER(vp), ci, ci, vp]),
        "ph_transition_dissolve": (ci, [vp, ci, vp, vp, cf, ci, ci, vp]),
        "ph_transition_wipe": (ci, [vp, ci, vp, vp, vp, ci, ci, vp]),
        "ph_mixer": (ci, [vp, ci, vp, vp, cf, ci, ci, vp]),
        "ph_wipe": (ci, [vp, ci, vp, vp, cf, ci, ci, vp]),
        "ph_fused_v210_combine": (ci, [vp, ci, ci, C.POINTER(vp), vp, cu, cu, vp, vp, vp, vp, vp]),
        "ph_fused_v210_combine_batch": (ci, [vp, ci, ci, ci, C.POINTER(vp), C.POINTER(vp), cu, cu, vp, vp, vp, vp, vp]),
        "ph_colour_gamma2linear_lut": (ci, [C.c_char_p, f32p]),
        "ph_colour_linear2gamma_lut": (ci, [C.c_char_p, f32p]),
        "ph_colour_ycbcr2rgb_matrix": (ci, [C.c_char_p, ci, ci, ci, ci, f32p]),
        "ph_colour_rgb2ycbcr_matrix": (ci, [C.c_char_p, ci, ci, ci, ci, f32p]),
        "ph_colour_rgb2rgb_matrix": (ci, [C.c_char_p, C.c_char_p, f32p]),
        "ph_transform_matrix": (ci, [ci, ci, ci, ci, cd, cd, cd, cd, cd, cd, cd, f32p]),
        "ph_lut_register": (ci, [vp, vp, f32p]),
        "ph_lut_unregister": (ci, [vp, vp]),
        "ph_lut_query": (ci, [vp, vp, C.POINTER(cu), C.POINTER(cu), C.POINTER(cu)]),
        "ph_lut_layout_of": (ci, [f32p, C.POINTER(LutLayout), vp, C.c_size_t]),
        "ph_compose_up_write_v210_pair": (ci, [vp, ci, ci, C.POINTER(PhImageLayer), C.POINTER(PhImageLayer), vp, vp, cu, cu, cu, vp, vp]),
        "ph_compose_up_write_v210_batch": (ci, [vp, ci, ci, ci, C.POINTER(C.POINTER(PhImageLayer)), C.POINTER(vp), cu, cu, cu, vp, vp]),
        "ph_pack_read_batch": (ci, [vp, ci, ci, ci, C.POINTER(vp * 3), C.POINTER(vp), cu, cu, vp, vp, vp]),
        "ph_ctx_set_option": (ci, [vp, C.c_char_p, ci]),
        "ph_pack_plane_bytes": (ci, [ci, cu, cu, C.POINTER(cs)]),
        "ph_pack_read": (ci, [vp, ci, ci, C.POINTER(vp), vp, cu, cu, vp, vp, vp]),
        "ph_pack_write": (ci, [vp, ci, ci, vp, C.POINTER(vp), cu, cu, cu, vp, vp]),
        "ph_compose_write_v210": (ci, [vp, ci, ci, C.POINTER(PhLayer), vp, cu, cu, cu, vp, vp]),
        "ph_compose_wipe_write_v210": (ci, [vp, ci, ci, C.POINTER(PhLayer), C.POINTER(PhLayerWipe), vp, cu, cu, cu, vp, vp]),
        "ph_compose_up_write_v210": (ci, [vp, ci, ci, C.POINTER(PhImageLayer), vp, cu, cu, cu, vp, vp]),
        "ph_chan_compose_v210": (ci, [vp, ci, ci, C.POINTER(PhChanLayer), vp, cu, cu, cu, vp, vp, vp, vp, vp]),
        "ph_chan_compose": (ci, [vp, ci, ci, C.POINTER(PhChanLayer), ci, C.POINTER(C.c_void_p), cu, cu, cu, vp, vp, vp, vp, vp]),
        "ph_chan_compose_batch": (ci, [vp, ci, ci, C.POINTER(PhChanJob), cu, cu, vp, vp, vp, vp, vp]),
        "ph_ctx_host_pool_stats": (ci, [vp, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(C.c_uint64)]),
        "ph_event_record_timed": (ci, [vp, ci, C.POINTER(vp)]),
        "ph_event_elapsed_us": (ci, [vp, vp, C.POINTER(cu)]),
        "ph_run_programs": (ci, [vp, ci, C.POINTER(vp), C.POINTER(C.POINTER(PhArg)), C.POINTER(ci), ci]),
        "ph_route_unique_id": (ci, [vp]),
        "ph_route_init": (ci, [vp, vp, ci, ci, C.POINTER(vp)]),
        "ph_route_destroy": (ci, [vp]),
        "ph_route_group_begin": (ci, [vp]),
        "ph_route_group_end": (ci, [vp]),
        "ph_route_send": (ci, [vp, vp, cs, ci]),
        "ph_route_recv": (ci, [vp, vp, cs, ci]),
        "ph_route_after_queue": (ci, [vp, ci]),
        "ph_queue_after_route": (ci, [vp, ci]),
        "ph_route_wait": (ci, [vp]),
        "ph_route_stream": (vp, [vp]),
        "ph_route_comm_count": (ci, [vp, C.POINTER(C.c_int)]),
        "ph_run_programs_progress": (ci, [C.POINTER(ci)]),
        "ph_buf_reuse": (ci, [vp]),
        "ph_image_unpack_rgb": (ci, [vp, ci, vp, ci, ci]),
        "ph_trace_begin": (ci, [ci]),
        "ph_trace_end": (ci, [C.c_char_p, cs]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(l, name)  # AttributeError here = the header and the library disagree
        fn.restype, fn.argtypes = res, args
    _lib = l
    return l


def check(rc, ctx=None):
    if rc != 0:
        msg = lib().ph_last_error(ctx)
        raise PhaneronError("libphaneron_hip error %d: %s" % (rc, msg.decode() if msg else "?"))


class trace:
    """`with capi.trace(dry_run=False) as t: <calls>` - t.route is the '+'-joined list of the kernels those calls launched on
    this thread (ph_trace_begin / ph_trace_end); dry_run: the calls choose their kernels but enqueue nothing."""

    def __init__(self, dry_run=False):
        self.dry_run, self.route = dry_run, None

    def __enter__(self):
        check(lib().ph_trace_begin(1 if self.dry_run else 0))
        return self

    def __exit__(self, *exc):
        buf = C.create_string_buffer(4096)
        rc = lib().ph_trace_end(buf, len(buf))
        if exc[0] is None:
            check(rc)
        self.route = buf.value.decode()
        return False

    @property
    def kernels(self):
        return self.route.split("+") if self.route else []


# ---- host colour maths (pure host code in the library; no device needed) ---------------------
def lut_layout(lut):
    """(layout dict, LDS image as uint8 array) of a 65536-entry f32 table as the table kernels hold it; (None, None) when the
    table does not compress exactly (ph_lut_layout_of; needs no device)."""
    lut = np.ascontiguousarray(lut, np.float32)
    assert lut.shape == (65536,)
    lay = LutLayout()
    n = lib().ph_lut_layout_of(lut, C.byref(lay), None, 0)
    if n < 0:
        check(n)
    if n == 0:
        return None, None
    image = np.zeros(n, np.uint8)
    if lib().ph_lut_layout_of(lut, C.byref(lay), image.ctypes.data_as(C.c_void_p), n) != n:
        check(-1)
    return {k: getattr(lay, k) for k, _ in LutLayout._fields_}, image


def gamma2linear_lut(colspec):
    out = np.empty(65536, np.float32)
    check(lib().ph_colour_gamma2linear_lut(colspec.encode(), out))
    return out


def linear2gamma_lut(colspec):
    out = np.empty(65536, np.float32)
    check(lib().ph_colour_linear2gamma_lut(colspec.encode(), out))
    return out


def ycbcr2rgb_matrix(colspec, num_bits=10, luma_black=64, luma_white=940, chroma_range=896):
    out = np.empty(12, np.float32)
    check(lib().ph_colour_ycbcr2rgb_matrix(colspec.encode(), num_bits, luma_black, luma_white, chroma_range, out))
    return out


def rgb2ycbcr_matrix(colspec, num_bits=10, luma_black=64, luma_white=940, chroma_range=896):
    out = np.empty(12, np.float32)
    check(lib().ph_colour_rgb2ycbcr_matrix(colspec.encode(), num_bits, luma_black, luma_white, chroma_range, out))
    return out


def rgb2rgb_matrix(src, dst):
    out = np.empty(9, np.float32)
    check(lib().ph_colour_rgb2rgb_matrix(src.encode(), dst.encode(), out))
    return out


def transform_matrix(width, height, flip_h=False, flip_v=False, anchor_x=0.0, anchor_y=0.0, scale_x=1.0,
                     scale_y=1.0, offset_x=0.0, offset_y=0.0, rotate=0.0):
    out = np.empty(9, np.float32)
    check(lib().ph_transform_matrix(width, height, int(flip_h), int(flip_v), anchor_x, anchor_y, scale_x, scale_y,
                                    offset_x, offset_y, rotate, out))
    return out


RESOLVED_HOW = {0: "tag", 1: "name", 2: "text", 3: "signature"}


def resolve_program(source, name):
    """Which precompiled kernel createProgram(source, {name}) selects - no context, no device.
    Returns (kernel id, pack format name or None, how: "tag" | "name" | "text" | "signature")."""
    buf = C.create_string_buffer(64)
    fmt, how = C.c_int(), C.c_int()
    src = source.encode() if isinstance(source, str) else source
    check(lib().ph_program_resolve(src, name.encode(), buf, 64, C.byref(fmt), C.byref(how)))
    names = {v: k for k, v in FORMATS.items()}
    return buf.value.decode(), names.get(fmt.value), RESOLVED_HOW[how.value]


def v210_pitch_bytes(width):
    return int(lib().ph_v210_pitch_bytes(width))


FORMATS = {"v210": 0, "yuv422p10": 1, "yuv422p8": 2, "yuv420p": 3, "nv12": 4, "rgba8": 5, "bgra8": 6}
# (numBits, lumaBlack, lumaWhite, chromaRange) of each format's Reader/Writer; None = RGB (no YCbCr matrix)
FORMAT_RANGE = {"v210": (10, 64, 940, 896), "yuv422p10": (10, 64, 940, 896), "yuv422p8": (8, 16, 235, 224),
                "yuv420p": (8, 16, 235, 224), "nv12": (8, 16, 235, 224), "rgba8": None, "bgra8": None}


def pack_plane_bytes(fmt, width, height):
    b = (C.c_size_t * 3)()
    n = lib().ph_pack_plane_bytes(FORMATS[fmt], width, height, b)
    if n < 0:
        check(n)
    return [int(b[i]) for i in range(n)]


# ---- device side ---------------------------------------------------------------------------------
def _ptr(t):
    """Device pointer of a torch CUDA tensor (or a raw int)."""
    return C.c_void_p(t if isinstance(t, int) else t.data_ptr())


class Context:
    """One GPU: three in-order HIP streams (load / process / unload) and a buffer pool -
    the nodencl `clContext` of the reference (src/index.ts:94-108)."""

    def __init__(self, device_index=0):
        h = C.c_void_p()
        check(lib().ph_ctx_create(device_index, C.byref(h)))
        self.h = h
        self.device_index = device_index

    def close(self):
        if self.h:
            lib().ph_ctx_destroy(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def info(self):
        v, d = C.create_string_buffer(128), C.create_string_buffer(256)
        check(lib().ph_ctx_info(self.h, v, 128, d, 256), self.h)
        return v.value.decode(), d.value.decode()

    def stream_ptr(self, queue=QUEUE_PROCESS):
        return lib().ph_ctx_stream(self.h, queue)

    def torch_stream(self, queue=QUEUE_PROCESS):
        import torch
        return torch.cuda.ExternalStream(self.stream_ptr(queue), device=torch.device("cuda", self.device_index))

    def wait(self, queue=QUEUE_PROCESS):
        check(lib().ph_wait_finish(self.h, queue), self.h)

    def register_lut(self, device_lut, host_lut):
        """Tell the library the host contents of a device gamma LUT so its kernels can keep an exact
        compressed copy in LDS.  Returns True if the table is LDS-capable."""
        rc = lib().ph_lut_register(self.h, _ptr(device_lut), np.ascontiguousarray(host_lut, np.float32))
        if rc < 0:
            check(rc, self.h)
        return rc == 1

    def unregister_lut(self, device_lut):
        check(lib().ph_lut_unregister(self.h, _ptr(device_lut)), self.h)

    def lut_info(self, device_lut):
        b, t, s = C.c_uint32(), C.c_uint32(), C.c_uint32()
        check(lib().ph_lut_query(self.h, _ptr(device_lut), C.byref(b), C.byref(t), C.byref(s)), self.h)
        return dict(lds_bytes=b.value, index_bias=t.value, blocks_per_octave_log2=s.value)

    def set_option(self, name, value):
        check(lib().ph_ctx_set_option(self.h, name.encode(), int(value)), self.h)

    # typed kernels on torch tensors / raw pointers ------------------------------------------------
    def v210_read(self, src, dst, width, height, col_matrix, lut, gamut, queue=QUEUE_PROCESS):
        check(lib().ph_v210_read(self.h, queue, _ptr(src), _ptr(dst), width, height, _ptr(col_matrix), _ptr(lut),
                                 _ptr(gamut)), self.h)

    def v210_read_batch(self, srcs, dsts, width, height, col_matrix, lut, gamut, queue=QUEUE_PROCESS):
        """several frames of one size and colour recipe in one launch; dsts[i] = v210_read(srcs[i])"""
        n = len(srcs)
        ins = (C.c_void_p * n)(*[_ptr(b).value for b in srcs])
        outs = (C.c_void_p * n)(*[_ptr(b).value for b in dsts])
        check(lib().ph_v210_read_batch(self.h, queue, n, ins, outs, width, height, _ptr(col_matrix), _ptr(lut), _ptr(gamut)), self.h)

    def v210_write(self, src, dst, width, height, interlace, col_matrix, lut, queue=QUEUE_PROCESS):
        check(lib().ph_v210_write(self.h, queue, _ptr(src), _ptr(dst), width, height, interlace, _ptr(col_matrix),
                                  _ptr(lut)), self.h)

    def yadif(self, prev, cur, nxt, dst, width, height, parity, tff, skip_spatial=False, queue=QUEUE_PROCESS):
        check(lib().ph_yadif(self.h, queue, _ptr(prev), _ptr(cur), _ptr(nxt), width, height, int(parity), int(tff),
                             int(skip_spatial), _ptr(dst)), self.h)

    def yadif_pair(self, prev, cur, nxt, dst_parity0, dst_parity1, width, height, tff, skip_spatial=False, queue=QUEUE_PROCESS):
        """both fields of one frame in one pass: dst_parity0 / dst_parity1 = yadif(..., parity=0 / 1)"""
        check(lib().ph_yadif_pair(self.h, queue, _ptr(prev), _ptr(cur), _ptr(nxt), width, height, int(tff), int(skip_spatial),
                                  _ptr(dst_parity0), _ptr(dst_parity1)), self.h)

    def v210_yadif_pair(self, sources, width, height, tff, skip_spatial, col_matrix, lut, gamut, queue=QUEUE_PROCESS, rgb=False,
                        prepare_only=False, packing="v210"):
        """sources: [(prev, cur, next, dst_parity0, dst_parity1)] - v210 windows in, both de-interlaced fields out (f32 RGBA,
        or with rgb=True packed f32 RGB, 12 bytes per pixel); == v210_read x 3 -> yadif x 2 per source, as one kernel.
        packing "yuv422p10" / "yuv422p8" / "yuv420p": prev, cur, next are (y, u, v) plane triples, "nv12": (y, cbcr) pairs (ph_yadif_pair_packed)"""
        arr = (PhDeintSource * len(sources))()
        for i, s in enumerate(sources):
            if packing == "v210":
                arr[i].prev, arr[i].cur, arr[i].next = (_ptr(b).value for b in s[:3])
            else:  # (nv12: two planes per frame - Y and the interleaved CbCr plane)
                (arr[i].prev, arr[i].prev_u, arr[i].prev_v), (arr[i].cur, arr[i].cur_u, arr[i].cur_v), (arr[i].next, arr[i].next_u, arr[i].next_v) = (
                    (tuple(_ptr(p).value for p in frame) + (None,))[:3] for frame in s[:3])
            arr[i].out_parity0, arr[i].out_parity1 = _ptr(s[3]).value, _ptr(s[4]).value
        args = (self.h, queue, len(sources), arr, FORMATS[packing], width, height, int(tff), int(skip_spatial), IMG_RGB_F32 if rgb else IMG_RGBA_F32,
                _ptr(col_matrix), _ptr(lut), _ptr(gamut))
        if prepare_only:
            fn, h = lib().ph_yadif_pair_packed, self.h

            def job(_keep=(sources,)):
                check(fn(*args), h)
            return job
        check(lib().ph_yadif_pair_packed(*args), self.h)

    def compose_up_write_v210_pair(self, layers_a, layers_b, dst_a, dst_b, out_w, out_h, interlace, wr_cm, wr_lut, queue=QUEUE_PROCESS, rgb=False,
                                   prepare_only=False):
        """Both fields of a de-interlaced frame in one launch (ph_compose_up_write_v210_pair): layers_a -> dst_a and layers_b -> dst_b, the
        two sets differing in their tensors only.  Arguments as compose_up_write_v210."""
        import numpy as np
        keep = []

        def arr_of(layers):
            arr = (PhImageLayer * len(layers))()
            for i, (t, w, h, m) in enumerate(layers):
                mh = np.ascontiguousarray(m, np.float32)
                keep.append(mh)
                arr[i].data, arr[i].width, arr[i].height = _ptr(t).value, w, h
                arr[i].format = IMG_RGB_F32 if rgb else IMG_RGBA_F32
                arr[i].matrix9_host = mh.ctypes.data_as(C.POINTER(C.c_float))
            return arr
        assert len(layers_a) == len(layers_b)
        aa, ab = arr_of(layers_a), arr_of(layers_b)
        args = (self.h, queue, len(layers_a), aa, ab, _ptr(dst_a), _ptr(dst_b), out_w, out_h, interlace, _ptr(wr_cm), _ptr(wr_lut))
        if prepare_only:
            fn, h = lib().ph_compose_up_write_v210_pair, self.h

            def job(_keep=(keep, layers_a, layers_b, dst_a, dst_b, aa, ab)):
                check(fn(*args), h)
            return job
        check(lib().ph_compose_up_write_v210_pair(*args), self.h)

    def compose_up_write_v210_batch(self, layer_sets, dsts, out_w, out_h, interlace, wr_cm, wr_lut, queue=QUEUE_PROCESS, rgb=False, prepare_only=False):
        """1 .. 4 sets of layers that differ in their tensors only, each into its own frame, in one launch (ph_compose_up_write_v210_batch).
        Arguments as compose_up_write_v210."""
        import numpy as np
        keep = []
        arrs = []
        for layers in layer_sets:
            arr = (PhImageLayer * len(layers))()
            for i, (t, w, h, m) in enumerate(layers):
                mh = np.ascontiguousarray(m, np.float32)
                keep.append(mh)
                arr[i].data, arr[i].width, arr[i].height = _ptr(t).value, w, h
                arr[i].format = IMG_RGB_F32 if rgb else IMG_RGBA_F32
                arr[i].matrix9_host = mh.ctypes.data_as(C.POINTER(C.c_float))
            arrs.append(arr)
        sets = (C.POINTER(PhImageLayer) * len(arrs))(*[C.cast(a, C.POINTER(PhImageLayer)) for a in arrs])
        outs = (C.c_void_p * len(dsts))(*[_ptr(d).value for d in dsts])
        args = (self.h, queue, len(layer_sets), len(layer_sets[0]), sets, outs, out_w, out_h, interlace, _ptr(wr_cm), _ptr(wr_lut))
        if prepare_only:
            fn, h = lib().ph_compose_up_write_v210_batch, self.h

            def job(_keep=(keep, arrs, layer_sets, dsts, sets, outs)):
                check(fn(*args), h)
            return job
        check(lib().ph_compose_up_write_v210_batch(*args), self.h)

    def compose_up_write_v210(self, layers, dst, out_w, out_h, interlace, wr_cm, wr_lut, queue=QUEUE_PROCESS, rgb=False, prepare_only=False):
        """The 2 x 2-block compositor for layers enlarged 2x or more (ph_compose_up_write_v210).  layers: [(tensor, width,
        height, matrix)] with matrix = nine host floats (transform_matrix); rgb=True: the tensors are packed f32 RGB."""
        import numpy as np
        arr = (PhImageLayer * len(layers))()
        keep = []
        for i, (t, w, h, m) in enumerate(layers):
            mh = np.ascontiguousarray(m, np.float32)
            keep.append(mh)
            arr[i].data, arr[i].width, arr[i].height = _ptr(t).value, w, h
            arr[i].format = IMG_RGB_F32 if rgb else IMG_RGBA_F32
            arr[i].matrix9_host = mh.ctypes.data_as(C.POINTER(C.c_float))
        args = (self.h, queue, len(layers), arr, _ptr(dst), out_w, out_h, interlace, _ptr(wr_cm), _ptr(wr_lut))
        if prepare_only:
            fn, h = lib().ph_compose_up_write_v210, self.h

            def job(_keep=(keep, layers, dst)):
                check(fn(*args), h)
            return job
        check(lib().ph_compose_up_write_v210(*args), self.h)

    def image_unpack_rgb(self, image, width, height, queue=QUEUE_PROCESS):
        """a packed f32 RGB image (12 bytes per pixel) expanded in place into the f32 RGBA image its buffer is sized for (ph_image_unpack_rgb)"""
        check(lib().ph_image_unpack_rgb(self.h, queue, _ptr(image), width, height), self.h)

    def transform(self, src, in_w, in_h, matrix, dst, out_w, out_h, queue=QUEUE_PROCESS):
        check(lib().ph_transform(self.h, queue, _ptr(src), in_w, in_h, _ptr(matrix), _ptr(dst), out_w, out_h), self.h)

    def resize(self, src, in_w, in_h, scale, offset_x, offset_y, flip, dst, out_w, out_h, queue=QUEUE_PROCESS):
        check(lib().ph_resize(self.h, queue, _ptr(src), in_w, in_h, scale, offset_x, offset_y, _ptr(flip), _ptr(dst),
                              out_w, out_h), self.h)

    def combine(self, layers, dst, width, height, queue=QUEUE_PROCESS):
        arr = (C.c_void_p * len(layers))(*[_ptr(l).value for l in layers])
        check(lib().ph_combine(self.h, queue, len(layers), arr, width, height, _ptr(dst)), self.h)

    def transition_dissolve(self, in0, in1, mix, dst, width, height, queue=QUEUE_PROCESS):
        check(lib().ph_transition_dissolve(self.h, queue, _ptr(in0), _ptr(in1), mix, width, height, _ptr(dst)), self.h)

    def transition_wipe(self, in0, in1, mask, dst, width, height, queue=QUEUE_PROCESS):
        check(lib().ph_transition_wipe(self.h, queue, _ptr(in0), _ptr(in1), _ptr(mask), width, height, _ptr(dst)),
              self.h)

    def mixer(self, in0, in1, mix, dst, width, height, queue=QUEUE_PROCESS):
        check(lib().ph_mixer(self.h, queue, _ptr(in0), _ptr(in1), mix, width, height, _ptr(dst)), self.h)

    def wipe(self, in0, in1, wipe, dst, width, height, queue=QUEUE_PROCESS):
        check(lib().ph_wipe(self.h, queue, _ptr(in0), _ptr(in1), wipe, width, height, _ptr(dst)), self.h)

    def pack_read(self, fmt, planes, dst, width, height, col_matrix, lut, gamut, queue=QUEUE_PROCESS):
        arr = (C.c_void_p * 3)(*([_ptr(p).value for p in planes] + [None] * (3 - len(planes))))
        check(lib().ph_pack_read(self.h, queue, FORMATS[fmt], arr, _ptr(dst), width, height,
                                 None if col_matrix is None else _ptr(col_matrix), _ptr(lut), _ptr(gamut)), self.h)

    def pack_read_batch(self, fmt, frames, dsts, width, height, col_matrix, lut, gamut, queue=QUEUE_PROCESS):
        """several frames of one format, size and Loader recipe in one launch (ph_pack_read_batch): frames = [planes, ...] as pack_read takes them"""
        arr = ((C.c_void_p * 3) * len(frames))()
        for i, planes in enumerate(frames):
            for k, p in enumerate(planes):
                arr[i][k] = _ptr(p).value
        outs = (C.c_void_p * len(dsts))(*[_ptr(d).value for d in dsts])
        check(lib().ph_pack_read_batch(self.h, queue, FORMATS[fmt], len(frames), arr, outs, width, height,
                                       None if col_matrix is None else _ptr(col_matrix), _ptr(lut), _ptr(gamut)), self.h)

    def pack_write(self, fmt, src, planes, width, height, interlace, col_matrix, lut, queue=QUEUE_PROCESS):
        arr = (C.c_void_p * 3)(*([_ptr(p).value for p in planes] + [None] * (3 - len(planes))))
        check(lib().ph_pack_write(self.h, queue, FORMATS[fmt], _ptr(src), arr, width, height, interlace,
                                  None if col_matrix is None else _ptr(col_matrix), _ptr(lut)), self.h)

    def compose_write_v210(self, layers, dst, out_w, out_h, interlace, wr_cm, wr_lut, queue=QUEUE_PROCESS):
        """layers: list of (rgba tensor, width, height, matrix tensor or None)"""
        arr = (PhLayer * len(layers))()
        for i, (t, w, h, m) in enumerate(layers):
            arr[i].rgba, arr[i].width, arr[i].height = _ptr(t).value, w, h
            arr[i].matrix9 = _ptr(m).value if m is not None else None
        check(lib().ph_compose_write_v210(self.h, queue, len(layers), arr, _ptr(dst), out_w, out_h, interlace,
                                          _ptr(wr_cm), _ptr(wr_lut)), self.h)

    def compose_wipe_write_v210(self, layers, wipes, dst, out_w, out_h, interlace, wr_cm, wr_lut, queue=QUEUE_PROCESS):
        """layers as compose_write_v210; wipes: per layer None or (incoming rgba tensor, mask rgba tensor)"""
        arr = (PhLayer * len(layers))()
        for i, (t, w, h, m) in enumerate(layers):
            arr[i].rgba, arr[i].width, arr[i].height = _ptr(t).value, w, h
            arr[i].matrix9 = _ptr(m).value if m is not None else None
        wp = (PhLayerWipe * len(layers))()
        for i, wv in enumerate(wipes):
            if wv is not None:
                wp[i].incoming_rgba = _ptr(wv[0]).value if wv[0] is not None else None
                wp[i].mask_rgba = _ptr(wv[1]).value if wv[1] is not None else None
        check(lib().ph_compose_wipe_write_v210(self.h, queue, len(layers), arr, wp, _ptr(dst), out_w, out_h, interlace,
                                               _ptr(wr_cm), _ptr(wr_lut)), self.h)

    @staticmethod
    def _chan_layers(layers):
        """the ph_chan_layer array of a list of layer dicts (chan_compose_v210) + what has to stay alive with it"""
        import numpy as np
        arr = (PhChanLayer * len(layers))()
        keep = []

        def fill(dst_src, spec):
            t, w, h, m = spec[:4]
            kind = spec[4] if len(spec) > 4 else "v210"
            if kind in SRC_PLANAR:  # t: the planes (nv12: two); an optional sixth element: the source's own Loader matrix (device)
                planes = [_ptr(p).value for p in t]
                dst_src.data, dst_src.data_u = planes[0], planes[1]
                dst_src.data_v = planes[2] if len(planes) > 2 else None
                if len(spec) > 5 and spec[5] is not None:
                    dst_src.col_matrix12 = _ptr(spec[5]).value
            else:
                dst_src.data = _ptr(t).value
            dst_src.width, dst_src.height = w, h
            dst_src.format = SRC_PLANAR.get(kind, {"rgba": SRC_RGBA_F32, "rgba8": SRC_RGBA8, "bgra8": SRC_BGRA8}.get(kind, SRC_V210))
            if m is not None:
                mh = np.ascontiguousarray(m, np.float32)
                keep.append(mh)
                dst_src.matrix9_host = mh.ctypes.data_as(C.POINTER(C.c_float))
        for i, L in enumerate(layers):
            fill(arr[i].src, L["src"])
            arr[i].transition = {"cut": TRANSITION_CUT, "dissolve": TRANSITION_DISSOLVE, "wipe": TRANSITION_WIPE}[L.get("transition", "cut")]
            arr[i].mix = float(L.get("mix", 0.0))
            if L.get("incoming") is not None:
                fill(arr[i].incoming, L["incoming"])
            if L.get("mask") is not None:
                fill(arr[i].mask, L["mask"])
        return arr, keep

    def chan_compose_v210(self, layers, dst, out_w, out_h, interlace, rd_cm, rd_lut, rd_gm, wr_cm, wr_lut, queue=QUEUE_PROCESS,
                          prepare_only=False, out_fmt="v210"):
        """The channel compositor straight from v210 sources (ph_chan_compose_v210).  layers: list of dicts
        {src: SOURCE, transition: "cut" | "dissolve" | "wipe", mix: float, incoming: SOURCE, mask: SOURCE}; a SOURCE is
        (tensor, width, height, matrix) or (tensor, width, height, matrix, "rgba") or ((y, u, v) plane tensors, width, height,
        matrix, "yuv422p10" | "yuv422p8" | "yuv420p" | "nv12"[, own Loader matrix tensor]) - matrix: nine host floats
        (transform_matrix) or None for 1:1; format v210 unless "rgba" (f32 RGBA image), a planar pack format, or "rgba8" / "bgra8"
        (tensor of packed 8-bit pixels)."""
        arr, keep = self._chan_layers(layers)
        if out_fmt != "v210":  # dst: the planes of the packed frame (ph_chan_compose); wr_cm None for rgba8 / bgra8
            planes = (C.c_void_p * 3)(*([_ptr(p).value for p in dst] + [None] * (3 - len(dst))))
            keep.append(planes)
            args = (self.h, queue, len(layers), arr, FORMATS[out_fmt], planes, out_w, out_h, interlace, _ptr(rd_cm), _ptr(rd_lut), _ptr(rd_gm),
                    None if wr_cm is None else _ptr(wr_cm), _ptr(wr_lut))
            fn = lib().ph_chan_compose
        else:
            args = (self.h, queue, len(layers), arr, _ptr(dst), out_w, out_h, interlace, _ptr(rd_cm), _ptr(rd_lut), _ptr(rd_gm), _ptr(wr_cm), _ptr(wr_lut))
            fn = lib().ph_chan_compose_v210
        if prepare_only:  # a caller that replays the same job (a bench loop) skips the marshalling: job() launches it
            h = self.h

            def job(_keep=(keep, layers, dst)):
                check(fn(*args), h)
            return job
        check(fn(*args), self.h)

    def chan_compose_batch(self, jobs, out_w, out_h, rd_cm, rd_lut, rd_gm, wr_cm, wr_lut, queue=QUEUE_PROCESS, prepare_only=False):
        """Several channels' frames in one launch (ph_chan_compose_batch).  jobs: list of (layers, dst, interlace) - layers as
        chan_compose_v210 takes them, dst the job's v210 frame."""
        arr = (PhChanJob * len(jobs))()
        keep = []
        for j, (layers, dst, interlace) in enumerate(jobs):
            la, k = self._chan_layers(layers)
            keep.append((la, k, layers, dst))
            arr[j].n, arr[j].layers, arr[j].out, arr[j].interlace = len(layers), la, _ptr(dst).value, interlace
        args = (self.h, queue, len(jobs), arr, out_w, out_h, _ptr(rd_cm), _ptr(rd_lut), _ptr(rd_gm), _ptr(wr_cm), _ptr(wr_lut))
        fn, h = lib().ph_chan_compose_batch, self.h
        if prepare_only:

            def job(_keep=(keep, arr)):
                check(fn(*args), h)
            return job
        check(fn(*args), h)

    def fused_v210_combine(self, layers, dst, width, height, rd_cm, rd_lut, rd_gm, wr_cm, wr_lut,
                           queue=QUEUE_PROCESS, prepare_only=False):
        arr = (C.c_void_p * len(layers))(*[_ptr(l).value for l in layers])
        args = (self.h, queue, len(layers), arr, _ptr(dst), width, height, _ptr(rd_cm), _ptr(rd_lut), _ptr(rd_gm), _ptr(wr_cm), _ptr(wr_lut))
        if prepare_only:  # a caller that replays the same job (a bench loop over a ring of frame sets): job() is the C call alone
            fn, h = lib().ph_fused_v210_combine, self.h

            def job(_keep=(arr, layers, dst)):
                check(fn(*args), h)
            return job
        check(lib().ph_fused_v210_combine(*args), self.h)

    def fused_v210_combine_batch(self, jobs, dsts, width, height, rd_cm, rd_lut, rd_gm, wr_cm, wr_lut, queue=QUEUE_PROCESS):
        """jobs: list of per-frame layer lists (equal length); dsts: one output per job"""
        n = len(jobs[0])
        flat = [_ptr(l).value for job in jobs for l in job]
        arr = (C.c_void_p * len(flat))(*flat)
        outs = (C.c_void_p * len(dsts))(*[_ptr(d).value for d in dsts])
        check(lib().ph_fused_v210_combine_batch(self.h, queue, len(jobs), n, arr, outs, width, height, _ptr(rd_cm),
                                                _ptr(rd_lut), _ptr(rd_gm), _ptr(wr_cm), _ptr(wr_lut)), self.h)

    # ---- nodencl-shaped surface: buffers, programs, queues -----------------------------------------
    def create_buffer(self, nbytes, access="readwrite", svm="coarse", dims=None, owner=""):
        """`clContext.createBuffer(bytes, access, svmType, imageDims?, owner?)`"""
        return Buffer(self, nbytes, access, svm, dims, owner)

    def create_program(self, source, name, global_work_items, work_items_per_group=0):
        """`clContext.createProgram(kernelSrc, {name, globalWorkItems, workItemsPerGroup})`"""
        return Program(self, source, name, global_work_items, work_items_per_group)

    @staticmethod
    def _args(params):
        n = len(params)
        arr = (PhArg * n)()
        keep = []
        for i, (k, v) in enumerate(params.items()):
            kb = k.encode()
            keep.append(kb)
            arr[i].name = kb
            if isinstance(v, Buffer):
                arr[i].kind, arr[i].v.buf = ARG_BUF, v.h
            elif isinstance(v, bool):
                arr[i].kind, arr[i].v.u32 = ARG_U32, int(v)
            elif isinstance(v, int):
                if v < 0:
                    arr[i].kind, arr[i].v.i32 = ARG_I32, v
                else:
                    arr[i].kind, arr[i].v.u32 = ARG_U32, v
            elif isinstance(v, float):
                arr[i].kind, arr[i].v.f32 = ARG_F32, v
            else:
                raise TypeError("kernel parameter %r: unsupported value %r" % (k, type(v)))
        return arr, keep

    def run_programs(self, jobs, queue=QUEUE_PROCESS):
        """Several jobs in one call (ph_run_programs): jobs = [(program, params), ...] as run_program takes them."""
        marshalled = [self._args(params) for _, params in jobs]
        progs = (C.c_void_p * len(jobs))(*[p.h for p, _ in jobs])
        args = (C.POINTER(PhArg) * len(jobs))(*[C.cast(a, C.POINTER(PhArg)) for a, _ in marshalled])
        counts = (C.c_int * len(jobs))(*[len(params) for _, params in jobs])
        check(lib().ph_run_programs(self.h, len(jobs), progs, args, counts, queue), self.h)

    def run_program(self, program, params, queue=QUEUE_PROCESS, check_only=False):
        """`clContext.runProgram(program, params, queue)`: params maps kernel argument NAMES to a
        Buffer, an int or a float (floats must be Python floats).  Returns the RunTimings in us.
        check_only: examine the job as a launch would and enqueue nothing (ph_check_program)."""
        arr, keep = self._args(params)
        n = len(params)
        if check_only:  # ph_check_program: the checks of a launch, nothing enqueued
            check(lib().ph_check_program(self.h, program.h, arr, n, queue), self.h)
            return None
        t = RunTimings()
        check(lib().ph_run_program(self.h, program.h, arr, n, queue, C.byref(t)), self.h)
        return {"dataToKernel": t.data_to_kernel, "kernelExec": t.kernel_exec, "totalTime": t.total_time}

    def record(self, fn, queue=QUEUE_PROCESS):
        """Record the launches `fn()` issues on `queue` into a replayable Graph (ph_graph_*)."""
        check(lib().ph_graph_begin(self.h, queue), self.h)
        try:
            fn()
        finally:
            g = C.c_void_p()
            rc = lib().ph_graph_end(self.h, queue, C.byref(g))
        check(rc, self.h)
        return Graph(self, g)

    def queue_idle(self, queue=QUEUE_PROCESS):
        r = lib().ph_queue_query(self.h, queue)
        if r < 0:
            check(r, self.h)
        return bool(r)

    def queue_wait_queue(self, waiter, signal):
        check(lib().ph_queue_wait_queue(self.h, waiter, signal), self.h)

    def record_event(self, queue):
        return Event(self, queue)

    def host_pool_stats(self):
        """pinned host mirrors: {in_use, pooled, peak_in_use} bytes and `pins` = hipHostMalloc calls so far (ph_ctx_host_pool_stats)"""
        a, b, c, n = C.c_size_t(), C.c_size_t(), C.c_size_t(), C.c_uint64()
        check(lib().ph_ctx_host_pool_stats(self.h, C.byref(a), C.byref(b), C.byref(c), C.byref(n)), self.h)
        return {"in_use": a.value, "pooled": b.value, "peak_in_use": c.value, "pins": n.value}

    def buffer_stats(self):
        a, b, c = C.c_size_t(), C.c_size_t(), C.c_size_t()
        check(lib().ph_ctx_buffer_stats(self.h, C.byref(a), C.byref(b), C.byref(c)), self.h)
        return {"live_buffers": a.value, "live_bytes": b.value, "pooled_bytes": c.value}


_ACCESS = {"readonly": 0, "writeonly": 1, "readwrite": 2}
_SVM = {"none": 0, "coarse": 1, "fine": 2}


class Buffer:
    """A nodencl `OpenCLBuffer`: ref-counted device block from the context's pool + pinned host mirror."""

    def __init__(self, ctx, nbytes, access="readwrite", svm="coarse", dims=None, owner=""):
        h = C.c_void_p()
        w, hh = dims if dims else (0, 0)
        check(lib().ph_buf_create(ctx.h, nbytes, _ACCESS[access], _SVM[svm], w, hh, owner.encode(), C.byref(h)), ctx.h)
        self.ctx, self.h, self.nbytes, self.timestamp = ctx, h, nbytes, 0

    def add_ref(self):
        return lib().ph_buf_addref(self.h)

    def release(self):
        return lib().ph_buf_release(self.h)

    def refcount(self):
        return lib().ph_buf_refcount(self.h)

    def device_ptr(self):
        return lib().ph_buf_device_ptr(self.h)

    def host(self, dtype="uint8"):
        """numpy view of the pinned host mirror"""
        import numpy as np
        p = lib().ph_buf_host_ptr(self.h)
        if not p:
            check(-1, self.ctx.h)
        a = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(self.nbytes,))
        return a.view(dtype)

    def host_access(self, direction, queue=QUEUE_LOAD, src=None):
        """`buf.hostAccess(dir, queue, src?)`"""
        d = {"readonly": HOST_READONLY, "writeonly": HOST_WRITEONLY, "none": HOST_NONE}[direction]
        if src is not None:
            import numpy as np
            a = np.ascontiguousarray(src).view(np.uint8).reshape(-1)
            check(lib().ph_buf_host_access(self.h, d, queue, a.ctypes.data, a.nbytes), self.ctx.h)
        else:
            check(lib().ph_buf_host_access(self.h, d, queue, None, 0), self.ctx.h)

    def download_async(self, queue=QUEUE_UNLOAD):
        check(lib().ph_buf_download_async(self.h, queue), self.ctx.h)


def route_unique_id():
    """128 bytes made by ONE rank; hand them to every other rank before Route(...)"""
    buf = C.create_string_buffer(128)
    check(lib().ph_route_unique_id(buf))
    return buf.raw


class Route:
    """Cross-GPU frame hand-off of the ROUTE producer (ph_route_*): RCCL point-to-point on a communication stream
    of its own, ordered against the context's queues on the device."""

    def __init__(self, ctx, unique_id, rank, world):
        h = C.c_void_p()
        check(lib().ph_route_init(ctx.h, C.c_char_p(unique_id), rank, world, C.byref(h)), ctx.h)
        self.ctx, self.h, self.rank, self.world = ctx, h, rank, world

    def group(self):
        route = self

        class _Group:
            def __enter__(self):
                check(lib().ph_route_group_begin(route.h))

            def __exit__(self, *a):
                check(lib().ph_route_group_end(route.h))
        return _Group()

    def send(self, tensor, peer, nbytes=None):
        check(lib().ph_route_send(self.h, _ptr(tensor), nbytes if nbytes is not None else tensor.numel() * tensor.element_size(), peer))

    def recv(self, tensor, peer, nbytes=None):
        check(lib().ph_route_recv(self.h, _ptr(tensor), nbytes if nbytes is not None else tensor.numel() * tensor.element_size(), peer))

    def after_queue(self, queue=QUEUE_PROCESS):
        check(lib().ph_route_after_queue(self.h, queue))

    def queue_after(self, queue=QUEUE_PROCESS):
        check(lib().ph_queue_after_route(self.h, queue))

    def wait(self):
        check(lib().ph_route_wait(self.h))

    def comm_count(self):
        """ranks RCCL itself counts in this route's communicator (ncclCommCount)"""
        n = C.c_int()
        check(lib().ph_route_comm_count(self.h, C.byref(n)))
        return n.value

    def destroy(self):
        if self.h:
            lib().ph_route_destroy(self.h)
            self.h = None


class Event:
    """A recorded point in a queue (ph_event): wait() blocks the host until it has been reached."""

    def __init__(self, ctx, queue):
        h = C.c_void_p()
        check(lib().ph_event_record(ctx.h, queue, C.byref(h)), ctx.h)
        self.ctx, self.h = ctx, h

    def wait(self):
        check(lib().ph_event_wait(self.h), self.ctx.h)

    def done(self):
        r = lib().ph_event_query(self.h)
        if r < 0:
            check(r, self.ctx.h)
        return bool(r)

    def destroy(self):
        if self.h:
            lib().ph_event_destroy(self.h)
            self.h = None


class Graph:
    """A recorded batch of launches: launch() replays it as one submission."""

    def __init__(self, ctx, h):
        self.ctx, self.h = ctx, h

    def launch(self, queue=QUEUE_PROCESS):
        check(lib().ph_graph_launch(self.h, queue), self.ctx.h)

    def destroy(self):
        if self.h:
            lib().ph_graph_destroy(self.h)
            self.h = None


class Program:
    def __init__(self, ctx, source, name, global_work_items, work_items_per_group=0):
        g = global_work_items if isinstance(global_work_items, (list, tuple)) else [global_work_items]
        arr = (C.c_uint32 * len(g))(*[int(x) for x in g])
        h = C.c_void_p()
        check(lib().ph_program_create(ctx.h, source.encode() if source is not None else None, name.encode(), arr, len(g),
                                      int(work_items_per_group), C.byref(h)), ctx.h)
        self.ctx, self.h = ctx, h

    def kernel(self):
        return lib().ph_program_kernel(self.h).decode()

    def destroy(self):
        if self.h:
            lib().ph_program_destroy(self.h)
            self.h = None
