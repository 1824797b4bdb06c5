"""ctypes binding of oracle/liboracle.so (the CPU restatement) and, when present, of
oracle/_ref/libphaneron_ref.so (the reference's own kernel text compiled for x86).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg as the *checker*.  Nothing under phaneron_amd/ imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_u32p = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")
_u16p = np.ctypeslib.ndpointer(dtype=np.uint16, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")


def _cpu_has(*flags):
    try:
        with open("/proc/cpuinfo") as f:
            txt = f.read()
        line = next(l for l in txt.splitlines() if l.startswith("flags"))
        have = set(line.split(":", 1)[1].split())
        return all(fl in have for fl in flags)
    except Exception:
        return False


def build(force=False):
    """Compile liboracle.so with gcc (idempotent).  Returns the path of the library to load."""
    if _cpu_has("avx2", "fma"):
        name, arch = "liboracle.so", None
    else:  # host CPU without FMA/AVX2: same source, generic code
        name, arch = "liboracle_generic.so", ""
    path = os.path.join(_HERE, name)
    srcs = [os.path.join(_HERE, f) for f in ("phaneron_oracle.c", "phaneron_oracle_formats.c", "phaneron_oracle.h")]
    stale = (not os.path.exists(path)) or os.path.getmtime(path) < max(os.path.getmtime(f) for f in srcs)
    if force or stale:
        cmd = ["make", "-C", _HERE, "OUT=" + name]
        if arch is not None:
            cmd.append("ARCH=" + arch)
        subprocess.run(cmd, check=True, capture_output=True)
    return path


_lib = None


def lib():
    global _lib
    if _lib is None:
        l = C.CDLL(build())
        l.orc_gamma2linear_lut.argtypes = [C.c_char_p, _f32p]
        l.orc_linear2gamma_lut.argtypes = [C.c_char_p, _f32p]
        l.orc_ycbcr2rgb_matrix.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, _f32p]
        l.orc_rgb2ycbcr_matrix.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, _f32p]
        l.orc_rgb2rgb_matrix.argtypes = [C.c_char_p, C.c_char_p, _f32p]
        l.orc_transform_matrix.argtypes = [C.c_int] * 4 + [C.c_double] * 7 + [_f32p]
        l.orc_transform_matrix.restype = None
        l.orc_v210_pitch_pixels.argtypes = [C.c_uint32]
        l.orc_v210_pitch_pixels.restype = C.c_uint32
        l.orc_v210_pitch_bytes.argtypes = [C.c_uint32]
        l.orc_v210_pitch_bytes.restype = C.c_uint32
        l.orc_v210_fill_ramp.argtypes = [_u8p, C.c_uint32, C.c_uint32]
        l.orc_v210_fill_ramp.restype = None
        l.orc_v210_read.argtypes = [_u32p, _f32p, C.c_uint32, C.c_uint32, _f32p, _f32p, _f32p]
        l.orc_v210_read.restype = None
        l.orc_v210_write.argtypes = [_f32p, _u32p, C.c_uint32, C.c_uint32, C.c_uint32, _f32p, _f32p]
        l.orc_v210_write.restype = None
        l.orc_yadif.argtypes = [_f32p, _f32p, _f32p] + [C.c_int] * 5 + [_f32p]
        l.orc_yadif.restype = None
        l.orc_transform.argtypes = [_f32p, C.c_int, C.c_int, _f32p, _f32p, C.c_int, C.c_int]
        l.orc_transform.restype = None
        l.orc_resize.argtypes = [_f32p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, _f32p, _f32p, C.c_int, C.c_int]
        l.orc_resize.restype = None
        l.orc_combine.argtypes = [C.c_int, C.POINTER(C.c_void_p), C.c_int, C.c_int, _f32p]
        l.orc_transition_dissolve.argtypes = [_f32p, _f32p, C.c_float, C.c_int, C.c_int, _f32p]
        l.orc_transition_dissolve.restype = None
        l.orc_mixer.argtypes = [_f32p, _f32p, C.c_float, C.c_int, C.c_int, _f32p]
        l.orc_mixer.restype = None
        l.orc_transition_wipe.argtypes = [_f32p, _f32p, _f32p, C.c_int, C.c_int, _f32p]
        l.orc_transition_wipe.restype = None
        l.orc_wipe.argtypes = [_f32p, _f32p, C.c_float, C.c_int, C.c_int, _f32p]
        l.orc_wipe.restype = None
        l.orc_pipeline_v210_combine.argtypes = [C.c_int, C.POINTER(C.c_void_p), _u32p, C.c_uint32, C.c_uint32,
                                                _f32p, _f32p, _f32p, _f32p, _f32p, _f32p]
        l.orc_pack_pitch.argtypes = [C.c_int, C.c_uint32]
        l.orc_pack_pitch.restype = C.c_uint32
        l.orc_pack_plane_bytes.argtypes = [C.c_int, C.c_uint32, C.c_uint32, C.POINTER(C.c_size_t)]
        l.orc_pack_read.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, _f32p, C.c_uint32, C.c_uint32,
                                    C.c_void_p, _f32p, _f32p]
        l.orc_pack_write.argtypes = [C.c_int, _f32p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32,
                                     C.c_uint32, C.c_void_p, _f32p]
        l.orc_num_threads.restype = C.c_int
        l.orc_set_num_threads.argtypes = [C.c_int]
        l.orc_prim_dot4.argtypes = [_f32p, _f32p, _f32p, C.c_size_t]
        l.orc_prim_dot3.argtypes = [_f32p, _f32p, _f32p, C.c_size_t]
        l.orc_prim_convert_range.argtypes = [C.c_int, C.c_uint32, C.c_uint32, _u16p]
        for n in ("orc_prim_dot4", "orc_prim_dot3", "orc_prim_convert_range"):
            getattr(l, n).restype = None
        l.orc_set_num_threads.restype = None
        _lib = l
    return _lib


def _ptr_array(arrs):
    return (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])


# ---- colour maths --------------------------------------------------------------------------
def gamma2linear_lut(colspec):
    out = np.empty(65536, np.float32)
    lib().orc_gamma2linear_lut(colspec.encode(), out)
    return out


def linear2gamma_lut(colspec):
    out = np.empty(65536, np.float32)
    lib().orc_linear2gamma_lut(colspec.encode(), out)
    return out


def ycbcr2rgb_matrix(colspec, num_bits=10, luma_black=64, luma_white=940, chr_range=896):
    out = np.empty(12, np.float32)
    lib().orc_ycbcr2rgb_matrix(colspec.encode(), num_bits, luma_black, luma_white, chr_range, out)
    return out


def rgb2ycbcr_matrix(colspec, num_bits=10, luma_black=64, luma_white=940, chr_range=896):
    out = np.empty(12, np.float32)
    lib().orc_rgb2ycbcr_matrix(colspec.encode(), num_bits, luma_black, luma_white, chr_range, out)
    return out


def rgb2rgb_matrix(src, dst):
    out = np.empty(9, np.float32)
    lib().orc_rgb2rgb_matrix(src.encode(), dst.encode(), out)
    return out


def transform_matrix(width, height, flip_h=False, flip_v=False, anchor_x=0.0, anchor_y=0.0, scale_x=1.0,
                     scale_y=1.0, offset_x=0.0, offset_y=0.0, rotate=0.0):
    out = np.empty(9, np.float32)
    lib().orc_transform_matrix(width, height, int(flip_h), int(flip_v), anchor_x, anchor_y, scale_x, scale_y,
                               offset_x, offset_y, rotate, out)
    return out


# ---- v210 -----------------------------------------------------------------------------------
def v210_pitch_bytes(width):
    return int(lib().orc_v210_pitch_bytes(width))


def v210_fill_ramp(width, height):
    buf = np.zeros(v210_pitch_bytes(width) * height, np.uint8)
    lib().orc_v210_fill_ramp(buf, width, height)
    return buf.view(np.uint32)


def v210_read(words, width, height, col_matrix, lut, gamut):
    out = np.zeros(width * height * 4, np.float32)
    lib().orc_v210_read(np.ascontiguousarray(words, np.uint32), out, width, height,
                        np.ascontiguousarray(col_matrix, np.float32), lut, np.ascontiguousarray(gamut, np.float32))
    return out.reshape(height, width, 4)


def v210_write(rgba, width, height, interlace, col_matrix, lut, out=None):
    if out is None:
        out = np.zeros(v210_pitch_bytes(width) * height // 4, np.uint32)
    lib().orc_v210_write(np.ascontiguousarray(rgba, np.float32).reshape(-1), out, width, height, interlace,
                         np.ascontiguousarray(col_matrix, np.float32), lut)
    return out


# ---- other pack formats ------------------------------------------------------------------------
FORMATS = {"v210": 0, "yuv422p10": 1, "yuv422p8": 2, "yuv420p": 3, "nv12": 4, "rgba8": 5, "bgra8": 6}
# Reader/Writer constants of each format: (numBits, lumaBlack, lumaWhite, chromaRange), None = RGB
FORMAT_RANGE = {"v210": (10, 64, 940, 896), "yuv422p10": (10, 64, 940, 896), "yuv422p8": (8, 16, 235, 224),
                "yuv420p": (8, 16, 235, 224), "nv12": (8, 16, 235, 224), "rgba8": None, "bgra8": None}


def pack_plane_bytes(fmt, width, height):
    b = (C.c_size_t * 3)()
    n = lib().orc_pack_plane_bytes(FORMATS[fmt], width, height, b)
    return [int(b[i]) for i in range(n)]


def _planes3(planes):
    ps = [np.ascontiguousarray(p) for p in planes]
    ptrs = [p.ctypes.data for p in ps] + [None] * (3 - len(ps))
    return ps, ptrs


def pack_read(fmt, planes, width, height, col_matrix, lut, gamut):
    """planes: list of uint8 arrays (raw plane bytes).  col_matrix None for the RGB formats."""
    ps, ptrs = _planes3(planes)
    out = np.zeros(width * height * 4, np.float32)
    cm = None if col_matrix is None else np.ascontiguousarray(col_matrix, np.float32)
    rc = lib().orc_pack_read(FORMATS[fmt], ptrs[0], ptrs[1], ptrs[2], out, width, height,
                             None if cm is None else cm.ctypes.data, lut, np.ascontiguousarray(gamut, np.float32))
    assert rc == 0
    return out.reshape(height, width, 4)


def pack_write(fmt, rgba, width, height, interlace, col_matrix, lut, planes=None):
    """Returns the list of plane byte arrays (uint8).  planes: pre-filled destinations (copied)."""
    sizes = pack_plane_bytes(fmt, width, height)
    ps = [np.zeros(s, np.uint8) for s in sizes] if planes is None else [np.array(p, np.uint8, copy=True) for p in planes]
    ptrs = [p.ctypes.data for p in ps] + [None] * (3 - len(ps))
    cm = None if col_matrix is None else np.ascontiguousarray(col_matrix, np.float32)
    rc = lib().orc_pack_write(FORMATS[fmt], np.ascontiguousarray(rgba, np.float32).reshape(-1), ptrs[0], ptrs[1], ptrs[2],
                              width, height, interlace, None if cm is None else cm.ctypes.data, lut)
    assert rc == 0
    return ps


# ---- image ops --------------------------------------------------------------------------------
def _img(a):
    return np.ascontiguousarray(a, np.float32).reshape(-1)


def yadif(prev, cur, nxt, parity, tff, skip_spatial=False):
    h, w, _ = cur.shape
    out = np.zeros(w * h * 4, np.float32)
    lib().orc_yadif(_img(prev), _img(cur), _img(nxt), w, h, int(parity), int(tff), int(skip_spatial), out)
    return out.reshape(h, w, 4)


def transform(img, m9, out_w, out_h):
    ih, iw, _ = img.shape
    out = np.zeros(out_w * out_h * 4, np.float32)
    lib().orc_transform(_img(img), iw, ih, np.ascontiguousarray(m9, np.float32), out, out_w, out_h)
    return out.reshape(out_h, out_w, 4)


def resize(img, scale, offset_x, offset_y, flip_h, flip_v, out_w, out_h):
    ih, iw, _ = img.shape
    flip = np.array([1.0 if flip_h else 0.0, -1.0 if flip_h else 1.0, 1.0 if flip_v else 0.0,
                     -1.0 if flip_v else 1.0], np.float32)  # resize.ts:85-90
    out = np.zeros(out_w * out_h * 4, np.float32)
    lib().orc_resize(_img(img), iw, ih, scale, offset_x, offset_y, flip, out, out_w, out_h)
    return out.reshape(out_h, out_w, 4)


def combine(layers):
    h, w, _ = layers[0].shape
    ls = [_img(l) for l in layers]
    out = np.zeros(w * h * 4, np.float32)
    rc = lib().orc_combine(len(ls), _ptr_array(ls), w, h, out)
    if rc != 0:
        raise ValueError("Combine requires at least 2 layers")
    return out.reshape(h, w, 4)


def _two(fn, a, b, s):
    h, w, _ = a.shape
    out = np.zeros(w * h * 4, np.float32)
    fn(_img(a), _img(b), s, w, h, out)
    return out.reshape(h, w, 4)


def transition_dissolve(in0, in1, mix):
    return _two(lib().orc_transition_dissolve, in0, in1, mix)


def mixer(in0, in1, mix):
    return _two(lib().orc_mixer, in0, in1, mix)


def wipe(in0, in1, w_):
    return _two(lib().orc_wipe, in0, in1, w_)


def transition_wipe(in0, in1, mask):
    h, w, _ = in0.shape
    out = np.zeros(w * h * 4, np.float32)
    lib().orc_transition_wipe(_img(in0), _img(in1), _img(mask), w, h, out)
    return out.reshape(h, w, 4)


def pipeline_v210_combine(layers, width, height, rd_cm, rd_lut, rd_gm, wr_cm, wr_lut, scratch=None):
    n = len(layers)
    ls = [np.ascontiguousarray(l, np.uint32) for l in layers]
    out = np.zeros(v210_pitch_bytes(width) * height // 4, np.uint32)
    if scratch is None:
        scratch = np.empty((n + 1) * width * height * 4, np.float32)
    rc = lib().orc_pipeline_v210_combine(n, _ptr_array(ls), out, width, height,
                                         np.ascontiguousarray(rd_cm, np.float32), rd_lut,
                                         np.ascontiguousarray(rd_gm, np.float32),
                                         np.ascontiguousarray(wr_cm, np.float32), wr_lut, scratch)
    if rc != 0:
        raise ValueError("bad layer count")
    return out


def effective_cpus():
    """CPUs this process may really use: min(affinity mask, cgroup CPU quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, q // period))
        except Exception:
            pass
    return n


def num_threads():
    return int(lib().orc_num_threads())


def set_num_threads(n):
    lib().orc_set_num_threads(int(n))


# ---- the built-in semantics the oracle assumes, in bulk (pinned to AMD's device library by the tests) ----
CONVERTS = {"convert_ushort_sat_rte": 0, "convert_ushort_sat_rtz": 1, "convert_ushort_sat": 2,
            "convert_uchar_sat_rte": 3, "convert_ushort_sat_rtz(round(x))": 4}


def prim_dot(a, b):
    """a, b: (n, 3) or (n, 4) float32 -> (n,) float32 with the oracle's dot3 / dot4."""
    a, b = np.ascontiguousarray(a, np.float32), np.ascontiguousarray(b, np.float32)
    out = np.empty(a.shape[0], np.float32)
    (lib().orc_prim_dot4 if a.shape[1] == 4 else lib().orc_prim_dot3)(a, b, out, a.shape[0])
    return out


def prim_convert_range(which, first_bits, n):
    out = np.empty(n, np.uint16)
    lib().orc_prim_convert_range(which, first_bits, n, out)
    return out


def ref_builtin_dot(r, a, b):
    a, b = np.ascontiguousarray(a, np.float32), np.ascontiguousarray(b, np.float32)
    out = np.empty(a.shape[0], np.float32)
    (r.ref_builtin_dot4 if a.shape[1] == 4 else r.ref_builtin_dot3)(a, b, out, a.shape[0])
    return out


def ref_builtin_convert_range(r, which, first_bits, n):
    out = np.empty(n, np.uint16)
    r.ref_builtin_convert_range(which, first_bits, n, out)
    return out


def ref_builtin_convert_list(r, which, bits):
    bits = np.ascontiguousarray(bits, np.uint32)
    out = np.empty(bits.size, np.uint16)
    r.ref_builtin_convert_list(which, bits, bits.size, out)
    return out


# ---- oracle/_ref: the reference's own kernel text (build container only) -----------------------
_ref = None


def _cpu_has(*want):
    try:
        flags = next(l for l in open("/proc/cpuinfo") if l.startswith("flags")).split()
    except Exception:
        return False
    return all(w in flags for w in want)


def ref_available():
    # the device-library object inside it is built with -mfma (its dot() must fuse like the GPU does)
    return os.path.exists(os.path.join(_HERE, "_ref", "libphaneron_ref.so")) and _cpu_has("fma")


def _bind_ref(path):
    if True:
        r = C.CDLL(path)
        r.ref_v210_read.argtypes = [_u32p, _f32p, C.c_uint, C.c_uint, _f32p, _f32p, _f32p]
        r.ref_v210_write.argtypes = [_f32p, _u32p, C.c_uint, C.c_uint, C.c_uint, _f32p, _f32p]
        r.ref_yadif.argtypes = [_f32p, _f32p, _f32p] + [C.c_int] * 5 + [_f32p]
        r.ref_transform.argtypes = [_f32p, C.c_int, C.c_int, _f32p, _f32p, C.c_int, C.c_int]
        r.ref_resize.argtypes = [_f32p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, _f32p, _f32p, C.c_int, C.c_int]
        r.ref_mixer.argtypes = [_f32p, _f32p, C.c_float, C.c_int, C.c_int, _f32p]
        r.ref_wipe.argtypes = [_f32p, _f32p, C.c_float, C.c_int, C.c_int, _f32p]
        r.ref_transition_dissolve.argtypes = [_f32p, _f32p, C.c_float, C.c_int, C.c_int, _f32p]
        r.ref_transition_wipe.argtypes = [_f32p, _f32p, _f32p, C.c_int, C.c_int, _f32p]
        r.ref_combine.argtypes = [C.c_int, C.POINTER(C.c_void_p), C.c_int, C.c_int, _f32p]
        r.ref_pack_read.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, _f32p, C.c_uint, C.c_uint, C.c_void_p,
                                    _f32p, _f32p]
        r.ref_pack_write.argtypes = [C.c_int, _f32p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint, C.c_uint, C.c_uint,
                                     C.c_void_p, _f32p]
        for n in ("ref_v210_read", "ref_v210_write", "ref_yadif", "ref_transform", "ref_resize", "ref_mixer",
                  "ref_wipe", "ref_transition_dissolve", "ref_transition_wipe"):
            getattr(r, n).restype = None
        r.ref_pipeline_v210_combine.argtypes = [C.c_int, C.POINTER(C.c_void_p), _u32p, C.c_uint, C.c_uint, _f32p, _f32p,
                                                _f32p, _f32p, _f32p, _f32p]
        r.ref_set_num_threads.argtypes = [C.c_int]
        r.ref_builtin_dot4.argtypes = [_f32p, _f32p, _f32p, C.c_size_t]
        r.ref_builtin_dot3.argtypes = [_f32p, _f32p, _f32p, C.c_size_t]
        r.ref_builtin_fma.argtypes = [_f32p, _f32p, _f32p, _f32p, C.c_size_t]
        r.ref_builtin_convert_range.argtypes = [C.c_int, C.c_uint32, C.c_uint32, _u16p]
        r.ref_builtin_convert_list.argtypes = [C.c_int, _u32p, C.c_uint32, _u16p]
        for n in ("ref_builtin_dot4", "ref_builtin_dot3", "ref_builtin_fma", "ref_builtin_convert_range",
                  "ref_builtin_convert_list"):
            getattr(r, n).restype = None
    return r


def ref():
    """ctypes handle of the reference-kernel library (only where oracle/_ref was built)."""
    global _ref
    if _ref is None:
        _ref = _bind_ref(os.path.join(_HERE, "_ref", "libphaneron_ref.so"))
    return _ref


_ref_fast = None


def have_ref_fast():
    """The -O3 -mavx2 -mfma build of the same reference kernels, usable on this host's CPU?"""
    return os.path.exists(os.path.join(_HERE, "_ref", "libphaneron_ref_fast.so")) and _cpu_has("avx2", "fma")


def ref_fast():
    global _ref_fast
    if _ref_fast is None:
        _ref_fast = _bind_ref(os.path.join(_HERE, "_ref", "libphaneron_ref_fast.so"))
    return _ref_fast


def ref_pipeline_v210_combine(r, layers, width, height, rd_cm, rd_lut, rd_gm, wr_cm, wr_lut, scratch=None):
    """The reference kernels' own chain (read x n -> combine_n -> write) through library `r`."""
    n = len(layers)
    ls = [np.ascontiguousarray(l, np.uint32) for l in layers]
    out = np.zeros(v210_pitch_bytes(width) * height // 4, np.uint32)
    if scratch is None:
        scratch = np.empty((n + 1) * width * height * 4, np.float32)
    rc = r.ref_pipeline_v210_combine(n, _ptr_array(ls), out, width, height, np.ascontiguousarray(rd_cm, np.float32),
                                     rd_lut, np.ascontiguousarray(rd_gm, np.float32),
                                     np.ascontiguousarray(wr_cm, np.float32), wr_lut, scratch)
    if rc != 0:
        raise ValueError("bad layer count")
    return out
