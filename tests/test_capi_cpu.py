"""CPU-side checks of the product library: it loads, exports every symbol include/*.h declares,
its host colour maths equals the reference goldens, and it refuses to run without a GPU."""
import ctypes
import glob
import json
import os
import re

import numpy as np
import pytest

from phaneron_amd import build, capi

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
HM = json.load(open(os.path.join(ROOT, "tests", "golden", "host_maths.json")))


def hexes(a):
    return ["%08x" % v for v in np.ascontiguousarray(a, np.float32).view(np.uint32)]


def declared_symbols():
    names = set()
    for h in glob.glob(os.path.join(ROOT, "include", "*.h")):
        txt = re.sub(r"/\*.*?\*/", "", open(h).read(), flags=re.S)
        names.update(re.findall(r"\b(ph_[a-z0-9_]+)\s*\(", txt))
    return sorted(names)


def test_library_builds_and_exports_every_declared_symbol():
    path = build.build()
    lib = ctypes.CDLL(path)
    syms = declared_symbols()
    assert len(syms) >= 40
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, missing
    assert sorted(capi.EXPORTS) == syms  # the ctypes binding covers the whole header


def test_abi_version():
    assert capi.lib().ph_abi_version() == 8


@pytest.mark.parametrize("spec", ["601-625", "601_525", "709", "2020", "sRGB", "bogus"])
def test_product_colour_maths_matches_reference(spec):
    import hashlib
    assert hashlib.sha256(capi.gamma2linear_lut(spec).tobytes()).hexdigest() == HM["lut"][spec]["g2l_sha256"]
    assert hashlib.sha256(capi.linear2gamma_lut(spec).tobytes()).hexdigest() == HM["lut"][spec]["l2g_sha256"]
    for rng, a in (("10", (10, 64, 940, 896)), ("8", (8, 16, 235, 224))):
        assert hexes(capi.ycbcr2rgb_matrix(spec, *a)) == HM["ycbcr2rgb"]["%s/%s" % (spec, rng)]
        assert hexes(capi.rgb2ycbcr_matrix(spec, *a)) == HM["rgb2ycbcr"]["%s/%s" % (spec, rng)]
    for dst in ["601-625", "601_525", "709", "2020", "sRGB", "bogus"]:
        assert hexes(capi.rgb2rgb_matrix(spec, dst)) == HM["rgb2rgb"]["%s->%s" % (spec, dst)]


@pytest.mark.parametrize("i", range(len(HM["transform"])))
def test_product_transform_matrix_matches_reference(i):
    t = HM["transform"][i]
    p = t["params"]
    m = capi.transform_matrix(t["width"], t["height"], p.get("flipH", False), p.get("flipV", False),
                              p.get("anchorX", 0.0), p.get("anchorY", 0.0), p.get("scaleX", 1.0), p.get("scaleY", 1.0),
                              p.get("offsetX", 0.0), p.get("offsetY", 0.0), p.get("rotate", 0.0))
    assert hexes(m) == t["matrix"]


def test_pitch():
    for w, want in ((1920, 5120), (3840, 10240), (1280, 3456), (720, 1920), (100, 384)):
        assert capi.v210_pitch_bytes(w) == want


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(capi.PhaneronError, match="no HIP device"):
        capi.Context(0)


def test_product_never_imports_oracle():
    for path in glob.glob(os.path.join(ROOT, "phaneron_amd", "**", "*"), recursive=True) + \
            glob.glob(os.path.join(ROOT, "node", "**", "*"), recursive=True):
        if os.sep + "test" + os.sep in path:
            continue  # node/test/ is test infrastructure (it drives the checker on purpose)
        if os.path.isfile(path) and path.endswith((".py", ".cpp", ".hip", ".h", ".js", ".c")):
            assert "oracle" not in open(path).read().replace("no oracle", ""), path


def test_argument_validation_needs_no_device():
    """Every typed entry point rejects NULL / zero / out-of-range arguments before it touches the device,
    with a message that names the problem (no compute call is made here: there is no GPU)."""
    import ctypes as C
    l = capi.lib()
    one = C.c_void_p(16)  # a non-NULL pointer that is never dereferenced on these paths
    arr3 = (C.c_void_p * 3)(16, 16, 16)
    cases = [
        (lambda: l.ph_v210_read(None, 1, None, one, 1920, 1080, one, one, one), "ph_v210_read"),
        (lambda: l.ph_v210_write(None, 1, one, one, 1920, 1080, 2, one, one), "interlace must be 0, 1 or 3"),
        (lambda: l.ph_fused_v210_combine(None, 1, 9, arr3, one, 1920, 1080, one, one, one, one, one), "1..8 layers"),
        (lambda: l.ph_fused_v210_combine(None, 1, 3, arr3, one, 1281, 720, one, one, one, one, one), "is odd"),
        (lambda: l.ph_fused_v210_combine(None, 1, 3, arr3, one, 1280, 720, one, one, one, one, one), "ctx is NULL"),  # (1280 is a served width)
        (lambda: l.ph_combine(None, 1, 1, arr3, 64, 64, one), "between 2 and 8"),
        (lambda: l.ph_yadif(None, 1, None, one, one, 64, 64, 0, 1, 0, one), "ph_yadif"),
        (lambda: l.ph_transform(None, 1, one, 0, 64, one, one, 64, 64), "ph_transform"),
        (lambda: l.ph_pack_read(None, 1, 99, arr3, one, 64, 64, one, one, one), "ph_pack_read"),
        (lambda: l.ph_queue_wait_queue(None, 0, 1), "ctx is NULL"),
        (lambda: l.ph_buf_host_access(None, 0, 0, None, 0), "NULL buffer"),
        (lambda: l.ph_ctx_set_option(None, b"lds_lut", 1), "NULL"),
    ]
    for call, needle in cases:
        rc = call()
        assert rc < 0, needle
        assert needle in l.ph_last_error(None).decode(), (needle, l.ph_last_error(None))
    # zero-height frames are a no-op, not an error (an empty field)
    assert l.ph_v210_read(None, 1, one, one, 1920, 0, one, one, one) == 0
    assert l.ph_fused_v210_combine(None, 1, 2, arr3, one, 1920, 0, one, one, one, one, one) == 0


def test_pack_geometry():
    want = {"yuv422p10": [1920 * 2 * 1080, 1920 * 1080, 1920 * 1080], "yuv422p8": [1920 * 1080, 960 * 1080, 960 * 1080],
            "yuv420p": [1920 * 1080, 960 * 540, 960 * 540], "nv12": [1920 * 1080, 1920 * 540], "rgba8": [1920 * 4 * 1080],
            "bgra8": [1920 * 4 * 1080], "v210": [5120 * 1080]}
    for fmt, sizes in want.items():
        assert capi.pack_plane_bytes(fmt, 1920, 1080) == sizes, fmt
    assert capi.pack_plane_bytes("yuv422p8", 718, 480) == [720 * 480, 360 * 480, 360 * 480]  # pitch rounds up to 8


# ---- createProgram: which precompiled kernel a (source text, kernel name) pair selects -------------------
PACK_FORMATS = ["v210", "yuv422p10", "yuv422p8", "yuv420p", "nv12", "rgba8", "bgra8"]
IMAGE_KERNELS = ["yadif", "transform", "resize", "transition_dissolve", "transition_wipe", "mixer", "wipe"] + \
                ["combine_%d" % n for n in range(2, 9)]
REF_CL = os.path.join(ROOT, "oracle", "_ref", "work", "cl")
needs_ref_text = pytest.mark.skipif(not os.path.isdir(REF_CL), reason="the reference's kernel text exists only in the build container")


def _want_kernel(fmt, name):
    return "%s_%s" % (fmt, name)


def test_program_resolution_by_tag_and_by_name_needs_no_device():
    for fmt in PACK_FORMATS:
        for name in ("read", "write"):
            assert capi.resolve_program("phaneron:" + fmt, name) == (_want_kernel(fmt, name), fmt, "tag")
    for k in IMAGE_KERNELS:
        assert capi.resolve_program("", k) == (k, None, "name")
    assert capi.resolve_program("phaneron:x", "fused_v210_combine_4")[0] == "fused_v210_combine_4"
    for k in ("chan_compose_v210_4", "compose_up_write_v210_2", "compose_write_v210_3", "v210_yadif_pair_4", "v210_read_batch_5"):
        assert capi.resolve_program("phaneron:fused", k) == (k, None, "tag") and capi.resolve_program("", k)[2] == "name"
    for src, name, needle in (("", "sharpen", "unknown kernel"), ("", "combine_9", "layers are built"), ("", "combine_4x", "plain layer count"),
                              ("", "chan_compose_v210_", "plain layer count"), ("", "compose_up_write_v210_9", "layers are built"),
                              ("", "combine_1", "layers are built"), ("phaneron:v211", "read", "cannot tell which pack format"),
                              ("__kernel void read(__global float* a) {}", "read", "cannot tell which pack format")):
        with pytest.raises(capi.PhaneronError, match=needle):
            capi.resolve_program(src, name)


@needs_ref_text
def test_reference_kernel_text_selects_the_right_kernel():
    """The strings the reference passes at packer.ts:98 and imageProcess.ts:69, unchanged: every pack-format
    source is recognised by its exact text, every image kernel by its name."""
    for fmt in PACK_FORMATS:
        src = open(os.path.join(REF_CL, fmt + ".cl")).read()
        for name in ("read", "write"):
            assert capi.resolve_program(src, name) == (_want_kernel(fmt, name), fmt, "text"), (fmt, name)
        # re-indented / CRLF text is the same program
        assert capi.resolve_program(src.replace("\n", "\r\n").replace("  ", "\t"), "read")[2] == "text"
    for stem, names in (("yadif", ["yadif"]), ("transform", ["transform"]), ("resize", ["resize"]), ("mix", ["mixer"]),
                        ("wipe", ["wipe"]), ("transition_dissolve", ["transition_dissolve"]),
                        ("transition_wipe", ["transition_wipe"])):
        src = open(os.path.join(REF_CL, stem + ".cl")).read()
        for name in names:
            assert capi.resolve_program(src, name) == (name, None, "name")
    for n in range(2, 9):
        src = open(os.path.join(REF_CL, "combine_%d.cl" % n)).read()
        assert capi.resolve_program(src, "combine_%d" % n) == ("combine_%d" % n, None, "name")


@needs_ref_text
def test_edited_kernel_text_is_still_told_apart_by_signature_and_body():
    """A maintainer's edited copy no longer matches a fingerprint: the argument list decides, plus one probe of
    the kernel BODY where two formats share a signature.  Comments must not fool it (an rgba8 source that
    mentions bgra, a 4:2:2 source that mentions inOffUV)."""
    decoys = "// bgra inOffUV outOffUV ushort8 inputC colMatrix uint4\n/* bgra8.s2 = x; inOffUV */\n"
    for fmt in PACK_FORMATS:
        src = decoys + open(os.path.join(REF_CL, fmt + ".cl")).read() + "\n// edited\n"
        for name in ("read", "write"):
            assert capi.resolve_program(src, name) == (_want_kernel(fmt, name), fmt, "signature"), (fmt, name)


# ---- the LDS form of a gamma table (ph_lut_layout_of: host code of the product, no device) --------------------------------
def _lds_lookup(layout, image, idx):
    """The table kernels' lookup (phaneron_amd/csrc/ph_ldslut.h) restated with numpy on the LDS image: every float operation of it
    is exact, so float64 arithmetic narrowed to float32 reproduces it."""
    magic = np.float64(12582912.0)  # 1.5 * 2^23: y = idx + magic is what the rounding add leaves
    y = idx.astype(np.float64) + magic
    a_scale = np.float64(layout["a_scale"])
    fs = (y * a_scale - (magic - layout["index_bias"]) * a_scale).astype(np.float32)
    assert np.array_equal(fs.astype(np.float64), (idx.astype(np.float64) + layout["index_bias"]) * a_scale)  # exact
    a_addr = (fs.view(np.uint32) >> np.uint32(layout["shift"] - 2)) & np.uint32(0xFFFFFFFC)
    d_addr = np.uint32(layout["delta_off"]) + 2 * idx.astype(np.uint32)
    assert a_addr.min() >= layout["hole"] and a_addr.max() + 4 <= layout["delta_off"] and d_addr.max() + 2 <= layout["lds_bytes"]
    anchors = image[: layout["delta_off"]].view(np.uint32)
    deltas = image[layout["delta_off"]:].view(np.uint16)
    return anchors[a_addr >> 2] + deltas[idx].astype(np.uint32)


@pytest.mark.parametrize("spec", ["709", "2020", "601-625", "601-525", "sRGB"])
@pytest.mark.parametrize("direction", ["gamma2linear", "linear2gamma"])
def test_lds_form_of_the_gamma_tables_is_exact(spec, direction):
    lut = capi.gamma2linear_lut(spec) if direction == "gamma2linear" else capi.linear2gamma_lut(spec)
    layout, image = capi.lut_layout(lut)
    assert layout is not None, "the %s %s table must fit the LDS" % (spec, direction)
    assert layout["lds_bytes"] == image.size <= 160 * 1024 and layout["hole"] == 4 << (23 - layout["shift"])
    assert not image[: layout["hole"]].any()
    idx = np.arange(65536, dtype=np.uint32)
    assert np.array_equal(_lds_lookup(layout, image, idx), lut.view(np.uint32))


def test_a_table_that_does_not_compress_stays_plain():
    rng = np.random.default_rng(7)
    layout, image = capi.lut_layout(rng.random(65536, dtype=np.float32))  # neighbouring entries orders of magnitude apart
    assert layout is None and image is None
    ramp = (np.arange(65536, dtype=np.float32) / np.float32(65535.0)).astype(np.float32)  # a linear ramp does compress
    layout, image = capi.lut_layout(ramp)
    assert layout is not None and np.array_equal(_lds_lookup(layout, image, np.arange(65536, dtype=np.uint32)), ramp.view(np.uint32))
