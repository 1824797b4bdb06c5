"""The CPU restatement (oracle/) against golden vectors produced by running the REFERENCE
itself (tests/golden/gen_golden.py): bit-for-bit on every case.  CPU only."""
import hashlib
import json
import os

import numpy as np
import pytest

import cases
import frames
from oracle import orc

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")
HM = json.load(open(os.path.join(GOLD, "host_maths.json")))
KAT = json.load(open(os.path.join(GOLD, "kat.json")))
NPZ = np.load(os.path.join(GOLD, "kernels.npz"))
SPECS = ["601-625", "601_525", "709", "2020", "sRGB", "bogus"]
RANGES = {"10": (10, 64, 940, 896), "8": (8, 16, 235, 224)}


def hexes(a):
    return ["%08x" % v for v in np.ascontiguousarray(a, np.float32).view(np.uint32)]


def bits_equal(a, b):
    return np.array_equal(np.ascontiguousarray(a, np.float32).reshape(-1).view(np.uint32),
                          np.ascontiguousarray(b, np.float32).reshape(-1).view(np.uint32))


# ---- host maths (colourMaths.ts, transform.ts) ------------------------------------------------
@pytest.mark.parametrize("spec", SPECS)
def test_luts_match_reference(spec):
    g2l, l2g = orc.gamma2linear_lut(spec), orc.linear2gamma_lut(spec)
    assert hashlib.sha256(g2l.tobytes()).hexdigest() == HM["lut"][spec]["g2l_sha256"]
    assert hashlib.sha256(l2g.tobytes()).hexdigest() == HM["lut"][spec]["l2g_sha256"]
    assert hexes(g2l[::257]) == HM["lut"][spec]["g2l_every257"]
    assert hexes(l2g[::257]) == HM["lut"][spec]["l2g_every257"]


@pytest.mark.parametrize("spec", SPECS)
@pytest.mark.parametrize("rng", sorted(RANGES))
def test_ycbcr_matrices_match_reference(spec, rng):
    a = RANGES[rng]
    assert hexes(orc.ycbcr2rgb_matrix(spec, *a)) == HM["ycbcr2rgb"]["%s/%s" % (spec, rng)]
    assert hexes(orc.rgb2ycbcr_matrix(spec, *a)) == HM["rgb2ycbcr"]["%s/%s" % (spec, rng)]


@pytest.mark.parametrize("src", SPECS)
def test_gamut_matrices_match_reference(src):
    for dst in SPECS:
        assert hexes(orc.rgb2rgb_matrix(src, dst)) == HM["rgb2rgb"]["%s->%s" % (src, dst)], (src, dst)


def test_survey_bit_patterns():
    # SURVEY.md 8(a4) quotes these from the reference; keep them as literal known answers
    assert " ".join(hexes(orc.ycbcr2rgb_matrix("709"))) == (
        "3a95a025 00000000 3ae65eea bf7912ee 3a95a025 b95b3912 ba08f5b5 3e9a5bf0 3a95a025 3b07b951 00000000 bf911353")
    assert " ".join(hexes(orc.rgb2ycbcr_matrix("2020"))) == (
        "4366200d 44147b64 424fc986 427fffff c2fa3792 c3a1721c 43e00000 44000000 43e00000 c3cdfbe7 c21020c7 44000000")
    assert " ".join(hexes(orc.rgb2rgb_matrix("709", "2020"))) == (
        "3f209d89 3ea897c9 3d31690c 3d8d82e3 3f6b66ff 3c3a28f3 3c864755 3db44051 3f6545bb")


@pytest.mark.parametrize("i", range(len(HM["transform"])))
def test_transform_matrix_matches_reference(i):
    t = HM["transform"][i]
    p = t["params"]
    m = orc.transform_matrix(t["width"], t["height"], p.get("flipH", False), p.get("flipV", False),
                             p.get("anchorX", 0.0), p.get("anchorY", 0.0), p.get("scaleX", 1.0), p.get("scaleY", 1.0),
                             p.get("offsetX", 0.0), p.get("offsetY", 0.0), p.get("rotate", 0.0))
    assert hexes(m) == t["matrix"]


@pytest.mark.parametrize("dims", sorted(HM["ramp"]))
def test_ramp_matches_reference_fillbuf(dims):
    w, h = (int(v) for v in dims.split("x"))
    assert hashlib.sha256(orc.v210_fill_ramp(w, h).tobytes()).hexdigest() == HM["ramp"][dims]
    if w % 6 == 0:
        assert np.array_equal(frames.v210_ramp(w, h), orc.v210_fill_ramp(w, h))


# ---- kernels ------------------------------------------------------------------------------------
def run_oracle(c, inp):
    op = c["op"]
    if op == "v210_read":
        return orc.v210_read(inp["words"], c["w"], c["h"], orc.ycbcr2rgb_matrix(c["spec"]),
                             orc.gamma2linear_lut(c["spec"]), orc.rgb2rgb_matrix(c["spec"], c["out_spec"]))
    if op == "v210_write":
        return orc.v210_write(inp["rgba"], c["w"], c["h"], c["interlace"], orc.rgb2ycbcr_matrix(c["spec"]),
                              orc.linear2gamma_lut(c["spec"]), out=inp["dst"].copy())
    if op == "pack_read":
        rng = orc.FORMAT_RANGE[c["fmt"]]
        cm = None if rng is None else orc.ycbcr2rgb_matrix(c["spec"], *rng)
        return orc.pack_read(c["fmt"], inp["planes"], c["w"], c["h"], cm, orc.gamma2linear_lut(c["spec"]),
                             orc.rgb2rgb_matrix(c["spec"], c["out_spec"]))
    if op == "pack_write":
        rng = orc.FORMAT_RANGE[c["fmt"]]
        cm = None if rng is None else orc.rgb2ycbcr_matrix(c["spec"], *rng)
        return np.concatenate(orc.pack_write(c["fmt"], inp["rgba"], c["w"], c["h"], c["interlace"], cm,
                                             orc.linear2gamma_lut(c["spec"]), planes=inp["dst"]))
    if op == "yadif":
        return orc.yadif(inp["prev"], inp["cur"], inp["next"], c["parity"], c["tff"], c["skip"])
    if op == "transform":
        t = HM["transform"][c["tp"]]
        p = t["params"]
        m = orc.transform_matrix(c["mw"], c["mh"], p.get("flipH", False), p.get("flipV", False),
                                 p.get("anchorX", 0.0), p.get("anchorY", 0.0), p.get("scaleX", 1.0),
                                 p.get("scaleY", 1.0), p.get("offsetX", 0.0), p.get("offsetY", 0.0),
                                 p.get("rotate", 0.0))
        return orc.transform(inp["img"], m, c["ow"], c["oh"])
    if op == "resize":
        return orc.resize(inp["img"], c["scale"], c["ox"], c["oy"], c["fh"], c["fv"], c["ow"], c["oh"])
    if op == "combine":
        return orc.combine(inp["layers"])
    if op == "dissolve":
        return orc.transition_dissolve(inp["in0"], inp["in1"], c["mix"])
    if op == "mixer":
        return orc.mixer(inp["in0"], inp["in1"], c["mix"])
    if op == "wipe":
        return orc.wipe(inp["in0"], inp["in1"], c["wipe"])
    if op == "twipe":
        return orc.transition_wipe(inp["in0"], inp["in1"], inp["mask"])
    raise KeyError(op)


@pytest.mark.parametrize("name", [c["name"] for c in cases.CASES])
def test_kernel_matches_reference(name):
    c = cases.BY_NAME[name]
    got = run_oracle(c, cases.inputs(c))
    want = NPZ[name]
    if got.dtype in (np.uint32, np.uint8):
        assert np.array_equal(got.reshape(-1), want.reshape(-1))
    else:
        bad = np.flatnonzero(got.reshape(-1).view(np.uint32) != want.reshape(-1).view(np.uint32))
        assert bad.size == 0, "%d of %d floats differ, first at %d: %r vs %r" % (
            bad.size, want.size, bad[0], got.reshape(-1)[bad[0]], want.reshape(-1)[bad[0]])


def test_write_leaves_other_field_untouched():
    for name in ("write_rand_96x6_709_top", "write_rand_96x6_2020_bottom"):
        c = cases.BY_NAME[name]
        out = NPZ[name].reshape(c["h"], -1)
        keep = 1 if c["interlace"] == 1 else 0
        assert (out[keep::2] == cases.POISON).all() and not (out[1 - keep::2] == cases.POISON).any()


def test_known_answer_1080p_ramp_roundtrip():
    """The reference's implied v210 KAT: ramp -> read(709->709) -> write(709) == input."""
    w, h = 1920, 1080
    ramp = frames.v210_ramp(w, h)
    assert hashlib.sha256(ramp.tobytes()).hexdigest() == KAT["ramp_1080p_sha256"]
    rgba = orc.v210_read(ramp, w, h, orc.ycbcr2rgb_matrix("709"), orc.gamma2linear_lut("709"),
                         orc.rgb2rgb_matrix("709", "709"))
    assert hashlib.sha256(rgba.tobytes()).hexdigest() == KAT["ramp_1080p_read709_rgba_sha256"]
    back = orc.v210_write(rgba, w, h, 0, orc.rgb2ycbcr_matrix("709"), orc.linear2gamma_lut("709"))
    assert np.array_equal(back, ramp)


FMT_KATS = [("yuv422p10", 1920, 1080, "709"), ("yuv420p", 1920, 1080, "709"), ("nv12", 1920, 1080, "709"),
            ("yuv422p8", 718, 480, "709"), ("rgba8", 1920, 1080, "sRGB"), ("bgra8", 1920, 1080, "sRGB")]


@pytest.mark.parametrize("fmt,w,h,spec", FMT_KATS)
def test_reference_roundtrip_scripts(fmt, w, h, spec):
    """src/process/test/{nv12,yuv420p,yuv422p8,yuv422p10}Test.ts (and the same for the RGBA formats):
    test pattern -> ToRGBA -> FromRGBA; the reference kernels reproduce the input byte for byte."""
    planes = frames.pack_ramp(fmt, w, h)
    assert hashlib.sha256(np.concatenate(planes).tobytes()).hexdigest() == HM["ramp_fmt"]["%s/%dx%d" % (fmt, w, h)]
    rng = orc.FORMAT_RANGE[fmt]
    rcm = None if rng is None else orc.ycbcr2rgb_matrix(spec, *rng)
    wcm = None if rng is None else orc.rgb2ycbcr_matrix(spec, *rng)
    rgba = orc.pack_read(fmt, planes, w, h, rcm, orc.gamma2linear_lut(spec), orc.rgb2rgb_matrix(spec, spec))
    assert hashlib.sha256(rgba.tobytes()).hexdigest() == KAT["%s_%dx%d_rgba_sha256" % (fmt, w, h)]
    back = orc.pack_write(fmt, rgba, w, h, 0, wcm, orc.linear2gamma_lut(spec))
    assert hashlib.sha256(np.concatenate(back).tobytes()).hexdigest() == KAT["%s_%dx%d_back_sha256" % (fmt, w, h)]
    assert KAT["%s_%dx%d_roundtrip_identical" % (fmt, w, h)] is True
    assert all(np.array_equal(a, b) for a, b in zip(planes, back))


@pytest.mark.parametrize("key", sorted(HM["format_geometry"]))
def test_format_geometry_matches_reference(key):
    fmt, dims = key.split("/")
    w, h = (int(v) for v in dims.split("x"))
    g = HM["format_geometry"][key]
    assert orc.pack_plane_bytes(fmt, w, h) == g["numBytes"] == frames.pack_plane_bytes(fmt, w, h)
    if not g["isRGB"]:
        assert orc.FORMAT_RANGE[fmt] == (g["numBits"], g["lumaBlack"], g["lumaWhite"], g["chromaRange"])
    else:
        assert orc.FORMAT_RANGE[fmt] is None


def test_pipeline_chain_equals_separate_ops():
    w, h = 96, 8
    layers = [frames.v210_random(w, h, frames.layer_seed(0, i)) for i in range(4)]
    rd = (orc.ycbcr2rgb_matrix("709"), orc.gamma2linear_lut("709"), orc.rgb2rgb_matrix("709", "2020"))
    wr = (orc.rgb2ycbcr_matrix("2020"), orc.linear2gamma_lut("2020"))
    got = orc.pipeline_v210_combine(layers, w, h, *rd, *wr)
    rgba = [orc.v210_read(l, w, h, *rd) for l in layers]
    want = orc.v210_write(orc.combine(rgba), w, h, 0, *wr)
    assert np.array_equal(got, want)
    # alpha is 1 everywhere after a v210 read, so "over" leaves the top layer (SURVEY 8d note)
    assert np.array_equal(got, orc.v210_write(rgba[3], w, h, 0, *wr))


def test_oracle_is_thread_count_invariant():
    w, h = 96, 16
    src = frames.v210_random(w, h, 5)
    rd = (orc.ycbcr2rgb_matrix("709"), orc.gamma2linear_lut("709"), orc.rgb2rgb_matrix("709", "709"))
    orc.set_num_threads(1)
    a = orc.v210_read(src, w, h, *rd)
    orc.set_num_threads(0)
    b = orc.v210_read(src, w, h, *rd)
    assert bits_equal(a, b)


@pytest.mark.skipif(not orc.ref_available(), reason="oracle/_ref is only built where the reference checkout exists")
def test_reference_kernel_chain_equals_oracle_chain_any_build_any_thread_count():
    """bench.py's cpu_baseline "reference" leg: the reference kernels' read x4 -> combine_4 -> write
    (both builds of oracle/_ref, serial and threaded) give the oracle pipeline's words."""
    w, h = 1920, 6
    layers = [frames.v210_random(w, h, frames.layer_seed(2, i)) for i in range(4)]
    rd = (orc.ycbcr2rgb_matrix("709"), orc.gamma2linear_lut("709"), orc.rgb2rgb_matrix("709", "2020"))
    wr = (orc.rgb2ycbcr_matrix("2020"), orc.linear2gamma_lut("2020"))
    want = orc.pipeline_v210_combine(layers, w, h, *rd, *wr)
    libs = [orc.ref()] + ([orc.ref_fast()] if orc.have_ref_fast() else [])
    for r in libs:
        for threads in (1, 3):
            r.ref_set_num_threads(threads)
            assert np.array_equal(orc.ref_pipeline_v210_combine(r, layers, w, h, *rd, *wr), want)
        r.ref_set_num_threads(1)


# ---- the OpenCL built-ins: oracle restatement vs AMD's own device-library bodies ------------------------
# oracle/_ref links dot / fma / convert_*_sat* taken from /opt/rocm/amdgcn/bitcode/{opencl,ocml}.bc
# (oracle/refbuild/devlib_builtins.py: function bodies unchanged, retargeted to x86-64).  These tests pin
# the oracle's own spelling of those built-ins (phaneron_oracle.c "OpenCL built-in semantics") to them;
# tests/test_builtins_gpu.py runs the same bitcode natively on the MI355X against the product's primitives.
def _wild_floats(rng, shape):
    """float32 values covering every exponent, both signs, denormals, zeros, infinities and NaNs."""
    bits = rng.integers(0, 1 << 32, size=shape, dtype=np.uint64).astype(np.uint32)
    tame = (rng.standard_normal(shape) * 10.0 ** rng.integers(-6, 6, size=shape)).astype(np.float32)
    pick = rng.random(shape) < 0.5
    return np.where(pick, bits.view(np.float32), tame)


@pytest.mark.skipif(not orc.ref_available(), reason="oracle/_ref is only built where the reference checkout exists")
@pytest.mark.parametrize("k", [3, 4])
def test_oracle_dot_equals_device_library_dot(k):
    rng = np.random.default_rng(1234 + k)
    n = 4_000_000
    a, b = _wild_floats(rng, (n, k)), _wild_floats(rng, (n, k))
    # colour-matrix-like operands too: code values against small coefficients
    a[: n // 4] = rng.integers(0, 1024, size=(n // 4, k)).astype(np.float32)
    b[: n // 4] = (rng.standard_normal((n // 4, k)) * 0.01).astype(np.float32)
    want = orc.ref_builtin_dot(orc.ref(), a, b)
    got = orc.prim_dot(a, b)
    nan = np.isnan(want)
    assert np.array_equal(np.isnan(got), nan)
    assert np.array_equal(got.view(np.uint32)[~nan], want.view(np.uint32)[~nan])
    if orc.have_ref_fast():  # the -O3 -mavx2 build links the same bodies
        assert np.array_equal(orc.ref_builtin_dot(orc.ref_fast(), a, b).view(np.uint32)[~nan], want.view(np.uint32)[~nan])


@pytest.mark.skipif(not orc.ref_available(), reason="oracle/_ref is only built where the reference checkout exists")
@pytest.mark.parametrize("name", sorted(orc.CONVERTS))
def test_oracle_converts_equal_device_library_over_all_floats(name):
    """All 2^32 float bit patterns (NaNs, infinities, denormals, ties included), in chunks of 2^26."""
    which = orc.CONVERTS[name]
    r = orc.ref()
    r.ref_set_num_threads(orc.effective_cpus())
    try:
        step = 1 << 26
        for first in range(0, 1 << 32, step):
            want = orc.ref_builtin_convert_range(r, which, first, step)
            got = orc.prim_convert_range(which, first, step)
            if not np.array_equal(got, want):
                i = int(np.flatnonzero(got != want)[0])
                raise AssertionError("%s(bits 0x%08x): oracle %d, device library %d" % (name, first + i, got[i], want[i]))
    finally:
        r.ref_set_num_threads(1)
