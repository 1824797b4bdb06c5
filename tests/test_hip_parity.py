"""GPU parity tests (-m gpu): every HIP kernel, called through the C ABI, against
  (a) the committed golden vectors produced by running the reference (bit-exact), and
  (b) the oracle on larger seeded inputs (bit-exact), and
  (c) size-independent properties at BASELINE.json's full sizes (round trips, top-layer-wins).
north_star tolerance: bit-exact for v210 pack/unpack, <= 1 ULP f32 for colour/mix maths; every
kernel here is held to 0 ULP (explicit fma chains make that reachable)."""
import hashlib
import json
import os

import numpy as np
import pytest

import cases
import frames
from oracle import orc

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
HM = json.load(open(os.path.join(GOLD, "host_maths.json")))
KAT = json.load(open(os.path.join(GOLD, "kat.json")))
NPZ = np.load(os.path.join(GOLD, "kernels.npz"))


@pytest.fixture(params=["lds_lut", "global_lut"], autouse=True)
def lut_path(request):
    """Every test runs twice: with the gamma LUTs served from LDS (default) and with the
    global-gather kernels; both must be bit-identical to the reference."""
    import hip_harness as hh
    hh.ctx().set_option("lds_lut", request.param == "lds_lut")
    yield request.param
    hh.ctx().set_option("lds_lut", True)


def assert_bits(got, want, what=""):
    got, want = np.ascontiguousarray(got).reshape(-1), np.ascontiguousarray(want).reshape(-1)
    if got.dtype == np.uint8:
        assert np.array_equal(got, want), "%s: %d bytes differ" % (what, int((got != want).sum()))
        return
    g = got.view(np.uint32)
    w = want.view(np.uint32)
    assert g.shape == w.shape, (g.shape, w.shape)
    bad = np.flatnonzero(g != w)
    assert bad.size == 0, "%s: %d of %d words differ, first at %d: %08x vs %08x" % (
        what, bad.size, w.size, bad[0], g[bad[0]], w[bad[0]])


@pytest.mark.parametrize("name", [c["name"] for c in cases.CASES])
def test_kernel_matches_reference_golden(name):
    import hip_harness as hh
    c = cases.BY_NAME[name]
    got = hh.run_case(c, cases.inputs(c), HM)
    assert_bits(got, NPZ[name], name)


def test_native_library_is_loaded():
    import hip_harness as hh
    hh.ctx()
    with open("/proc/self/maps") as f:
        assert "libphaneron_hip.so" in f.read()
    vendor, device = hh.ctx().info()
    assert "gfx950" in device, device


def test_known_answer_1080p_ramp_roundtrip():
    """The reference's implied KAT: ramp -> read(709->709) -> write(709) is byte-identical."""
    import torch
    import hip_harness as hh
    w, h = 1920, 1080
    ramp = frames.v210_ramp(w, h)
    k = hh.ctx()
    cm, lut, gm = hh.ColourParams.reader("709", "709")
    wcm, wlut = hh.ColourParams.writer("709")
    src = hh.dev(ramp)
    rgba = torch.zeros(w * h * 4, dtype=torch.float32, device="cuda")
    back = torch.zeros_like(src)
    k.v210_read(src, rgba, w, h, cm, lut, gm)
    k.v210_write(rgba, back, w, h, 0, wcm, wlut)
    assert hashlib.sha256(hh.host(rgba).tobytes()).hexdigest() == KAT["ramp_1080p_read709_rgba_sha256"]
    assert np.array_equal(hh.host(back, np.uint32), ramp)


@pytest.mark.parametrize("w,h,spec,out_spec,legal", [(1920, 1080, "709", "2020", True), (3840, 270, "2020", "709", False),
                                                     (1280, 72, "601-625", "709", True), (100, 7, "709", "709", False)])
def test_v210_read_vs_oracle(w, h, spec, out_spec, legal):
    import torch
    import hip_harness as hh
    words = frames.v210_random(w, h, 1234 + w, legal=legal)
    cm, lut, gm = hh.ColourParams.reader(spec, out_spec)
    out = torch.zeros(w * h * 4, dtype=torch.float32, device="cuda")
    hh.ctx().v210_read(hh.dev(words), out, w, h, cm, lut, gm)
    want = orc.v210_read(words, w, h, orc.ycbcr2rgb_matrix(spec), orc.gamma2linear_lut(spec),
                         orc.rgb2rgb_matrix(spec, out_spec))
    assert_bits(hh.host(out), want, "v210_read %dx%d" % (w, h))


@pytest.mark.parametrize("w,h,spec,interlace", [(1920, 1080, "709", 0), (1920, 540, "2020", 1), (3840, 270, "2020", 3),
                                                (1280, 72, "709", 0), (100, 6, "709", 3), (98, 5, "709", 0)])
def test_v210_write_vs_oracle(w, h, spec, interlace):
    import hip_harness as hh
    rgba = frames.rgba_random(w, h, 4321 + w, -0.05, 1.05)
    dst = np.full(frames.v210_pitch_bytes(w) * h // 4, cases.POISON, np.uint32)
    wcm, wlut = hh.ColourParams.writer(spec)
    out = hh.dev(dst)
    hh.ctx().v210_write(hh.dev(rgba), out, w, h, interlace, wcm, wlut)
    want = orc.v210_write(rgba, w, h, interlace, orc.rgb2ycbcr_matrix(spec), orc.linear2gamma_lut(spec), out=dst.copy())
    assert_bits(hh.host(out, np.uint32), want, "v210_write %dx%d il=%d" % (w, h, interlace))


@pytest.mark.parametrize("w,h", [(1920, 540), (301, 33), (3, 2)])
def test_yadif_vs_oracle(w, h):
    import torch
    import hip_harness as hh
    p, c, n = (frames.rgba_random(w, h, 900 + i) for i in range(3))
    for parity in (0, 1):
        for tff in (0, 1):
            out = torch.zeros(w * h * 4, dtype=torch.float32, device="cuda")
            hh.ctx().yadif(hh.dev(p), hh.dev(c), hh.dev(n), out, w, h, parity, tff, False)
            assert_bits(hh.host(out), orc.yadif(p, c, n, parity, tff, False), "yadif %dx%d p%d t%d" % (w, h, parity, tff))


def test_image_store_policy_changes_no_bit():
    """ph_ctx_set_option("stream_images"): streamed or cached image stores, same images (read, yadif, transform, combine)"""
    import torch
    import hip_harness as hh
    from phaneron_amd import capi
    w, h = 384, 40
    k = hh.ctx()
    cm, lut, gm = hh.ColourParams.reader("709", "2020")
    src = [hh.dev(frames.v210_random(w, h, 6100 + i)) for i in range(3)]
    m = hh.dev(capi.transform_matrix(w, h, scale_x=0.75, scale_y=0.75, rotate=0.03))
    results = []
    try:
        for policy in (0, 1):
            k.set_option("stream_images", policy)
            rgba = [torch.zeros(w * h * 4, dtype=torch.float32, device="cuda") for _ in range(3)]
            for s_, r in zip(src, rgba):
                k.v210_read(s_, r, w, h, cm, lut, gm)
            de, xf, cb = (torch.zeros(w * h * 4, dtype=torch.float32, device="cuda") for _ in range(3))
            k.yadif(rgba[0], rgba[1], rgba[2], de, w, h, 1, 1, False)
            k.transform(de, w, h, m, xf, w, h)
            k.combine([rgba[0], xf], cb, w, h)
            results.append([hh.host(t).copy() for t in (rgba[0], de, xf, cb)])
    finally:
        k.set_option("stream_images", 0)
    for a, b in zip(*results):
        assert_bits(a, b, "store policy")
    want = orc.v210_read(frames.v210_random(w, h, 6100), w, h, orc.ycbcr2rgb_matrix("709"), orc.gamma2linear_lut("709"), orc.rgb2rgb_matrix("709", "2020"))
    assert_bits(results[1][0], want, "streamed v210 read")
    with pytest.raises(Exception, match="unknown option"):
        k.set_option("stream_everything", 1)


@pytest.mark.parametrize("w,h,n", [(1920, 1080, 5), (96, 7, 3), (1282, 3, 2), (6, 1, 8), (3840, 64, 1)])
def test_v210_read_batch_vs_oracle(w, h, n):
    """n frames in one launch == n single reads (both the LDS-table kernel and, for ragged widths, the gather kernel)"""
    import torch
    import hip_harness as hh
    rcm, rlut, rgm = hh.ColourParams.reader("709", "2020")
    srcs = [frames.v210_random(w, h, 7000 + 13 * i + w) for i in range(n)]
    outs = [torch.full((w * h * 4,), float("nan"), dtype=torch.float32, device="cuda") for _ in range(n)]
    hh.ctx().v210_read_batch([hh.dev(s) for s in srcs], outs, w, h, rcm, rlut, rgm)
    for i in range(n):
        want = orc.v210_read(srcs[i], w, h, orc.ycbcr2rgb_matrix("709"), orc.gamma2linear_lut("709"), orc.rgb2rgb_matrix("709", "2020"))
        assert_bits(hh.host(outs[i]), want, "v210_read_batch %dx%d frame %d of %d" % (w, h, i, n))
    with pytest.raises(Exception, match="1..8 frames"):
        hh.ctx().v210_read_batch([], [], w, h, rcm, rlut, rgm)


@pytest.mark.parametrize("w,h", [(1920, 540), (301, 33), (3, 2), (250, 17), (64, 1)])
def test_yadif_pair_vs_oracle(w, h):
    """both fields of a frame in one pass == the filter run once per parity (yadif.ts:100-145, send_field)"""
    import torch
    import hip_harness as hh
    p, c, n = (frames.rgba_random(w, h, 950 + i) for i in range(3))
    for tff in (0, 1):
        for skip in (False, True):
            out = [torch.full((w * h * 4,), float("nan"), dtype=torch.float32, device="cuda") for _ in range(2)]
            hh.ctx().yadif_pair(hh.dev(p), hh.dev(c), hh.dev(n), out[0], out[1], w, h, tff, skip)
            for parity in (0, 1):
                assert_bits(hh.host(out[parity]), orc.yadif(p, c, n, parity, tff, skip),
                            "yadif_pair %dx%d p%d t%d s%d" % (w, h, parity, tff, skip))
    with pytest.raises(Exception, match="same buffer"):
        hh.ctx().yadif_pair(hh.dev(p), hh.dev(c), hh.dev(n), out[0], out[0], w, h, 1, False)


@pytest.mark.parametrize("w,h,n", [(1920, 270, 2), (96, 33, 3), (6, 2, 1), (348, 17, 1), (354, 64, 4), (1920, 1, 1)])
def test_v210_yadif_pair_vs_oracle(w, h, n, lut_path):
    """the fused de-interlacing reader == ToRGBA on the three window frames, then Yadif once per parity"""
    import torch
    import hip_harness as hh
    rcm, rlut, rgm = hh.ColourParams.reader("709", "2020")
    if lut_path == "global_lut":  # the kernel exists in the LDS-table form only and says so
        with pytest.raises(Exception, match="no LDS form"):
            hh.ctx().v210_yadif_pair([(rlut, rlut, rlut, rlut, rgm)], w, h, 1, False, rcm, rlut, rgm)
        return
    o_args = (orc.ycbcr2rgb_matrix("709"), orc.gamma2linear_lut("709"), orc.rgb2rgb_matrix("709", "2020"))
    for tff, skip in ((1, False), (0, False), (1, True)):
        wins = [[frames.v210_random(w, h, 8100 + 31 * l + 7 * i + w + tff) for i in range(3)] for l in range(n)]
        outs = [[torch.full((w * h * 4,), float("nan"), dtype=torch.float32, device="cuda") for _ in range(2)] for _ in range(n)]
        hh.ctx().v210_yadif_pair([(hh.dev(wins[l][0]), hh.dev(wins[l][1]), hh.dev(wins[l][2]), outs[l][0], outs[l][1]) for l in range(n)],
                                 w, h, tff, skip, rcm, rlut, rgm)
        for l in range(n):
            p, c, nx = (orc.v210_read(f, w, h, *o_args) for f in wins[l])
            for parity in (0, 1):
                assert_bits(hh.host(outs[l][parity]), orc.yadif(p, c, nx, parity, tff, skip),
                            "v210_yadif_pair %dx%d layer %d p%d t%d s%d" % (w, h, l, parity, tff, skip))
    with pytest.raises(Exception, match="multiple of 6"):
        hh.ctx().v210_yadif_pair([(hh.dev(wins[0][0]), hh.dev(wins[0][1]), hh.dev(wins[0][2]), outs[0][0], outs[0][1])], 100, 4, 1, False, rcm, rlut, rgm)
    with pytest.raises(Exception, match="same buffer"):
        hh.ctx().v210_yadif_pair([(hh.dev(wins[0][0]), hh.dev(wins[0][1]), hh.dev(wins[0][2]), outs[0][0], outs[0][0])], w, h, 1, False, rcm, rlut, rgm)


@pytest.mark.parametrize("fmt", ["yuv422p10", "yuv422p8", "yuv420p", "nv12"])
@pytest.mark.parametrize("w,h,n", [(1920, 64, 2), (100, 33, 3), (346, 17, 1), (2, 2, 1)])
def test_planar_yadif_pair_vs_oracle(fmt, w, h, n, lut_path):
    """windows of interlaced FILE frames (planar 4:2:2, what decoders of XDCAM / ProRes material hand over): the fused de-interlacing
    reader == that format's ToRGBA on the three window frames (the 8-bit format with its own Loader matrix), then Yadif per parity"""
    import torch
    import hip_harness as hh
    from phaneron_amd import capi
    if lut_path == "global_lut":  # the fused de-interlacing reader exists in the LDS-table form only (its refusal is checked with v210 above)
        return
    if fmt in ("yuv420p", "nv12"):
        h += h & 1  # (4:2:0: a chroma line serves two luma lines - even heights)
    _, rlut, rgm = hh.ColourParams.reader("709", "2020")
    rng = orc.FORMAT_RANGE[fmt]
    rcm = hh.dev(capi.ycbcr2rgb_matrix("709", *rng))
    o_args = (orc.ycbcr2rgb_matrix("709", *rng), orc.gamma2linear_lut("709"), orc.rgb2rgb_matrix("709", "2020"))
    for tff, skip, rgb in ((1, False, False), (0, False, False), (1, True, True)):
        wins = [[frames.pack_random(fmt, w, h, 9100 + 31 * l + 7 * i + w + tff) for i in range(3)] for l in range(n)]
        px = 3 if rgb else 4
        outs = [[torch.full((w * h * px,), float("nan"), dtype=torch.float32, device="cuda") for _ in range(2)] for _ in range(n)]
        dev = lambda frame: tuple(hh.dev(np.ascontiguousarray(p)) for p in frame)
        hh.ctx().v210_yadif_pair([(dev(wins[l][0]), dev(wins[l][1]), dev(wins[l][2]), outs[l][0], outs[l][1]) for l in range(n)],
                                 w, h, tff, skip, rcm, rlut, rgm, rgb=rgb, packing=fmt)
        for l in range(n):
            p, c, nx = (orc.pack_read(fmt, f, w, h, *o_args) for f in wins[l])
            for parity in (0, 1):
                want = orc.yadif(p, c, nx, parity, tff, skip)
                got = hh.host(outs[l][parity])
                if rgb:
                    want = np.ascontiguousarray(want[..., :3])
                assert_bits(got, want, "%s yadif pair %dx%d layer %d p%d t%d s%d" % (fmt, w, h, l, parity, tff, skip))
    with pytest.raises(Exception, match="multiple of 2"):
        hh.ctx().v210_yadif_pair([(dev(wins[0][0]), dev(wins[0][1]), dev(wins[0][2]), outs[0][0], outs[0][1])], 101, 4, 1, False, rcm, rlut, rgm, packing=fmt)


@pytest.mark.parametrize("iw,ih,ow,oh,kw", [
    (1920, 1080, 3840, 2160, {}),
    (960, 540, 960, 540, dict(scale_x=0.5, scale_y=0.5, offset_x=0.25, offset_y=-0.25)),
    (320, 180, 320, 180, dict(rotate=0.1, anchor_x=0.2, anchor_y=-0.1, flip_h=True)),
])
def test_transform_vs_oracle(iw, ih, ow, oh, kw):
    import torch
    import hip_harness as hh
    from phaneron_amd import capi
    img = frames.rgba_random(iw, ih, 77)
    m = capi.transform_matrix(ow, oh, **kw)
    assert_bits(m, orc.transform_matrix(ow, oh, **kw), "matrix")
    out = torch.zeros(ow * oh * 4, dtype=torch.float32, device="cuda")
    hh.ctx().transform(hh.dev(img), iw, ih, hh.dev(m), out, ow, oh)
    assert_bits(hh.host(out), orc.transform(img, m, ow, oh), "transform")


def test_resize_vs_oracle():
    import torch
    import hip_harness as hh
    img = frames.rgba_random(640, 360, 78)
    for scale, ox, oy, fh, fv, ow, oh in [(1.0, 0, 0, 0, 0, 1280, 720), (0.5, 0.3, -0.2, 1, 0, 640, 360),
                                          (1.7, -1.0, 1.0, 0, 1, 333, 111)]:
        flip = np.array([1.0 if fh else 0.0, -1.0 if fh else 1.0, 1.0 if fv else 0.0, -1.0 if fv else 1.0], np.float32)
        out = torch.zeros(ow * oh * 4, dtype=torch.float32, device="cuda")
        hh.ctx().resize(hh.dev(img), 640, 360, scale, ox, oy, hh.dev(flip), out, ow, oh)
        assert_bits(hh.host(out), orc.resize(img, scale, ox, oy, fh, fv, ow, oh), "resize")


@pytest.mark.parametrize("n", [2, 3, 4, 6, 8])
def test_combine_vs_oracle_1080p(n):
    import torch
    import hip_harness as hh
    w, h = 1920, 1080 // 4
    layers = [frames.rgba_random(w, h, 500 + i) for i in range(n)]
    out = torch.zeros(w * h * 4, dtype=torch.float32, device="cuda")
    hh.ctx().combine([hh.dev(l) for l in layers], out, w, h)
    assert_bits(hh.host(out), orc.combine(layers), "combine_%d" % n)


def test_lut_registry_reports_lds_form():
    import hip_harness as hh
    for spec in ("709", "2020", "601-625", "sRGB"):
        _, lut, _ = hh.ColourParams.reader(spec, "709")
        info = hh.ctx().lut_info(lut)
        assert 0 < info["lds_bytes"] <= 160 * 1024, info
        _, wlut = hh.ColourParams.writer(spec)
        assert 0 < hh.ctx().lut_info(wlut)["lds_bytes"] <= 160 * 1024


def test_incompressible_lut_falls_back_to_gather_kernels():
    """A table that is not locally smooth cannot be held exactly in LDS: it must stay plain and
    the global-gather kernels must still give the exact answer."""
    import torch
    import hip_harness as hh
    from phaneron_amd import capi
    rng = np.random.default_rng(7)
    lut = rng.random(65536, dtype=np.float32)
    dlut = hh.dev(lut)
    assert hh.ctx().register_lut(dlut, lut) is False
    assert hh.ctx().lut_info(dlut)["lds_bytes"] == 0
    w, h = 96, 4
    words = frames.v210_random(w, h, 3)
    cm, gm = capi.ycbcr2rgb_matrix("709"), capi.rgb2rgb_matrix("709", "709")
    out = torch.zeros(w * h * 4, dtype=torch.float32, device="cuda")
    hh.ctx().v210_read(hh.dev(words), out, w, h, hh.dev(cm), dlut, hh.dev(np.concatenate([gm, np.zeros(3, np.float32)])))
    assert_bits(hh.host(out), orc.v210_read(words, w, h, cm, lut, gm), "plain LUT")
    hh.ctx().unregister_lut(dlut)


def test_combine_rejects_single_layer():
    import torch
    import hip_harness as hh
    from phaneron_amd import capi
    t = torch.zeros(16, dtype=torch.float32, device="cuda")
    with pytest.raises(capi.PhaneronError):
        hh.ctx().combine([t], t, 2, 2)


@pytest.mark.parametrize("n,w,h,rspec,wspec", [(1, 96, 4, "709", "709"), (2, 1920, 64, "709", "709"),
                                               (4, 1920, 270, "709", "2020"), (8, 480, 32, "2020", "709")])
def test_fused_pipeline_vs_oracle_chain(n, w, h, rspec, wspec):
    """The fused kernel must equal the reference's job batch read x n -> combine_n -> write."""
    import torch
    import hip_harness as hh
    layers = [frames.v210_random(w, h, frames.layer_seed(0, i), legal=(i % 2 == 0)) for i in range(n)]
    cm, lut, gm = hh.ColourParams.reader(rspec, wspec)
    wcm, wlut = hh.ColourParams.writer(wspec)
    out = torch.zeros(frames.v210_pitch_bytes(w) * h // 4, dtype=torch.int32, device="cuda")
    hh.ctx().fused_v210_combine([hh.dev(l) for l in layers], out, w, h, cm, lut, gm, wcm, wlut)
    want = orc.pipeline_v210_combine(layers, w, h, orc.ycbcr2rgb_matrix(rspec), orc.gamma2linear_lut(rspec),
                                     orc.rgb2rgb_matrix(rspec, wspec), orc.rgb2ycbcr_matrix(wspec),
                                     orc.linear2gamma_lut(wspec))
    assert_bits(hh.host(out, np.uint32), want, "fused n=%d" % n)


@pytest.mark.parametrize("n,w,h", [(4, 1280, 36), (1, 1280, 5), (3, 100, 7), (8, 52, 3), (2, 1302, 4), (4, 50, 2), (2, 2, 1)])
def test_fused_pipeline_ragged_widths(n, w, h):
    """Widths that are not a multiple of 48 (1280: the reference's 720p50, src/config.ts:43-54; 100 and 52: a tail of 4 pixels; 1302: whole
    quads, then cleared slots; 50, 2: less than a block): lines addressed by pitch, the tail quad read with the fourth vector
    component 0 (v210.ts:88-93) and written from truncated indices with round() (v210.ts:173-184), the slots behind it cleared
    (v210.ts:131-136) - all of it in the fused kernel, against the oracle's chain.  Illegal codes included; the output is poisoned first."""
    import torch
    import hip_harness as hh
    layers = [frames.v210_random(w, h, frames.layer_seed(6, i), legal=(i % 2 == 1)) for i in range(n)]
    cm, lut, gm = hh.ColourParams.reader("709", "2020")
    wcm, wlut = hh.ColourParams.writer("2020")
    out = torch.full((frames.v210_pitch_bytes(w) * h // 4,), 0x2AAAAAAA, dtype=torch.int32, device="cuda")
    hh.ctx().fused_v210_combine([hh.dev(l) for l in layers], out, w, h, cm, lut, gm, wcm, wlut)
    want = orc.pipeline_v210_combine(layers, w, h, orc.ycbcr2rgb_matrix("709"), orc.gamma2linear_lut("709"), orc.rgb2rgb_matrix("709", "2020"),
                                     orc.rgb2ycbcr_matrix("2020"), orc.linear2gamma_lut("2020"))
    assert_bits(hh.host(out, np.uint32), want, "fused n=%d %dx%d" % (n, w, h))


def test_fused_pipeline_ragged_width_shows_every_layer():
    """the same with the lower layers made visible (a reader table whose entry 0 is +Inf poisons what `combine` would drop) and a
    non-standard matrix, so that the tail's missing offset column shows in every layer"""
    import torch
    import hip_harness as hh
    w, h, n = 1280, 9, 4
    layers = [frames.v210_random(w, h, frames.layer_seed(8, i), legal=False) for i in range(n)]
    m = (orc.ycbcr2rgb_matrix("709").reshape(3, 4) * np.array([[1.0], [1.03125], [0.96875]], np.float32)).reshape(-1).astype(np.float32)
    m[1], m[10] = np.float32(1.5e-4), np.float32(-2.5e-4)
    lut = orc.gamma2linear_lut("709").copy()
    lut[0] = np.inf
    gm = orc.rgb2rgb_matrix("709", "2020")
    dlut = hh.dev(lut)
    assert hh.ctx().register_lut(dlut, lut)
    wcm, wlut = hh.ColourParams.writer("2020")
    out = torch.zeros(frames.v210_pitch_bytes(w) * h // 4, dtype=torch.int32, device="cuda")
    hh.ctx().fused_v210_combine([hh.dev(l) for l in layers], out, w, h, hh.dev(m), dlut, hh.dev(np.concatenate([gm, np.zeros(3, np.float32)])), wcm, wlut)
    want = orc.pipeline_v210_combine(layers, w, h, m, lut, gm, orc.rgb2ycbcr_matrix("2020"), orc.linear2gamma_lut("2020"))
    assert_bits(hh.host(out, np.uint32), want, "ragged, every layer visible")
    hh.ctx().unregister_lut(dlut)


@pytest.mark.parametrize("shape", ["full", "negative_luma_gain", "cb_in_red_only", "standard_with_negative_zero"])
def test_fused_pipeline_with_non_standard_matrices(shape):
    """The fused kernel has a fast path for the matrix shape colourMaths produces (one luma gain, no Cb
    in R, no Cr in B) chosen on the device from the coefficient values.  Matrices that miss the shape
    by one coefficient - or hit it with a -0 - must still equal the oracle chain, bit for bit."""
    import torch
    import hip_harness as hh
    w, h, n = 1920, 10, 3
    layers = [frames.v210_random(w, h, frames.layer_seed(5, i), legal=(i != 1)) for i in range(n)]
    m = orc.ycbcr2rgb_matrix("709").copy()
    if shape == "full":          # every coefficient non-zero, three different luma gains
        m = (m.reshape(3, 4) * np.array([[1.0], [1.03125], [0.96875]], np.float32)).reshape(-1).astype(np.float32)
        m[1], m[10] = np.float32(1.5e-4), np.float32(-2.5e-4)
    elif shape == "negative_luma_gain":  # same gain in all rows, but negative: Y * m0 can be -0
        m[0] = m[4] = m[8] = np.float32(-m[0])
    elif shape == "cb_in_red_only":
        m[1] = np.float32(3.0e-4)
    else:                         # the standard shape with a NEGATIVE zero where the zeros are
        m[1], m[10] = np.float32(-0.0), np.float32(-0.0)
    gm = orc.rgb2rgb_matrix("709", "2020")
    lut = orc.gamma2linear_lut("709")
    _, dlut, dgm = hh.ColourParams.reader("709", "2020")
    wcm, wlut = hh.ColourParams.writer("2020")
    out = torch.zeros(frames.v210_pitch_bytes(w) * h // 4, dtype=torch.int32, device="cuda")
    hh.ctx().fused_v210_combine([hh.dev(l) for l in layers], out, w, h, hh.dev(m), dlut, dgm, wcm, wlut)
    want = orc.pipeline_v210_combine(layers, w, h, m, lut, gm, orc.rgb2ycbcr_matrix("2020"), orc.linear2gamma_lut("2020"))
    assert_bits(hh.host(out, np.uint32), want, shape)


@pytest.mark.parametrize("case", ["pip_1080", "upscale2x", "single_layer", "interlaced", "all_direct", "ragged_1280", "ragged_100_field", "ragged_1302"])
def test_compose_write_vs_oracle_chain(case, lut_path):
    """ph_compose_write_v210 == transform x N -> combine_N -> v210 write of the oracle."""
    import torch
    import hip_harness as hh
    from phaneron_amd import capi
    if lut_path != "lds_lut":
        pytest.skip("the fused compositor exists only in the LDS-LUT form")
    k = hh.ctx()
    if case == "pip_1080":
        ow, oh, il = 960, 270, 0
        specs = [(960, 270, None), (960, 270, dict(scale_x=0.5, scale_y=0.5, offset_x=-0.25, offset_y=-0.25)),
                 (480, 135, dict(scale_x=0.5, scale_y=0.5, offset_x=0.25, offset_y=0.25, rotate=0.05)), (960, 270, {})]
    elif case == "upscale2x":
        ow, oh, il = 960, 540, 0
        specs = [(480, 270, {}) for _ in range(4)]
    elif case == "single_layer":
        ow, oh, il = 192, 40, 0
        specs = [(96, 20, dict(flip_h=True))]
    elif case == "ragged_1280":  # 720p50's width: a tail quad of two pixels and two cleared slots on every line (v210.ts:131-136,166-193)
        ow, oh, il = 1280, 18, 0
        specs = [(1280, 18, None), (640, 9, dict(scale_x=0.5, scale_y=0.5, offset_x=0.25)), (1280, 18, dict(rotate=0.03))]
    elif case == "ragged_100_field":  # a tail of four pixels, one field
        ow, oh, il = 100, 12, 3
        specs = [(100, 12, None), (50, 6, dict(scale_x=0.8, scale_y=0.8))]
    elif case == "ragged_1302":  # whole quads only, then cleared slots
        ow, oh, il = 1302, 5, 0
        specs = [(1302, 5, {})]
    elif case == "all_direct":  # no layer sampled: the one-load-per-layer variant of the kernel
        ow, oh, il = 480, 50, 1
        specs = [(480, 50, None) for _ in range(3)]
    else:
        ow, oh, il = 480, 64, 3
        specs = [(480, 64, None), (240, 32, dict(scale_x=0.75, scale_y=0.75))]
    imgs = [frames.rgba_random(w, h, 9000 + i) for i, (w, h, _) in enumerate(specs)]
    mats = [None if kw is None else capi.transform_matrix(ow, oh, **kw) for (_, _, kw) in specs]
    wcm, wlut = hh.ColourParams.writer("2020")
    dst0 = np.full(frames.v210_pitch_bytes(ow) * oh // 4, cases.POISON, np.uint32)
    out = hh.dev(dst0)
    layers = [(hh.dev(im), w, h, None if m is None else hh.dev(m)) for im, (w, h, _), m in zip(imgs, specs, mats)]
    k.compose_write_v210(layers, out, ow, oh, il, wcm, wlut)
    xf = [im if m is None else orc.transform(im, m, ow, oh) for im, m in zip(imgs, mats)]
    comb = xf[0] if len(xf) == 1 else orc.combine(xf)
    want = orc.v210_write(comb, ow, oh, il, orc.rgb2ycbcr_matrix("2020"), orc.linear2gamma_lut("2020"), out=dst0.copy())
    assert_bits(hh.host(out, np.uint32), want, "compose_write " + case)


def test_compose_write_refuses_unregistered_lut():
    import torch
    import hip_harness as hh
    from phaneron_amd import capi
    t = torch.zeros(96 * 4 * 4, dtype=torch.float32, device="cuda")
    raw = hh.dev(capi.linear2gamma_lut("709"))  # not registered: no LDS form
    with pytest.raises(capi.PhaneronError, match="LDS form"):
        hh.ctx().compose_write_v210([(t, 96, 4, None), (t, 96, 4, None)], torch.zeros(96 * 4, dtype=torch.int32, device="cuda"),
                                    96, 4, 0, hh.ColourParams.writer("709")[0], raw)


FMT_KATS = [("yuv422p10", 1920, 1080, "709"), ("yuv420p", 1920, 1080, "709"), ("nv12", 1920, 1080, "709"),
            ("yuv422p8", 718, 480, "709"), ("rgba8", 1920, 1080, "sRGB"), ("bgra8", 1920, 1080, "sRGB")]


@pytest.mark.parametrize("fmt,w,h,spec", FMT_KATS)
def test_reference_roundtrip_scripts_on_gpu(fmt, w, h, spec):
    """The reference's src/process/test/*Test.ts scripts: test pattern -> ToRGBA -> FromRGBA ->
    Buffer.compare == 0, at their own sizes (1920x1080; 718 wide for yuv422p8)."""
    import torch
    import hip_harness as hh
    planes = frames.pack_ramp(fmt, w, h)
    k = hh.ctx()
    rcm, rlut, rgm = hh.ColourParams.fmt_reader(fmt, spec, spec)
    wcm, wlut = hh.ColourParams.fmt_writer(fmt, spec)
    rgba = torch.zeros(w * h * 4, dtype=torch.float32, device="cuda")
    dplanes = [hh.dev(p) for p in planes]
    back = [torch.zeros_like(p) for p in dplanes]
    k.pack_read(fmt, dplanes, rgba, w, h, rcm, rlut, rgm)
    k.pack_write(fmt, rgba, back, w, h, 0, wcm, wlut)
    assert hashlib.sha256(hh.host(rgba).tobytes()).hexdigest() == KAT["%s_%dx%d_rgba_sha256" % (fmt, w, h)]
    for a, b in zip(planes, back):
        assert np.array_equal(a, hh.host(b))


@pytest.mark.parametrize("fmt", ["yuv422p10", "yuv422p8", "yuv420p", "nv12", "rgba8", "bgra8"])
def test_pack_read_batch_equals_the_oracle_per_frame(fmt, lut_path):
    """ph_pack_read_batch: 2, 5 and 8 frames of one format and size in one launch (several channels' clips of a tick) - every image the
    oracle's reader of that frame; sizes that do not fill the chip and a ragged width"""
    import torch
    import hip_harness as hh
    rng = orc.FORMAT_RANGE[fmt]
    rcm, rlut, rgm = hh.ColourParams.fmt_reader(fmt, "709", "2020")
    o_args = (None if rng is None else orc.ycbcr2rgb_matrix("709", *rng), orc.gamma2linear_lut("709"), orc.rgb2rgb_matrix("709", "2020"))
    for (w, h, n) in ((384, 54, 2), (1920, 64, 5), (702 if fmt not in ("rgba8", "bgra8") else 704, 10, 8)):
        frames_ = [frames.pack_random(fmt, w, h, 8800 + 13 * i + w) for i in range(n)]
        outs = [torch.full((w * h * 4,), float("nan"), dtype=torch.float32, device="cuda") for _ in range(n)]
        hh.ctx().pack_read_batch(fmt, [[hh.dev(p) for p in f] for f in frames_], outs, w, h, rcm, rlut, rgm)
        for i in range(n):
            assert_bits(hh.host(outs[i]), orc.pack_read(fmt, frames_[i], w, h, *o_args), "%s batch read %dx%d frame %d of %d" % (fmt, w, h, i, n))


@pytest.mark.parametrize("fmt", ["yuv422p10", "yuv422p8", "yuv420p", "nv12", "rgba8", "bgra8"])
def test_pack_formats_vs_oracle_1080(fmt):
    """Full-range random planes at 1920x270 (+ an odd tail width), both fields, against the oracle."""
    import torch
    import hip_harness as hh
    for w, h in ((1920, 270), (718, 10)):
        if fmt in ("rgba8", "bgra8") and w == 718:
            w = 704
        planes = frames.pack_random(fmt, w, h, 77)
        rng = orc.FORMAT_RANGE[fmt]
        spec, ospec = "709", "2020"
        rcm, rlut, rgm = hh.ColourParams.fmt_reader(fmt, spec, ospec)
        out = torch.zeros(w * h * 4, dtype=torch.float32, device="cuda")
        hh.ctx().pack_read(fmt, [hh.dev(p) for p in planes], out, w, h, rcm, rlut, rgm)
        want = orc.pack_read(fmt, planes, w, h, None if rng is None else orc.ycbcr2rgb_matrix(spec, *rng),
                             orc.gamma2linear_lut(spec), orc.rgb2rgb_matrix(spec, ospec))
        assert_bits(hh.host(out), want, "%s read %dx%d" % (fmt, w, h))
        rgba = frames.rgba_random(w, h, 78, -0.05, 1.05)
        wcm, wlut = hh.ColourParams.fmt_writer(fmt, ospec)
        for il in (0, 1, 3):
            dst = [np.full(n, 0x5A, np.uint8) for n in frames.pack_plane_bytes(fmt, w, h)]
            dplanes = [hh.dev(p) for p in dst]
            hh.ctx().pack_write(fmt, hh.dev(rgba), dplanes, w, h, il, wcm, wlut)
            want = orc.pack_write(fmt, rgba, w, h, il, None if rng is None else orc.rgb2ycbcr_matrix(ospec, *rng),
                                  orc.linear2gamma_lut(ospec), planes=dst)
            for i, (g, wnt) in enumerate(zip(dplanes, want)):
                assert np.array_equal(hh.host(g), wnt), "%s write il=%d plane %d %dx%d" % (fmt, il, i, w, h)


def test_fused_equals_unfused_kernels_2160p():
    """Full BASELINE size: the fused kernel against the separate HIP kernels (already pinned to
    the oracle above), plus the size-independent property that with alpha == 1 everywhere the
    composite equals the top layer alone."""
    import torch
    import hip_harness as hh
    w, h, n = 3840, 2160, 4
    k = hh.ctx()
    layers = [hh.dev(frames.v210_random(w, h, frames.layer_seed(0, i))) for i in range(n)]
    cm, lut, gm = hh.ColourParams.reader("709", "2020")
    wcm, wlut = hh.ColourParams.writer("2020")
    fused = torch.zeros(frames.v210_pitch_bytes(w) * h // 4, dtype=torch.int32, device="cuda")
    k.fused_v210_combine(layers, fused, w, h, cm, lut, gm, wcm, wlut)
    rgba = [torch.empty(w * h * 4, dtype=torch.float32, device="cuda") for _ in range(n)]
    for l, r in zip(layers, rgba):
        k.v210_read(l, r, w, h, cm, lut, gm)
    comb = torch.empty(w * h * 4, dtype=torch.float32, device="cuda")
    k.combine(rgba, comb, w, h)
    unfused = torch.zeros_like(fused)
    k.v210_write(comb, unfused, w, h, 0, wcm, wlut)
    top = torch.zeros_like(fused)
    k.v210_write(rgba[n - 1], top, w, h, 0, wcm, wlut)
    k.wait()
    torch.cuda.synchronize()
    assert torch.equal(fused, unfused)
    assert torch.equal(fused, top)
    # and against the CPU side at the full size, every word: the reference's own kernels compiled for x86 (oracle/_ref,
    # ~1 s for 4 x 2160p on the host threads) when that build travelled with the snapshot, the oracle's restatement if not
    hl = [frames.v210_random(w, h, frames.layer_seed(0, i)) for i in range(n)]
    colour = (orc.ycbcr2rgb_matrix("709"), orc.gamma2linear_lut("709"), orc.rgb2rgb_matrix("709", "2020"),
              orc.rgb2ycbcr_matrix("2020"), orc.linear2gamma_lut("2020"))
    if orc.have_ref_fast():
        r = orc.ref_fast()
        r.ref_set_num_threads(orc.effective_cpus())
        want = orc.ref_pipeline_v210_combine(r, hl, w, h, *colour)
    else:
        want = orc.pipeline_v210_combine(hl, w, h, *colour)
    assert np.array_equal(hh.host(fused, np.uint32), np.asarray(want).view(np.uint32).reshape(-1)), "fused 4 x 2160p differs from the CPU side"


@pytest.mark.parametrize("w,h,n", [(7680, 4320, 2), (7680, 1082, 3), (48, 1, 3), (96, 3, 4)])
def test_fused_tile_walk_equals_unfused_kernels(w, h, n):
    """Sizes that make a workgroup walk several tiles (4320p: 21 600 quads per CU = 3 full tiles of
    6144 + a partial one), a ragged last tile, and frames smaller than one slice: the fused kernel
    against the separate HIP kernels (pinned to the oracle elsewhere)."""
    import torch
    import hip_harness as hh
    k = hh.ctx()
    words = frames.v210_pitch_bytes(w) * h // 4
    g = torch.Generator(device="cuda").manual_seed(w * 31 + h)
    layers = [torch.randint(0, 2 ** 30, (words,), dtype=torch.int32, device="cuda", generator=g) for _ in range(n)]
    cm, lut, gm = hh.ColourParams.reader("709", "2020")
    wcm, wlut = hh.ColourParams.writer("2020")
    fused = torch.zeros(words, dtype=torch.int32, device="cuda")
    k.fused_v210_combine(layers, fused, w, h, cm, lut, gm, wcm, wlut)
    rgba = [torch.empty(w * h * 4, dtype=torch.float32, device="cuda") for _ in range(n)]
    for l, r in zip(layers, rgba):
        k.v210_read(l, r, w, h, cm, lut, gm)
    comb = torch.empty(w * h * 4, dtype=torch.float32, device="cuda")
    k.combine(rgba, comb, w, h)
    unfused = torch.zeros_like(fused)
    k.v210_write(comb, unfused, w, h, 0, wcm, wlut)
    k.wait()
    torch.cuda.synchronize()
    assert torch.equal(fused, unfused)


@pytest.mark.parametrize("jobs,n,w,h", [(2, 4, 1920, 1080), (4, 3, 1920, 270), (3, 2, 96, 5), (8, 1, 480, 36)])
def test_fused_batch_equals_separate_calls(jobs, n, w, h):
    """ph_fused_v210_combine_batch: several frames in one launch (the CUs are divided between the jobs)
    give exactly what one call per frame gives - including job counts that do not divide the CU count
    and frames smaller than a slice."""
    import torch
    import hip_harness as hh
    k = hh.ctx()
    words = frames.v210_pitch_bytes(w) * h // 4
    g = torch.Generator(device="cuda").manual_seed(jobs * 1000 + n)
    layers = [[torch.randint(0, 2 ** 30, (words,), dtype=torch.int32, device="cuda", generator=g) for _ in range(n)]
              for _ in range(jobs)]
    cm, lut, gm = hh.ColourParams.reader("709", "2020")
    wcm, wlut = hh.ColourParams.writer("2020")
    batched = [torch.zeros(words, dtype=torch.int32, device="cuda") for _ in range(jobs)]
    single = [torch.zeros(words, dtype=torch.int32, device="cuda") for _ in range(jobs)]
    k.fused_v210_combine_batch(layers, batched, w, h, cm, lut, gm, wcm, wlut)
    for j in range(jobs):
        k.fused_v210_combine(layers[j], single[j], w, h, cm, lut, gm, wcm, wlut)
    k.wait()
    torch.cuda.synchronize()
    for j in range(jobs):
        assert torch.equal(batched[j], single[j]), j


def test_roundtrip_2160p_properties():
    """Full UHD size, size-independent properties:
    (1) the reference's ramp pattern survives read(2020->2020) -> write(2020) byte for byte;
    (2) write -> read -> write is a fixed point (within 1 code) when pixel pairs share a colour
        (4:2:2 keeps chroma of even pixels only, so only pair-constant images can round-trip)."""
    import torch
    import hip_harness as hh
    w, h = 3840, 2160
    k = hh.ctx()
    cm, lut, gm = hh.ColourParams.reader("2020", "2020")
    wcm, wlut = hh.ColourParams.writer("2020")
    words = frames.v210_pitch_bytes(w) * h // 4
    ramp = frames.v210_ramp(w, h)
    assert hashlib.sha256(ramp.tobytes()).hexdigest() == HM["ramp"]["3840x2160"]
    rgba = torch.empty(w * h * 4, dtype=torch.float32, device="cuda")
    back = torch.zeros(words, dtype=torch.int32, device="cuda")
    k.v210_read(hh.dev(ramp), rgba, w, h, cm, lut, gm)
    k.v210_write(rgba, back, w, h, 0, wcm, wlut)
    assert np.array_equal(hh.host(back, np.uint32), ramp)

    pair = np.repeat(frames.rgba_random(w // 2, h, 99, 0.0, 1.0), 2, axis=1).copy()
    v1 = torch.zeros(words, dtype=torch.int32, device="cuda")
    v2 = torch.zeros(words, dtype=torch.int32, device="cuda")
    k.v210_write(hh.dev(pair), v1, w, h, 0, wcm, wlut)
    k.v210_read(v1, rgba, w, h, cm, lut, gm)
    k.v210_write(rgba, v2, w, h, 0, wcm, wlut)
    a = frames.v210_unpack_codes(hh.host(v1, np.uint32), w, h)
    b = frames.v210_unpack_codes(hh.host(v2, np.uint32), w, h)
    for p0, p1 in zip(a, b):
        assert np.abs(p0.astype(np.int64) - p1.astype(np.int64)).max() <= 1
