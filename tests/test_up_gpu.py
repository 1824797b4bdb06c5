"""GPU tests (-m gpu) of ph_compose_up_write_v210 (a 2 x 2 block of output pixels per lane from one 3 x 3 source patch per
layer, for layers enlarged 2x or more) and of the packed-RGB output of ph_v210_yadif_pair_fmt that feeds it.  Compared word
for word with the oracle's chain transform.ts -> combine.ts -> v210.ts write."""
import numpy as np
import pytest

import frames
from oracle import orc

pytestmark = pytest.mark.gpu


def m(ow, oh, **kw):
    from phaneron_amd import capi
    return capi.transform_matrix(ow, oh, **kw)


def run(layers, ow, oh, interlace=0, rgb=False, spec="2020", dst=None):
    """layers: [(rgba float array h x w x 4, matrix)]; returns (device words, oracle words)"""
    import torch
    import hip_harness as hh
    k = hh.ctx()
    wcm, wlut = hh.ColourParams.writer(spec)
    wr_o = (orc.rgb2ycbcr_matrix(spec), orc.linear2gamma_lut(spec))
    placed = [orc.transform(img, mat, ow, oh) for img, mat in layers]
    comb = placed[0] if len(placed) == 1 else orc.combine(placed)
    want = orc.v210_write(comb, ow, oh, interlace, *wr_o, out=None if dst is None else dst.copy())
    out = hh.dev(dst) if dst is not None else torch.zeros(frames.v210_pitch_bytes(ow) * oh // 4, dtype=torch.int32, device="cuda")
    dl = []
    for img, mat in layers:
        h, w, _ = img.shape
        data = np.ascontiguousarray(img[..., :3]) if rgb else img
        dl.append((hh.dev(data.reshape(-1)), w, h, mat))
    k.compose_up_write_v210(dl, out, ow, oh, interlace, wcm, wlut, rgb=rgb)
    return hh.host(out, np.uint32), np.asarray(want).reshape(-1)


def check(layers, ow, oh, what, **kw):
    got, want = run(layers, ow, oh, **kw)
    bad = np.flatnonzero(got != want)
    assert bad.size == 0, "%s: %d of %d words differ, first at word %d (line %d)" % (
        what, bad.size, got.size, bad[0], bad[0] // (frames.v210_pitch_bytes(ow) // 4))


def opaque(w, h, seed):
    img = frames.rgba_random(w, h, seed, -0.05, 1.05)
    img = img.reshape(h, w, 4).copy()
    img[..., 3] = 1.0
    return img


@pytest.mark.parametrize("rgb", [False, True], ids=["rgba", "packed-rgb"])
@pytest.mark.parametrize("n", [1, 2, 4])
def test_two_times_up_scale_of_full_frame_layers(n, rgb):
    """BASELINE config 3's compositor: 2x identity fill of every layer (transform.ts:54-55 without the half-pixel term: every
    second output pixel's first tap sits exactly on a texel boundary, where f32 noise decides which texel it is)"""
    sw, sh, ow, oh = 192, 54, 384, 108
    layers = [(opaque(sw, sh, 10 + l) if rgb else frames.rgba_random(sw, sh, 10 + l, -0.05, 1.05).reshape(sh, sw, 4), m(ow, oh)) for l in range(n)]
    check(layers, ow, oh, "%d layers 2x" % n, rgb=rgb)


@pytest.mark.parametrize("rgb", [False, True], ids=["rgba", "packed-rgb"])
def test_insets_with_borders_and_other_magnifications(rgb):
    """layers that cover part of the frame (the border colour and its alpha 0 take part), 3x and 2.5x, sources of several sizes"""
    ow, oh = 384, 120
    mk = (lambda w, h, s: opaque(w, h, s)) if rgb else (lambda w, h, s: frames.rgba_random(w, h, s, 0.0, 1.0).reshape(h, w, 4))
    layers = [(mk(192, 60, 1), m(ow, oh)),                                                    # 2x, full frame
              (mk(48, 15, 2), m(ow, oh, scale_x=0.5, scale_y=0.5, offset_x=-0.25, offset_y=0.2)),  # 4x, inset
              (mk(64, 20, 3), m(ow, oh, scale_x=0.5, scale_y=0.5, offset_x=0.3, offset_y=-0.3)),   # 3x, partly off screen
              (mk(96, 24, 4), m(ow, oh, scale_x=0.625, scale_y=0.5, offset_x=0.05, offset_y=0.1))]  # 2.5x
    check(layers, ow, oh, "insets", rgb=rgb)
    check(layers[1:], ow, oh, "insets without a background", rgb=rgb)


@pytest.mark.parametrize("interlace", [1, 3])
def test_field_outputs(interlace):
    """a field write takes every other line: a layer must be enlarged more than 2x vertically for its written rows to be less
    than a texel apart (here 4x)"""
    import hip_harness as hh
    from phaneron_amd import capi
    sw, sh, ow, oh = 96, 13, 192, 54
    dst = np.full(frames.v210_pitch_bytes(ow) * oh // 4, 0x2AAAAAAA, np.uint32)
    layers = [(opaque(sw, sh, 20), m(ow, oh)), (opaque(48, 6, 21), m(ow, oh, scale_x=0.5, scale_y=0.5, offset_x=0.2))]
    check(layers, ow, oh, "interlace %d" % interlace, interlace=interlace, rgb=True, dst=dst)
    with pytest.raises(capi.PhaneronError, match="must be enlarged"):  # exactly 2x vertically: a field's rows are a whole texel apart
        run([(opaque(96, 27, 22), m(ow, oh))], ow, oh, interlace=interlace, rgb=True)


@pytest.mark.parametrize("rgb", [False, True], ids=["rgba", "packed-rgb"])
@pytest.mark.parametrize("interlace", [0, 1, 3])
@pytest.mark.parametrize("shape", [(192, 54, 384, 108, 3), (96, 31, 288, 124, 2), (1920, 1080, 3840, 2160, 4)])
def test_both_fields_in_one_launch_equal_two_launches(shape, interlace, rgb):
    """ph_compose_up_write_v210_pair == ph_compose_up_write_v210 twice (and, at the small sizes, == the oracle's chain for each set)"""
    import torch
    import hip_harness as hh
    sw, sh, ow, oh, n = shape
    if interlace and oh * 0.99 <= 2 * sh:  # a field write needs more than 2x vertically (rows two lines apart)
        pytest.skip("not eligible as a field write")
    k = hh.ctx()
    wcm, wlut = hh.ColourParams.writer("2020")
    mat = m(ow, oh)
    words = frames.v210_pitch_bytes(ow) * oh // 4
    rng = np.random.default_rng(sw + interlace)
    sets = []
    for f in range(2):
        imgs = [opaque(sw, sh, 300 + 10 * f + l) if rgb else frames.rgba_random(sw, sh, 300 + 10 * f + l, -0.05, 1.05).reshape(sh, sw, 4) for l in range(n)]
        sets.append(imgs)
    dev = [[(hh.dev((np.ascontiguousarray(img[..., :3]) if rgb else img).reshape(-1)), sw, sh, mat) for img in imgs] for imgs in sets]
    fill = rng.integers(0, 2 ** 30, words, dtype=np.int64).astype(np.uint32)  # a field write leaves the other field's lines alone
    single = [hh.dev(fill.copy()) for _ in range(2)]
    pair = [hh.dev(fill.copy()) for _ in range(2)]
    for f in range(2):
        k.compose_up_write_v210(dev[f], single[f], ow, oh, interlace, wcm, wlut, rgb=rgb)
    k.compose_up_write_v210_pair(dev[0], dev[1], pair[0], pair[1], ow, oh, interlace, wcm, wlut, rgb=rgb)
    k.wait()
    for f in range(2):
        a, b = hh.host(single[f], np.uint32), hh.host(pair[f], np.uint32)
        bad = np.flatnonzero(a != b)
        assert bad.size == 0, "set %d: %d words differ from the single launch, first at %d" % (f, bad.size, bad[0])
    if ow <= 384:
        wr_o = (orc.rgb2ycbcr_matrix("2020"), orc.linear2gamma_lut("2020"))
        for f in range(2):
            placed = [orc.transform(img, mat, ow, oh) for img in sets[f]]
            want = orc.v210_write(placed[0] if n == 1 else orc.combine(placed), ow, oh, interlace, *wr_o, out=fill.copy())
            assert np.array_equal(hh.host(pair[f], np.uint32), np.asarray(want).reshape(-1)), "set %d differs from the oracle" % f
    with pytest.raises(Exception, match="same buffer"):
        k.compose_up_write_v210_pair(dev[0], dev[1], pair[0], pair[0], ow, oh, interlace, wcm, wlut, rgb=rgb)
    with pytest.raises(Exception, match="placed differently"):
        other = [(t, w, h, m(ow, oh, offset_x=0.01)) for t, w, h, _ in dev[1]]
        k.compose_up_write_v210_pair(dev[0], other, pair[0], pair[1], ow, oh, interlace, wcm, wlut, rgb=rgb)


def test_frame_shapes():
    """a row shorter than a wave step, rows that end in a short step, an odd number of rows (the last row has no partner),
    a frame smaller than the chip"""
    for ow, oh, sw, sh in ((48, 6, 24, 3), (192, 7, 96, 3), (336, 9, 100, 4), (3840, 26, 1920, 13)):
        layers = [(opaque(sw, sh, 30), m(ow, oh)), (opaque(sw // 2, max(sh // 2, 1), 31), m(ow, oh, scale_x=0.5, scale_y=0.5, offset_x=-0.2, offset_y=0.1))]
        check(layers, ow, oh, "%dx%d" % (ow, oh), rgb=True)


def test_placements_that_do_not_qualify_are_refused():
    import torch
    import hip_harness as hh
    from phaneron_amd import capi
    k = hh.ctx()
    wcm, wlut = hh.ColourParams.writer("709")
    ow, oh = 192, 54
    src = hh.dev(opaque(96, 27, 40).reshape(-1))
    out = torch.zeros(frames.v210_pitch_bytes(ow) * oh // 4, dtype=torch.int32, device="cuda")
    for kw in (dict(rotate=0.1), dict(flip_h=True), dict(scale_x=0.9, scale_y=0.9)):
        mat = capi.transform_matrix(ow, oh, **kw)
        src_w = 192 if "scale_x" in kw else 96  # a 192-wide source shrunk to 0.9: not enlarged
        with pytest.raises(capi.PhaneronError, match="must be enlarged"):
            k.compose_up_write_v210([(src, src_w, 27 if src_w == 96 else 13, mat)], out, ow, oh, 0, wcm, wlut)


def test_packed_rgb_fields_equal_the_rgba_fields():
    """ph_v210_yadif_pair_fmt(PH_IMG_RGB_F32): the same two de-interlaced fields without their constant alpha"""
    import torch
    import hip_harness as hh
    k = hh.ctx()
    w, h = 384, 40
    cm, lut, gm = hh.ColourParams.reader("709", "2020")
    win = [hh.dev(frames.v210_random(w, h, frames.layer_seed(7, i), legal=(i != 1))) for i in range(3)]
    for tff in (1, 0):
        rgba = [torch.zeros(w * h * 4, dtype=torch.float32, device="cuda") for _ in range(2)]
        rgb = [torch.zeros(w * h * 3, dtype=torch.float32, device="cuda") for _ in range(2)]
        k.v210_yadif_pair([(win[0], win[1], win[2], rgba[0], rgba[1])], w, h, tff, False, cm, lut, gm)
        k.v210_yadif_pair([(win[0], win[1], win[2], rgb[0], rgb[1])], w, h, tff, False, cm, lut, gm, rgb=True)
        for a, b in zip(rgba, rgb):
            a4 = hh.host(a).reshape(-1, 4)
            assert np.all(a4[:, 3] == 1.0)
            assert np.array_equal(a4[:, :3].view(np.uint32), hh.host(b).reshape(-1, 3).view(np.uint32))


def test_config3_route_at_full_size():
    """BASELINE config 3 as bench.py times it now: per frame ONE de-interlacing reader launch (packed RGB fields), per field
    ONE 2 x 2-block compositor launch - against the route of round 2 (RGBA fields, pixel-per-lane compositor), which
    tests/test_chains_gpu.py pins to the oracle at this size"""
    import torch
    import hip_harness as hh
    from phaneron_amd import capi
    k = hh.ctx()
    sw, sh, ow, oh = 1920, 1080, 3840, 2160
    cm, lut, gm = hh.ColourParams.reader("709", "2020")
    wcm, wlut = hh.ColourParams.writer("2020")
    mat_h = capi.transform_matrix(ow, oh)
    mat_d = hh.dev(mat_h)
    wins = [[hh.dev(frames.v210_random(sw, sh, frames.layer_seed(3, 4 * l + i))) for i in range(3)] for l in range(4)]
    rgba = [[torch.zeros(sw * sh * 4, dtype=torch.float32, device="cuda") for _ in range(2)] for _ in range(4)]
    rgb = [[torch.zeros(sw * sh * 3, dtype=torch.float32, device="cuda") for _ in range(2)] for _ in range(4)]
    k.v210_yadif_pair([(wins[l][0], wins[l][1], wins[l][2], rgba[l][0], rgba[l][1]) for l in range(4)], sw, sh, 1, False, cm, lut, gm)
    k.v210_yadif_pair([(wins[l][0], wins[l][1], wins[l][2], rgb[l][0], rgb[l][1]) for l in range(4)], sw, sh, 1, False, cm, lut, gm, rgb=True)
    words = frames.v210_pitch_bytes(ow) * oh // 4
    for parity in (0, 1):
        want = torch.zeros(words, dtype=torch.int32, device="cuda")
        got = torch.zeros(words, dtype=torch.int32, device="cuda")
        k.compose_write_v210([(rgba[l][parity], sw, sh, mat_d) for l in range(4)], want, ow, oh, 0, wcm, wlut)
        k.compose_up_write_v210([(rgb[l][parity], sw, sh, mat_h) for l in range(4)], got, ow, oh, 0, wcm, wlut, rgb=True)
        k.wait()
        torch.cuda.synchronize()
        assert torch.equal(got, want), "field of parity %d" % parity


def test_random_magnified_layers():
    """seeded random jobs: 1-5 layers of random sizes enlarged 1.1x-6x with random offsets (partly off screen included), both
    layouts, output sizes around the 126-column wave step and odd row counts"""
    r = np.random.default_rng(29092026)
    outs = [(48, 4), (96, 7), (144, 12), (240, 9), (288, 31), (384, 16), (768, 5)]
    for case in range(16):
        ow, oh = outs[case % len(outs)]
        rgb = bool(case & 1)
        layers = []
        for l in range(int(r.integers(1, 6))):
            fx, fy = float(r.choice([1.1, 1.5, 2.0, 2.0, 2.5, 3.0, 4.0, 6.0])), float(r.choice([1.2, 1.5, 2.0, 2.0, 2.5, 3.0, 4.0]))
            shown_w, shown_h = float(r.choice([1.0, 1.0, 0.5, 0.75])), float(r.choice([1.0, 1.0, 0.5]))
            sw, sh = max(int(ow * shown_w / fx), 1), max(int(oh * shown_h / fy), 1)
            img = opaque(sw, sh, int(r.integers(1, 1 << 30))) if rgb else frames.rgba_random(sw, sh, int(r.integers(1, 1 << 30)), -0.05, 1.05).reshape(sh, sw, 4)
            kw = dict(scale_x=shown_w, scale_y=shown_h, offset_x=float(r.uniform(-0.4, 0.4)) if shown_w < 1 or r.random() < 0.3 else 0.0,
                      offset_y=float(r.uniform(-0.4, 0.4)) if shown_h < 1 or r.random() < 0.3 else 0.0)
            mat = m(ow, oh, **kw)
            assert float(mat[0]) * sw <= 0.99 * ow and float(mat[4]) * sh <= 0.99 * oh
            layers.append((img, mat))
        if not layers:
            continue
        check(layers, ow, oh, "random magnified job %d: %dx%d, %d layers, %s" % (case, ow, oh, len(layers), "rgb" if rgb else "rgba"), rgb=rgb)


def test_8k_from_four_uhd_layers_equals_the_pixel_per_lane_compositor():
    """7680 x 4320 from four 2160p images (2x): the 2 x 2-block compositor against ph_compose_write_v210 (pixel per lane, pinned to
    the oracle's chain at small sizes), word for word, both image layouts"""
    import torch
    import hip_harness as hh
    ow, oh, sw, sh = 7680, 4320, 3840, 2160
    k = hh.ctx()
    wcm, wlut = hh.ColourParams.writer("2020")
    mats = [m(ow, oh), m(ow, oh, scale_x=0.75, scale_y=0.75, offset_x=0.1), m(ow, oh, scale_x=0.6, scale_y=0.55, offset_x=-0.25, offset_y=0.25), m(ow, oh)]
    gen = torch.Generator(device="cuda").manual_seed(99)
    imgs = [torch.rand(sw * sh * 4, device="cuda", generator=gen) * 1.1 - 0.05 for _ in range(4)]
    for im in imgs:
        im.view(-1, 4)[:, 3] = 1.0  # (the packed-RGB layout implies alpha 1)
    words = frames.v210_pitch_bytes(ow) * oh // 4
    want = torch.zeros(words, dtype=torch.int32, device="cuda")
    k.compose_write_v210([(im, sw, sh, hh.dev(np.asarray(mt, np.float32))) for im, mt in zip(imgs, mats)], want, ow, oh, 0, wcm, wlut)
    for rgb in (False, True):
        got = torch.zeros_like(want)
        data = [im.view(-1, 4)[:, :3].contiguous().view(-1) if rgb else im for im in imgs]
        k.compose_up_write_v210([(d, sw, sh, mt) for d, mt in zip(data, mats)], got, ow, oh, 0, wcm, wlut, rgb=rgb)
        k.wait()
        torch.cuda.synchronize()
        assert torch.equal(got, want), "packed RGB" if rgb else "RGBA"


@pytest.mark.parametrize("rgb", [False, True], ids=["rgba", "packed-rgb"])
@pytest.mark.parametrize("ow", [134, 136, 90, 1280, 1276, 1290])
def test_output_lines_with_tails(ow, rgb):
    """Round 5: output widths that are not a multiple of 48 (1280 x 720: src/config.ts:43-54) - whole quads, the tail quad of 2 or 4
    pixels with the reference's tail arithmetic (truncated table indices, round() and a truncating convert: v210.ts:166-193), the
    slots its writer clears up to the pitch (:131-136), lines addressed by pitch; the destination starts out poisoned"""
    oh = 14
    sw, sh = ow // 2 - 3, 6
    mk = (lambda w, h, s: opaque(w, h, s)) if rgb else (lambda w, h, s: frames.rgba_random(w, h, s, -0.05, 1.05).reshape(h, w, 4))
    layers = [(mk(sw, sh, 40), m(ow, oh)), (mk(max(sw // 3, 2), 3, 41), m(ow, oh, scale_x=0.5, scale_y=0.5, offset_x=0.26, offset_y=-0.1))]
    dst = np.full(frames.v210_pitch_bytes(ow) * oh // 4, 0x2AAAAAAA, np.uint32)
    check(layers, ow, oh, "%d wide" % ow, rgb=rgb, dst=dst)
    check(layers[:1], ow, oh, "%d wide, one layer" % ow, rgb=rgb, dst=dst)


@pytest.mark.parametrize("interlace", [1, 3])
def test_field_outputs_on_lines_with_tails(interlace):
    ow, oh = 1280, 36
    dst = np.full(frames.v210_pitch_bytes(ow) * oh // 4, 0x2AAAAAAA, np.uint32)
    layers = [(opaque(320, 8, 50), m(ow, oh)), (opaque(100, 4, 51), m(ow, oh, scale_x=0.5, scale_y=0.5, offset_x=0.2))]
    check(layers, ow, oh, "1280 wide, interlace %d" % interlace, interlace=interlace, rgb=True, dst=dst)


def test_720p_sources_enlarged_at_full_size():
    """1280 x 720 images enlarged 2x to 2560 x 1440 (lines that end in a tail quad of four pixels) and 3x to
    3840 x 2160, against the oracle's transform -> combine -> write; and two fields' frames of the 2560-wide shape in one launch"""
    import torch
    import hip_harness as hh
    sw, sh = 1280, 720
    imgs = [opaque(sw, sh, 60), opaque(sw, sh, 61)]
    for ow, oh in ((2560, 1440), (3840, 2160)):
        layers = [(imgs[0], m(ow, oh)), (imgs[1], m(ow, oh, scale_x=0.75, scale_y=0.75, offset_x=0.1, offset_y=-0.1))]
        check(layers, ow, oh, "1280x720 -> %dx%d" % (ow, oh), rgb=True)
    ow, oh = 2560, 1440
    k = hh.ctx()
    wcm, wlut = hh.ColourParams.writer("2020")
    mat = m(ow, oh)
    dev = [[(hh.dev(np.ascontiguousarray(img[..., :3]).reshape(-1)), sw, sh, mat)] for img in imgs]
    words = frames.v210_pitch_bytes(ow) * oh // 4
    single = [torch.zeros(words, dtype=torch.int32, device="cuda") for _ in range(2)]
    pair = [torch.zeros(words, dtype=torch.int32, device="cuda") for _ in range(2)]
    for f in range(2):
        k.compose_up_write_v210(dev[f], single[f], ow, oh, 0, wcm, wlut, rgb=True)
    k.compose_up_write_v210_pair(dev[0], dev[1], pair[0], pair[1], ow, oh, 0, wcm, wlut, rgb=True)
    k.wait()
    for f in range(2):
        assert torch.equal(single[f], pair[f]), "set %d: the pair launch differs from the single one" % f


@pytest.mark.parametrize("rgb", [False, True], ids=["rgba", "packed-rgb"])
@pytest.mark.parametrize("shape", [(192, 54, 384, 108, 2, 0), (96, 24, 288, 124, 1, 3), (128, 36, 1280, 72, 3, 0), (1280, 720, 1920, 1080, 1, 0)])
def test_several_channels_in_one_launch_equal_their_own_launches(shape, rgb):
    """ph_compose_up_write_v210_batch with 3 and 4 sets of layers of one shape == ph_compose_up_write_v210 per set (and, at the small sizes,
    == the oracle's chain for each set); more than four sets, a set placed differently and a shared output are refused"""
    import hip_harness as hh
    sw, sh, ow, oh, n, interlace = shape
    k = hh.ctx()
    wcm, wlut = hh.ColourParams.writer("709")
    mat = m(ow, oh, scale_x=0.9, scale_y=0.9, offset_x=0.02)
    words = frames.v210_pitch_bytes(ow) * oh // 4
    fill = np.random.default_rng(sw + n).integers(0, 2 ** 30, words, dtype=np.int64).astype(np.uint32)
    sets = [[opaque(sw, sh, 500 + 10 * f + l) if rgb else frames.rgba_random(sw, sh, 500 + 10 * f + l, -0.05, 1.05).reshape(sh, sw, 4) for l in range(n)] for f in range(4)]
    dev = [[(hh.dev((np.ascontiguousarray(img[..., :3]) if rgb else img).reshape(-1)), sw, sh, mat) for img in imgs] for imgs in sets]
    single = [hh.dev(fill.copy()) for _ in range(4)]
    for f in range(4):
        k.compose_up_write_v210(dev[f], single[f], ow, oh, interlace, wcm, wlut, rgb=rgb)
    for jobs in (3, 4):
        outs = [hh.dev(fill.copy()) for _ in range(jobs)]
        k.compose_up_write_v210_batch(dev[:jobs], outs, ow, oh, interlace, wcm, wlut, rgb=rgb)
        k.wait()
        for f in range(jobs):
            a, b = hh.host(single[f], np.uint32), hh.host(outs[f], np.uint32)
            bad = np.flatnonzero(a != b)
            assert bad.size == 0, "%d jobs, set %d: %d words differ from its own launch, first at %d" % (jobs, f, bad.size, bad[0])
    if ow <= 384:
        wr_o = (orc.rgb2ycbcr_matrix("709"), orc.linear2gamma_lut("709"))
        for f in range(4):
            placed = [orc.transform(img, mat, ow, oh) for img in sets[f]]
            want = orc.v210_write(placed[0] if n == 1 else orc.combine(placed), ow, oh, interlace, *wr_o, out=fill.copy())
            assert np.array_equal(hh.host(single[f], np.uint32), np.asarray(want).reshape(-1)), "set %d differs from the oracle" % f
    outs = [hh.dev(fill.copy()) for _ in range(5)]
    with pytest.raises(Exception, match="jobs"):
        k.compose_up_write_v210_batch(dev + dev[:1], outs, ow, oh, interlace, wcm, wlut, rgb=rgb)
    with pytest.raises(Exception, match="same output"):
        k.compose_up_write_v210_batch(dev[:3], [outs[0], outs[1], outs[0]], ow, oh, interlace, wcm, wlut, rgb=rgb)
    with pytest.raises(Exception, match="placed differently"):
        other = [(t, w, h, m(ow, oh, scale_x=0.9, scale_y=0.9)) for t, w, h, _ in dev[2]]
        k.compose_up_write_v210_batch([dev[0], dev[1], other], outs[:3], ow, oh, interlace, wcm, wlut, rgb=rgb)
