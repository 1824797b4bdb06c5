"""BASELINE.json configs 2 and 3 as whole chains, at full size, through the C ABI against the oracle - and the
headline kernel with its lower layers made visible.

  config 3  yadif.ts:88-145 (window, parity), producer/mixer.ts:209-223 (transform per layer),
            combiner.ts:219-254 (combine_N), io.ts:152-164 + v210.ts:113-195 (write):
            4 x 1920x1080 tff v210 -> read(709->2020) -> Yadif window of 3 frames, send_field ->
            transform to 3840x2160 -> combine_4 -> write('2020')
  config 2  4-layer 1080p50: full-frame ramp + three PiP layers (scale 0.5, offsets +-0.25), the top one
            inside a transition_wipe against a second source with a horizontal-ramp mask
            (transitioner.ts:165-176, transition.ts:66-74), combine_4, write
  headline  v210 readers always emit alpha = 1 (v210.ts:76), so `combine` keeps only the top layer and a
            wrong lower-layer pointer could not change a single output bit.  A reader LUT whose entry 0 is
            +Inf makes every lower-layer pixel that clamps to index 0 poison the accumulator
            (fma(Inf, 0, t) = NaN): then all N layers of every job show in the output.
Bit-exact everywhere.  Every route the library offers for a chain is checked: the reference-shaped one
(one kernel per operator) and the fused ones.
"""
import numpy as np
import pytest

import frames
from oracle import orc
from phaneron_amd import capi

pytestmark = pytest.mark.gpu


def _bits_equal(got, want, what):
    g = np.ascontiguousarray(got).reshape(-1).view(np.uint32)
    w = np.ascontiguousarray(want).reshape(-1).view(np.uint32)
    assert g.shape == w.shape, (what, g.shape, w.shape)
    bad = np.flatnonzero(g != w)
    assert bad.size == 0, "%s: %d of %d words differ, first at %d: %08x vs %08x" % (what, bad.size, w.size, bad[0], g[bad[0]], w[bad[0]])


def _img(w, h):
    import torch
    return torch.empty(w * h * 4, dtype=torch.float32, device="cuda")


def _v210_out(w, h):
    import torch
    return torch.full((frames.v210_pitch_bytes(w) * h // 4,), 0x2AAAAAAA, dtype=torch.int32, device="cuda")


# ---------------------------------------------------------------------------------------------------
# config 3
# ---------------------------------------------------------------------------------------------------
SW, SH, OW, OH = 1920, 1080, 3840, 2160


def _config3_matrices(pip):
    """producer/mixer.ts:209-223 derives the transform parameters from the layer's fill; identity fill is the
    plain 2x upscale.  pip: layers 1..3 are quarter-size insets, so every layer shows in the output."""
    if not pip:
        return [dict()] * 4
    return [dict()] + [dict(scale_x=0.5, scale_y=0.5, offset_x=ox, offset_y=oy)
                       for ox, oy in ((-0.25, -0.25), (0.25, -0.25), (0.25, 0.25))]


@pytest.fixture(scope="module")
def config3_sources():
    """Three consecutive 1080i frames per layer (prev / cur / next of the Yadif window, yadif.ts:88-113), the
    oracle's linear-2020 RGBA of each, and the same on the device."""
    import hip_harness as hh
    from phaneron_amd import capi
    rd_o = (orc.ycbcr2rgb_matrix("709"), orc.gamma2linear_lut("709"), orc.rgb2rgb_matrix("709", "2020"))
    cm, lut, gm = hh.ColourParams.reader("709", "2020")
    k = hh.ctx()
    words, rgba_o, rgba_d = [], [], []
    for layer in range(4):
        fw = [frames.v210_ramp(SW, SH) if (layer == 0 and t == 1) else frames.v210_random(SW, SH, frames.layer_seed(3, layer) + 97 * t)
              for t in range(3)]
        words.append(fw)
        rgba_o.append([orc.v210_read(f, SW, SH, *rd_o) for f in fw])
        dev_rgba = []
        for f in fw:
            out = _img(SW, SH)
            k.v210_read(hh.dev(f), out, SW, SH, cm, lut, gm)
            dev_rgba.append(out)
        rgba_d.append(dev_rgba)
    for layer in range(4):
        for t in range(3):
            _bits_equal(hh.host(rgba_d[layer][t]), rgba_o[layer][t], "read layer %d frame %d" % (layer, t))
    return words, rgba_o, rgba_d


@pytest.mark.parametrize("second_field,pip", [(False, False), (True, True)])
def test_config3_chain_one_field_full_size(config3_sources, second_field, pip):
    import hip_harness as hh
    from phaneron_amd import capi
    words, rgba_o, rgba_d = config3_sources
    tff = 1
    parity = tff ^ (0 if second_field else 1)  # yadif.ts:104: parity = tff ^ !isSecond
    mats = [capi.transform_matrix(OW, OH, **kw) for kw in _config3_matrices(pip)]
    mats_o = [orc.transform_matrix(OW, OH, **kw) for kw in _config3_matrices(pip)]
    for a, b in zip(mats, mats_o):
        _bits_equal(a, b, "transform matrix")
    # ---- oracle: operator by operator, as the reference's job queue runs them
    deint_o = [orc.yadif(rgba_o[l][0], rgba_o[l][1], rgba_o[l][2], parity, tff, False) for l in range(4)]
    up_o = [orc.transform(deint_o[l], mats_o[l], OW, OH) for l in range(4)]
    want = orc.v210_write(orc.combine(up_o), OW, OH, 0, orc.rgb2ycbcr_matrix("2020"), orc.linear2gamma_lut("2020"))
    # ---- device, reference-shaped route: yadif x4, transform x4, combine_4, write
    k = hh.ctx()
    wcm, wlut = hh.ColourParams.writer("2020")
    dmats = [hh.dev(m) for m in mats]
    deint = [_img(SW, SH) for _ in range(4)]
    up = [_img(OW, OH) for _ in range(4)]
    for l in range(4):
        k.yadif(rgba_d[l][0], rgba_d[l][1], rgba_d[l][2], deint[l], SW, SH, parity, tff, False)
        k.transform(deint[l], SW, SH, dmats[l], up[l], OW, OH)
    comb = _img(OW, OH)
    k.combine(up, comb, OW, OH)
    out = _v210_out(OW, OH)
    k.v210_write(comb, out, OW, OH, 0, wcm, wlut)
    for l in range(4):
        _bits_equal(hh.host(deint[l]), deint_o[l], "yadif layer %d" % l)
    _bits_equal(hh.host(out, np.uint32), want, "config 3, one kernel per operator")
    # ---- device, fused compositor: yadif x4, then [transform x4 -> combine_4 -> write] as one kernel
    out2 = _v210_out(OW, OH)
    k.compose_write_v210([(deint[l], SW, SH, dmats[l]) for l in range(4)], out2, OW, OH, 0, wcm, wlut)
    _bits_equal(hh.host(out2, np.uint32), want, "config 3, fused compositor")


@pytest.mark.parametrize("tff", [1, 0])
def test_config3_deinterlacing_reader_full_size(config3_sources, tff):
    """ph_v210_yadif_pair on config 3's four full-size windows in ONE launch (the strip height the launcher picks for
    4 x 1080 rows, every column block, both outputs) against the oracle's read -> yadif for both parities - and the
    whole best route of config 3: its outputs through the fused compositor equal the oracle chain's v210 frame.
    Then config 3 exactly as bench.py times it (packed-RGB fields, both fields' compositors in one launch) against the
    same oracle frames (yadif.ts:88-145, mixer.ts:209-223, combiner.ts:219-254, v210.ts:113-195)."""
    import hip_harness as hh
    from phaneron_amd import capi
    words, rgba_o, rgba_d = config3_sources
    k = hh.ctx()
    cm, lut, gm = hh.ColourParams.reader("709", "2020")
    outs = [[_img(SW, SH), _img(SW, SH)] for _ in range(4)]
    dwords = [[hh.dev(f) for f in words[l]] for l in range(4)]
    k.v210_yadif_pair([(dwords[l][0], dwords[l][1], dwords[l][2], outs[l][0], outs[l][1]) for l in range(4)], SW, SH, tff, False, cm, lut, gm)
    deint_o = [[orc.yadif(rgba_o[l][0], rgba_o[l][1], rgba_o[l][2], parity, tff, False) for parity in (0, 1)] for l in range(4)]
    for l in range(4):
        for parity in (0, 1):
            _bits_equal(hh.host(outs[l][parity]), deint_o[l][parity], "deinterlacing reader layer %d parity %d tff %d" % (l, parity, tff))
    # the route bench.py reports for config 3: one compositor launch per field on those outputs (one shared placement)
    m = capi.transform_matrix(OW, OH)
    dm = hh.dev(m)
    wcm, wlut = hh.ColourParams.writer("2020")
    wants = []
    for parity in (0, 1):
        out = _v210_out(OW, OH)
        k.compose_write_v210([(outs[l][parity], SW, SH, dm) for l in range(4)], out, OW, OH, 0, wcm, wlut)
        up_o = [orc.transform(deint_o[l][parity], orc.transform_matrix(OW, OH), OW, OH) for l in range(4)]
        want = orc.v210_write(orc.combine(up_o), OW, OH, 0, orc.rgb2ycbcr_matrix("2020"), orc.linear2gamma_lut("2020"))
        _bits_equal(hh.host(out, np.uint32), want, "config 3 best route, parity %d tff %d" % (parity, tff))
        wants.append(want)
    # ---- the route bench.py times NOW (`secondary`, config 3), against the ORACLE at full size: the de-interlacing reader
    # writing packed-RGB fields (ph_v210_yadif_pair_fmt, PH_IMG_RGB_F32), then both fields' 2 x 2-block compositors as ONE
    # launch (ph_compose_up_write_v210_pair) - and a launch per field (ph_compose_up_write_v210)
    import torch
    rgb = [[torch.zeros(SW * SH * 3, dtype=torch.float32, device="cuda") for _ in range(2)] for _ in range(4)]
    k.v210_yadif_pair([(dwords[l][0], dwords[l][1], dwords[l][2], rgb[l][0], rgb[l][1]) for l in range(4)], SW, SH, tff, False, cm, lut, gm, rgb=True)
    for l in range(4):
        for parity in (0, 1):
            _bits_equal(hh.host(rgb[l][parity]).reshape(-1, 3), np.ascontiguousarray(deint_o[l][parity].reshape(-1, 4)[:, :3]),
                        "packed-RGB field, layer %d parity %d tff %d" % (l, parity, tff))
    pair = [_v210_out(OW, OH), _v210_out(OW, OH)]
    k.compose_up_write_v210_pair([(rgb[l][0], SW, SH, m) for l in range(4)], [(rgb[l][1], SW, SH, m) for l in range(4)], pair[0], pair[1],
                                 OW, OH, 0, wcm, wlut, rgb=True)
    for parity in (0, 1):
        _bits_equal(hh.host(pair[parity], np.uint32), wants[parity], "config 3 as benched (pair launch), parity %d tff %d" % (parity, tff))
        single = _v210_out(OW, OH)
        k.compose_up_write_v210([(rgb[l][parity], SW, SH, m) for l in range(4)], single, OW, OH, 0, wcm, wlut, rgb=True)
        _bits_equal(hh.host(single, np.uint32), wants[parity], "config 3 as benched (launch per field), parity %d tff %d" % (parity, tff))


# ---------------------------------------------------------------------------------------------------
# config 2
# ---------------------------------------------------------------------------------------------------
def test_config2_chain_with_wipe_full_size():
    import hip_harness as hh
    from phaneron_amd import capi
    w, h = 1920, 1080
    rd_o = (orc.ycbcr2rgb_matrix("709"), orc.gamma2linear_lut("709"), orc.rgb2rgb_matrix("709", "709"))
    wr_o = (orc.rgb2ycbcr_matrix("709"), orc.linear2gamma_lut("709"))
    srcs = [frames.v210_ramp(w, h)] + [frames.v210_random(w, h, frames.layer_seed(2, l)) for l in (1, 2, 3)]
    second = frames.v210_random(w, h, frames.layer_seed(2, 4), legal=False)
    mask = frames.mask_ramp(w, h)
    kws = [dict()] + [dict(scale_x=0.5, scale_y=0.5, offset_x=ox, offset_y=oy)
                      for ox, oy in ((-0.25, -0.25), (0.25, -0.25), (0.25, 0.25))]
    # ---- oracle (transitioner.ts:165-176: the wipe runs between the layer's transformed frame and the second source)
    rgba_o = [orc.v210_read(s, w, h, *rd_o) for s in srcs]
    second_o = orc.v210_read(second, w, h, *rd_o)
    xf_o = [orc.transform(rgba_o[l], orc.transform_matrix(w, h, **kws[l]), w, h) for l in range(4)]
    trans_o = orc.transition_wipe(xf_o[3], second_o, mask)
    want = orc.v210_write(orc.combine([xf_o[0], xf_o[1], xf_o[2], trans_o]), w, h, 0, *wr_o)
    # ---- device, one kernel per operator (13 kernels)
    k = hh.ctx()
    cm, lut, gm = hh.ColourParams.reader("709", "709")
    wcm, wlut = hh.ColourParams.writer("709")
    dmats = [hh.dev(capi.transform_matrix(w, h, **kw)) for kw in kws]
    rgba = [_img(w, h) for _ in range(5)]
    for s, r in zip(srcs + [second], rgba):
        k.v210_read(hh.dev(s), r, w, h, cm, lut, gm)
    xf = [_img(w, h) for _ in range(4)]
    for l in range(4):
        k.transform(rgba[l], w, h, dmats[l], xf[l], w, h)
    trans, comb = _img(w, h), _img(w, h)
    dmask = hh.dev(mask)
    k.transition_wipe(xf[3], rgba[4], dmask, trans, w, h)
    _bits_equal(hh.host(trans), trans_o, "transition_wipe 1920x1080")
    k.combine([xf[0], xf[1], xf[2], trans], comb, w, h)
    out = _v210_out(w, h)
    k.v210_write(comb, out, w, h, 0, wcm, wlut)
    _bits_equal(hh.host(out, np.uint32), want, "config 2, one kernel per operator")
    # ---- device, fused compositor: the transitioned layer enters 1:1, the other three are sampled in the kernel
    out2 = _v210_out(w, h)
    k.compose_write_v210([(rgba[0], w, h, dmats[0]), (rgba[1], w, h, dmats[1]), (rgba[2], w, h, dmats[2]),
                          (trans, w, h, None)], out2, w, h, 0, wcm, wlut)
    _bits_equal(hh.host(out2, np.uint32), want, "config 2, fused compositor")
    # ---- device, the wipe inside the compositor: transform + transition_wipe + combine + write as one kernel
    out3 = _v210_out(w, h)
    k.compose_wipe_write_v210([(rgba[l], w, h, dmats[l]) for l in range(4)], [None, None, None, (rgba[4], dmask)],
                              out3, w, h, 0, wcm, wlut)
    _bits_equal(hh.host(out3, np.uint32), want, "config 2, compositor with the wipe inside")
    with pytest.raises(Exception, match="both the incoming image and the mask"):
        k.compose_wipe_write_v210([(rgba[0], w, h, dmats[0])], [(rgba[4], None)], out3, w, h, 0, wcm, wlut)
    # the PiP geometry really exposes every layer: each one alone changes the result
    for drop in range(4):
        layers = [xf_o[0], xf_o[1], xf_o[2], trans_o]
        layers[drop] = np.zeros_like(layers[drop])
        other = orc.v210_write(orc.combine(layers), w, h, 0, *wr_o)
        assert not np.array_equal(other, want), "layer %d does not show in the composite" % drop


# ---------------------------------------------------------------------------------------------------
# headline kernel: make layers 0..N-2 visible
# ---------------------------------------------------------------------------------------------------
def _poison_lut():
    lut = orc.gamma2linear_lut("709").copy()
    lut[0] = np.float32(np.inf)  # entry 0 is a block of its own in the LDS form (ph_lut.h), so it stays exact
    return lut


@pytest.mark.parametrize("n", [2, 4, 8])
@pytest.mark.parametrize("lds", [True, False])
def test_fused_headline_sees_every_layer(n, lds):
    """ph_fused_v210_combine with a reader table whose entry 0 is +Inf: a pixel of ANY layer whose R, G or B
    clamps to index 0 turns the accumulator into Inf, the next layer's fma(acc, 0, t) into NaN, and the
    writer maps NaN to index 0.  The oracle runs the same chain with the same table."""
    import torch
    import hip_harness as hh
    from phaneron_amd import capi
    w, h = 1920, 32
    k = hh.ctx()
    k.set_option("lds_lut", lds)
    try:
        lut = _poison_lut()
        dlut = hh.dev(lut)
        assert k.register_lut(dlut, lut) == 1, "the poisoned table must keep its LDS form"
        cm, gm = hh.dev(capi.ycbcr2rgb_matrix("709")), hh.dev(np.concatenate([capi.rgb2rgb_matrix("709", "2020"), np.zeros(3, np.float32)]))
        wcm, wlut = hh.ColourParams.writer("2020")
        layers = [frames.v210_random(w, h, frames.layer_seed(5, i), legal=False) for i in range(n)]
        out = _v210_out(w, h)
        k.fused_v210_combine([hh.dev(l) for l in layers], out, w, h, cm, dlut, gm, wcm, wlut)
        got = hh.host(out, np.uint32)
        rd_o = (orc.ycbcr2rgb_matrix("709"), lut, orc.rgb2rgb_matrix("709", "2020"))
        wr_o = (orc.rgb2ycbcr_matrix("2020"), orc.linear2gamma_lut("2020"))
        want = orc.pipeline_v210_combine(layers, w, h, *rd_o, *wr_o)
        _bits_equal(got, want, "fused %d layers, poisoned reader table" % n)
        # and the poison really reaches the output from every layer: swapping any one lower layer for another
        # frame changes the oracle's result
        for l in range(n - 1):
            alt = list(layers)
            alt[l] = frames.v210_random(w, h, 0xABC0 + l, legal=False)
            assert not np.array_equal(orc.pipeline_v210_combine(alt, w, h, *rd_o, *wr_o), want), l
        k.unregister_lut(dlut)
        del torch
    finally:
        k.set_option("lds_lut", True)


@pytest.mark.parametrize("jobs,n", [(2, 4), (3, 3), (8, 2)])
def test_fused_batch_sees_every_layer_of_every_job(jobs, n):
    """The same for ph_fused_v210_combine_batch: a wrong more_layers[job - 1][l] or output pointer for any job
    or layer would show."""
    import hip_harness as hh
    from phaneron_amd import capi
    w, h = 1920, 24
    k = hh.ctx()
    lut = _poison_lut()
    dlut = hh.dev(lut)
    assert k.register_lut(dlut, lut) == 1
    cm, gm = hh.dev(capi.ycbcr2rgb_matrix("709")), hh.dev(np.concatenate([capi.rgb2rgb_matrix("709", "2020"), np.zeros(3, np.float32)]))
    wcm, wlut = hh.ColourParams.writer("2020")
    rd_o = (orc.ycbcr2rgb_matrix("709"), lut, orc.rgb2rgb_matrix("709", "2020"))
    wr_o = (orc.rgb2ycbcr_matrix("2020"), orc.linear2gamma_lut("2020"))
    layers = [[frames.v210_random(w, h, frames.layer_seed(6 + j, i), legal=False) for i in range(n)] for j in range(jobs)]
    outs = [_v210_out(w, h) for _ in range(jobs)]
    k.fused_v210_combine_batch([[hh.dev(l) for l in job] for job in layers], outs, w, h, cm, dlut, gm, wcm, wlut)
    for j in range(jobs):
        _bits_equal(hh.host(outs[j], np.uint32), orc.pipeline_v210_combine(layers[j], w, h, *rd_o, *wr_o), "job %d" % j)
    k.unregister_lut(dlut)


# ---------------------------------------------------------------------------------------------------
# 720p50: the reference's third video format (src/config.ts:43-54).  1280 % 48 = 32 and 1280 % 6 = 2: every line ends
# in a tail quad and two cleared slots, so every operator's tail path is on the chain (v210.ts:84-110, 166-193)
# ---------------------------------------------------------------------------------------------------
HW, HH = 1280, 720


@pytest.fixture(scope="module")
def hd720_chain():
    """four 1280 x 720 v210 sources, the Mixer placements of config 2 (a full-frame layer through the identity fill and three
    quarter-size insets, producer/mixer.ts:209-223), and the oracle's chain read -> transform -> combine_4 (combiner.ts:219-254)"""
    rd_o = (orc.ycbcr2rgb_matrix("709"), orc.gamma2linear_lut("709"), orc.rgb2rgb_matrix("709", "709"))
    words = [frames.v210_random(HW, HH, frames.layer_seed(7, l), legal=(l != 2)) for l in range(4)]
    kws = [dict()] + [dict(scale_x=0.5, scale_y=0.5, offset_x=ox, offset_y=oy) for ox, oy in ((-0.25, -0.25), (0.25, -0.25), (0.25, 0.25))]
    mats = [capi.transform_matrix(HW, HH, **kw) for kw in kws]
    for kw, mat in zip(kws, mats):
        _bits_equal(mat, orc.transform_matrix(HW, HH, **kw), "transform matrix")
    rgba = [orc.v210_read(f, HW, HH, *rd_o) for f in words]
    placed = [orc.transform(rgba[l], mats[l], HW, HH) for l in range(4)]
    return words, mats, rgba, placed, orc.combine(placed)


@pytest.mark.parametrize("interlace", [0, 1, 3])
def test_720p_chain_full_size(hd720_chain, interlace):
    """read (tail) -> transform -> combine_4 -> write (tail, progressive and both fields) at 1280 x 720 against the oracle: one kernel
    per operator as the reference's job queue posts them, and the same frame from the channel compositor in ONE launch"""
    import hip_harness as hh
    words, mats, rgba_o, placed_o, comb_o = hd720_chain
    wr_o = (orc.rgb2ycbcr_matrix("709"), orc.linear2gamma_lut("709"))
    before = np.full(frames.v210_pitch_bytes(HW) * HH // 4, 0x2AAAAAAA, np.uint32)
    want = np.asarray(orc.v210_write(comb_o, HW, HH, interlace, *wr_o, out=before.copy())).reshape(-1)
    k = hh.ctx()
    cm, lut, gm = hh.ColourParams.reader("709", "709")
    wcm, wlut = hh.ColourParams.writer("709")
    dwords = [hh.dev(f) for f in words]
    # ---- one kernel per operator
    up = []
    for l in range(4):
        img = _img(HW, HH)
        k.v210_read(dwords[l], img, HW, HH, cm, lut, gm)
        if interlace == 0:
            _bits_equal(hh.host(img), rgba_o[l], "720p read, layer %d" % l)
        placed = _img(HW, HH)
        k.transform(img, HW, HH, hh.dev(mats[l]), placed, HW, HH)
        up.append(placed)
    comb = _img(HW, HH)
    k.combine(up, comb, HW, HH)
    out = hh.dev(before.copy())
    k.v210_write(comb, out, HW, HH, interlace, wcm, wlut)
    _bits_equal(hh.host(out, np.uint32), want, "720p, one kernel per operator, interlace %d" % interlace)
    # ---- the channel compositor straight from the v210 words
    out2 = hh.dev(before.copy())
    k.chan_compose_v210([dict(src=(dwords[l], HW, HH, mats[l])) for l in range(4)], out2, HW, HH, interlace, cm, lut, gm, wcm, wlut)
    _bits_equal(hh.host(out2, np.uint32), want, "720p, channel compositor, interlace %d" % interlace)


def test_720p_headline_shape_full_size(hd720_chain):
    """four 1:1 layers -> combine_4 -> write at 1280 x 720: the fused kernel and the channel compositor against the oracle chain,
    with the lower layers made visible (entry 0 of the reader table is +Inf)"""
    import hip_harness as hh
    words = hd720_chain[0]
    lut = orc.gamma2linear_lut("709").copy()
    lut[0] = np.inf
    cm_o, gm_o = orc.ycbcr2rgb_matrix("709"), orc.rgb2rgb_matrix("709", "2020")
    want = orc.pipeline_v210_combine(words, HW, HH, cm_o, lut, gm_o, orc.rgb2ycbcr_matrix("2020"), orc.linear2gamma_lut("2020"))
    k = hh.ctx()
    dlut = hh.dev(lut)
    assert k.register_lut(dlut, lut)
    cm, _, gm = hh.ColourParams.reader("709", "2020")
    wcm, wlut = hh.ColourParams.writer("2020")
    dwords = [hh.dev(f) for f in words]
    out = _v210_out(HW, HH)
    k.fused_v210_combine(dwords, out, HW, HH, cm, dlut, gm, wcm, wlut)
    _bits_equal(hh.host(out, np.uint32), want, "720p fused kernel")
    out2 = _v210_out(HW, HH)
    k.chan_compose_v210([dict(src=(d, HW, HH, None)) for d in dwords], out2, HW, HH, 0, cm, dlut, gm, wcm, wlut)
    _bits_equal(hh.host(out2, np.uint32), want, "720p channel compositor, 1:1 layers")
    k.unregister_lut(dlut)
