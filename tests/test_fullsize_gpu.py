"""GPU tests (-m gpu): the routes bench.py's `secondary` records time, and the channel kernel's instantiations, at FULL SIZE
and exactly as benched (tools/config_bench.py: the same entry points, shapes and placements) against the oracle's chain of the
reference's operators.  The one class of bug this code base has shipped lived only at full-size shares (DESIGN.md section 4,
step 11: waves recomputing a share's last quad), so every figure in the bench line names the test here that pins its bytes
(`parity_test` in the record).  Reference: ffmpegProducer.ts:395-442 (a decoder's frame at the clip's own size), mixer.ts:189-228
(the default fill), yadif.ts:88-145, combiner.ts:219-254, v210.ts:113-195."""
import numpy as np
import pytest

import frames
from oracle import orc
from test_chan_gpu import Src, both_routes, check, check_batch, colour, m, oracle_chain, PIP

pytestmark = pytest.mark.gpu

W, H = 1920, 1080


def route_of(layers, ow, oh, interlace=0):
    """the kernels ph_chan_compose_v210 picks for this frame (a dry run: nothing is launched)"""
    import torch
    import hip_harness as hh
    from phaneron_amd import capi
    from test_chan_gpu import device_layers
    _, _, rd_d, wr_d = colour("709", "709")
    out = torch.zeros(frames.v210_pitch_bytes(ow) * oh // 4, dtype=torch.int32, device="cuda")
    with capi.trace(dry_run=True) as t:
        hh.ctx().chan_compose_v210(device_layers(layers), out, ow, oh, interlace, *rd_d, *wr_d)
    return t.route


# ---- bench.py secondary f1 / f2: file playback ----------------------------------------------------------------------------------------
@pytest.mark.parametrize("cw,ch,name", [(1920, 1080, "f1"), (1280, 720, "f2")])
def test_file_playback_as_benched(cw, ch, name):
    """f1: a 1080p yuv420p clip under the default fill on a 1080p50 channel; f2: a 720p yuv420p clip filling it - ph_chan_compose_v210 on the
    decoder's planes with its own 8-bit Loader matrix, by the route the bench times (reader of the format + 2 x 2-block compositor in ONE launch),
    by the same two as two launches (option chan_enlarged = 2) and by the channel kernel (chan_enlarged = 0)"""
    clip = frames.pack_random("yuv420p", cw, ch, 1300 + cw)
    layers = [dict(src=Src(clip, cw, ch, m(W, H), fmt="yuv420p"))]
    assert route_of(layers, W, H) == "clip_up_write_v210<rgb>"
    both_routes(lambda route: check(layers, W, H, "%s: %dx%d yuv420p on %dx%d by the %s" % (name, cw, ch, W, H, route)))


# ---- bench.py secondary f3: 4 x 1080i50 -> yadif -> own size -> combine_4 -> 1080p50 ---------------------------------------------------
@pytest.mark.parametrize("tff", [1, 0])
def test_interlaced_sources_on_a_1080p_channel_as_benched(tff):
    """f3: the de-interlacing reader writing packed-RGB fields of four 1080i windows in ONE launch (ph_v210_yadif_pair_fmt), then both
    fields' frames from the 2 x 2-block compositor under the default fill in ONE launch (ph_compose_up_write_v210_pair) - and a launch
    per field; every output field against read -> yadif -> transform (default fill) -> combine_4 -> write of the oracle"""
    import torch
    import hip_harness as hh
    from phaneron_amd import capi
    rd_o, wr_o, rd_d, wr_d = colour("709", "709")
    k = hh.ctx()
    words = [[frames.v210_random(W, H, frames.layer_seed(1310 + l, t), legal=(l != 1)) for t in range(3)] for l in range(4)]
    dwords = [[hh.dev(f.reshape(-1)) for f in words[l]] for l in range(4)]
    rgb = [[torch.zeros(W * H * 3, dtype=torch.float32, device="cuda") for _ in range(2)] for _ in range(4)]
    k.v210_yadif_pair([(dwords[l][0], dwords[l][1], dwords[l][2], rgb[l][0], rgb[l][1]) for l in range(4)], W, H, tff, False, *rd_d, rgb=True)
    fill = capi.transform_matrix(W, H)
    pair = [torch.zeros(frames.v210_pitch_bytes(W) * H // 4, dtype=torch.int32, device="cuda") for _ in range(2)]
    k.compose_up_write_v210_pair([(rgb[l][0], W, H, fill) for l in range(4)], [(rgb[l][1], W, H, fill) for l in range(4)], pair[0], pair[1], W, H, 0, *wr_d, rgb=True)
    rgba_o = [[orc.v210_read(f, W, H, *rd_o) for f in words[l]] for l in range(4)]
    for parity in (0, 1):
        deint = [orc.yadif(rgba_o[l][0], rgba_o[l][1], rgba_o[l][2], parity, tff, False) for l in range(4)]
        for l in range(4):
            got = hh.host(rgb[l][parity]).reshape(-1, 3)
            assert np.array_equal(got.view(np.uint32), np.ascontiguousarray(deint[l].reshape(-1, 4)[:, :3]).view(np.uint32)), "field image layer %d parity %d" % (l, parity)
        placed = [orc.transform(deint[l], orc.transform_matrix(W, H), W, H) for l in range(4)]
        want = np.asarray(orc.v210_write(orc.combine(placed), W, H, 0, *wr_o)).reshape(-1)
        got = hh.host(pair[parity], np.uint32)
        assert np.array_equal(got, want), "f3 as benched (pair launch), parity %d tff %d: %d words differ" % (parity, tff, np.count_nonzero(got != want))
        single = torch.zeros_like(pair[0])
        k.compose_up_write_v210([(rgb[l][parity], W, H, fill) for l in range(4)], single, W, H, 0, *wr_d, rgb=True)
        assert np.array_equal(hh.host(single, np.uint32), want), "f3, a launch per field, parity %d tff %d" % (parity, tff)


# ---- the channel kernel's round-5 instantiations at 1920 x 1080 ------------------------------------------------------------------------
def test_planar_clips_with_shared_taps_full_size():
    """<4,0>: two 1080p yuv420p clips at their own scale, the upper one moved by a fraction of a pixel (nothing the compositor route
    takes: the channel kernel's tap-sharing instantiation), progressive and a field"""
    a, b = frames.pack_random("yuv420p", W, H, 1320), frames.pack_random("yuv420p", W, H, 1321)
    layers = [dict(src=Src(a, W, H, m(W, H), fmt="yuv420p")), dict(src=Src(b, W, H, m(W, H, offset_x=0.3 / W + 0.25, offset_y=0.4 / H), fmt="yuv420p"))]
    assert route_of(layers, W, H) == "chan_compose_v210<4,0>"
    check(layers, W, H, "two 1080p yuv420p clips, shared taps, full size", specs=("709", "2020"))
    check(layers, W, H, "the same, field 3", interlace=3, poison_dst=True)


def test_planar_clips_only_full_size():
    """<3,0>: a rotated yuv422p10 clip, a yuv422p10 inset and an f32 image - planar clips and images only, none at its own scale"""
    a, b = frames.pack_random("yuv422p10", W, H, 1330), frames.pack_random("yuv422p10", 960, 540, 1331)
    img = frames.rgba_random(W, H, 1332, -0.05, 1.05)
    layers = [dict(src=Src(a, W, H, m(W, H, scale_x=0.8, scale_y=0.8, rotate=0.05), fmt="yuv422p10")), dict(src=Src(b, 960, 540, m(W, H, **PIP[1]), fmt="yuv422p10")),
              dict(src=Src(img, W, H, m(W, H, rotate=0.01), fmt="rgba"))]
    assert route_of(layers, W, H) == "chan_compose_v210<3,0>"
    check(layers, W, H, "planar clips only, full size")


def test_planar_clip_with_a_v210_inset_full_size():
    """<2,0>: a 1080p yuv422p10 clip under the default fill with a v210 inset (the everything instantiation)"""
    a = frames.pack_random("yuv422p10", W, H, 1340)
    v = frames.v210_random(960, 540, frames.layer_seed(1341, 0))
    layers = [dict(src=Src(a, W, H, m(W, H), fmt="yuv422p10")), dict(src=Src(v, 960, 540, m(W, H, **PIP[2])))]
    assert route_of(layers, W, H) == "chan_compose_v210<2,0>"
    check(layers, W, H, "a yuv422p10 clip with a v210 inset, full size", specs=("709", "2020"))


@pytest.mark.parametrize("fmt", ["bgra8", "rgba8"])
def test_graphic_over_a_v210_clip_full_size(fmt):
    """<5,0>: a full-frame graphic with alpha over a live v210 clip, both of the channel's size (the graphic shares its taps)"""
    clip = frames.v210_random(W, H, frames.layer_seed(1350, 0))
    g = frames.pack_random(fmt, W, H, 1351)
    layers = [dict(src=Src(clip, W, H, m(W, H))), dict(src=Src(g, W, H, m(W, H), fmt=fmt))]
    assert route_of(layers, W, H) == "chan_compose_v210<5,0>"
    check(layers, W, H, "%s graphic over a v210 clip, full size" % fmt)


def test_four_channels_of_file_playback_in_one_call_full_size():
    """four channels each playing a 1080p yuv422p10 file (one batched read + one compositor launch for all four), and four playing
    720p yuv420p clips that fill their channels - ph_chan_compose_batch at full size, every frame against the oracle's chain and against
    its own single call"""
    for fmt, cw, ch in (("yuv422p10", W, H), ("yuv420p", 1280, 720)):
        jobs = [([dict(src=Src(frames.pack_random(fmt, cw, ch, 1360 + c), cw, ch, m(W, H), fmt=fmt))], 0, c) for c in range(4)]
        check_batch(jobs, W, H, "four channels of %s %dx%d playback in one call" % (fmt, cw, ch))


def test_four_720p_channels_in_one_launch_full_size():
    """bench.py secondary "720p50 x 4": four 1280 x 720 channels (a full-frame layer and three insets each; every line ends in a tail
    quad and cleared slots) in one launch of the batch kernel's tail instantiation, at full size"""
    from test_chan_gpu import pip_layers
    w, h = 1280, 720
    check_batch([(pip_layers(w, h, 1370 + c), 0, c) for c in range(4)], w, h, "four 1280x720 channels in one launch")


def test_four_channels_of_file_playback_with_an_inset_in_one_launch_full_size():
    """round 6: jobs with planar sources share the batch kernel's launches (its PLANAR instantiation).  Four channels each playing a 1080p
    yuv422p10 file with a quarter-size yuv420p inset (its own 8-bit Loader matrix), a fifth with a bgra8 graphic over a v210 clip and a
    sixth of v210 layers only, in ONE call: every frame against the oracle's chain and against its own single call"""
    import hip_harness as hh
    from phaneron_amd import capi
    jobs = []
    for c in range(4):
        full = frames.pack_random("yuv422p10", W, H, 1380 + c)
        inset = frames.pack_random("yuv420p", 960, 540, 1390 + c)
        jobs.append(([dict(src=Src(full, W, H, m(W, H), fmt="yuv422p10")), dict(src=Src(inset, 960, 540, m(W, H, **PIP[1 + c % 3]), fmt="yuv420p"))], 0, c))
    g = frames.pack_random("bgra8", W, H, 1395)
    jobs.append(([dict(src=Src(frames.v210_random(W, H, frames.layer_seed(1396, 0)), W, H, m(W, H, offset_x=0.3 / W))), dict(src=Src(g, W, H, m(W, H, offset_x=0.3 / W), fmt="bgra8"))], 0, 4))
    from test_chan_gpu import pip_layers
    jobs.append((pip_layers(W, H, 1397), 0, 5))
    # the route first (a dry run): one launch of the planar instantiation for all six
    import torch
    from test_chan_gpu import device_layers
    _, _, rd_d, wr_d = colour("709", "709")
    outs = [torch.zeros(frames.v210_pitch_bytes(W) * H // 4, dtype=torch.int32, device="cuda") for _ in jobs]
    with capi.trace(dry_run=True) as t:
        hh.ctx().chan_compose_batch([(device_layers(l), o, il) for (l, il, _), o in zip(jobs, outs)], W, H, *rd_d, *wr_d)
    assert t.route == "chan_compose_batch<2>x6", t.route
    check_batch(jobs, W, H, "four channels of file playback with an inset, a graphic over a clip, a v210 channel: one launch")


def test_one_launch_decoder_frames_random_shapes():
    """clip_up_write_v210_kernel (a decoder's frame in one launch) on seeded random shapes: clip and channel sizes around the kernel's units (126-column
    wave steps, row pairs, tiles that end mid-frame, channels narrower than one step, lines with tails), placements that push the clip partly
    off screen or leave a border, every planar / packed-RGB format, progressive and both fields - each against the oracle's chain, and the route
    checked to be the one-launch kernel (PH_FUZZ_SEED / PH_FUZZ_CASES: longer campaigns by hand)"""
    import os
    r = np.random.default_rng(int(os.environ.get("PH_FUZZ_SEED", "20261002")))
    fmts = ["yuv420p", "yuv422p10", "yuv422p8", "nv12", "rgba8", "bgra8"]
    outs = [(126, 8), (128, 10), (252, 6), (384, 54), (640, 36), (1280, 24), (1290, 14), (50, 4), (1920, 40), (2520, 12)]
    cases, ran = int(os.environ.get("PH_FUZZ_CASES", "40")), 0
    for case in range(cases):
        ow, oh = outs[case % len(outs)]
        fmt = fmts[int(r.integers(0, len(fmts)))]
        interlace = int(r.choice([0, 0, 1, 3])) if oh % 2 == 0 else 0
        # a clip no larger than the channel, enlarged by its placement (scale >= 1 / 0.98 per written row) or of the channel's size under the default fill
        if r.random() < 0.25 and not interlace:
            sw, sh, kw = ow, oh, dict()
        else:
            sw = max(2, int(ow * r.uniform(0.2, 0.9)) // 2 * 2)
            sh = max(2, int(oh * r.uniform(0.2, 0.9) / (2 if interlace else 1)) // 2 * 2)
            kw = dict(scale_x=float(r.choice([1.0, 0.8, 0.6])), scale_y=float(r.choice([1.0, 0.8, 0.6])), offset_x=float(r.uniform(-0.3, 0.3)), offset_y=float(r.uniform(-0.3, 0.3)))
        if fmt in ("yuv420p", "nv12"):
            sh += sh & 1
        if ow % 2 or sw % 2:
            continue
        clip = frames.pack_random(fmt, sw, sh, 5000 + case)
        layers = [dict(src=Src(clip, sw, sh, m(ow, oh, **kw), fmt=fmt))]
        route = route_of(layers, ow, oh, interlace)
        if not route.startswith("clip_up_write_v210"):
            continue  # (a shape the compositor does not take: another test's business)
        check(layers, ow, oh, "one-launch clip case %d: %s %dx%d on %dx%d il %d %r" % (case, fmt, sw, sh, ow, oh, interlace, kw), interlace=interlace,
              specs=[("709", "709"), ("709", "2020")][case % 2], poison_dst=bool(interlace))
        ran += 1
    assert ran >= cases // 2, "only %d of %d random shapes took the one-launch route" % (ran, cases)
