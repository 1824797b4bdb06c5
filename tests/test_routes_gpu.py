"""GPU tests (-m gpu): which of the library's routes makes a frame (ph_trace_begin / ph_trace_end, include/phaneron_hip.h).
The reference runs one kernel per job (clJobQueue.ts:126); this library folds a frame's jobs into fused launches and
chooses among several routes by the frame's shape - in ph_api.cpp (ph_chan_compose, ph_chan_compose_batch,
chan_layers_enlarged) and in the launchers (the channel kernel's instantiations).  These tests PIN program -> route for
representative frames of every kind, by dry runs (the calls choose, nothing is enqueued), so a change that silently moves a
shape to another kernel fails here before it shows up as a benchmark figure."""
import numpy as np
import pytest

import frames

pytestmark = pytest.mark.gpu


def m(ow, oh, **kw):
    from phaneron_amd import capi
    return capi.transform_matrix(ow, oh, **kw)


def colour(rspec="709", wspec="709"):
    import hip_harness as hh
    return hh.ColourParams.reader(rspec, wspec), hh.ColourParams.writer(wspec)


def v210(w, h):
    import torch
    return torch.zeros(frames.v210_pitch_bytes(w) * h // 4, dtype=torch.int32, device="cuda")


def planes(fmt, w, h):
    import torch
    from phaneron_amd import capi
    return tuple(torch.zeros(n, dtype=torch.uint8, device="cuda") for n in capi.pack_plane_bytes(fmt, w, h) if n)


def clip(fmt, w, h, ow, oh, **kw):
    """a layer dict as Context.chan_compose_v210 takes it: v210 words, a decoder's planes, a packed-RGB graphic or an f32 image"""
    import torch
    import hip_harness as hh
    from phaneron_amd import capi
    from oracle import orc
    if fmt == "v210":
        return dict(src=(v210(w, h), w, h, m(ow, oh, **kw)))
    if fmt == "rgba":
        return dict(src=(torch.zeros(w * h * 4, device="cuda"), w, h, m(ow, oh, **kw), "rgba"))
    p = planes(fmt, w, h)
    if fmt in ("rgba8", "bgra8"):
        return dict(src=(p[0], w, h, m(ow, oh, **kw), fmt))
    own = None if fmt == "yuv422p10" else hh.dev(capi.ycbcr2rgb_matrix("709", *orc.FORMAT_RANGE[fmt]))
    return dict(src=(p, w, h, m(ow, oh, **kw), fmt, own))


PIP = [dict(), dict(scale_x=0.5, scale_y=0.5, offset_x=-0.25, offset_y=-0.25), dict(scale_x=0.5, scale_y=0.5, offset_x=0.25, offset_y=-0.25),
       dict(scale_x=0.5, scale_y=0.5, offset_x=0.25, offset_y=0.25)]


def config2(w, h, wipe=True):
    import torch
    layers = [clip("v210", w, h, w, h, **PIP[l]) for l in range(4)]
    if wipe:
        layers[3].update(transition="wipe", incoming=(v210(w, h), w, h, None), mask=(torch.zeros(w * h * 4, device="cuda"), w, h, None, "rgba"))
    return layers


def chan(layers, ow, oh, interlace=0, out_format=None):
    import hip_harness as hh
    from phaneron_amd import capi
    rd, wr = colour()
    k = hh.ctx()
    with capi.trace(dry_run=True) as t:
        if out_format is None:
            k.chan_compose_v210(layers, v210(ow, oh), ow, oh, interlace, *rd, *wr)
        else:
            k.chan_compose_v210(layers, planes(out_format, ow, oh), ow, oh, interlace, *rd, *wr, out_fmt=out_format)
    return t.route


def batch(jobs, ow, oh):
    import hip_harness as hh
    from phaneron_amd import capi
    rd, wr = colour()
    k = hh.ctx()
    with capi.trace(dry_run=True) as t:
        k.chan_compose_batch([(layers, v210(ow, oh), il) for layers, il in jobs], ow, oh, *rd, *wr)
    return t.route


# (the channel kernel's name carries <phase-1 instantiation, output format>: 0 v210 / image programs on whole 48-pixel blocks, 1 the same with
#  line tails, 2 everything, 3 planar clips only, 4 planar clips with shared taps, 5 packed-RGB graphics over v210 clips; formats as PH_FMT_*)
ROUTES = [
    ("config 2: four v210 layers, three insets, a wipe (1080p)", lambda: chan(config2(1920, 1080), 1920, 1080), "chan_compose_v210<0,0>"),
    ("config 2 without the wipe, a field write", lambda: chan(config2(1920, 1080, wipe=False), 1920, 1080, interlace=3), "chan_compose_v210<0,0>"),
    ("the same program on a 1280 x 720 channel (lines with tails)", lambda: chan(config2(1280, 720), 1280, 720), "chan_compose_v210<1,0>"),
    ("a 1080p yuv422p10 clip under the default fill with a v210 inset", lambda: chan([clip("yuv422p10", 1920, 1080, 1920, 1080), clip("v210", 960, 540, 1920, 1080, **PIP[2])], 1920, 1080),
     "chan_compose_v210<2,0>"),
    ("two 1080p yuv420p clips, the upper one moved by a fraction of a pixel", lambda: chan([clip("yuv420p", 1920, 1080, 1920, 1080), clip("yuv420p", 1920, 1080, 1920, 1080, offset_x=0.3 / 1920)], 1920, 1080),
     "chan_compose_v210<4,0>"),
    ("a yuv422p10 clip under a placed yuv422p10 inset and an image", lambda: chan([clip("yuv422p10", 1920, 1080, 1920, 1080, scale_x=0.8, scale_y=0.8, rotate=0.05), clip("yuv422p10", 960, 540, 1920, 1080, **PIP[1]),
                                                                                    clip("rgba", 1920, 1080, 1920, 1080, rotate=0.01)], 1920, 1080), "chan_compose_v210<3,0>"),
    ("a bgra8 graphic over a v210 clip, both of the channel's size", lambda: chan([clip("v210", 1920, 1080, 1920, 1080), clip("bgra8", 1920, 1080, 1920, 1080)], 1920, 1080), "chan_compose_v210<5,0>"),
    ("a 1080p yuv420p clip on a 1080p channel (a file's frame under the default fill)", lambda: chan([clip("yuv420p", 1920, 1080, 1920, 1080)], 1920, 1080), "clip_up_write_v210<rgb>"),
    ("a 720p yuv420p clip filling a 1080p channel", lambda: chan([clip("yuv420p", 1280, 720, 1920, 1080)], 1920, 1080), "clip_up_write_v210<rgb>"),
    ("an enlarged bgra8 graphic (alpha travels with it)", lambda: chan([clip("bgra8", 1280, 720, 1920, 1080)], 1920, 1080), "clip_up_write_v210<rgba>"),
    ("one 720p v210 clip on a 1080p channel (not faster in one launch: two)", lambda: chan([clip("v210", 1280, 720, 1920, 1080)], 1920, 1080), "v210_read_lds+compose_up_write_v210"),
    ("a 720p yuv420p clip under a 1080p f32 image: the image is read as it is, so two launches", lambda: chan([clip("yuv420p", 1280, 720, 1920, 1080), clip("rgba", 1920, 1080, 1920, 1080)], 1920, 1080),
     "pack_read+compose_up_write_v210"),
    ("two 720p v210 clips on a 1080p channel", lambda: chan([clip("v210", 1280, 720, 1920, 1080), clip("v210", 1280, 720, 1920, 1080, scale_x=0.8, scale_y=0.8)], 1920, 1080),
     "v210_read_lds_batch+compose_up_write_v210"),
    ("one live v210 clip under the default fill, alone", lambda: chan([clip("v210", 1920, 1080, 1920, 1080)], 1920, 1080), "chan_compose_v210<0,0>"),
    ("config 2 into the encoder's frame (yuv422p8)", lambda: chan(config2(1920, 1080), 1920, 1080, out_format="yuv422p8"), "chan_compose_v210<0,2>"),
    ("config 2 into a yuv420p frame (the everything instantiation)", lambda: chan(config2(1920, 1080), 1920, 1080, out_format="yuv420p"), "chan_compose_v210<2,3>"),
    ("four channels of config 2 in one call", lambda: batch([(config2(1920, 1080, wipe=c == 1), 0) for c in range(4)], 1920, 1080), "chan_compose_batch<0>x4"),
    ("four channels each showing a live v210 clip", lambda: batch([([clip("v210", 1920, 1080, 1920, 1080)], 0) for _ in range(4)], 1920, 1080),
     "v210_read_lds_batch+compose_up_write_v210"),
    ("four channels of 1080p yuv422p10 playback", lambda: batch([([clip("yuv422p10", 1920, 1080, 1920, 1080)], 0) for _ in range(4)], 1920, 1080), "clip_up_write_v210<rgb>x4"),
    ("six channels of 720p yuv420p playback: four frames to a launch, then two", lambda: batch([([clip("yuv420p", 1280, 720, 1920, 1080)], 0) for _ in range(6)], 1920, 1080),
     "clip_up_write_v210<rgb>x4+clip_up_write_v210<rgb>x2"),
    ("four channels of 1080p yuv422p10 playback with a v210 inset each: the batch kernel's planar instantiation",
     lambda: batch([([clip("yuv422p10", 1920, 1080, 1920, 1080), clip("v210", 960, 540, 1920, 1080, **PIP[2])], 0) for _ in range(4)], 1920, 1080), "chan_compose_batch<2>x4"),
    ("eight channels of config 2 without wipes: one launch", lambda: batch([(config2(1920, 1080, wipe=False), 0) for _ in range(8)], 1920, 1080), "chan_compose_batch<0>x8"),
    ("eight channels of config 2 in mid-wipe (six ops each): two even launches, not 6 + 2", lambda: batch([(config2(1920, 1080), 0) for _ in range(8)], 1920, 1080),
     "chan_compose_batch<0>x4+chan_compose_batch<0>x4"),
]


@pytest.mark.parametrize("what,run,route", ROUTES, ids=[r[0] for r in ROUTES])
def test_program_to_route(what, run, route):
    assert run() == route, what


def test_the_headline_and_the_separate_operators():
    import torch
    import hip_harness as hh
    from phaneron_amd import capi
    rd, wr = colour()
    k = hh.ctx()
    w, h = 3840, 2160
    layers = [v210(w, h) for _ in range(4)]
    with capi.trace(dry_run=True) as t:
        k.fused_v210_combine(layers, v210(w, h), w, h, *rd, *wr)
    assert t.route == "fused_v210_combine_lds"
    with capi.trace(dry_run=True) as t:
        k.fused_v210_combine_batch([layers, layers], [v210(w, h), v210(w, h)], w, h, *rd, *wr)
    assert t.route == "fused_v210_combine_lds"
    img = torch.zeros(1920 * 1080 * 4, device="cuda")
    with capi.trace(dry_run=True) as t:
        k.v210_read(v210(1920, 1080), img, 1920, 1080, *rd)
        k.v210_write(img, v210(1920, 1080), 1920, 1080, 0, *wr)
    assert t.route == "v210_read_lds+v210_write_lds"


def test_a_dry_run_enqueues_nothing_and_a_traced_run_is_the_run():
    """the same call dry, traced and plain: one route; the dry run leaves the output as it was, the traced run writes what the plain one writes"""
    import torch
    import hip_harness as hh
    from phaneron_amd import capi
    rd, wr = colour()
    k = hh.ctx()
    w, h = 384, 54
    src = hh.dev(frames.v210_random(w, h, frames.layer_seed(1200, 0)).reshape(-1))
    layers = [dict(src=(src, w, h, m(w, h)))]
    words = frames.v210_pitch_bytes(w) * h // 4
    outs = [hh.dev(np.full(words, 0x2AAAAAAA, np.uint32)) for _ in range(3)]
    with capi.trace(dry_run=True) as dry:
        k.chan_compose_v210(layers, outs[0], w, h, 0, *rd, *wr)
    with capi.trace() as live:
        k.chan_compose_v210(layers, outs[1], w, h, 0, *rd, *wr)
    k.chan_compose_v210(layers, outs[2], w, h, 0, *rd, *wr)
    assert dry.route == live.route == "chan_compose_v210<0,0>"
    got = [hh.host(o, np.uint32) for o in outs]
    assert (got[0] == 0x2AAAAAAA).all()
    assert np.array_equal(got[1], got[2]) and not (got[1] == 0x2AAAAAAA).all()
    # tracing is per call pair: nothing is noted outside, and an end without a begin is refused
    with pytest.raises(capi.PhaneronError):
        capi.check(capi.lib().ph_trace_end(None, 0))
