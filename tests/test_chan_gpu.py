"""GPU tests (-m gpu) of ph_chan_compose_v210: a channel's whole frame - ToRGBA -> Mixer transform -> (Transitioner
dissolve / wipe) -> combine_N -> FromRGBA - as one kernel that samples the v210 words directly.  Every case is compared,
word for word, with the oracle's chain of the reference's operators (v210.ts read, transform.ts, transition.ts,
combine.ts, v210.ts write) on the same inputs."""
import os

import numpy as np
import pytest

import frames
from oracle import orc

pytestmark = pytest.mark.gpu


def colour(rspec, wspec):
    import hip_harness as hh
    rd_o = (orc.ycbcr2rgb_matrix(rspec), orc.gamma2linear_lut(rspec), orc.rgb2rgb_matrix(rspec, wspec))
    wr_o = (orc.rgb2ycbcr_matrix(wspec), orc.linear2gamma_lut(wspec))
    return rd_o, wr_o, hh.ColourParams.reader(rspec, wspec), hh.ColourParams.writer(wspec)


class Src:
    """one source on both sides: the oracle's RGBA (read, then placed) and the tuple the binding takes"""

    def __init__(self, data, w, h, matrix=None, fmt="v210", spec="709"):
        self.data, self.w, self.h, self.matrix, self.fmt, self.spec = data, w, h, matrix, fmt, spec  # spec: the reader's colour space (8-bit planar sources make their own matrix for it)

    def oracle(self, rd_o, ow, oh):
        if self.fmt in orc.FORMATS and self.fmt != "v210":  # data: the planes; a 10-bit 4:2:2 source shares the v210 Loader matrix, 8-bit ones have their own, RGB ones none
            rng = orc.FORMAT_RANGE[self.fmt]
            cm = rd_o[0] if self.fmt == "yuv422p10" else None if rng is None else orc.ycbcr2rgb_matrix(self.spec, *rng)
            img = orc.pack_read(self.fmt, [np.ascontiguousarray(p).view(np.uint8) for p in self.data], self.w, self.h, cm, rd_o[1], rd_o[2])
        else:
            img = orc.v210_read(self.data, self.w, self.h, *rd_o) if self.fmt == "v210" else self.data.reshape(self.h, self.w, 4)
        if self.matrix is None:
            assert (self.w, self.h) == (ow, oh)
            return img
        return orc.transform(img, self.matrix, ow, oh)

    def device(self):
        import hip_harness as hh
        if self.fmt in orc.FORMATS and self.fmt != "v210":
            from phaneron_amd import capi
            if orc.FORMAT_RANGE[self.fmt] is None:  # rgba8 / bgra8: one packed plane
                return (hh.dev(np.ascontiguousarray(self.data[0]).reshape(-1)), self.w, self.h, self.matrix, self.fmt)
            own = None if self.fmt == "yuv422p10" else hh.dev(capi.ycbcr2rgb_matrix(self.spec, *orc.FORMAT_RANGE[self.fmt]))
            return (tuple(hh.dev(np.ascontiguousarray(p).reshape(-1)) for p in self.data), self.w, self.h, self.matrix, self.fmt, own)
        t = hh.dev(self.data.reshape(-1))
        return (t, self.w, self.h, self.matrix) + (("rgba",) if self.fmt == "rgba" else ())


def oracle_chain(layers, ow, oh, interlace, rd_o, wr_o, dst=None):
    placed = []
    for L in layers:
        t = L["src"].oracle(rd_o, ow, oh)
        kind = L.get("transition", "cut")
        if kind == "dissolve":
            t = orc.transition_dissolve(t, L["incoming"].oracle(rd_o, ow, oh), L["mix"])
        elif kind == "wipe":
            t = orc.transition_wipe(t, L["incoming"].oracle(rd_o, ow, oh), L["mask"].oracle(rd_o, ow, oh))
        placed.append(t)
    comb = placed[0] if len(placed) == 1 else orc.combine(placed)  # one layer: the combiner passes it through (combiner.ts:213-217)
    return orc.v210_write(comb, ow, oh, interlace, *wr_o, out=dst)


def run_device(layers, ow, oh, interlace, rd_d, wr_d, dst=None):
    import torch
    import hip_harness as hh
    k = hh.ctx()
    words = frames.v210_pitch_bytes(ow) * oh // 4
    out = hh.dev(dst) if dst is not None else torch.zeros(words, dtype=torch.int32, device="cuda")
    dl = []
    for L in layers:
        d = dict(src=L["src"].device(), transition=L.get("transition", "cut"), mix=L.get("mix", 0.0))
        for role in ("incoming", "mask"):
            if L.get(role) is not None:
                d[role] = L[role].device()
        dl.append(d)
    k.chan_compose_v210(dl, out, ow, oh, interlace, *rd_d, *wr_d)
    return hh.host(out, np.uint32)


def check(layers, ow, oh, what, interlace=0, specs=("709", "709"), poison_dst=False):
    rd_o, wr_o, rd_d, wr_d = colour(*specs)
    dst = None
    if poison_dst:  # an interlaced write touches every other line only: the rest must stay as it was
        dst = np.full(frames.v210_pitch_bytes(ow) * oh // 4, 0x2AAAAAAA, np.uint32)
    want = oracle_chain(layers, ow, oh, interlace, rd_o, wr_o, None if dst is None else dst.copy())
    got = run_device(layers, ow, oh, interlace, rd_d, wr_d, dst)
    bad = np.flatnonzero(got != np.asarray(want).reshape(-1))
    assert bad.size == 0, "%s: %d of %d words differ, first at word %d (line %d)" % (
        what, bad.size, got.size, bad[0], bad[0] // (frames.v210_pitch_bytes(ow) // 4))


def m(ow, oh, **kw):
    from phaneron_amd import capi
    return capi.transform_matrix(ow, oh, **kw)


PIP = [dict(), dict(scale_x=0.5, scale_y=0.5, offset_x=-0.25, offset_y=-0.25), dict(scale_x=0.5, scale_y=0.5, offset_x=0.25, offset_y=-0.25),
       dict(scale_x=0.5, scale_y=0.5, offset_x=0.25, offset_y=0.25)]


def pip_layers(w, h, seed, n=4):
    first = frames.v210_ramp(w, h) if w % 6 == 0 else frames.v210_random(w, h, frames.layer_seed(seed, 0))  # (the ramp is defined for whole quads)
    srcs = [first] + [frames.v210_random(w, h, frames.layer_seed(seed, l)) for l in range(1, n)]
    return [dict(src=Src(srcs[l], w, h, m(w, h, **PIP[l]))) for l in range(n)]


@pytest.mark.parametrize("n", [1, 2, 3, 4])
def test_picture_in_picture_layers(n):
    """the Mixer placements of BASELINE config 2: a full-frame layer through the identity fill (NOT a copy: a half-pixel
    shift and 2 x 2 average, transform.ts:54-55) and quarter-size insets whose outside is the transparent border"""
    w, h = 384, 108
    check(pip_layers(w, h, 20 + n, n), w, h, "%d PiP layers %dx%d" % (n, w, h))


def test_wipe_and_dissolve_transitions():
    w, h = 384, 64
    second = frames.v210_random(w, h, frames.layer_seed(30, 1), legal=False)
    mask = frames.mask_ramp(w, h)
    layers = pip_layers(w, h, 31)
    # wipe on the top layer: incoming v210 taken 1:1, mask an f32 image (what tools/config_bench.py times)
    layers[3].update(transition="wipe", incoming=Src(second, w, h), mask=Src(mask, w, h, fmt="rgba"))
    check(layers, w, h, "wipe, 1:1 incoming, f32 mask")
    # as the reference's graph feeds the Transitioner: incoming and mask are Mixer outputs themselves (placed sources)
    vmask = frames.v210_ramp(w, h)
    layers[3].update(incoming=Src(second, w, h, m(w, h)), mask=Src(vmask, w, h, m(w, h, scale_x=1.5, scale_y=1.5)))
    check(layers, w, h, "wipe, placed v210 incoming and placed v210 mask")
    for mix in (1.0, 0.75, 1.0 / 3.0, 0.0):
        layers[1].update(transition="dissolve", mix=mix, incoming=Src(second, w, h, m(w, h, scale_x=0.5, scale_y=0.5)))
        check(layers, w, h, "dissolve mix %g on layer 1 + wipe on layer 3" % mix)


def test_one_to_one_sources_and_f32_layers():
    """sources without a transform are taken pixel for pixel (the headline's shape), f32 RGBA sources (a routed frame) mix in"""
    w, h = 192, 40
    v = [frames.v210_random(w, h, frames.layer_seed(40, l), legal=(l != 2)) for l in range(4)]
    check([dict(src=Src(x, w, h)) for x in v], w, h, "four 1:1 v210 layers")
    rgba = frames.rgba_random(w, h, 41, -0.05, 1.05)  # real alpha: the layers below show through
    check([dict(src=Src(v[0], w, h)), dict(src=Src(rgba, w, h, fmt="rgba"))], w, h, "v210 under a 1:1 f32 layer")
    small = frames.rgba_random(96, 20, 42, 0.0, 1.0)
    check([dict(src=Src(v[0], w, h, m(w, h))), dict(src=Src(small, 96, 20, m(w, h, scale_x=0.7, scale_y=0.7, offset_x=0.1), fmt="rgba")),
           dict(src=Src(v[1], w, h, m(w, h, scale_x=0.25, scale_y=0.25, offset_x=-0.3, offset_y=0.3)))], w, h, "placed f32 layer between v210 layers")


@pytest.mark.parametrize("kw", [dict(rotate=0.07), dict(rotate=-0.25, scale_x=0.8, scale_y=0.8), dict(flip_h=True), dict(flip_v=True, scale_x=2.0, scale_y=2.0),
                                dict(scale_x=3.0, scale_y=0.4, offset_x=0.2), dict(anchor_x=0.25, anchor_y=-0.25, rotate=0.4, scale_x=0.6, scale_y=0.6),
                                dict(offset_x=1.2), dict(scale_x=0.01, scale_y=0.01)],
                         ids=lambda k: ",".join("%s=%g" % kv for kv in k.items()))
def test_arbitrary_placements(kw):
    """rotations, flips, zooms, a layer moved wholly off screen, a layer shrunk to a few pixels - over a 1:1 background so
    that the placed layer's alpha edge is visible"""
    w, h = 384, 96
    bg, fg = frames.v210_random(w, h, frames.layer_seed(50, 0)), frames.v210_random(w, h, frames.layer_seed(50, 1), legal=False)
    check([dict(src=Src(bg, w, h)), dict(src=Src(fg, w, h, m(w, h, **kw)))], w, h, "placement %r" % (kw,))


def test_sources_of_other_sizes_and_colour_recipes():
    """sources smaller and larger than the output (the Mixer scales them to the consumer format), 709 -> 2020 and a
    non-standard reader matrix (the general dot-product path)"""
    ow, oh = 384, 120
    a = frames.v210_random(192, 60, frames.layer_seed(60, 0))      # up-scaled 2x
    b = frames.v210_random(768, 240, frames.layer_seed(60, 1))     # shrunk to half
    c = frames.v210_random(96, 36, frames.layer_seed(60, 2))       # odd aspect, placed small
    layers = [dict(src=Src(a, 192, 60, m(ow, oh))), dict(src=Src(b, 768, 240, m(ow, oh, scale_x=0.5, scale_y=0.5, offset_x=0.2, offset_y=-0.2))),
              dict(src=Src(c, 96, 36, m(ow, oh, scale_x=0.3, scale_y=0.3, offset_x=-0.3, offset_y=0.25)))]
    check(layers, ow, oh, "mixed source sizes 709 -> 709")
    check(layers, ow, oh, "mixed source sizes 709 -> 2020", specs=("709", "2020"))
    check(layers, ow, oh, "mixed source sizes 2020 -> 709", specs=("2020", "709"))


def test_general_reader_matrix_path():
    """a YCbCr matrix that is not of the standard shape (Cb feeds R) takes the kernel's general dot-product path"""
    import hip_harness as hh
    from phaneron_amd import capi
    w, h = 192, 24
    cm = orc.ycbcr2rgb_matrix("709").copy()
    cm[1] = np.float32(1.7e-5)
    rd_o = (cm, orc.gamma2linear_lut("709"), orc.rgb2rgb_matrix("709", "2020"))
    wr_o = (orc.rgb2ycbcr_matrix("2020"), orc.linear2gamma_lut("2020"))
    _, lut, gm = hh.ColourParams.reader("709", "2020")
    rd_d, wr_d = (hh.dev(cm), lut, gm), hh.ColourParams.writer("2020")
    layers = pip_layers(w, h, 70, 3)
    want = oracle_chain(layers, w, h, 0, rd_o, wr_o)
    got = run_device(layers, w, h, 0, rd_d, wr_d)
    assert np.array_equal(got, np.asarray(want).reshape(-1))


@pytest.mark.parametrize("interlace", [1, 3])
def test_field_outputs(interlace):
    """FromRGBA's field writes (v210.ts:126-127): every other line, the rest of the destination untouched"""
    w, h = 384, 54
    check(pip_layers(w, h, 80 + interlace), w, h, "interlace %d" % interlace, interlace=interlace, poison_dst=True)


def test_config2_full_size_with_wipe():
    """BASELINE config 2 at its real size through the one kernel: ramp full frame, three quarter-size insets, a wipe on
    the top layer against a second source by a horizontal-ramp mask"""
    w, h = 1920, 1080
    layers = pip_layers(w, h, 2)
    second = frames.v210_random(w, h, frames.layer_seed(2, 4), legal=False)
    layers[3].update(transition="wipe", incoming=Src(second, w, h), mask=Src(frames.mask_ramp(w, h), w, h, fmt="rgba"))
    check(layers, w, h, "config 2 at 1920x1080")


def test_frame_sizes_that_do_not_fill_the_chip():
    """fewer chunks than waves, a single row, and a 2160p output from 1080p sources (more than one index-frame pass per lane)"""
    for w, h in ((192, 1), (192, 5), (576, 3)):
        check(pip_layers(w, h, 90), w, h, "%dx%d" % (w, h))
    ow, oh = 3840, 256
    a = frames.v210_random(1920, 128, frames.layer_seed(91, 0))
    b = frames.v210_random(1920, 128, frames.layer_seed(91, 1))
    check([dict(src=Src(a, 1920, 128, m(ow, oh))), dict(src=Src(b, 1920, 128, m(ow, oh, scale_x=0.5, scale_y=0.5, offset_x=0.25)))], ow, oh,
          "1920x128 -> 3840x256", specs=("709", "2020"))


def test_refusals():
    import torch
    import hip_harness as hh
    from phaneron_amd import capi
    k = hh.ctx()
    rd_d, wr_d = hh.ColourParams.reader("709", "709"), hh.ColourParams.writer("709")
    w, h = 384, 8
    src = hh.dev(frames.v210_random(w, h, 1))
    out = torch.zeros(frames.v210_pitch_bytes(w) * h // 4, dtype=torch.int32, device="cuda")
    ok = dict(src=(src, w, h, None))
    with pytest.raises(capi.PhaneronError, match="needs an even width"):  # (ragged even widths are served: test_ragged_output_widths)
        k.chan_compose_v210([dict(src=(src, 331, h, None))], out, 331, h, 0, *rd_d, *wr_d)
    with pytest.raises(capi.PhaneronError, match="not the output size"):
        k.chan_compose_v210([dict(src=(src, 192, h, None))], out, w, h, 0, *rd_d, *wr_d)
    with pytest.raises(capi.PhaneronError, match="incoming source is empty"):
        k.chan_compose_v210([dict(ok, transition="dissolve", mix=0.5)], out, w, h, 0, *rd_d, *wr_d)
    with pytest.raises(capi.PhaneronError, match="mask is empty"):
        k.chan_compose_v210([dict(ok, transition="wipe", incoming=(src, w, h, None))], out, w, h, 0, *rd_d, *wr_d)
    with pytest.raises(capi.PhaneronError, match="an odd width"):
        k.chan_compose_v210([dict(src=(src, 101, h, capi.transform_matrix(w, h)))], out, w, h, 0, *rd_d, *wr_d)
    with pytest.raises(capi.PhaneronError, match="1..8 layers"):
        k.chan_compose_v210([ok] * 9, out, w, h, 0, *rd_d, *wr_d)
    plain = hh.dev(orc.linear2gamma_lut("709"))  # never registered: no LDS form
    with pytest.raises(capi.PhaneronError, match="writer gamma LUT has no LDS form"):
        k.chan_compose_v210([ok], out, w, h, 0, *rd_d, wr_d[0], plain)


def random_layers(r, ow, oh, n_layers, with_planar=False):
    """n_layers random layers for an ow x oh channel: source sizes and formats, placements, transitions (see test_random_channel_programs)"""
    def source(must_fill=False):
        rgba = r.random() < 0.25
        one_to_one = r.random() < 0.3
        if one_to_one:
            w, h = ow, oh
        else:
            w, h = int(r.choice([48, 96, 192, 288, 384])), int(r.integers(2, 40))
        seed = int(r.integers(1, 1 << 30))
        planar = None if rgba or not with_planar or r.random() < 0.6 else str(r.choice(["yuv422p10", "yuv422p8", "yuv420p", "nv12", "rgba8", "bgra8"]))
        if planar:
            if not one_to_one:
                h += h & 1  # (4:2:0: an even height)
            elif oh & 1 and planar in ("yuv420p", "nv12"):
                planar = "yuv422p8"
        data = frames.rgba_random(w, h, seed, -0.05, 1.05) if rgba else frames.pack_random(planar, w, h, seed) if planar else frames.v210_random(w, h, seed, legal=bool(r.random() < 0.7))
        mat = None
        if not one_to_one or r.random() < 0.5:
            kw = dict(scale_x=float(r.choice([0.3, 0.5, 1.0, 1.0, 1.7, 2.0])), scale_y=float(r.choice([0.3, 0.5, 1.0, 1.0, 1.7, 2.0])),
                      offset_x=float(r.uniform(-0.6, 0.6)), offset_y=float(r.uniform(-0.6, 0.6)))
            if r.random() < 0.3:
                kw["rotate"] = float(r.uniform(-0.5, 0.5))
            if r.random() < 0.2:
                kw["flip_h"] = True
            if must_fill:
                kw = dict()
            mat = m(ow, oh, **kw)
        return Src(data, w, h, mat, "rgba" if rgba else planar or "v210")
    layers = []
    for l in range(n_layers):
        L = dict(src=source(must_fill=(l == 0 and r.random() < 0.5)))
        t = r.random()
        if t < 0.2:
            L.update(transition="dissolve", mix=float(r.choice([0.0, 0.25, 1.0 / 3.0, 1.0])), incoming=source())
        elif t < 0.4:
            L.update(transition="wipe", incoming=source(), mask=source())
        layers.append(L)
    return layers


def test_random_channel_programs():
    """seeded random channels: output sizes around the kernel's units (192-column chunks, row pairs, fewer chunks than waves),
    1-6 layers of random source sizes and formats, random placements (scales either side of 1, rotations, flips, offsets that
    push layers partly or wholly off screen), random transitions with placed or 1:1 partners, every interlace mode"""
    r = np.random.default_rng(int(os.environ.get("PH_FUZZ_SEED", "20260929")))  # (PH_FUZZ_SEED / PH_FUZZ_CASES: longer campaigns by hand)
    sizes = [(192, 2), (192, 9), (384, 33), (576, 17), (768, 6), (960, 20)]
    for case in range(int(os.environ.get("PH_FUZZ_CASES", "18"))):
        ow, oh = sizes[case % len(sizes)]
        interlace = int(r.choice([0, 0, 1, 3]))
        layers = random_layers(r, ow, oh, int(r.integers(1, 7)))
        check(layers, ow, oh, "random channel %d: %dx%d il %d, %d layers" % (case, ow, oh, interlace, len(layers)), interlace=interlace,
              specs=[("709", "709"), ("709", "2020"), ("2020", "709")][case % 3], poison_dst=bool(interlace))


def test_dissolve_against_a_smaller_rotated_source_over_the_same_frame_taken_plain():
    """one v210 frame used twice - pixel for pixel as the bottom layer and, through the identity fill, as the outgoing side
    of a dissolve whose incoming side is a half-size source, rotated (the chain node/defer.js folds into this kernel)"""
    ow, oh = 384, 108
    a = frames.v210_random(ow, oh, frames.layer_seed(95, 0))
    b = frames.v210_random(192, 54, frames.layer_seed(95, 1))
    for mix in (0.75, 0.2):
        layers = [dict(src=Src(a, ow, oh)),
                  dict(src=Src(a, ow, oh, m(ow, oh)), transition="dissolve", mix=mix, incoming=Src(b, 192, 54, m(ow, oh, scale_x=0.8, scale_y=0.8, rotate=0.05)))]
        check(layers, ow, oh, "dissolve %g, half-size rotated incoming" % mix)


def test_8k_channel_equals_the_separate_kernels():
    """Beyond UHD (7680 x 4320: 88 MB v210 frames, 530 MB images, a 265 MB index frame): the one-kernel channel against the
    separate HIP kernels - read, transform, combine_3, write, each pinned to the oracle at sizes it finishes in seconds - word
    for word.  Layers: an 8K frame taken pixel for pixel, a 2160p frame enlarged by the default fill, an 8K frame shrunk and rotated."""
    import torch
    import hip_harness as hh
    from phaneron_amd import capi
    ow, oh = 7680, 4320
    k = hh.ctx()
    a = frames.v210_random(ow, oh, frames.layer_seed(97, 0))
    b = frames.v210_random(3840, 2160, frames.layer_seed(97, 1))
    c = frames.v210_random(ow, oh, frames.layer_seed(97, 2), legal=False)
    _, _, rd_d, wr_d = colour("709", "2020")
    mats = [None, m(ow, oh), m(ow, oh, scale_x=0.4, scale_y=0.4, rotate=0.07, offset_x=0.2, offset_y=-0.15)]
    srcs = [(a, ow, oh), (b, 3840, 2160), (c, ow, oh)]
    dev = [hh.dev(s[0]) for s in srcs]
    words = frames.v210_pitch_bytes(ow) * oh // 4
    got = torch.zeros(words, dtype=torch.int32, device="cuda")
    k.chan_compose_v210([dict(src=(d, s[1], s[2], mt)) for d, s, mt in zip(dev, srcs, mats)], got, ow, oh, 0, *rd_d, *wr_d)
    placed = []
    for d, s, mt in zip(dev, srcs, mats):
        img = torch.empty(s[1] * s[2] * 4, dtype=torch.float32, device="cuda")
        k.v210_read(d, img, s[1], s[2], *rd_d)
        if mt is not None:
            out = torch.empty(ow * oh * 4, dtype=torch.float32, device="cuda")
            k.transform(img, s[1], s[2], hh.dev(np.asarray(mt, np.float32)), out, ow, oh)
            k.wait()
            img = out
        placed.append(img)
    comb = torch.empty(ow * oh * 4, dtype=torch.float32, device="cuda")
    k.combine(placed, comb, ow, oh)
    want = torch.zeros_like(got)
    k.v210_write(comb, want, ow, oh, 0, *wr_d)
    k.wait()
    torch.cuda.synchronize()
    assert torch.equal(got, want)


def test_planar_ten_bit_sources():
    """yuv422p10le frames (what a ProRes / DNxHD decoder hands over: ffmpegProducer.ts:410-412) as sources of the channel kernel:
    pixel for pixel, placed, enlarged from a smaller frame, as the incoming side of a dissolve and as a wipe's mask, beside v210
    and image layers - against the oracle's yuv422p10 reader (every 16-bit code, legal or not) followed by the chain of operators"""
    w, h = 384, 54
    p = [frames.pack_random("yuv422p10", w, h, 300 + i) for i in range(4)]
    small = frames.pack_random("yuv422p10", 200, 30, 310)  # (a width that is not a multiple of 8: the planes' lines are padded)
    small = [(frames.splitmix64(311 + i, x.size // 2) % np.uint64(65536)).astype(np.uint16).view(np.uint8) for i, x in enumerate(small)]  # every 16-bit word
    v = frames.v210_random(w, h, frames.layer_seed(98, 0))
    check([dict(src=Src(p[0], w, h, fmt="yuv422p10"))], w, h, "one planar layer, pixel for pixel")
    check([dict(src=Src(p[0], w, h, fmt="yuv422p10")), dict(src=Src(p[1], w, h, m(w, h, **PIP[1]), fmt="yuv422p10")),
           dict(src=Src(v, w, h, m(w, h, **PIP[2]))), dict(src=Src(small, 200, 30, m(w, h, scale_x=0.6, scale_y=0.6, rotate=0.05, offset_x=0.2), fmt="yuv422p10"))],
          w, h, "planar and v210 layers, placed", specs=("709", "2020"))
    rgba = frames.rgba_random(w, h, 320, -0.05, 1.05)
    layers = [dict(src=Src(v, w, h)),
              dict(src=Src(p[2], w, h, m(w, h), fmt="yuv422p10"), transition="dissolve", mix=0.4, incoming=Src(small, 200, 30, m(w, h, scale_x=1.5, scale_y=1.5), fmt="yuv422p10")),
              dict(src=Src(rgba, w, h, fmt="rgba"), transition="wipe", incoming=Src(p[3], w, h, fmt="yuv422p10"), mask=Src(p[0], w, h, m(w, h, scale_x=2.0, scale_y=2.0), fmt="yuv422p10"))]
    check(layers, w, h, "planar sources inside transitions")
    check(layers, w, h, "planar sources inside transitions, field 3", interlace=3, poison_dst=True)


@pytest.mark.parametrize("fmt", ["yuv422p8", "yuv420p", "nv12"])
def test_planar_eight_bit_sources(fmt):
    """the 8-bit planar formats of file decoders (ffmpegProducer.ts:398-408) as sources: their code ranges are not the 10-bit ones, so
    each brings its own Loader matrix (col_matrix12) while the gamma table and the gamut matrix are the call's - pixel for pixel,
    placed (4:2:0: a chroma line serves two luma lines, also across the bilinear taps), odd sizes' padded lines, beside v210 and
    planar 10-bit layers, inside a dissolve; against the oracle's reader of that format followed by the chain"""
    w, h = 384, 54
    a = frames.pack_random(fmt, w, h, 400)
    b = frames.pack_random(fmt, 204, 38, 401)  # lines padded to a multiple of 8 samples
    v = frames.v210_random(w, h, frames.layer_seed(99, 0))
    p10 = frames.pack_random("yuv422p10", w, h, 402)
    check([dict(src=Src(a, w, h, fmt=fmt))], w, h, "%s pixel for pixel" % fmt)
    check([dict(src=Src(a, w, h, fmt=fmt, spec="709")), dict(src=Src(b, 204, 38, m(w, h, scale_x=0.5, scale_y=0.7, offset_x=-0.2, rotate=0.03), fmt=fmt)),
           dict(src=Src(v, w, h, m(w, h, **PIP[2]))), dict(src=Src(p10, w, h, m(w, h, **PIP[3]), fmt="yuv422p10"))], w, h, "%s beside v210 and yuv422p10" % fmt)
    layers = [dict(src=Src(v, w, h)), dict(src=Src(a, w, h, m(w, h), fmt=fmt), transition="dissolve", mix=0.3, incoming=Src(b, 204, 38, m(w, h, scale_x=1.7, scale_y=1.7), fmt=fmt))]
    check(layers, w, h, "%s inside a dissolve" % fmt)
    check(layers, w, h, "%s inside a dissolve, 709 -> 2020, field 1" % fmt, interlace=1, poison_dst=True, specs=("709", "2020"))


@pytest.mark.parametrize("fmt", ["rgba8", "bgra8"])
def test_packed_rgb_sources_with_alpha(fmt):
    """stills and graphics (rgba8 / bgra8: every byte, alpha too, through the gamma table - rgba8.ts:49-62): a logo with real alpha
    over v210 and planar layers, pixel for pixel and placed, as the incoming side of a dissolve and as a wipe's mask"""
    w, h = 384, 54
    g = frames.pack_random(fmt, w, h, 500)
    small = frames.pack_random(fmt, 100, 30, 501)
    v = frames.v210_random(w, h, frames.layer_seed(100, 0))
    p10 = frames.pack_random("yuv422p10", w, h, 502)
    check([dict(src=Src(g, w, h, fmt=fmt))], w, h, "%s pixel for pixel" % fmt)
    check([dict(src=Src(v, w, h)), dict(src=Src(p10, w, h, m(w, h, **PIP[1]), fmt="yuv422p10")), dict(src=Src(g, w, h, fmt=fmt)),
           dict(src=Src(small, 100, 30, m(w, h, scale_x=0.3, scale_y=0.5, offset_x=0.3, offset_y=-0.2, rotate=-0.04), fmt=fmt))], w, h, "%s over v210 and yuv422p10" % fmt,
          specs=("709", "2020"))
    layers = [dict(src=Src(v, w, h)), dict(src=Src(g, w, h, m(w, h), fmt=fmt), transition="dissolve", mix=0.6, incoming=Src(small, 100, 30, m(w, h, scale_x=2.0, scale_y=2.0), fmt=fmt)),
              dict(src=Src(p10, w, h, fmt="yuv422p10"), transition="wipe", incoming=Src(v, w, h, m(w, h, **PIP[3])), mask=Src(g, w, h, fmt=fmt))]
    check(layers, w, h, "%s inside transitions" % fmt)


@pytest.mark.parametrize("w,h,interlace", [(720, 576, 0), (720, 576, 1), (720, 486, 3), (48, 4, 0), (240, 7, 0), (1440, 30, 0), (336, 9, 3)])
def test_widths_that_are_not_whole_chunks(w, h, interlace):
    """SD frames (720 x 576 / 486, field writes) and other widths that are multiples of 48 but not of the kernel's 192-pixel chunk: a
    row's last chunk is short - its spare lanes park nothing, its spare quads write nothing, the lines' ends are what the oracle wrote"""
    layers = pip_layers(w, h, 600 + w + interlace, 3)
    layers.append(dict(src=Src(frames.pack_random("yuv420p", w, h + (h & 1), 601), w, h + (h & 1), m(w, h, scale_x=0.6, scale_y=0.6, offset_x=0.2, offset_y=0.2), fmt="yuv420p")))
    check(layers, w, h, "%dx%d interlace %d" % (w, h, interlace), interlace=interlace, poison_dst=interlace != 0)


def placed_and_combined(layers, ow, oh, rd_o):
    """the oracle's composite of a layer list (no transitions) as an RGBA image"""
    placed = [L["src"].oracle(rd_o, ow, oh) for L in layers]
    return placed[0] if len(placed) == 1 else orc.combine(placed)


@pytest.mark.parametrize("fmt", ["rgba8", "bgra8", "yuv422p8", "yuv422p10", "yuv420p", "nv12"])
@pytest.mark.parametrize("interlace", [0, 1, 3])
def test_other_output_formats(fmt, interlace):
    """the channel's packed frame in the formats of the reference's other consumers - rgba8 / bgra8 for the screen
    (screenConsumer.ts:131), yuv422p8 for an encoder (ffmpegConsumer.ts:144), yuv422p10, and the 4:2:0 Writers (yuv420p.ts, nv12.ts:
    chroma from the upper line of a line pair) - whole frames and single fields (the other field's lines stay as they were), from
    v210, planar and graphics sources: against the oracle's chain ending in that format's writer"""
    import torch
    import hip_harness as hh
    from phaneron_amd import capi
    w, h = 384, 54
    rd_o, _, rd_d, _ = colour("709", "709")
    layers = pip_layers(w, h, 700 + interlace, 3)
    layers.append(dict(src=Src(frames.pack_random("bgra8", 100, 30, 701), 100, 30, m(w, h, scale_x=0.3, scale_y=0.5, offset_x=0.3, offset_y=-0.2), fmt="bgra8")))
    layers.append(dict(src=Src(frames.pack_random("yuv420p", w, h, 702), w, h, m(w, h, **PIP[3]), fmt="yuv420p")))
    check_format(layers, w, h, fmt, "%s interlace %d" % (fmt, interlace), interlace)


def check_format(layers, w, h, fmt, what, interlace=0):
    """the channel's frame in another wire format against the oracle's chain ending in that format's writer; the planes are
    poisoned first (a field write leaves the other field's lines alone)"""
    import torch
    import hip_harness as hh
    from phaneron_amd import capi
    rd_o, _, rd_d, _ = colour("709", "709")
    rng = orc.FORMAT_RANGE[fmt]
    wcm_o = None if rng is None else orc.rgb2ycbcr_matrix("709", *rng)
    wlut_o = orc.linear2gamma_lut("709")
    before = [np.full(n, 0x5A, np.uint8) for n in frames.pack_plane_bytes(fmt, w, h)]
    want = orc.pack_write(fmt, placed_and_combined(layers, w, h, rd_o), w, h, interlace, wcm_o, wlut_o, planes=before)
    k = hh.ctx()
    dst = [hh.dev(b.copy()) for b in before]
    wlut_d = hh.ColourParams.writer("709")[1]
    wcm_d = None if rng is None else hh.dev(capi.rgb2ycbcr_matrix("709", *rng))
    dl = [dict(src=L["src"].device()) for L in layers]
    k.chan_compose_v210(dl, dst, w, h, interlace, *rd_d, wcm_d, wlut_d, out_fmt=fmt)
    k.wait()
    torch.cuda.synchronize()
    for i, (g, wnt) in enumerate(zip(dst, want)):
        got = hh.host(g, np.uint8)
        bad = np.flatnonzero(got != wnt)
        assert bad.size == 0, "%s plane %d: %d of %d bytes differ, first at %d" % (what, i, bad.size, got.size, bad[0])


# ---- widths that are not a multiple of 48: 1280 x 720 is one of the reference's three video formats (src/config.ts:43-54) ------------------
@pytest.mark.parametrize("w,h", [(1280, 24), (100, 9), (52, 6), (1302, 4), (50, 3)])
def test_ragged_output_widths(w, h):
    """a channel whose frames end in a tail quad (1280 % 6 = 2, 100 % 6 = 4) and cleared slots (1302: whole quads only): sources of
    the same ragged width taken 1:1 and placed (their tail pixels are read without the matrix's offset column, v210.ts:88-93), the
    writer's tail arithmetic (v210.ts:173-184), lines by pitch, both fields; the output is poisoned first"""
    v = [frames.v210_random(w, h, frames.layer_seed(80, l), legal=(l != 1)) for l in range(4)]
    check([dict(src=Src(x, w, h)) for x in v[:2]], w, h, "1:1 layers %dx%d" % (w, h), poison_dst=True)
    check(pip_layers(w, h, 81), w, h, "PiP layers %dx%d" % (w, h), specs=("709", "2020"), poison_dst=True)
    check([dict(src=Src(v[0], w, h, m(w, h, scale_x=1.3, scale_y=1.3, rotate=0.05))), dict(src=Src(v[1], w, h, m(w, h, scale_x=0.6, scale_y=0.7, offset_x=0.2)))],
          w, h, "rotated, enlarged and shrunk %dx%d" % (w, h), poison_dst=True)
    if h % 2 == 0:
        for interlace in (1, 3):
            check(pip_layers(w, h, 82, 3), w, h, "field %d of %dx%d" % (interlace, w, h), interlace=interlace, poison_dst=True)


def test_ragged_sources_on_regular_outputs_and_the_other_way_round():
    """a 1280-wide source shown on a 1920 channel (only its reads meet a tail) and 1920-wide sources on a 1280 channel (only the write
    does); a transition whose incoming source and mask are ragged; an f32 layer in between"""
    ow, oh = 1920, 20
    hd = frames.v210_random(1280, 16, frames.layer_seed(83, 0), legal=False)
    full = frames.v210_random(ow, oh, frames.layer_seed(83, 1))
    check([dict(src=Src(full, ow, oh)), dict(src=Src(hd, 1280, 16, m(ow, oh, scale_x=0.9, scale_y=0.9)))], ow, oh, "720-line source on an HD channel")
    ow, oh = 1280, 18
    a, b = frames.v210_random(1920, 24, frames.layer_seed(84, 0)), frames.v210_random(1280, 18, frames.layer_seed(84, 1))
    mask = frames.v210_ramp(96, 6)
    rgba = frames.rgba_random(ow, oh, 85, -0.05, 1.05)
    layers = [dict(src=Src(a, 1920, 24, m(ow, oh))), dict(src=Src(rgba, ow, oh, fmt="rgba")),
              dict(src=Src(b, 1280, 18, m(ow, oh, scale_x=0.5, scale_y=0.5, offset_x=0.25)), transition="wipe", incoming=Src(b, 1280, 18), mask=Src(mask, 96, 6, m(ow, oh))),
              dict(src=Src(a, 1920, 24, m(ow, oh, scale_x=0.4, scale_y=0.4, offset_x=-0.3, offset_y=0.3)), transition="dissolve", mix=0.25, incoming=Src(b, 1280, 18, m(ow, oh, scale_x=0.4, scale_y=0.4, offset_x=-0.3, offset_y=0.3)))]
    check(layers, ow, oh, "HD sources, an image, a wipe and a dissolve on a 1280 channel", specs=("709", "2020"), poison_dst=True)


@pytest.mark.parametrize("fmt", ["rgba8", "bgra8", "yuv422p8", "yuv422p10", "yuv420p", "nv12"])
def test_other_output_formats_at_1280(fmt):
    """the screen's and an encoder's frames of a 1280-wide channel (widths in multiples of 8 for the planar writers)"""
    w, h = 1280, 12
    v = [frames.v210_random(w, h, frames.layer_seed(86, l)) for l in range(2)]
    layers = [dict(src=Src(v[0], w, h)), dict(src=Src(v[1], w, h, m(w, h, scale_x=0.5, scale_y=0.5, offset_x=0.1)))]
    check_format(layers, w, h, fmt, "1280-wide %s frame" % fmt)


# ---- several channels' frames per launch (ph_chan_compose_batch, round 5) -----------------------------------------------------

def device_layers(layers):
    dl = []
    for L in layers:
        d = dict(src=L["src"].device(), transition=L.get("transition", "cut"), mix=L.get("mix", 0.0))
        for role in ("incoming", "mask"):
            if L.get(role) is not None:
                d[role] = L[role].device()
        dl.append(d)
    return dl


def check_batch(jobs, ow, oh, what, specs=("709", "709")):
    """jobs: list of (layers, interlace, out_slot) - jobs with the same out_slot write into one frame (its two fields).  Every frame
    against the oracle's chain, word for word, and against the same jobs posted one call each."""
    import torch
    import hip_harness as hh
    rd_o, wr_o, rd_d, wr_d = colour(*specs)
    k = hh.ctx()
    words = frames.v210_pitch_bytes(ow) * oh // 4
    slots = sorted({s for _, _, s in jobs})
    want = {s: np.full(words, 0x2AAAAAAA, np.uint32) for s in slots}
    for layers, interlace, s in jobs:
        want[s] = np.asarray(oracle_chain(layers, ow, oh, interlace, rd_o, wr_o, want[s])).reshape(-1)
    outs = {s: hh.dev(np.full(words, 0x2AAAAAAA, np.uint32)) for s in slots}
    single = {s: hh.dev(np.full(words, 0x2AAAAAAA, np.uint32)) for s in slots}
    dls = [device_layers(layers) for layers, _, _ in jobs]
    k.chan_compose_batch([(dl, outs[s], il) for dl, (_, il, s) in zip(dls, jobs)], ow, oh, *rd_d, *wr_d)
    for dl, (_, il, s) in zip(dls, jobs):
        k.chan_compose_v210(dl, single[s], ow, oh, il, *rd_d, *wr_d)
    for s in slots:
        got, one = hh.host(outs[s], np.uint32), hh.host(single[s], np.uint32)
        bad = np.flatnonzero(got != want[s])
        assert bad.size == 0, "%s: frame %d: %d of %d words differ from the oracle chain, first at word %d (line %d)" % (
            what, s, bad.size, got.size, bad[0], bad[0] // (frames.v210_pitch_bytes(ow) // 4))
        assert np.array_equal(got, one), "%s: frame %d: the batch differs from the separate calls" % (what, s)


def channel_variants(w, h, seed):
    """four channels as a gallery would run them: PiP, PiP in mid-wipe, a dissolve against a placed source under an f32 layer, one plain layer"""
    second = frames.v210_random(w, h, frames.layer_seed(seed, 9), legal=False)
    a = pip_layers(w, h, seed)
    b = pip_layers(w, h, seed + 1)
    b[3].update(transition="wipe", incoming=Src(second, w, h), mask=Src(frames.mask_ramp(w, h), w, h, fmt="rgba"))
    c = pip_layers(w, h, seed + 2, 2)
    c[1].update(transition="dissolve", mix=0.375, incoming=Src(second, w, h, m(w, h, scale_x=0.5, scale_y=0.5, rotate=0.1)))
    c.append(dict(src=Src(frames.rgba_random(w, h, seed + 3, -0.05, 1.05), w, h, fmt="rgba")))
    d = [dict(src=Src(frames.v210_random(w, h, frames.layer_seed(seed, 7)), w, h))]
    return [a, b, c, d]


def test_chan_batch_equals_separate_calls():
    """four different channels' frames in one launch = the oracle's chain per channel = four separate calls"""
    w, h = 384, 108
    check_batch([(layers, 0, i) for i, layers in enumerate(channel_variants(w, h, 300))], w, h, "4 channels %dx%d" % (w, h))
    check_batch([(layers, 0, i) for i, layers in enumerate(channel_variants(w, h, 310))], w, h, "4 channels 709 -> 2020", specs=("709", "2020"))


def test_chan_batch_fields_and_mixed_kinds():
    """both fields of a frame as two jobs of one launch; fields and frames in one call (split into launches inside); eight jobs; nine jobs"""
    w, h = 384, 54
    v = channel_variants(w, h, 320)
    check_batch([(v[0], 1, 0), (v[1], 3, 0)], w, h, "two fields of one frame from two programs")
    check_batch([(v[0], 1, 0), (v[0], 3, 0), (v[2], 1, 1), (v[2], 3, 1)], w, h, "two channels' field pairs")
    check_batch([(v[0], 0, 0), (v[1], 1, 1), (v[2], 3, 1), (v[3], 0, 2)], w, h, "frames and fields in one call")
    more = channel_variants(w, h, 330)
    check_batch([(x, 0, i) for i, x in enumerate(v + more)], w, h, "eight jobs")
    check_batch([(x, 0, i) for i, x in enumerate(v + more + [v[1]])], w, h, "nine jobs: two launches")


def test_chan_batch_same_frame_twice_keeps_call_order():
    """two jobs that write the same lines of one frame are not put into one launch: the later job wins, as with separate calls"""
    w, h = 192, 12
    v = channel_variants(w, h, 340)
    check_batch([(v[0], 0, 0), (v[3], 0, 0)], w, h, "same frame twice")
    check_batch([(v[0], 1, 0), (v[3], 0, 0)], w, h, "a field, then the whole frame")


def test_chan_batch_lines_with_tails():
    """1280-wide channels (src/config.ts:43-54): tail quads and cleared slots in every job; a narrower ragged source in one of them"""
    w, h = 1280, 24
    v = channel_variants(w, h, 350)
    small = frames.v210_random(332, 10, frames.layer_seed(350, 5))
    v[3].append(dict(src=Src(small, 332, 10, m(w, h, scale_x=0.4, scale_y=0.4, offset_x=0.2))))
    check_batch([(x, 0, i) for i, x in enumerate(v)], w, h, "4 channels 1280 wide")
    check_batch([(v[0], 1, 0), (v[3], 3, 0)], w, h, "1280 wide, the two fields")


def test_chan_batch_with_a_planar_job_in_the_middle():
    """a job the batch kernel does not take (planar sources) runs in its turn through the one-job kernel; the jobs around it are batched"""
    w, h = 384, 32
    v = channel_variants(w, h, 360)
    planar = [dict(src=Src(frames.pack_random("yuv422p10", w, h, 361), w, h, fmt="yuv422p10")), v[3][0]]
    check_batch([(v[0], 0, 0), (planar, 0, 1), (v[1], 0, 2), (v[2], 0, 3)], w, h, "planar job between batched ones")


def test_chan_batch_full_size_four_1080p_channels():
    """the reference's deployment: four 1080p channels in one context (src/index.ts:45-71) - one launch, each frame = the oracle's chain"""
    w, h = 1920, 1080
    check_batch([(layers, 0, i) for i, layers in enumerate(channel_variants(w, h, 370))], w, h, "4 channels 1920x1080")


def test_chan_batch_random_calls():
    """seeded random calls of ph_chan_compose_batch: 2 - 10 jobs of 1 - 5 random layers each (40 ops and 8 jobs per launch: longer calls split inside),
    frames and fields mixed, two jobs now and then the two fields of ONE frame, output widths with and without tails - every frame
    against the oracle's chain and against the same jobs posted one call each"""
    r = np.random.default_rng(int(os.environ.get("PH_FUZZ_SEED", "20260930")))
    sizes = [(192, 9), (384, 33), (576, 16), (100, 9), (1280, 6), (960, 20)]
    for case in range(int(os.environ.get("PH_FUZZ_CASES", "30"))):
        ow, oh = sizes[case % len(sizes)]
        jobs, slot = [], 0
        while len(jobs) < int(r.integers(2, 11)):
            planar = case % 2 == 1  # every other call has decoders' frames and graphics among its sources (those jobs run in their turn, or by the route)
            if r.random() < 0.25:  # both fields of one frame, from two programs
                jobs.append((random_layers(r, ow, oh, int(r.integers(1, 4)), planar), 1, slot))
                jobs.append((random_layers(r, ow, oh, int(r.integers(1, 4)), planar), 3, slot))
            elif r.random() < 0.3:  # a run of channels showing clips of one shape under one placement (the read + compositor route's groups)
                fmt = str(r.choice(["v210", "yuv420p", "bgra8"])) if planar else "v210"
                cw, ch = (ow, oh) if r.random() < 0.5 else (ow // 2 // 6 * 6 or 6, max(2, oh // 2 // 2 * 2))
                if fmt == "yuv420p":
                    ch += ch & 1
                    if (cw, ch) != (ow, oh + (oh & 1)) and cw == ow:
                        cw, ch = ow // 2 // 6 * 6 or 6, max(2, oh // 2 // 2 * 2)
                if (cw, ch) == (ow, oh) or cw < ow:
                    for _ in range(int(r.integers(2, 6))):
                        seed = int(r.integers(1, 1 << 30))
                        data = frames.v210_random(cw, ch, seed) if fmt == "v210" else frames.pack_random(fmt, cw, ch, seed)
                        jobs.append(([dict(src=Src(data, cw, ch, m(ow, oh), fmt=fmt))], 0, slot))
                        slot += 1
                    continue
                jobs.append((random_layers(r, ow, oh, 1, planar), 0, slot))
            else:
                jobs.append((random_layers(r, ow, oh, int(r.integers(1, 6)), planar), int(r.choice([0, 0, 0, 1, 3])), slot))
            slot += 1
        check_batch(jobs, ow, oh, "random call %d: %dx%d, %d jobs" % (case, ow, oh, len(jobs)), specs=[("709", "709"), ("709", "2020")][case % 2])


def both_routes(fn):
    """run fn() with frames of enlarged clips made by every route there is: the one-launch form where it applies (clips in their wire formats:
    reader + 2 x 2-block compositor in one kernel; the default), read + 2 x 2-block compositor as two launches (option chan_enlarged = 2), and
    the channel kernel (chan_enlarged = 0)"""
    import hip_harness as hh
    k = hh.ctx()
    try:
        for on, name in ((1, "one-launch reader + 2x2-block compositor (where it applies)"), (2, "read + 2x2-block compositor, two launches"), (0, "channel kernel")):
            k.set_option("chan_enlarged", on)
            fn(name)
    finally:
        k.set_option("chan_enlarged", 1)


@pytest.mark.parametrize("sw,sh,ow,oh,interlace", [(128, 36, 192, 54, 0), (96, 30, 192, 54, 1), (96, 30, 192, 54, 3), (100, 24, 384, 33, 0), (128, 36, 1280, 10, 0),
                                                   (720, 48, 1920, 24, 0), (52, 7, 100, 9, 3)])
def test_enlarged_clips_on_both_routes(sw, sh, ow, oh, interlace):
    """the reference's everyday case - a clip smaller than its channel, uploaded at its own size, filling the frame through the Mixer's transform
    (ffmpegProducer.ts:395-442, mixer.ts:189-228): one clip, two clips of one size (one batched read), three of different sizes and
    placements - by ph_v210_read + ph_compose_up_write_v210 and by the channel kernel, each against the oracle's chain"""
    def clip(seed, w, h, **kw):
        return dict(src=Src(frames.v210_random(w, h, frames.layer_seed(seed, w + h), legal=bool(seed & 1)), w, h, m(ow, oh, **kw)))
    cases = [[clip(400, sw, sh)],
             [clip(401, sw, sh), clip(402, sw, sh, scale_x=0.8, scale_y=0.8, offset_x=0.1, offset_y=-0.1)],
             [clip(403, sw, sh), clip(404, sw // 2 // 2 * 2, max(2, sh // 2), scale_x=0.7, scale_y=0.9, offset_x=-0.15), clip(405, sw, sh, scale_x=0.6, scale_y=0.6, offset_y=0.2)]]
    for i, layers in enumerate(cases):
        both_routes(lambda route: check(layers, ow, oh, "%d enlarged clips %dx%d on %dx%d il %d by the %s" % (len(layers), sw, sh, ow, oh, interlace, route),
                                        interlace=interlace, poison_dst=bool(interlace)))


def test_enlarged_clips_at_full_size_and_in_a_batch_call():
    """1280 x 720 clips on a 1920 x 1080 channel at full size (both routes), and such frames among the jobs of ph_chan_compose_batch
    (they are made in their turn, on their own; the other jobs still share a launch)"""
    w, h, ow, oh = 1280, 720, 1920, 1080
    layers = [dict(src=Src(frames.v210_random(w, h, frames.layer_seed(410, l)), w, h, m(ow, oh, scale_x=1.0 - 0.2 * l, scale_y=1.0 - 0.2 * l))) for l in range(2)]
    both_routes(lambda route: check(layers, ow, oh, "two 720p clips on a 1080p channel by the %s" % route))
    sw, sh, ow, oh = 128, 36, 384, 54
    v = channel_variants(ow, oh, 420)
    small = [dict(src=Src(frames.v210_random(sw, sh, frames.layer_seed(421, 0)), sw, sh, m(ow, oh)))]
    both_routes(lambda route: check_batch([(v[0], 0, 0), (small, 0, 1), (v[1], 0, 2), (small, 1, 3), (small, 3, 3), (v[3], 0, 4)], ow, oh,
                                          "a batch call with frames of enlarged clips among its jobs, those by the %s" % route))


def test_channels_of_enlarged_clips_share_their_launches():
    """ph_chan_compose_batch with four channels that show clips of ONE size under ONE placement (one batched read, one launch of the
    2 x 2-block compositor for all four), a fifth under another placement, both fields of a sixth, and the same jobs with the option off"""
    sw, sh, ow, oh = 128, 36, 384, 54
    clip = lambda seed, **kw: [dict(src=Src(frames.v210_random(sw, sh, frames.layer_seed(seed, 0)), sw, sh, m(ow, oh, **kw)))]
    two = lambda seed: [dict(src=Src(frames.v210_random(sw, sh, frames.layer_seed(seed, l)), sw, sh, m(ow, oh, scale_x=1.0 - 0.3 * l, scale_y=1.0 - 0.3 * l))) for l in range(2)]
    jobs = [(clip(430 + c), 0, c) for c in range(4)] + [(clip(435, scale_x=0.8, scale_y=0.8), 0, 4), (clip(436), 1, 5), (clip(437), 3, 5)] + [(two(440 + c), 0, 6 + c) for c in range(3)]
    both_routes(lambda route: check_batch(jobs, ow, oh, "channels of enlarged clips in one call, by the %s" % route))


@pytest.mark.parametrize("fmt", ["yuv422p10", "yuv422p8", "yuv420p", "nv12"])
def test_planar_clips_at_their_own_scale(fmt):
    """a file decoder's frame on a channel of its format under the Mixer's default fill (ffmpegProducer.ts:398-412, mixer.ts:189-228) - the
    everyday case - shares its taps between the pixels of a pair and between neighbouring lanes as v210 clips do: alone, two such clips
    (the upper one moved by whole and by fractional pixels), under an inset, in both fields, on frames that do not fill the chip and
    on a 1280-wide channel (lines with tails on the output side)"""
    for w, h in ((384, 54), (720, 60), (1280, 18)):
        a, b = frames.pack_random(fmt, w, h, 600 + w), frames.pack_random(fmt, w, h, 601 + w)
        v = frames.v210_random(w // 2 // 6 * 6, h // 2 // 2 * 2, frames.layer_seed(97, w))
        fill = dict(src=Src(a, w, h, m(w, h), fmt=fmt))
        # (a frame of nothing but decoders' frames under the default fill is made by read + 2 x 2-block compositor unless the option is off)
        both_routes(lambda route: check([fill], w, h, "%s %dx%d under the default fill by the %s" % (fmt, w, h, route)))
        both_routes(lambda route: check([fill, dict(src=Src(b, w, h, m(w, h), fmt=fmt))], w, h, "%s: two clips under the default fill by the %s" % (fmt, route), specs=("709", "2020")))
        check([fill, dict(src=Src(b, w, h, m(w, h, offset_x=8.0 / w, offset_y=-4.0 / h), fmt=fmt))], w, h, "%s: a second clip moved by whole pixels" % fmt)
        check([fill, dict(src=Src(b, w, h, m(w, h, offset_x=0.3 / w + 0.25, offset_y=0.4 / h), fmt=fmt)),
               dict(src=Src(v, w // 2 // 6 * 6, h // 2 // 2 * 2, m(w, h, **PIP[2])))], w, h, "%s: a clip moved by a fraction of a pixel, a v210 inset on top" % fmt, specs=("709", "2020"))
        for interlace in (1, 3):
            check([fill], w, h, "%s %dx%d under the default fill, field %d" % (fmt, w, h, interlace), interlace=interlace, poison_dst=True)
        # a graphic with alpha over the clip, both of the channel's size; an enlarged clip under such a graphic
        g = frames.pack_random("bgra8", w, h, 610 + w)
        small = frames.pack_random(fmt, w // 2 // 2 * 2, h // 2 // 2 * 2, 611 + w)
        both_routes(lambda route: check([fill, dict(src=Src(g, w, h, m(w, h), fmt="bgra8"))], w, h, "%s clip under a bgra8 graphic by the %s" % (fmt, route)))
        both_routes(lambda route: check([dict(src=Src(small, w // 2 // 2 * 2, h // 2 // 2 * 2, m(w, h), fmt=fmt)), dict(src=Src(g, w, h, m(w, h), fmt="bgra8"))], w, h,
                                        "an enlarged %s clip under a bgra8 graphic by the %s" % (fmt, route)))


def test_random_channel_programs_with_planar_clips():
    """the random channels of test_random_channel_programs with file decoders' frames among the sources (yuv422p10 / yuv422p8 / yuv420p / nv12:
    the kernel's planar instantiations - everything, clips only, clips with shared taps), fields included"""
    r = np.random.default_rng(int(os.environ.get("PH_FUZZ_SEED", "20261001")))
    sizes = [(192, 2), (192, 10), (384, 33), (576, 18), (768, 6), (960, 20), (100, 8)]
    for case in range(int(os.environ.get("PH_FUZZ_CASES", "21"))):
        ow, oh = sizes[case % len(sizes)]
        interlace = int(r.choice([0, 0, 1, 3]))
        layers = random_layers(r, ow, oh, int(r.integers(1, 6)), with_planar=True)
        if case % 6 == 3:  # graphics over v210 clips, some at the channel's size: the graphics' instantiation
            layers = [dict(src=Src(frames.v210_random(ow, oh, frames.layer_seed(7100, case)), ow, oh, m(ow, oh)))]
            for i in range(int(r.integers(1, 4))):
                f, full = str(r.choice(["rgba8", "bgra8"])), r.random() < 0.6
                gw, gh = (ow, oh) if full else (int(r.choice([48, 100, 192])), int(r.integers(2, 30)))
                kw = dict(offset_x=float(r.choice([0.0, 1.0, 0.4])) / ow, offset_y=float(r.choice([0.0, -1.0, 0.7])) / oh) if full else dict(scale_x=0.4, scale_y=0.5, offset_x=float(r.uniform(-0.4, 0.4)))
                layers.append(dict(src=Src(frames.pack_random(f, gw, gh, 7200 + case + i), gw, gh, m(ow, oh, **kw), fmt=f)))
        if case % 3 == 0:  # a program of planar clips at their own scale: the tap-sharing instantiation
            h2 = oh + (oh & 1)
            layers = [dict(src=Src(frames.pack_random(f, ow, h2 if f in ("yuv420p", "nv12") else oh, 7000 + case + i), ow, h2 if f in ("yuv420p", "nv12") else oh,
                                   m(ow, oh, offset_x=float(i) / ow, offset_y=0.5 * i / oh), fmt=f)) for i, f in enumerate(r.choice(["yuv422p10", "yuv420p", "nv12", "yuv422p8"], 2))]
        check(layers, ow, oh, "random channel with planar clips %d: %dx%d il %d, %d layers" % (case, ow, oh, interlace, len(layers)), interlace=interlace,
              specs=[("709", "709"), ("709", "2020")][case % 2], poison_dst=bool(interlace))


@pytest.mark.parametrize("fmt", ["yuv422p10", "yuv422p8", "yuv420p", "nv12", "rgba8", "bgra8"])
def test_enlarged_decoder_frames_on_both_routes(fmt):
    """a file smaller than its channel - a 720p H.264 clip (yuv420p) on a 1080 channel - by the reader of its format (ph_pack_read) + the
    2 x 2-block compositor and by the channel kernel: one clip, two clips, a clip beside a v210 clip, fields, a 1280-wide channel"""
    for (sw, sh, ow, oh) in ((128, 36, 192, 54), (200, 30, 384, 54), (96, 24, 1280, 30)):
        a, b = frames.pack_random(fmt, sw, sh, 800 + sw), frames.pack_random(fmt, sw, sh, 801 + sw)
        v = frames.v210_random(sw // 6 * 6, sh, frames.layer_seed(95, sw))
        one = [dict(src=Src(a, sw, sh, m(ow, oh), fmt=fmt))]
        two = one + [dict(src=Src(b, sw, sh, m(ow, oh, scale_x=0.8, scale_y=0.8, offset_x=0.1), fmt=fmt))]
        mixed = one + [dict(src=Src(v, sw // 6 * 6, sh, m(ow, oh, scale_x=0.7, scale_y=0.7, offset_y=-0.1)))]
        for what, layers in (("one", one), ("two", two), ("beside a v210 clip", mixed)):
            both_routes(lambda route: check(layers, ow, oh, "%s enlarged %s clip(s) %dx%d on %dx%d by the %s" % (what, fmt, sw, sh, ow, oh, route)))
        both_routes(lambda route: check(two, ow, oh, "two enlarged %s clips, field 3, by the %s" % (fmt, route), interlace=3, poison_dst=True, specs=("709", "2020")))


@pytest.mark.parametrize("out", ["yuv422p8", "rgba8", "yuv422p10"])
@pytest.mark.parametrize("src", ["v210", "yuv420p", "yuv422p10"])
def test_other_consumers_frames_from_every_kind_of_program(src, out):
    """the encoder's and the screen's frames (ffmpegConsumer.ts:144 yuv422p8, screenConsumer.ts:131 rgba8; yuv422p10 for comparison: the
    "everything" instantiation) from programs of v210 clips, of planar clips under the default fill (shared taps) and of placed planar
    clips: the kernel's lean instantiations with another writer behind them"""
    w, h = 384, 54
    def clip(seed, ww, hh_, **kw):
        data = frames.v210_random(ww, hh_, frames.layer_seed(seed, 0)) if src == "v210" else frames.pack_random(src, ww, hh_, seed)
        return dict(src=Src(data, ww, hh_, m(w, h, **kw), fmt=src))
    for interlace in (0, 3):
        check_format([clip(900, w, h)], w, h, out, "%s clip under the default fill -> %s il %d" % (src, out, interlace), interlace=interlace)
        check_format([clip(901, w, h), clip(902, 192, 30, **PIP[1]), clip(903, 192, 30, scale_x=0.4, scale_y=0.4, rotate=0.1, offset_x=0.2)], w, h, out,
                     "%s clips placed -> %s il %d" % (src, out, interlace), interlace=interlace)


def test_finished_images_on_both_routes():
    """f32 images among the layers of a frame the compositor route takes (de-interlaced fields at their own size: 1080i sources on a 1080 channel;
    an enlarged image; an image beside an enlarged clip): the compositor reads them as they are - against the oracle's chain, both routes"""
    w, h = 384, 54
    img = lambda seed, ww=w, hh_=h: Src(frames.rgba_random(ww, hh_, seed, -0.05, 1.05), ww, hh_, m(w, h), fmt="rgba")
    clip = Src(frames.v210_random(192, 30, frames.layer_seed(94, 0)), 192, 30, m(w, h, scale_x=0.9, scale_y=0.9))
    cases = [("four images under the default fill", [dict(src=img(950 + l)) for l in range(4)]),
             ("an enlarged image", [dict(src=img(955, 128, 36))]),
             ("an image under the default fill beside an enlarged v210 clip", [dict(src=img(956)), dict(src=clip)]),
             ("an enlarged yuv420p clip under an image", [dict(src=Src(frames.pack_random("yuv420p", 192, 30, 957), 192, 30, m(w, h), fmt="yuv420p")), dict(src=img(958))])]
    for what, layers in cases:
        both_routes(lambda route: check(layers, w, h, "%s by the %s" % (what, route), specs=("709", "2020")))
    both_routes(lambda route: check(cases[1][1], w, h, "an enlarged image, field 3, by the %s" % route, interlace=3, poison_dst=True))


@pytest.mark.parametrize("fmt", ["rgba8", "bgra8"])
def test_graphics_over_v210_clips(fmt):
    """a logo / lower third / full-frame graphic with alpha (a packed-RGB frame) over a live or v210 clip - the instantiation for v210 clips,
    graphics and images, in which a graphic of the channel's size shares its taps (its alpha travels through a halo slot of its own): the
    graphic under the default fill, moved by whole and by fractional pixels, two graphics, a small placed one, over an image, fields,
    frames that do not fill the chip and a 1280-wide channel"""
    for w, h in ((384, 54), (720, 60), (1280, 18)):
        clip = dict(src=Src(frames.v210_random(w, h, frames.layer_seed(93, w)), w, h, m(w, h)))
        g, g2 = frames.pack_random(fmt, w, h, 970 + w), frames.pack_random(fmt, w, h, 971 + w)
        small = frames.pack_random(fmt, 100, 20, 972 + w)
        check([clip, dict(src=Src(g, w, h, m(w, h), fmt=fmt))], w, h, "%s graphic %dx%d over a v210 clip, both under the default fill" % (fmt, w, h))
        check([clip, dict(src=Src(g, w, h, m(w, h, offset_x=6.0 / w, offset_y=-2.0 / h), fmt=fmt))], w, h, "%s graphic moved by whole pixels" % fmt, specs=("709", "2020"))
        check([clip, dict(src=Src(g, w, h, m(w, h, offset_x=0.3 / w, offset_y=0.6 / h), fmt=fmt)), dict(src=Src(g2, w, h, m(w, h), fmt=fmt))], w, h,
              "two %s graphics, one moved by a fraction of a pixel" % fmt)
        check([clip, dict(src=Src(small, 100, 20, m(w, h, scale_x=0.3, scale_y=0.4, offset_x=0.3, offset_y=0.3, rotate=0.05), fmt=fmt)), dict(src=Src(g, w, h, m(w, h), fmt=fmt))], w, h,
              "a small placed %s graphic under a full-frame one" % fmt)
        check([dict(src=Src(frames.rgba_random(w, h, 973 + w, -0.05, 1.05), w, h, fmt="rgba")), dict(src=Src(g, w, h, m(w, h), fmt=fmt))], w, h, "%s graphic over an image" % fmt)
        for interlace in (1, 3):
            check([clip, dict(src=Src(g, w, h, m(w, h), fmt=fmt))], w, h, "%s graphic over a v210 clip, field %d" % (fmt, interlace), interlace=interlace, poison_dst=True)
    check_format([dict(src=Src(frames.v210_random(384, 54, frames.layer_seed(92, 0)), 384, 54, m(384, 54))), dict(src=Src(frames.pack_random(fmt, 384, 54, 975), 384, 54, m(384, 54), fmt=fmt))],
                 384, 54, "yuv422p8", "%s graphic over a v210 clip into the encoder's frame" % fmt)


def test_channels_of_v210_clips_under_the_default_fill_share_the_route():
    """several channels each showing a v210 clip of the channel's size under the default fill (live sources): in one call their frames go by one
    batched read and one launch of the 2 x 2-block compositor; one such frame alone, or between frames of other shapes, stays the channel kernel's.
    Every frame against the oracle's chain and against its own single call (the channel kernel), with the option on and off"""
    w, h = 384, 54
    live = lambda seed: [dict(src=Src(frames.v210_random(w, h, frames.layer_seed(seed, 0), legal=bool(seed & 1)), w, h, m(w, h)))]
    two = lambda seed: [dict(src=Src(frames.v210_random(w, h, frames.layer_seed(seed, l)), w, h, m(w, h))) for l in range(2)]
    v = channel_variants(w, h, 460)
    jobs = [(live(461), 0, 0), (live(462), 0, 1), (live(463), 0, 2), (v[0], 0, 3), (live(464), 0, 4), (v[3], 0, 5), (two(465), 0, 6), (two(467), 0, 7),
            (live(469), 0, 8), (live(470), 0, 9), (live(471), 0, 10), (live(472), 0, 11), (live(473), 0, 12)]
    both_routes(lambda route: check_batch(jobs, w, h, "channels of v210 clips under the default fill, those by the %s" % route))
    w, h = 1920, 1080
    both_routes(lambda route: check_batch([(live(480 + c), 0, c) for c in range(4)], w, h, "four 1080p channels of live clips by the %s" % route))


def test_chan_batch_keeps_its_tables_across_the_jobs_it_hands_on():
    """regression (found by the seeded campaign, seed 31 case 62): ph_chan_compose_batch makes its launches' arguments late, after jobs it
    handed on - a frame of an enlarged image to the compositor (one table looked up), a single job to the one-job kernel (two) - and
    must still name ITS reader and writer tables in them: an enlarged f32 field, a frame on its own, then two fields in one launch"""
    w, h = 576, 16
    img = dict(src=Src(frames.rgba_random(288, 2, 990, -0.05, 1.05), 288, 2, m(w, h), fmt="rgba"))
    v = channel_variants(w, h, 991)
    jobs = [([img], 1, 0), (v[2], 3, 0), (v[1], 0, 1), (v[0], 1, 2), (v[3], 3, 2), ([img], 0, 3), (v[0], 0, 4), (v[1], 0, 5)]
    both_routes(lambda route: check_batch(jobs, w, h, "jobs handed on between launches, images by the %s" % route, specs=("709", "2020")))
