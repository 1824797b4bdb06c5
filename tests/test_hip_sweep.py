"""GPU test (-m gpu): seeded random sweeps of sizes and parameters, every result against the oracle
bit for bit.  The fixed cases of test_hip_parity.py pin the reference; these look for edges nobody
thought of: widths around the block / strip / chunk sizes of the kernels (250-column yadif strips,
192-pixel compositor chunks, 48-pixel v210 blocks, 8-pixel planar octets), odd heights, both fields,
every interlace mode, arbitrary transform parameters."""
import numpy as np
import pytest

import cases
import frames
from oracle import orc

pytestmark = pytest.mark.gpu


def bits_eq(got, want, what):
    g = np.ascontiguousarray(got).reshape(-1).view(np.uint32 if got.dtype.itemsize == 4 else got.dtype)
    w = np.ascontiguousarray(want).reshape(-1).view(g.dtype)
    assert g.shape == w.shape, (what, g.shape, w.shape)
    bad = np.flatnonzero(g != w)
    assert bad.size == 0, "%s: %d of %d differ, first at %d" % (what, bad.size, w.size, bad[0] if bad.size else -1)


def rng_for(name):
    return np.random.default_rng(sum(ord(c) * (i + 1) for i, c in enumerate(name)))  # stable across runs (hash() is salted)


def test_yadif_random_sizes():
    import torch
    import hip_harness as hh
    r = rng_for("yadif")
    sizes = [(250, 16), (251, 17), (249, 15), (500, 33), (501, 1), (1, 40), (7, 2), (256, 31), (750, 18)]
    sizes += [(int(r.integers(1, 700)), int(r.integers(1, 70))) for _ in range(8)]
    for (w, h) in sizes:
        p, c, n = (frames.rgba_random(w, h, 3000 + 7 * w + h + i) for i in range(3))
        dp, dc, dn = hh.dev(p), hh.dev(c), hh.dev(n)
        for parity in (0, 1):
            tff, skip = int(r.integers(0, 2)), bool(r.integers(0, 2))
            out = torch.zeros(w * h * 4, dtype=torch.float32, device="cuda")
            hh.ctx().yadif(dp, dc, dn, out, w, h, parity, tff, skip)
            bits_eq(hh.host(out), orc.yadif(p, c, n, parity, tff, skip), "yadif %dx%d p%d t%d s%d" % (w, h, parity, tff, skip))


def test_yadif_pair_random_sizes():
    import torch
    import hip_harness as hh
    r = rng_for("yadif pair")
    sizes = [(250, 16), (251, 17), (249, 15), (500, 33), (501, 1), (1, 40), (7, 2), (256, 31), (750, 18), (33, 3)]
    sizes += [(int(r.integers(1, 700)), int(r.integers(1, 70))) for _ in range(8)]
    for (w, h) in sizes:
        p, c, n = (frames.rgba_random(w, h, 3100 + 7 * w + h + i) for i in range(3))
        tff, skip = int(r.integers(0, 2)), bool(r.integers(0, 2))
        out = [torch.zeros(w * h * 4, dtype=torch.float32, device="cuda") for _ in range(2)]
        hh.ctx().yadif_pair(hh.dev(p), hh.dev(c), hh.dev(n), out[0], out[1], w, h, tff, skip)
        for parity in (0, 1):
            bits_eq(hh.host(out[parity]), orc.yadif(p, c, n, parity, tff, skip), "yadif_pair %dx%d p%d t%d s%d" % (w, h, parity, tff, skip))


def test_v210_read_write_random_sizes():
    import torch
    import hip_harness as hh
    r = rng_for("v210")
    widths = [48, 96, 6, 12, 90, 1278, 1280, 1920, 2, 4, 50, 100, 1000]
    for w in widths:
        h = int(r.integers(1, 9))
        words = frames.v210_random(w, h, 5000 + w, legal=bool(r.integers(0, 2)))
        cm, lut, gm = hh.ColourParams.reader("709", "2020")
        out = torch.zeros(w * h * 4, dtype=torch.float32, device="cuda")
        hh.ctx().v210_read(hh.dev(words), out, w, h, cm, lut, gm)
        bits_eq(hh.host(out), orc.v210_read(words, w, h, orc.ycbcr2rgb_matrix("709"), orc.gamma2linear_lut("709"),
                                            orc.rgb2rgb_matrix("709", "2020")), "v210_read %dx%d" % (w, h))
        if w % 48 and h > 1:
            continue  # the reference's write overlaps lines there (DESIGN deviations): single lines only
        for il in ((0,) if h < 2 else (0, 1, 3)):
            rgba = frames.rgba_random(w, h, 6000 + w + il, -0.1, 1.1)
            dst = np.full(frames.v210_pitch_bytes(w) * h // 4, cases.POISON, np.uint32)
            wcm, wlut = hh.ColourParams.writer("2020")
            o = hh.dev(dst)
            hh.ctx().v210_write(hh.dev(rgba), o, w, h, il, wcm, wlut)
            bits_eq(hh.host(o, np.uint32), orc.v210_write(rgba, w, h, il, orc.rgb2ycbcr_matrix("2020"),
                                                           orc.linear2gamma_lut("2020"), out=dst.copy()),
                    "v210_write %dx%d il%d" % (w, h, il))


def test_transform_random_parameters():
    import torch
    import hip_harness as hh
    from phaneron_amd import capi
    r = rng_for("transform")
    for k in range(12):
        iw, ih = int(r.integers(2, 200)), int(r.integers(2, 120))
        ow, oh = int(r.integers(1, 260)), int(r.integers(1, 140))
        kw = dict(flip_h=bool(r.integers(0, 2)), flip_v=bool(r.integers(0, 2)), anchor_x=float(r.uniform(-0.5, 0.5)),
                  anchor_y=float(r.uniform(-0.5, 0.5)), scale_x=float(r.uniform(0.2, 3.0)), scale_y=float(r.uniform(0.2, 3.0)),
                  offset_x=float(r.uniform(-1, 1)), offset_y=float(r.uniform(-1, 1)), rotate=float(r.uniform(-1, 1)))
        img = frames.rgba_random(iw, ih, 7000 + k)
        m = capi.transform_matrix(ow, oh, **kw)
        bits_eq(m, orc.transform_matrix(ow, oh, **kw), "matrix %d" % k)
        out = torch.zeros(ow * oh * 4, dtype=torch.float32, device="cuda")
        hh.ctx().transform(hh.dev(img), iw, ih, hh.dev(m), out, ow, oh)
        bits_eq(hh.host(out), orc.transform(img, m, ow, oh), "transform %d: %dx%d -> %dx%d %r" % (k, iw, ih, ow, oh, kw))


def test_compose_random_layers():
    import hip_harness as hh
    from phaneron_amd import capi
    r = rng_for("compose")
    hh.ctx().set_option("lds_lut", True)
    for k in range(8):
        ow, oh = 48 * int(r.integers(1, 9)), int(r.integers(2, 40))
        n = int(r.integers(1, 6))
        il = int(r.choice([0, 1, 3])) if oh >= 2 else 0
        specs = []
        for l in range(n):
            if r.integers(0, 3) == 0:
                specs.append((ow, oh, None))
            else:
                specs.append((int(r.integers(2, 300)), int(r.integers(2, 100)),
                              dict(scale_x=float(r.uniform(0.3, 2.0)), scale_y=float(r.uniform(0.3, 2.0)),
                                   offset_x=float(r.uniform(-0.5, 0.5)), offset_y=float(r.uniform(-0.5, 0.5)),
                                   rotate=float(r.uniform(-0.2, 0.2)))))
        imgs = [frames.rgba_random(w, h, 8000 + 10 * k + i) for i, (w, h, _) in enumerate(specs)]
        mats = [None if kw is None else capi.transform_matrix(ow, oh, **kw) for (_, _, kw) in specs]
        wcm, wlut = hh.ColourParams.writer("709")
        dst0 = np.full(frames.v210_pitch_bytes(ow) * oh // 4, cases.POISON, np.uint32)
        out = hh.dev(dst0)
        layers = [(hh.dev(im), w, h, None if m is None else hh.dev(m)) for im, (w, h, _), m in zip(imgs, specs, mats)]
        hh.ctx().compose_write_v210(layers, out, ow, oh, il, wcm, wlut)
        xf = [im if m is None else orc.transform(im, m, ow, oh) for im, m in zip(imgs, mats)]
        comb = xf[0] if len(xf) == 1 else orc.combine(xf)
        want = orc.v210_write(comb, ow, oh, il, orc.rgb2ycbcr_matrix("709"), orc.linear2gamma_lut("709"), out=dst0.copy())
        bits_eq(hh.host(out, np.uint32), want, "compose %d: %dx%d n=%d il=%d" % (k, ow, oh, n, il))


def test_compose_buffer_addressed_taps():
    """The compositor's buffer-addressed sampler (out_w % 192 == 0): border taps answered by the addressing hardware,
    row taps hoisted for unrotated placements, one shared placement for all layers, 1:1 layers mixed in - against the
    oracle chain, including placements that put most or all taps outside the source."""
    import hip_harness as hh
    from phaneron_amd import capi
    r = rng_for("compose taps")
    hh.ctx().set_option("lds_lut", True)
    wcm, wlut = hh.ColourParams.writer("2020")
    extreme = [dict(scale_x=0.02, scale_y=0.02), dict(scale_x=40.0, scale_y=40.0), dict(offset_x=3.0, offset_y=-2.5),
               dict(scale_x=1e-6, scale_y=1e6), dict(flip_h=True, flip_v=True, scale_x=0.7, scale_y=1.3),
               dict(rotate=0.25), dict(rotate=0.5, scale_x=2.0, scale_y=2.0), dict(offset_x=0.5, offset_y=0.5),
               dict(scale_x=1.0, scale_y=1.0, offset_x=-1.0 / 384, offset_y=1.0 / 24)]
    for k in range(14):
        ow, oh = 192 * int(r.integers(1, 4)), int(r.integers(2, 24))
        n = int(r.integers(1, 7))
        il = int(r.choice([0, 1, 3]))
        shared = k % 3 == 0 and n > 1          # all layers through ONE matrix buffer and one source size
        specs = []
        sw, sh = int(r.integers(2, 260)), int(r.integers(2, 90))
        kw_shared = extreme[k % len(extreme)] if k % 2 else dict(scale_x=float(r.uniform(0.5, 2.5)), scale_y=float(r.uniform(0.5, 2.5)))
        for l in range(n):
            if shared:
                specs.append((sw, sh, kw_shared))
            elif r.integers(0, 4) == 0:
                specs.append((ow, oh, None))
            elif r.integers(0, 2) == 0:
                specs.append((int(r.integers(1, 260)), int(r.integers(1, 90)), extreme[int(r.integers(0, len(extreme)))]))
            else:                                  # unrotated: the hoisted-row path when every sampled layer is
                specs.append((int(r.integers(2, 260)), int(r.integers(2, 90)),
                              dict(scale_x=float(r.uniform(0.3, 3.0)), scale_y=float(r.uniform(0.3, 3.0)),
                                   offset_x=float(r.uniform(-0.6, 0.6)), offset_y=float(r.uniform(-0.6, 0.6)))))
        imgs = [frames.rgba_random(w, h, 8600 + 10 * k + i) for i, (w, h, _) in enumerate(specs)]
        mats = [None if kw is None else capi.transform_matrix(ow, oh, **kw) for (_, _, kw) in specs]
        one = hh.dev(mats[0]) if shared else None
        dst0 = np.full(frames.v210_pitch_bytes(ow) * oh // 4, cases.POISON, np.uint32)
        out = hh.dev(dst0)
        layers = [(hh.dev(im), w, h, None if m is None else (one if shared else hh.dev(m))) for im, (w, h, _), m in zip(imgs, specs, mats)]
        hh.ctx().compose_write_v210(layers, out, ow, oh, il, wcm, wlut)
        xf = [im if m is None else orc.transform(im, m, ow, oh) for im, m in zip(imgs, mats)]
        comb = xf[0] if len(xf) == 1 else orc.combine(xf)
        want = orc.v210_write(comb, ow, oh, il, orc.rgb2ycbcr_matrix("2020"), orc.linear2gamma_lut("2020"), out=dst0.copy())
        bits_eq(hh.host(out, np.uint32), want, "compose taps %d: %dx%d n=%d il=%d shared=%s %r" % (k, ow, oh, n, il, shared, [s[2] for s in specs]))


def test_compose_with_wipes_inside():
    """ph_compose_wipe_write_v210: wipes on sampled and on 1:1 layers, several per frame, interlaced outputs - against
    transform -> transition_wipe -> combine -> write of the oracle."""
    import hip_harness as hh
    from phaneron_amd import capi
    r = rng_for("compose wipes")
    hh.ctx().set_option("lds_lut", True)
    wcm, wlut = hh.ColourParams.writer("709")
    for k in range(8):
        ow, oh = 192 * int(r.integers(1, 3)), int(r.integers(2, 20))
        n = int(r.integers(1, 6))
        il = int(r.choice([0, 1, 3]))
        specs, wipes = [], []
        for l in range(n):
            if r.integers(0, 3) == 0:
                specs.append((ow, oh, None))
            else:
                specs.append((int(r.integers(2, 200)), int(r.integers(2, 60)),
                              dict(scale_x=float(r.uniform(0.4, 2.5)), scale_y=float(r.uniform(0.4, 2.5)), offset_x=float(r.uniform(-0.4, 0.4)),
                                   offset_y=float(r.uniform(-0.4, 0.4)), rotate=float(r.choice([0.0, 0.0, 0.1])))))
            wipes.append(r.integers(0, 2) == 1 or (l == n - 1 and not any(wipes)))
        imgs = [frames.rgba_random(w, h, 8800 + 10 * k + i) for i, (w, h, _) in enumerate(specs)]
        incoming = [frames.rgba_random(ow, oh, 8900 + 10 * k + i, -0.05, 1.05) if wv else None for i, wv in enumerate(wipes)]
        masks = [frames.rgba_random(ow, oh, 8950 + 10 * k + i) if wv else None for i, wv in enumerate(wipes)]
        mats = [None if kw is None else capi.transform_matrix(ow, oh, **kw) for (_, _, kw) in specs]
        dst0 = np.full(frames.v210_pitch_bytes(ow) * oh // 4, cases.POISON, np.uint32)
        out = hh.dev(dst0)
        layers = [(hh.dev(im), w, h, None if m is None else hh.dev(m)) for im, (w, h, _), m in zip(imgs, specs, mats)]
        dw = [None if not wv else (hh.dev(incoming[i]), hh.dev(masks[i])) for i, wv in enumerate(wipes)]
        hh.ctx().compose_wipe_write_v210(layers, dw, out, ow, oh, il, wcm, wlut)
        xf = [im if m is None else orc.transform(im, m, ow, oh) for im, m in zip(imgs, mats)]
        xf = [orc.transition_wipe(x, incoming[i], masks[i]) if wipes[i] else x for i, x in enumerate(xf)]
        comb = xf[0] if len(xf) == 1 else orc.combine(xf)
        want = orc.v210_write(comb, ow, oh, il, orc.rgb2ycbcr_matrix("709"), orc.linear2gamma_lut("709"), out=dst0.copy())
        bits_eq(hh.host(out, np.uint32), want, "compose wipes %d: %dx%d n=%d il=%d wipes=%r" % (k, ow, oh, n, il, wipes))
    with pytest.raises(Exception, match="192"):
        t = hh.dev(frames.rgba_random(96, 4, 1))
        hh.ctx().compose_wipe_write_v210([(t, 96, 4, None)], [(t, t)], hh.dev(np.zeros(96 * 4, np.uint32)), 96, 4, 0, wcm, wlut)


@pytest.mark.parametrize("fmt", ["yuv422p10", "yuv422p8", "yuv420p", "nv12", "rgba8", "bgra8"])
def test_pack_formats_random_sizes(fmt):
    import torch
    import hip_harness as hh
    r = rng_for("pack" + fmt)
    rgb = fmt in ("rgba8", "bgra8")
    widths = [64, 128, 192] if rgb else [8, 16, 64, 70, 72, 74, 76, 78, 250, 256, 258, 1920]
    for w in widths:
        h = 2 * int(r.integers(1, 6))
        planes = frames.pack_random(fmt, w, h, 9000 + w)
        cm, lut, gm = hh.ColourParams.fmt_reader(fmt, "709", "2020")
        out = torch.zeros(w * h * 4, dtype=torch.float32, device="cuda")
        hh.ctx().pack_read(fmt, [hh.dev(p) for p in planes], out, w, h, cm, lut, gm)
        rng = orc.FORMAT_RANGE[fmt]
        ocm = None if rng is None else orc.ycbcr2rgb_matrix("709", *rng)
        bits_eq(hh.host(out), orc.pack_read(fmt, planes, w, h, ocm, orc.gamma2linear_lut("709"),
                                            orc.rgb2rgb_matrix("709", "2020")), "%s read %dx%d" % (fmt, w, h))
        for il in (0, 1, 3):
            rgba = frames.rgba_random(w, h, 9500 + w + il, -0.1, 1.1)
            wcm, wlut = hh.ColourParams.fmt_writer(fmt, "2020")
            dst = [np.full(nb, 0xA5, np.uint8) for nb in frames.pack_plane_bytes(fmt, w, h)]
            dplanes = [hh.dev(d) for d in dst]
            hh.ctx().pack_write(fmt, hh.dev(rgba), dplanes, w, h, il, wcm, wlut)
            owcm = None if rng is None else orc.rgb2ycbcr_matrix("2020", *rng)
            want = orc.pack_write(fmt, rgba, w, h, il, owcm, orc.linear2gamma_lut("2020"), planes=[d.copy() for d in dst])
            for i, (gp, wp) in enumerate(zip(dplanes, want)):
                bits_eq(hh.host(gp), wp, "%s write %dx%d il%d plane %d" % (fmt, w, h, il, i))


def test_special_float_values_in_write_paths():
    """NaN, +-Inf, -0, denormals and huge values in the float RGBA input of the writers: the reference's
    convert_ushort_sat_rte sends NaN to 0 and saturates the rest; the packed outputs must match the
    oracle exactly.  (Float outputs are compared with NaN == NaN: payload bits are not specified.)"""
    import torch
    import hip_harness as hh
    w, h = 96, 4
    rgba = frames.rgba_random(w, h, 4711, -0.2, 1.2)
    flat = rgba.reshape(-1)
    specials = np.array([np.nan, np.inf, -np.inf, -0.0, 1e-42, -1e-42, 3.0e38, -3.0e38, 1.0, 0.0, 0.5 / 65535, 1.5 / 65535,
                         2.5 / 65535, 65534.5 / 65535], np.float32)
    idx = np.random.default_rng(5).choice(flat.size, size=600, replace=False)
    flat[idx] = specials[np.arange(600) % specials.size]
    for il in (0, 1, 3):
        dst = np.full(frames.v210_pitch_bytes(w) * h // 4, cases.POISON, np.uint32)
        wcm, wlut = hh.ColourParams.writer("709")
        o = hh.dev(dst)
        hh.ctx().v210_write(hh.dev(rgba), o, w, h, il, wcm, wlut)
        bits_eq(hh.host(o, np.uint32), orc.v210_write(rgba, w, h, il, orc.rgb2ycbcr_matrix("709"), orc.linear2gamma_lut("709"),
                                                       out=dst.copy()), "v210_write specials il%d" % il)
    for fmt in ("yuv422p10", "nv12", "rgba8"):
        rng = orc.FORMAT_RANGE[fmt]
        wcm, wlut = hh.ColourParams.fmt_writer(fmt, "709")
        dst = [np.full(nb, 0xA5, np.uint8) for nb in frames.pack_plane_bytes(fmt, w, h)]
        dplanes = [hh.dev(d) for d in dst]
        hh.ctx().pack_write(fmt, hh.dev(rgba), dplanes, w, h, 0, wcm, wlut)
        want = orc.pack_write(fmt, rgba, w, h, 0, None if rng is None else orc.rgb2ycbcr_matrix("709", *rng),
                              orc.linear2gamma_lut("709"), planes=[d.copy() for d in dst])
        for i, (gp, wp) in enumerate(zip(dplanes, want)):
            bits_eq(hh.host(gp), wp, "%s write specials plane %d" % (fmt, i))
    # float kernels: same values, NaN positions must agree and every non-NaN word must be identical
    other = frames.rgba_random(w, h, 4712)
    out = torch.zeros(w * h * 4, dtype=torch.float32, device="cuda")
    for name, run, want in (
            ("combine", lambda: hh.ctx().combine([hh.dev(other), hh.dev(rgba), hh.dev(other)], out, w, h),
             orc.combine([other, rgba, other])),
            ("dissolve", lambda: hh.ctx().transition_dissolve(hh.dev(rgba), hh.dev(other), 0.3, out, w, h),
             orc.transition_dissolve(rgba, other, 0.3))):
        run()
        got = hh.host(out).reshape(-1)
        want = np.asarray(want, np.float32).reshape(-1)
        assert np.array_equal(np.isnan(got), np.isnan(want)), name
        ok = ~np.isnan(want)
        assert np.array_equal(got[ok].view(np.uint32), want[ok].view(np.uint32)), name
