"""Golden kernel cases: one table drives gen_golden.py (reference run), the oracle tests
(CPU) and the HIP parity tests (GPU).  Inputs are regenerated from seeds (tests/frames.py);
kernels.npz holds only the reference OUTPUT of each case under its `name`.

Colour parameters (matrices, LUTs) are taken from the reference's own host maths run under
node at generation time; tests take them from whichever implementation they are checking -
host_maths.json pins those separately.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import frames  # noqa: E402

POISON = 0xDEADBEEF  # pre-fill of v210 write destinations: untouched lines must keep it


def _c(name, op, **kw):
    d = dict(name=name, op=op)
    d.update(kw)
    return d


CASES = [
    # ---- v210 read (v210.ts:25-111) ----------------------------------------------------------
    _c("read_ramp_1920x2_709", "v210_read", w=1920, h=2, src="ramp", spec="709", out_spec="709"),
    _c("read_rand_96x4_709_2020_illegal", "v210_read", w=96, h=4, src="rand_full", seed=11, spec="709", out_spec="2020"),
    _c("read_rand_1280x3_601_709", "v210_read", w=1280, h=3, src="rand_legal", seed=12, spec="601-625", out_spec="709"),
    _c("read_rand_100x2_2020_709_tail4", "v210_read", w=100, h=2, src="rand_legal", seed=13, spec="2020", out_spec="709"),
    _c("read_rand_3840x1_2020_2020", "v210_read", w=3840, h=1, src="rand_legal", seed=14, spec="2020", out_spec="2020"),
    _c("read_rand_48x5_sRGBlut_709", "v210_read", w=48, h=5, src="rand_full", seed=15, spec="sRGB", out_spec="709"),
    # ---- v210 write (v210.ts:113-195) ---------------------------------------------------------
    _c("write_rand_1920x2_709", "v210_write", w=1920, h=2, seed=21, lo=-0.1, hi=1.1, spec="709", interlace=0),
    _c("write_rand_96x6_709_top", "v210_write", w=96, h=6, seed=22, lo=0.0, hi=1.0, spec="709", interlace=1),
    _c("write_rand_96x6_2020_bottom", "v210_write", w=96, h=6, seed=23, lo=0.0, hi=1.0, spec="2020", interlace=3),
    _c("write_rand_100x1_709_tail4", "v210_write", w=100, h=1, seed=24, lo=0.0, hi=1.0, spec="709", interlace=0),
    _c("write_rand_1280x1_601_tail2", "v210_write", w=1280, h=1, seed=25, lo=0.0, hi=1.0, spec="601_525", interlace=0),
    _c("write_rand_3840x1_2020", "v210_write", w=3840, h=1, seed=26, lo=-0.5, hi=1.5, spec="2020", interlace=0),
    _c("write_specials_96x4_709", "v210_write", w=96, h=4, seed=27, lo=0.0, hi=1.0, spec="709", interlace=0, specials=True),
    # ---- yadif (yadifCl.ts:105-167) -----------------------------------------------------------
] + [
    _c("yadif_64x16_p%d_t%d_s%d" % (p, t, s), "yadif", w=64, h=16, seed=31, parity=p, tff=t, skip=s)
    for p in (0, 1) for t in (0, 1) for s in (0, 1)
] + [
    _c("yadif_9x5_p1_t1_s0_edges", "yadif", w=9, h=5, seed=32, parity=1, tff=1, skip=0),
    _c("yadif_5x3_p0_t1_s0_edges", "yadif", w=5, h=3, seed=33, parity=0, tff=1, skip=0),
    # ---- transform (transform.ts:36-59); `tp` indexes host_maths.json["transform"] ------------
    _c("transform_identity_48x27", "transform", iw=48, ih=27, ow=48, oh=27, seed=41, tp=0, mw=1920, mh=1080),
    _c("transform_upscale2_32x18", "transform", iw=32, ih=18, ow=64, oh=36, seed=42, tp=0, mw=1920, mh=1080),
    _c("transform_pip_48x27", "transform", iw=48, ih=27, ow=64, oh=36, seed=43, tp=1, mw=3840, mh=2160),
    _c("transform_pip_anchor_48x27", "transform", iw=48, ih=27, ow=48, oh=27, seed=44, tp=2, mw=1920, mh=1080),
    _c("transform_rot45_48x27", "transform", iw=48, ih=27, ow=48, oh=27, seed=45, tp=3, mw=1920, mh=1080),
    _c("transform_rot_scale_48x27", "transform", iw=48, ih=27, ow=40, oh=30, seed=46, tp=4, mw=1920, mh=1080),
    _c("transform_fliph_40x24", "transform", iw=40, ih=24, ow=40, oh=24, seed=47, tp=5, mw=1280, mh=720),
    _c("transform_flipv_rot_40x24", "transform", iw=40, ih=24, ow=40, oh=24, seed=48, tp=6, mw=1280, mh=720),
    _c("transform_flip_zoom_64x36", "transform", iw=64, ih=36, ow=64, oh=36, seed=49, tp=7, mw=64, mh=36),
    # ---- resize (resize.ts:35-59) --------------------------------------------------------------
    _c("resize_unity_48x27", "resize", iw=48, ih=27, ow=48, oh=27, seed=51, scale=1.0, ox=0.0, oy=0.0, fh=0, fv=0),
    _c("resize_half_off_48x27", "resize", iw=48, ih=27, ow=48, oh=27, seed=52, scale=0.5, ox=0.25, oy=-0.125, fh=0, fv=0),
    _c("resize_x2_32x18", "resize", iw=32, ih=18, ow=64, oh=36, seed=53, scale=1.0, ox=0.0, oy=0.0, fh=0, fv=0),
    _c("resize_zoom_fliph_48x27", "resize", iw=48, ih=27, ow=48, oh=27, seed=54, scale=2.0, ox=-0.5, oy=0.5, fh=1, fv=0),
    _c("resize_flipv_48x27", "resize", iw=48, ih=27, ow=40, oh=20, seed=55, scale=1.25, ox=1.0, oy=-1.0, fh=0, fv=1),
    _c("resize_flip_both_48x27", "resize", iw=48, ih=27, ow=48, oh=27, seed=56, scale=0.75, ox=0.1, oy=0.2, fh=1, fv=1),
    # ---- combine_N (combine.ts:24-68) ----------------------------------------------------------
] + [
    _c("combine_%d_32x16" % n, "combine", w=32, h=16, seed=60 + n, n=n) for n in (2, 3, 4, 5, 8)
] + [
    _c("combine_4_opaque_top_32x16", "combine", w=32, h=16, seed=69, n=4, alpha=1.0),
    # ---- transitions / mix / wipe --------------------------------------------------------------
    _c("dissolve_0.3_32x16", "dissolve", w=32, h=16, seed=71, mix=0.3),
    _c("dissolve_0_32x16", "dissolve", w=32, h=16, seed=72, mix=0.0),
    _c("dissolve_1_32x16", "dissolve", w=32, h=16, seed=73, mix=1.0),
    _c("dissolve_frame7of24_32x16", "dissolve", w=32, h=16, seed=74, mix=1.0 - 7 / 24),
    _c("twipe_ramp_32x16", "twipe", w=32, h=16, seed=75, mask="ramp"),
    _c("twipe_rand_32x16", "twipe", w=32, h=16, seed=76, mask="rand"),
    _c("mixer_0.62_32x16", "mixer", w=32, h=16, seed=77, mix=0.62),
    _c("wipe_0.37_32x16", "wipe", w=32, h=16, seed=78, wipe=0.37),
    _c("wipe_0_32x16", "wipe", w=32, h=16, seed=79, wipe=0.0),
    _c("wipe_1_32x16", "wipe", w=32, h=16, seed=80, wipe=1.0),
    _c("wipe_0.5_33x7", "wipe", w=33, h=7, seed=81, wipe=0.5),
]

# ---- the other pack formats: read and write, full-range random planes, tail widths ----
FMT_SPECS = {"yuv422p10": ("709", "709"), "yuv422p8": ("601-625", "709"), "yuv420p": ("709", "2020"),
             "nv12": ("709", "709"), "rgba8": ("sRGB", "709"), "bgra8": ("sRGB", "sRGB")}
for _i, (_f, (_sp, _osp)) in enumerate(FMT_SPECS.items()):
    _widths = (128, 64) if _f in ("rgba8", "bgra8") else (128, 78, 76, 74)   # 78 % 8 = 6, 76 % 8 = 4, 74 % 8 = 2
    for _w in _widths:
        CASES.append(_c("fmt_read_%s_%dx4" % (_f, _w), "pack_read", fmt=_f, w=_w, h=4, seed=300 + 10 * _i, spec=_sp,
                        out_spec=_osp))
        for _il in (0, 1, 3):
            if _il and _w != _widths[0]:
                continue
            CASES.append(_c("fmt_write_%s_%dx4_il%d" % (_f, _w, _il), "pack_write", fmt=_f, w=_w, h=4,
                            seed=400 + 10 * _i + _il, lo=-0.1, hi=1.1, spec=_osp, interlace=_il))

CASES.append(_c("fmt_write_yuv422p10_96x4_specials", "pack_write", fmt="yuv422p10", w=96, h=4, seed=499, lo=0.0, hi=1.0,
                spec="709", interlace=0, specials=True))
CASES.append(_c("fmt_write_rgba8_128x4_specials", "pack_write", fmt="rgba8", w=128, h=4, seed=498, lo=0.0, hi=1.0,
                spec="709", interlace=0, specials=True))

BY_NAME = {c["name"]: c for c in CASES}


def v210_source(c):
    if c["src"] == "ramp":
        return frames.v210_ramp(c["w"], c["h"])
    return frames.v210_random(c["w"], c["h"], c["seed"], legal=(c["src"] == "rand_legal"))


def inputs(c):
    """Regenerate the input arrays of a case (dict of numpy arrays)."""
    op = c["op"]
    if op == "v210_read":
        return dict(words=v210_source(c))
    if op == "v210_write":
        dst = np.full(frames.v210_pitch_bytes(c["w"]) * c["h"] // 4, POISON, np.uint32)
        rgba = frames.rgba_specials(c["w"], c["h"], c["seed"]) if c.get("specials") else \
            frames.rgba_random(c["w"], c["h"], c["seed"], c["lo"], c["hi"])
        return dict(rgba=rgba, dst=dst)
    if op == "pack_read":
        return dict(planes=frames.pack_random(c["fmt"], c["w"], c["h"], c["seed"]))
    if op == "pack_write":
        dst = [np.full(n, 0xA5, np.uint8) for n in frames.pack_plane_bytes(c["fmt"], c["w"], c["h"])]
        rgba = frames.rgba_specials(c["w"], c["h"], c["seed"]) if c.get("specials") else \
            frames.rgba_random(c["w"], c["h"], c["seed"], c["lo"], c["hi"])
        return dict(rgba=rgba, dst=dst)
    if op == "yadif":
        s = c["seed"] * 1000
        return dict(prev=frames.rgba_random(c["w"], c["h"], s + 1), cur=frames.rgba_random(c["w"], c["h"], s + 2),
                    next=frames.rgba_random(c["w"], c["h"], s + 3))
    if op in ("transform", "resize"):
        return dict(img=frames.rgba_random(c["iw"], c["ih"], c["seed"] * 1000))
    if op == "combine":
        return dict(layers=[frames.rgba_random(c["w"], c["h"], c["seed"] * 1000 + i, alpha=c.get("alpha"))
                            for i in range(c["n"])])
    if op in ("dissolve", "mixer", "wipe"):
        s = c["seed"] * 1000
        return dict(in0=frames.rgba_random(c["w"], c["h"], s + 1), in1=frames.rgba_random(c["w"], c["h"], s + 2))
    if op == "twipe":
        s = c["seed"] * 1000
        mask = frames.mask_ramp(c["w"], c["h"]) if c["mask"] == "ramp" else frames.rgba_random(c["w"], c["h"], s + 3)
        return dict(in0=frames.rgba_random(c["w"], c["h"], s + 1), in1=frames.rgba_random(c["w"], c["h"], s + 2),
                    mask=mask)
    raise KeyError(op)
