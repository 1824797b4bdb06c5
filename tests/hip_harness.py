"""Helpers for the -m gpu tests: run cases through the C ABI on cuda:0 with torch as the
device-memory plumbing.  Colour parameters come from the product's own host maths
(ph_colour_*), never from the oracle."""
import numpy as np
import torch

from phaneron_amd import capi

_ctx = None


class _TorchOrdered:
    """The library launches on its OWN streams; torch fills / copies / random generators run on
    torch's.  Nothing orders the two, so a `torch.zeros` output can be zeroed AFTER the kernel under
    test wrote it.  Every call made through this proxy first drains the device (the product does not
    need this: its callers own their ordering, e.g. bench.py synchronises once before timing).

    The launches are asynchronous, so the tensors handed to a call must outlive it: a temporary such as
    `k.v210_read(hh.dev(frame), ...)` would otherwise go back to torch's caching allocator the moment the call returns
    and be handed out - and overwritten - by the next `hh.dev()` while the kernel is still reading it (seen once as
    a flaky 1080p chain test).  The proxy therefore holds on to the arguments of a call until the device has been
    drained again."""

    def __init__(self, c):
        self._c = c
        self._alive = []

    def __getattr__(self, name):
        attr = getattr(self._c, name)
        if not callable(attr):
            return attr

        def ordered(*args, **kw):
            torch.cuda.synchronize()
            del self._alive[:]
            self._alive.append((args, kw))
            return attr(*args, **kw)
        return ordered


def ctx():
    global _ctx
    if _ctx is None:
        _ctx = _TorchOrdered(capi.Context(0))
    return _ctx


def dev(a):
    """numpy -> cuda tensor (bit-preserving: uint32 travels as int32)."""
    a = np.ascontiguousarray(a)
    if a.dtype == np.uint32:
        a = a.view(np.int32)
    t = torch.from_numpy(a).cuda()
    torch.cuda.synchronize()
    return t


def host(t, dtype=None):
    ctx().wait()
    torch.cuda.synchronize()
    a = t.cpu().numpy()
    return a.view(dtype) if dtype is not None else a


class ColourParams:
    """Device-resident matrices / LUTs for a (read spec -> working spec -> write spec) set."""
    _cache = {}

    @classmethod
    def reader(cls, spec, out_spec):
        key = ("r", spec, out_spec)
        if key not in cls._cache:
            lut = capi.gamma2linear_lut(spec)
            dlut = dev(lut)
            ctx().register_lut(dlut, lut)  # lets the library keep the exact LDS form of the table
            cls._cache[key] = (dev(capi.ycbcr2rgb_matrix(spec)), dlut,
                               dev(np.concatenate([capi.rgb2rgb_matrix(spec, out_spec), np.zeros(3, np.float32)])))
        return cls._cache[key]

    @classmethod
    def fmt_reader(cls, fmt, spec, out_spec):
        """(colMatrix or None, lut, gamut) for a pack format: matrices use the format's own code ranges"""
        key = ("fr", fmt, spec, out_spec)
        if key not in cls._cache:
            rng = capi.FORMAT_RANGE[fmt]
            _, lut, _ = cls.reader(spec, out_spec)
            cm = None if rng is None else dev(capi.ycbcr2rgb_matrix(spec, *rng))
            cls._cache[key] = (cm, lut, dev(np.concatenate([capi.rgb2rgb_matrix(spec, out_spec), np.zeros(3, np.float32)])))
        return cls._cache[key]

    @classmethod
    def fmt_writer(cls, fmt, spec):
        key = ("fw", fmt, spec)
        if key not in cls._cache:
            rng = capi.FORMAT_RANGE[fmt]
            _, lut = cls.writer(spec)
            cls._cache[key] = (None if rng is None else dev(capi.rgb2ycbcr_matrix(spec, *rng)), lut)
        return cls._cache[key]

    @classmethod
    def writer(cls, spec):
        key = ("w", spec)
        if key not in cls._cache:
            lut = capi.linear2gamma_lut(spec)
            dlut = dev(lut)
            ctx().register_lut(dlut, lut)
            cls._cache[key] = (dev(capi.rgb2ycbcr_matrix(spec)), dlut)
        return cls._cache[key]


def run_case(c, inp, hm=None):
    """Run one tests/golden/cases.py case on the GPU; returns a numpy array shaped like the golden."""
    k = ctx()
    op = c["op"]
    if op == "v210_read":
        cm, lut, gm = ColourParams.reader(c["spec"], c["out_spec"])
        out = torch.zeros(c["h"] * c["w"] * 4, dtype=torch.float32, device="cuda")
        src = dev(inp["words"])
        k.v210_read(src, out, c["w"], c["h"], cm, lut, gm)
        return host(out)
    if op == "v210_write":
        cm, lut = ColourParams.writer(c["spec"])
        out = dev(inp["dst"])
        src = dev(inp["rgba"])
        k.v210_write(src, out, c["w"], c["h"], c["interlace"], cm, lut)
        return host(out, np.uint32)
    if op == "pack_read":
        cm, lut, gm = ColourParams.fmt_reader(c["fmt"], c["spec"], c["out_spec"])
        out = torch.zeros(c["h"] * c["w"] * 4, dtype=torch.float32, device="cuda")
        planes = [dev(p) for p in inp["planes"]]
        k.pack_read(c["fmt"], planes, out, c["w"], c["h"], cm, lut, gm)
        return host(out)
    if op == "pack_write":
        cm, lut = ColourParams.fmt_writer(c["fmt"], c["spec"])
        planes = [dev(p) for p in inp["dst"]]
        src = dev(inp["rgba"])
        k.pack_write(c["fmt"], src, planes, c["w"], c["h"], c["interlace"], cm, lut)
        return np.concatenate([host(p) for p in planes])
    if op == "yadif":
        out = torch.zeros(c["h"] * c["w"] * 4, dtype=torch.float32, device="cuda")
        p, cu, n = dev(inp["prev"]), dev(inp["cur"]), dev(inp["next"])
        k.yadif(p, cu, n, out, c["w"], c["h"], c["parity"], c["tff"], c["skip"])
        return host(out)
    if op == "transform":
        t = hm["transform"][c["tp"]]
        p = t["params"]
        m = capi.transform_matrix(c["mw"], c["mh"], p.get("flipH", False), p.get("flipV", False),
                                  p.get("anchorX", 0.0), p.get("anchorY", 0.0), p.get("scaleX", 1.0),
                                  p.get("scaleY", 1.0), p.get("offsetX", 0.0), p.get("offsetY", 0.0),
                                  p.get("rotate", 0.0))
        out = torch.zeros(c["oh"] * c["ow"] * 4, dtype=torch.float32, device="cuda")
        src, md = dev(inp["img"]), dev(m)
        k.transform(src, c["iw"], c["ih"], md, out, c["ow"], c["oh"])
        return host(out)
    if op == "resize":
        flip = np.array([1.0 if c["fh"] else 0.0, -1.0 if c["fh"] else 1.0, 1.0 if c["fv"] else 0.0,
                         -1.0 if c["fv"] else 1.0], np.float32)
        out = torch.zeros(c["oh"] * c["ow"] * 4, dtype=torch.float32, device="cuda")
        src, fd = dev(inp["img"]), dev(flip)
        k.resize(src, c["iw"], c["ih"], c["scale"], c["ox"], c["oy"], fd, out, c["ow"], c["oh"])
        return host(out)
    if op == "combine":
        ls = [dev(l) for l in inp["layers"]]
        out = torch.zeros(c["h"] * c["w"] * 4, dtype=torch.float32, device="cuda")
        k.combine(ls, out, c["w"], c["h"])
        return host(out)
    if op in ("dissolve", "mixer", "wipe"):
        a, b = dev(inp["in0"]), dev(inp["in1"])
        out = torch.zeros(c["h"] * c["w"] * 4, dtype=torch.float32, device="cuda")
        if op == "dissolve":
            k.transition_dissolve(a, b, c["mix"], out, c["w"], c["h"])
        elif op == "mixer":
            k.mixer(a, b, c["mix"], out, c["w"], c["h"])
        else:
            k.wipe(a, b, c["wipe"], out, c["w"], c["h"])
        return host(out)
    if op == "twipe":
        a, b, m = dev(inp["in0"]), dev(inp["in1"]), dev(inp["mask"])
        out = torch.zeros(c["h"] * c["w"] * 4, dtype=torch.float32, device="cuda")
        k.transition_wipe(a, b, m, out, c["w"], c["h"])
        return host(out)
    raise KeyError(op)
