#!/usr/bin/env python3
"""BASELINE config 3's two kernels timed alone: the de-interlacing reader (RGBA and packed-RGB fields) and the compositors
(pixel-per-lane on RGBA, 2 x 2 blocks on RGBA and on packed RGB).  python tools/up_bench.py [reps] [which]"""
import json
import os
import sys

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    import torch
    from phaneron_amd import capi
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    which = sys.argv[2] if len(sys.argv) > 2 else "all"
    ctx = capi.Context(0)
    stream = ctx.torch_stream()
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    sw, sh, ow, oh, R = 1920, 1080, 3840, 2160, 6
    NL = int(os.environ.get("PH_UP_LAYERS", "4"))  # layers of the compositor legs (the reader legs always take four windows)
    rd = [dev(capi.ycbcr2rgb_matrix("709")), dev(capi.gamma2linear_lut("709")), dev(np.concatenate([capi.rgb2rgb_matrix("709", "2020"), np.zeros(3, np.float32)]))]
    wr = [dev(capi.rgb2ycbcr_matrix("2020")), dev(capi.linear2gamma_lut("2020"))]
    torch.cuda.synchronize()
    ctx.register_lut(rd[1], capi.gamma2linear_lut("709"))
    ctx.register_lut(wr[1], capi.linear2gamma_lut("2020"))
    words = capi.v210_pitch_bytes(sw) * sh // 4
    src = [[torch.randint(0, 2 ** 30, (words,), dtype=torch.int32, device="cuda") for _ in range(4)] for _ in range(R)]
    out = torch.empty(capi.v210_pitch_bytes(ow) * oh // 4, dtype=torch.int32, device="cuda")
    out2 = torch.empty_like(out)
    rgba = [[[torch.rand(sw * sh * 4, device="cuda") for _ in range(2)] for _ in range(4)] for _ in range(2)]  # two sets: defeat the caches a little
    rgb = [[[torch.rand(sw * sh * 3, device="cuda") for _ in range(2)] for _ in range(4)] for _ in range(2)]
    mh = capi.transform_matrix(ow, oh)
    md = dev(mh)
    torch.cuda.synchronize()

    def timeit(fn):
        import time
        i, t0 = 0, time.perf_counter()
        while i < 4 or time.perf_counter() - t0 < 0.15:  # until the chip's clocks have settled (tools/config_bench.py timeit)
            fn(i)
            i += 1
            if i % 64 == 0:
                ctx.wait()
        ctx.wait()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for i in range(reps):
            fn(i)
        e1.record(stream)
        ctx.wait()
        return round(1e3 * e0.elapsed_time(e1) / reps, 2)

    def win(i, l):
        return (src[i % R][l], src[(i + 1) % R][l], src[(i + 2) % R][l])
    res = {}
    if which in ("all", "deint"):
        res["deint_rgba_us_per_frame"] = timeit(lambda i: ctx.v210_yadif_pair([win(i, l) + (rgba[i & 1][l][0], rgba[i & 1][l][1]) for l in range(4)], sw, sh, 1, False, *rd))
        res["deint_rgb_us_per_frame"] = timeit(lambda i: ctx.v210_yadif_pair([win(i, l) + (rgb[i & 1][l][0], rgb[i & 1][l][1]) for l in range(4)], sw, sh, 1, False, *rd, rgb=True))
    if which in ("all", "compose"):
        res["compose_px_rgba_us_per_field"] = timeit(lambda i: ctx.compose_write_v210([(rgba[i & 1][l][(i >> 1) & 1], sw, sh, md) for l in range(NL)], out, ow, oh, 0, *wr))
        jobs_a = [ctx.compose_up_write_v210([(rgba[s][l][p], sw, sh, mh) for l in range(NL)], out, ow, oh, 0, *wr, prepare_only=True) for s in range(2) for p in range(2)]
        res["compose_up_rgba_us_per_field"] = timeit(lambda i: jobs_a[i & 3]())
    if which in ("all", "compose", "up", "up_single"):  # up_single: one launch per field only (the counters of tools/pmc_kernel.sh are means per dispatch:
        # a two-field launch among them reads as write amplification - round 3's "36.7 MB for a 22.1 MB frame" was that)
        jobs_b = [ctx.compose_up_write_v210([(rgb[s][l][p], sw, sh, mh) for l in range(NL)], out, ow, oh, 0, *wr, rgb=True, prepare_only=True) for s in range(2) for p in range(2)]
        res["compose_up_rgb_us_per_field"] = timeit(lambda i: jobs_b[i & 3]())
    if which in ("all", "compose", "up"):
        pair_jobs = [ctx.compose_up_write_v210_pair([(rgb[s][l][0], sw, sh, mh) for l in range(NL)], [(rgb[s][l][1], sw, sh, mh) for l in range(NL)], out, out2,
                                                    ow, oh, 0, *wr, rgb=True, prepare_only=True) for s in range(2)]
        res["compose_up_rgb_pair_us_per_field"] = round(timeit(lambda i: pair_jobs[i & 1]()) / 2, 2)
    print(json.dumps(res), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
