#!/usr/bin/env python3
"""What the reference's own formats make of BASELINE config 3: 4 x 1080i50 sources de-interlaced (yadif, send_field) and shown at their own size
on a 1080p50 channel (Mixer's default fill, no enlargement) -> combine_4 -> v210.  Per output field: half a de-interlacing reader launch
(ph_v210_yadif_pair, both fields of four layers per frame) + one channel-kernel launch on the four field images.
  python tools/config3b_bench.py [reps]"""
import json
import os
import sys
import time

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    import torch
    from phaneron_amd import capi
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    ctx = capi.Context(0)
    stream = ctx.torch_stream()
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    w, h, R = 1920, 1080, 6
    rd = [dev(capi.ycbcr2rgb_matrix("709")), dev(capi.gamma2linear_lut("709")), dev(np.concatenate([capi.rgb2rgb_matrix("709", "709"), np.zeros(3, np.float32)]))]
    wr = [dev(capi.rgb2ycbcr_matrix("709")), dev(capi.linear2gamma_lut("709"))]
    torch.cuda.synchronize()
    ctx.register_lut(rd[1], capi.gamma2linear_lut("709"))
    ctx.register_lut(wr[1], capi.linear2gamma_lut("709"))
    words = capi.v210_pitch_bytes(w) * h // 4
    srcs = [[torch.randint(0, 2 ** 30, (words,), dtype=torch.int32, device="cuda") for _ in range(4)] for _ in range(R)]
    fields = [[torch.empty(w * h * 4, dtype=torch.float32, device="cuda") for _ in range(2)] for _ in range(4)]
    out = torch.empty(words, dtype=torch.int32, device="cuda")
    mat = capi.transform_matrix(w, h)
    win = [[srcs[k % R][l] for k in range(3)] for l in range(4)]
    chan = [ctx.chan_compose_v210([dict(src=(fields[l][p], w, h, mat, "rgba")) for l in range(4)], out, w, h, 0, *rd, *wr, prepare_only=True) for p in range(2)]
    torch.cuda.synchronize()

    def step(i):
        if not (i & 1):
            s = srcs[(i // 2) % R]
            for l in range(4):
                win[l] = [win[l][1], win[l][2], s[l]]
            ctx.v210_yadif_pair([(win[l][0], win[l][1], win[l][2], fields[l][0], fields[l][1]) for l in range(4)], w, h, 1, False, *rd)
        chan[1 ^ (0 if (i & 1) else 1)]()

    def timeit(fn, n):
        i, t0 = 0, time.perf_counter()
        while i < 4 or time.perf_counter() - t0 < 0.15:
            fn(i)
            i += 1
            if i % 64 == 0:
                ctx.wait()
        i += i & 1
        ctx.wait()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for k in range(i, i + n):
            fn(k)
        e1.record(stream)
        ctx.wait()
        return e0.elapsed_time(e1) / n
    ctx.set_option("chan_enlarged", 0)  # the channel kernel itself (the product's call sends these frames to the 2 x 2-block compositor: below)
    both = timeit(step, reps)
    only_chan = timeit(lambda i: chan[i & 1](), reps)
    ctx.set_option("chan_enlarged", 1)
    routed = timeit(step, reps)
    # the same frames by the 2 x 2-block compositor, which takes the default fill of frame-size images beside the enlargements: a launch per field
    # on RGBA fields, and the reader writing packed-RGB fields with both fields' frames as ONE launch (config 3's route)
    outs = [torch.empty(words, dtype=torch.int32, device="cuda") for _ in range(2)]
    up = [ctx.compose_up_write_v210([(fields[l][p], w, h, mat) for l in range(4)], out, w, h, 0, *wr, prepare_only=True) for p in range(2)]
    rgb = [[torch.empty(w * h * 3, dtype=torch.float32, device="cuda") for _ in range(2)] for _ in range(4)]
    pair = ctx.compose_up_write_v210_pair([(rgb[l][0], w, h, mat) for l in range(4)], [(rgb[l][1], w, h, mat) for l in range(4)], outs[0], outs[1], w, h, 0, *wr, rgb=True, prepare_only=True)

    def step_up(i):
        if not (i & 1):
            s = srcs[(i // 2) % R]
            for l in range(4):
                win[l] = [win[l][1], win[l][2], s[l]]
            ctx.v210_yadif_pair([(win[l][0], win[l][1], win[l][2], fields[l][0], fields[l][1]) for l in range(4)], w, h, 1, False, *rd)
        up[1 ^ (0 if (i & 1) else 1)]()

    def step_pair(i):
        if not (i & 1):
            s = srcs[(i // 2) % R]
            for l in range(4):
                win[l] = [win[l][1], win[l][2], s[l]]
            ctx.v210_yadif_pair([(win[l][0], win[l][1], win[l][2], rgb[l][0], rgb[l][1]) for l in range(4)], w, h, 1, False, *rd, rgb=True)
            pair()
    ctx.set_option("chan_enlarged", 0)
    chan[0](); ctx.wait()
    ref = out.clone()
    up[0](); ctx.wait()
    same = bool(torch.equal(ref, out))
    ctx.set_option("chan_enlarged", 1)
    both_up, both_pair = timeit(step_up, reps), timeit(step_pair, reps)
    ctx.set_option("chan_enlarged", 0)
    algo = 4 * 3 * capi.v210_pitch_bytes(w) * h // 2 + capi.v210_pitch_bytes(w) * h  # per field: half of 4 x 3 window frames in + one v210 frame out
    print(json.dumps({"config": "3b: 4 x 1080i50 -> yadif -> own size on a 1080p50 channel -> combine_4 -> v210 (per output field)", "us_per_field_by_channel_kernel": round(1e3 * both, 2),
                      "channel_kernel_alone_us": round(1e3 * only_chan, 2), "us_per_field_as_the_product_routes_it": round(1e3 * routed, 2),
                      "us_per_field_by_2x2_block_compositor": round(1e3 * both_up, 2),
                      "us_per_field_packed_rgb_fields_pair_launch": round(1e3 * both_pair, 2), "compositor_frame_equals_channel_kernel_frame": same, "fields_per_sec": round(1e3 / both_pair, 1), "x_realtime_50fps": round(1e3 / both_pair / 50, 1),
                      "algorithmic_bytes": algo, "hbm_frac": round(algo / both_pair / 1e6 / 8000.0, 4)}))
    ctx.close()


if __name__ == "__main__":
    main()
