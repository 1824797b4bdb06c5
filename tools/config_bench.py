#!/usr/bin/env python3
"""Secondary workloads of BASELINE.json (configs 2 and 3) run as the reference's own job
structure - one kernel per operator, f32 RGBA frames between them - through the C ABI.
These are what a drop-in phaneron executes today; the fused headline path is bench.py.

  config 2: 1 channel, 4 layers 1080p50: read x4 -> transform (full / PiP x3) -> wipe transition
            on layer 4 (second source + ramp mask) -> combine_4 -> write
  config 3: 1 channel, 4 layers -> 2160p50 from 1080i50: read(709->2020) (one new frame per layer
            per two fields) -> yadif send_field -> transform 1080->2160 -> combine_4 -> write(2020)
Prints one JSON line per config with fields/frames per second and algorithmic GB/s
(SURVEY.md 8d byte counts).
"""
import json
import os
import sys

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from phaneron_amd import capi


def main():
    ctx = capi.Context(0)
    stream = ctx.torch_stream()
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()

    def colour(rspec, wspec):
        rd = [dev(capi.ycbcr2rgb_matrix(rspec)), dev(capi.gamma2linear_lut(rspec)),
              dev(np.concatenate([capi.rgb2rgb_matrix(rspec, wspec), np.zeros(3, np.float32)]))]
        wr = [dev(capi.rgb2ycbcr_matrix(wspec)), dev(capi.linear2gamma_lut(wspec))]
        torch.cuda.synchronize()
        ctx.register_lut(rd[1], capi.gamma2linear_lut(rspec))
        ctx.register_lut(wr[1], capi.linear2gamma_lut(wspec))
        return rd, wr

    def v210(w, h, n):
        words = capi.v210_pitch_bytes(w) * h // 4
        return [torch.randint(0, 2 ** 30, (words,), dtype=torch.int32, device="cuda") & 0x3FFFFFFF for _ in range(n)]

    def img(w, h, n=1):
        return [torch.empty(w * h * 4, dtype=torch.float32, device="cuda") for _ in range(n)]

    def timeit(fn, reps):
        for i in range(3):
            fn(i)
        ctx.wait()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for i in range(reps):
            fn(i)
        e1.record(stream)
        ctx.wait()
        return e0.elapsed_time(e1) / reps

    # ---------------- config 2 -------------------------------------------------------------
    w, h = 1920, 1080
    rd, wr = colour("709", "709")
    R = 8
    src = [v210(w, h, 5) for _ in range(R)]          # 4 layers + the wipe's second source
    rgba = img(w, h, 5)
    xf = img(w, h, 4)
    trans, comb = img(w, h)[0], img(w, h)[0]
    out = torch.empty(capi.v210_pitch_bytes(w) * h // 4, dtype=torch.int32, device="cuda")
    mask = torch.zeros(h, w, 4, device="cuda")
    mask[..., 0] = torch.linspace(0, 1, w, device="cuda")[None, :]
    mask = mask.reshape(-1).contiguous()
    mats = [dev(capi.transform_matrix(w, h))] + [
        dev(capi.transform_matrix(w, h, scale_x=0.5, scale_y=0.5, offset_x=ox, offset_y=oy))
        for ox, oy in ((-0.25, -0.25), (0.25, -0.25), (0.25, 0.25))]
    torch.cuda.synchronize()

    def config2(i):
        s = src[i % R]
        for l in range(5):
            ctx.v210_read(s[l], rgba[l], w, h, *rd)
        for l in range(4):
            ctx.transform(rgba[l], w, h, mats[l], xf[l], w, h)
        ctx.transition_wipe(xf[3], rgba[4], mask, trans, w, h)
        ctx.combine([xf[0], xf[1], xf[2], trans], comb, w, h)
        ctx.v210_write(comb, out, w, h, 0, *wr)

    def config2_fused(i):  # same frame with transform x4 + combine_4 + write as one kernel
        s = src[i % R]
        for l in range(5):
            ctx.v210_read(s[l], rgba[l], w, h, *rd)
        ctx.transform(rgba[3], w, h, mats[3], xf[3], w, h)
        ctx.transition_wipe(xf[3], rgba[4], mask, trans, w, h)
        ctx.compose_write_v210([(rgba[0], w, h, mats[0]), (rgba[1], w, h, mats[1]), (rgba[2], w, h, mats[2]),
                                (trans, w, h, None)], out, w, h, 0, *wr)

    # the same two batches recorded once per ring slot and replayed as one submission each (ph_graph_*)
    graphs = [ctx.record(lambda i=i: config2(i)) for i in range(R)]
    graphs_f = [ctx.record(lambda i=i: config2_fused(i)) for i in range(R)]
    ms_g = timeit(lambda i: graphs[i % R].launch(), 200)
    ms_gf = timeit(lambda i: graphs_f[i % R].launch(), 200)
    print(json.dumps({"config": "2 as a recorded batch (hipGraph replay, 13 kernels/frame)", "ms_per_frame": round(ms_g, 4),
                      "frames_per_sec": round(1e3 / ms_g, 1)}), flush=True)
    print(json.dumps({"config": "2 fused compositor as a recorded batch (hipGraph replay, 8 kernels/frame)",
                      "ms_per_frame": round(ms_gf, 4), "frames_per_sec": round(1e3 / ms_gf, 1)}), flush=True)
    ms_f = timeit(config2_fused, 200)
    print(json.dumps({"config": "2 (fused compositor: read x5, transform, wipe, compose+write = 8 kernels/frame)",
                      "ms_per_frame": round(ms_f, 4), "frames_per_sec": round(1e3 / ms_f, 1)}), flush=True)
    ms = timeit(config2, 200)
    algo = 7 * capi.v210_pitch_bytes(w) * h  # 4 layers + second source + mask-as-v210-equivalent + 1 out
    print(json.dumps({"config": "2: 4-layer 1080p50 transform+wipe+combine (13 kernels/frame)", "ms_per_frame": round(ms, 4),
                      "frames_per_sec": round(1e3 / ms, 1), "algorithmic_MB": round(algo / 1e6, 1),
                      "algorithmic_GBps": round(algo / ms / 1e6, 1), "x_realtime_50fps": round(1e3 / ms / 50, 1)}), flush=True)

    # ---------------- config 3 -------------------------------------------------------------
    sw, sh, ow, oh = 1920, 1080, 3840, 2160
    rd, wr = colour("709", "2020")
    srcs = [v210(sw, sh, 4) for _ in range(R)]
    win = [img(sw, sh, 3) for _ in range(4)]         # prev / cur / next per layer
    deint = img(sw, sh, 4)
    up = img(ow, oh, 4)
    comb = img(ow, oh)[0]
    out = torch.empty(capi.v210_pitch_bytes(ow) * oh // 4, dtype=torch.int32, device="cuda")
    m = dev(capi.transform_matrix(ow, oh))
    torch.cuda.synchronize()

    def config3(i):  # one output FIELD; a new source frame is unpacked every second field
        s = srcs[(i // 2) % R]
        second = i & 1
        for l in range(4):
            if not second:
                win[l] = [win[l][1], win[l][2], win[l][0]]
                ctx.v210_read(s[l], win[l][2], sw, sh, *rd)
            ctx.yadif(win[l][0], win[l][1], win[l][2], deint[l], sw, sh, 1 ^ (0 if second else 1), 1, False)
            ctx.transform(deint[l], sw, sh, m, up[l], ow, oh)
        ctx.combine(up, comb, ow, oh)
        ctx.v210_write(comb, out, ow, oh, 0, *wr)

    def config3_fused(i):  # yadif per layer, then upscale x4 + combine_4 + write as one kernel
        s = srcs[(i // 2) % R]
        second = i & 1
        for l in range(4):
            if not second:
                win[l] = [win[l][1], win[l][2], win[l][0]]
                ctx.v210_read(s[l], win[l][2], sw, sh, *rd)
            ctx.yadif(win[l][0], win[l][1], win[l][2], deint[l], sw, sh, 1 ^ (0 if second else 1), 1, False)
        ctx.compose_write_v210([(deint[l], sw, sh, m) for l in range(4)], out, ow, oh, 0, *wr)

    mh = capi.transform_matrix(ow, oh)

    def config3_field(i):  # read (new frames only), then yadif + upscale + combine_4 + write as ONE kernel
        s = srcs[(i // 2) % R]
        second = i & 1
        for l in range(4):
            if not second:
                win[l] = [win[l][1], win[l][2], win[l][0]]
                ctx.v210_read(s[l], win[l][2], sw, sh, *rd)
        ctx.fused_field_v210([dict(prev=win[l][0], cur=win[l][1], next=win[l][2], width=sw, height=sh, matrix=m, matrix_host=mh,
                                   deinterlace=True, parity=1 ^ (0 if second else 1), tff=1) for l in range(4)], out, ow, oh, *wr)

    def config3_hybrid(i):  # read (new frames only), yadif per layer, then upscale + combine_4 + write from LDS windows
        s = srcs[(i // 2) % R]
        second = i & 1
        for l in range(4):
            if not second:
                win[l] = [win[l][1], win[l][2], win[l][0]]
                ctx.v210_read(s[l], win[l][2], sw, sh, *rd)
            ctx.yadif(win[l][0], win[l][1], win[l][2], deint[l], sw, sh, 1 ^ (0 if second else 1), 1, False)
        ctx.fused_field_v210([dict(cur=deint[l], width=sw, height=sh, matrix=m, matrix_host=mh, deinterlace=False) for l in range(4)],
                             out, ow, oh, *wr)

    algo3 = 4 * 3 * capi.v210_pitch_bytes(sw) * sh + capi.v210_pitch_bytes(ow) * oh  # 88 473 600
    ms_h = timeit(config3_hybrid, 300)
    print(json.dumps({"config": "3 (read x4 every other field, yadif x4, ONE windowed upscale/combine kernel + index->v210 kernel)",
                      "ms_per_field": round(ms_h, 4), "fields_per_sec": round(1e3 / ms_h, 1), "algorithmic_MB": round(algo3 / 1e6, 1),
                      "algorithmic_GBps": round(algo3 / ms_h / 1e6, 1), "roofline_frac": round(algo3 / ms_h / 1e6 / 8000.0, 4),
                      "x_realtime_50fps": round(1e3 / ms_h / 50, 1)}), flush=True)
    ms_ff = timeit(config3_field, 300)
    print(json.dumps({"config": "3 (field pipeline: read x4 every other field + ONE fused yadif/upscale/combine/write kernel)",
                      "ms_per_field": round(ms_ff, 4), "fields_per_sec": round(1e3 / ms_ff, 1), "algorithmic_MB": round(algo3 / 1e6, 1),
                      "algorithmic_GBps": round(algo3 / ms_ff / 1e6, 1), "roofline_frac": round(algo3 / ms_ff / 1e6 / 8000.0, 4),
                      "x_realtime_50fps": round(1e3 / ms_ff / 50, 1)}), flush=True)
    ms_f = timeit(config3_fused, 200)
    print(json.dumps({"config": "3 (fused compositor: read, yadif x4, compose+write = 5-9 kernels/field)",
                      "ms_per_field": round(ms_f, 4), "fields_per_sec": round(1e3 / ms_f, 1),
                      "algorithmic_GBps": round((4 * 3 * capi.v210_pitch_bytes(sw) * sh + capi.v210_pitch_bytes(ow) * oh) / ms_f / 1e6, 1)}), flush=True)
    ms = timeit(config3, 200)
    algo = 4 * 3 * capi.v210_pitch_bytes(sw) * sh + capi.v210_pitch_bytes(ow) * oh  # 88 473 600
    print(json.dumps({"config": "3: 4 x 1080i50 -> yadif -> 2x upscale -> combine_4 -> 2160p50 (12 kernels/field)",
                      "ms_per_field": round(ms, 4), "fields_per_sec": round(1e3 / ms, 1),
                      "algorithmic_MB": round(algo / 1e6, 1), "algorithmic_GBps": round(algo / ms / 1e6, 1),
                      "x_realtime_50fps": round(1e3 / ms / 50, 1)}), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
