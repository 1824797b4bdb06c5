#!/usr/bin/env python3
"""BASELINE config 2 through ph_chan_compose_v210 alone (one launch per frame): the timing loop tools/pmc_kernel.sh profiles.
  python tools/chan_bench.py [reps] [mask: rgba|v210] [variant: wipe|nowipe|layer0|insets|inset|overlay] [sources: v210|yuv422p10|yuv422p8|yuv420p|nv12]
  PH_CHAN_BENCH_JOBS=C: C channels' frames per call of ph_chan_compose_batch (PH_CHAN_BATCH=0: the same call, every job through the
  one-job kernel - the A/B of round 5); PH_CHAN_BENCH_W / _H: another frame size"""
import json
import os
import sys

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    import torch
    from phaneron_amd import capi
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    mask_kind = sys.argv[2] if len(sys.argv) > 2 else "rgba"
    variant = sys.argv[3] if len(sys.argv) > 3 else "wipe"
    packing = sys.argv[4] if len(sys.argv) > 4 else "v210"  # yuv422p10: the sources are planar 10-bit frames (file decoders' format)
    ctx = capi.Context(0)
    stream = ctx.torch_stream()
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    # (another height: how much of a frame's time is the last, partial round of chunks)
    w, h, R = int(os.environ.get("PH_CHAN_BENCH_W", "1920")), int(os.environ.get("PH_CHAN_BENCH_H", "1080")), 8
    C = int(os.environ.get("PH_CHAN_BENCH_JOBS", "1"))
    IL = int(os.environ.get("PH_CHAN_BENCH_INTERLACE", "0"))  # 1 / 3: a field write (a 1080i50 channel's consumer writes a field per 50p frame: macadamConsumer.ts:165)
    rd = [dev(capi.ycbcr2rgb_matrix("709")), dev(capi.gamma2linear_lut("709")), dev(np.concatenate([capi.rgb2rgb_matrix("709", "709"), np.zeros(3, np.float32)]))]
    wr = [dev(capi.rgb2ycbcr_matrix("709")), dev(capi.linear2gamma_lut("709"))]
    torch.cuda.synchronize()
    ctx.register_lut(rd[1], capi.gamma2linear_lut("709"))
    ctx.register_lut(wr[1], capi.linear2gamma_lut("709"))
    words = capi.v210_pitch_bytes(w) * h // 4
    src = [[torch.randint(0, 2 ** 30, (words,), dtype=torch.int32, device="cuda") for _ in range(6)] for _ in range(R)]
    out = torch.empty(words, dtype=torch.int32, device="cuda")
    out_fmt = os.environ.get("PH_CHAN_BENCH_OUT", "v210")  # another consumer's frame: yuv422p8 (ffmpegConsumer.ts:144), rgba8 (screenConsumer.ts:131), yuv422p10
    if out_fmt in ("yuv422p8", "yuv422p10"):
        dt = torch.uint8 if out_fmt == "yuv422p8" else torch.int16
        out = (torch.empty(w * h, dtype=dt, device="cuda"), torch.empty(w * h // 2, dtype=dt, device="cuda"), torch.empty(w * h // 2, dtype=dt, device="cuda"))
    elif out_fmt in ("rgba8", "bgra8"):
        out = (torch.empty(w * h, dtype=torch.int32, device="cuda"),)
    kind = ()
    if packing != "v210":  # planar sources (file decoders' formats); the 8-bit ones bring a Loader matrix of their own
        pitch = (w + 7) // 8 * 8
        wide = packing == "yuv422p10"
        plane = lambda n: torch.randint(0, 1024, (n,), dtype=torch.int16, device="cuda") if wide else torch.randint(0, 256, (n,), dtype=torch.uint8, device="cuda")
        sizes = {"yuv422p10": (pitch * h, pitch // 2 * h, pitch // 2 * h), "yuv422p8": (pitch * h, pitch // 2 * h, pitch // 2 * h),
                 "yuv420p": (pitch * h, pitch * h // 4, pitch * h // 4), "nv12": (pitch * h, pitch * h // 2)}[packing]
        src = [[tuple(plane(n) for n in sizes) for _ in range(6)] for _ in range(R)]
        own = None if wide else dev(capi.ycbcr2rgb_matrix("709", 8, 16, 235, 224))
        kind = (packing, own)
    overlay = torch.randint(0, 2 ** 31 - 1, (w * h,), dtype=torch.int32, device="cuda")
    mask = torch.zeros(h, w, 4, device="cuda")
    mask[..., 0] = torch.linspace(0, 1, w, device="cuda")[None, :]
    mask = mask.reshape(-1).contiguous()
    mats = [capi.transform_matrix(w, h)] + [capi.transform_matrix(w, h, scale_x=0.5, scale_y=0.5, offset_x=ox, offset_y=oy)
                                            for ox, oy in ((-0.25, -0.25), (0.25, -0.25), (0.25, 0.25))]
    torch.cuda.synchronize()

    def layers(s):
        ls = [dict(src=(s[l], w, h, mats[l]) + kind) for l in range(4)]
        if variant == "wipe":
            ls[3].update(transition="wipe", incoming=(s[4], w, h, None) + kind,
                         mask=(mask, w, h, None, "rgba") if mask_kind == "rgba" else (s[5], w, h, None) + kind)
        elif variant == "layer0":
            ls = ls[:1]
        elif variant == "insets":
            ls = ls[1:]
        elif variant == "inset":  # file playback with one inset: the full-frame clip and a quarter-size one
            ls = ls[:2]
        elif variant == "overlay":  # a clip under a full-frame graphic with alpha (a bgra8 frame: png / html overlays come as packed RGB)
            ls = [ls[0], dict(src=(overlay, w, h, mats[0], "bgra8"))]
        return ls
    if C > 1:  # C channels per launch: each job its own sources (rotated through the ring) and its own output
        outs = [torch.empty(words, dtype=torch.int32, device="cuda") for _ in range(C)]
        jobs = [ctx.chan_compose_batch([(layers(src[(i + j) % R]), outs[j], 0) for j in range(C)], w, h, *rd, *wr, prepare_only=True) for i in range(R)]
    else:
        jobs = [ctx.chan_compose_v210(layers(s), out, w, h, IL, *rd, *((None, wr[1]) if out_fmt in ("rgba8", "bgra8") else wr), prepare_only=True, out_fmt=out_fmt) for s in src]
    import time
    i, t0 = 0, time.perf_counter()
    while i < 8 or time.perf_counter() - t0 < 0.15:  # until the chip's clocks have settled (tools/config_bench.py timeit)
        jobs[i % R]()
        i += 1
        if i % 64 == 0:
            ctx.wait()
    ctx.wait()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for i in range(reps):
        jobs[i % R]()
    e1.record(stream)
    ctx.wait()
    print(json.dumps({"kernel": "chan_compose_v210", "width": w, "height": h, "variant": variant, "mask": mask_kind, "sources": packing, "out": out_fmt, "interlace": IL, "jobs_per_launch": C,
                      "batch_kernel": os.environ.get("PH_CHAN_BATCH", "1") != "0" and C > 1, "us_per_launch": round(1e3 * e0.elapsed_time(e1) / reps, 2),
                      "us_per_frame": round(1e3 * e0.elapsed_time(e1) / reps / C, 2)}), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
