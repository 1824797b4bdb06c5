#!/usr/bin/env python3
"""A clip smaller than its channel (the reference uploads clips at their own size and lets the Mixer's transform fill the channel:
src/producer/ffmpegProducer.ts:395-442, mixer.ts:189-228): N v210 layers of sw x sh shown full-frame on an ow x oh channel, three routes:
  chan       ph_chan_compose_v210 on the v210 sources with context option chan_enlarged = 0: the channel kernel, every tap converted
  routed     the same call as the product makes it (chan_enlarged = 1): one frame of wire-format clips = reader + 2 x 2-block compositor in ONE launch
  two_launch the same call with chan_enlarged = 2: read + 2 x 2-block compositor as two launches, scratch images (round 5's route)
  read+chan  ph_v210_read per layer (v210 -> f32 image, once per SOURCE pixel) + ph_chan_compose_v210 on the f32 images
  read+up    ph_v210_read per layer + ph_compose_up_write_v210 (the 2 x 2-block compositor on f32 images)
  python tools/enlarge_bench.py [reps] [layers] [sw] [sh] [ow] [oh]"""
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
    a = sys.argv[1:]
    reps = int(a[0]) if len(a) > 0 else 300
    n = int(a[1]) if len(a) > 1 else 1
    sw, sh = (int(a[2]), int(a[3])) if len(a) > 3 else (1280, 720)
    ow, oh = (int(a[4]), int(a[5])) if len(a) > 5 else (1920, 1080)
    ctx = capi.Context(0)
    stream = ctx.torch_stream()
    dev = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
    rd = [dev(capi.ycbcr2rgb_matrix("709")), dev(capi.gamma2linear_lut("709")), dev(np.concatenate([capi.rgb2rgb_matrix("709", "709"), np.zeros(3, np.float32)]))]
    wr = [dev(capi.rgb2ycbcr_matrix("709")), dev(capi.linear2gamma_lut("709"))]
    torch.cuda.synchronize()
    ctx.register_lut(rd[1], capi.gamma2linear_lut("709"))
    ctx.register_lut(wr[1], capi.linear2gamma_lut("709"))
    R = int(os.environ.get("PH_ENLARGE_RING", "6"))  # sets of sources / images the routes rotate through
    swords, owords = capi.v210_pitch_bytes(sw) * sh // 4, capi.v210_pitch_bytes(ow) * oh // 4
    src = [[torch.randint(0, 2 ** 30, (swords,), dtype=torch.int32, device="cuda") for _ in range(n)] for _ in range(R)]
    packing = os.environ.get("PH_ENLARGE_FORMAT", "v210")  # a decoder's planar frames instead (yuv422p10 | yuv422p8 | yuv420p | nv12): routes chan / routed only
    kind = ()
    if packing != "v210":
        pitch = (sw + 7) // 8 * 8
        wide = packing == "yuv422p10"
        plane = lambda k: torch.randint(0, 1024, (k,), dtype=torch.int16, device="cuda") if wide else torch.randint(0, 256, (k,), dtype=torch.uint8, device="cuda")
        sizes = {"yuv422p10": (pitch * sh, pitch // 2 * sh, pitch // 2 * sh), "yuv422p8": (pitch * sh, pitch // 2 * sh, pitch // 2 * sh),
                 "yuv420p": (pitch * sh, pitch * sh // 4, pitch * sh // 4), "nv12": (pitch * sh, pitch * sh // 2)}[packing]
        src = [[tuple(plane(k) for k in sizes) for _ in range(n)] for _ in range(R)]
        kind = (packing, None if wide else dev(capi.ycbcr2rgb_matrix("709", 8, 16, 235, 224)))
    img = [[torch.empty(sw * sh * 4, dtype=torch.float32, device="cuda") for _ in range(n)] for _ in range(R)]
    out = [torch.empty(owords, dtype=torch.int32, device="cuda") for _ in range(3)]
    # layer l: the whole clip over the whole channel, each a little smaller than the one below so that all of them show
    mats = [capi.transform_matrix(ow, oh, scale_x=1.0 - 0.1 * l, scale_y=1.0 - 0.1 * l) for l in range(n)]
    torch.cuda.synchronize()
    C = int(os.environ.get("PH_ENLARGE_CHANNELS", "1"))  # C channels' frames per ph_chan_compose_batch call (routes chan / routed only)
    if C > 1:
        outs = [torch.empty(owords, dtype=torch.int32, device="cuda") for _ in range(C)]
        chan_jobs = [[ctx.chan_compose_batch([([dict(src=(src[(i + c) % R][l], sw, sh, mats[l]) + kind) for l in range(n)], outs[c], 0) for c in range(C)], ow, oh, *rd, *wr, prepare_only=True)]
                     for i in range(R)]
        out[0] = outs[0]
    else:
        chan_jobs = [[ctx.chan_compose_v210([dict(src=(s[l], sw, sh, mats[l]) + kind) for l in range(n)], out[0], ow, oh, 0, *rd, *wr, prepare_only=True)] for s in src]
    routes = {"chan": chan_jobs, "routed": chan_jobs, "two_launch": chan_jobs}
    if C == 1 and packing == "v210":
        routes["read+chan"] = [[(lambda s=s, im=im: [ctx.v210_read(s[l], im[l], sw, sh, *rd) for l in range(n)]),
                                ctx.chan_compose_v210([dict(src=(im[l], sw, sh, mats[l], "rgba")) for l in range(n)], out[1], ow, oh, 0, *rd, *wr, prepare_only=True)]
                               for s, im in zip(src, img)]
        routes["read+up"] = [[(lambda s=s, im=im: [ctx.v210_read(s[l], im[l], sw, sh, *rd) for l in range(n)]),
                              ctx.compose_up_write_v210([(im[l], sw, sh, mats[l]) for l in range(n)], out[2], ow, oh, 0, *wr, prepare_only=True)]
                             for s, im in zip(src, img)]
    res = {}
    only = os.environ.get("PH_ENLARGE_ONLY")  # one route alone (for a rocprofv3 kernel trace of it)
    if only:
        routes = {k: v for k, v in routes.items() if k in ("chan", only)}
    for name, jobs in routes.items():
        ctx.set_option("chan_enlarged", 0 if name == "chan" else 2 if name == "two_launch" else 1)
        i, t0 = 0, time.perf_counter()
        while i < 8 or time.perf_counter() - t0 < 0.15:
            for j in jobs[i % R]:
                j()
            i += 1
            if i % 64 == 0:
                ctx.wait()
        ctx.wait()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for i in range(reps):
            for j in jobs[i % R]:
                j()
        e1.record(stream)
        ctx.wait()
        res[name] = round(1e3 * e0.elapsed_time(e1) / reps / C, 2)
        if name == "chan":
            kept = out[0].clone()
        elif name == "routed":
            same_routed = bool(torch.equal(kept, out[0]))
    same = {k: bool(torch.equal(kept, out[i])) for k, i in (("two_launch", 0), ("read+chan", 1), ("read+up", 2)) if k in res}
    if "routed" in res:
        same["routed"] = same_routed
    print(json.dumps({"bench": "enlarge", "channels_per_call": C, "format": packing, "layers": n, "source": [sw, sh], "channel": [ow, oh], "us_per_frame": res, "same_frame_as_chan": same}), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
