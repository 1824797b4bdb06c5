#!/usr/bin/env python3
"""Per-kernel timings at UHD size through the C ABI (HIP events on the library's stream):
GB/s of compulsory traffic for every kernel of the path.  Writes one JSON line per kernel."""
import json
import os
import sys

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from phaneron_amd import capi


def main():
    w, h = 3840, 2160
    only_formats = [a for a in sys.argv[1:] if not a.startswith('-')]
    ctx = capi.Context(0)
    stream = ctx.torch_stream()
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    rd = [dev(capi.ycbcr2rgb_matrix("709")), dev(capi.gamma2linear_lut("709")),
          dev(np.concatenate([capi.rgb2rgb_matrix("709", "2020"), np.zeros(3, np.float32)]))]
    wr = [dev(capi.rgb2ycbcr_matrix("2020")), dev(capi.linear2gamma_lut("2020"))]
    torch.cuda.synchronize()
    ctx.register_lut(rd[1], capi.gamma2linear_lut("709"))
    ctx.register_lut(wr[1], capi.linear2gamma_lut("2020"))
    if os.environ.get("PH_BENCH_GLOBAL_LUT"):
        ctx.set_option("lds_lut", 0)
    # a kernel timed alone has no consumer on the device: image outputs streamed past the caches, as a caller that
    # knows this would ask for (PH_BENCH_CACHED_IMAGES=1 times the library's default, which assumes a consumer follows)
    streamed = 0 if os.environ.get("PH_BENCH_CACHED_IMAGES") else 1
    ctx.set_option("stream_images", streamed)
    words = capi.v210_pitch_bytes(w) * h // 4
    R = 6  # ring to defeat the 256 MiB Infinity Cache
    v = [torch.randint(0, 2 ** 30, (words,), dtype=torch.int32, device="cuda") & 0x3FFFFFFF for _ in range(R)]
    img = [torch.rand(w * h * 4, dtype=torch.float32, device="cuda") for _ in range(R)]
    out_img = [torch.empty(w * h * 4, dtype=torch.float32, device="cuda") for _ in range(2)]
    out_v = [torch.empty(words, dtype=torch.int32, device="cuda") for _ in range(2)]
    src1080 = [torch.rand(1920 * 1080 * 4, dtype=torch.float32, device="cuda") for _ in range(R)]
    m = dev(capi.transform_matrix(w, h))
    flip = dev(np.array([0, 1, 0, 1], np.float32))
    torch.cuda.synchronize()
    vb, ib = words * 4, w * h * 16

    def timeit(name, fn, bytes_, reps=30):
        if only_formats and not name.startswith("pack_"):
            return
        for i in range(3):
            fn(i)
        ctx.wait()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for i in range(reps):
            fn(i)
        e1.record(stream)
        ctx.wait()
        ms = e0.elapsed_time(e1) / reps
        print(json.dumps({"kernel": name, "ms": round(ms, 4), "algorithmic_MB": round(bytes_ / 1e6, 1),
                          "GBps": round(bytes_ / ms / 1e6, 1), "image_stores": "streamed" if streamed else "cached"}), flush=True)

    timeit("v210_read 2160p", lambda i: ctx.v210_read(v[i % R], out_img[i % 2], w, h, *rd), vb + ib)
    timeit("v210_write 2160p", lambda i: ctx.v210_write(img[i % R], out_v[i % 2], w, h, 0, *wr), vb + ib)
    timeit("combine_4 2160p", lambda i: ctx.combine([img[(i + j) % R] for j in range(4)], out_img[i % 2], w, h), 5 * ib)
    timeit("combine_2 2160p", lambda i: ctx.combine([img[(i + j) % R] for j in range(2)], out_img[i % 2], w, h), 3 * ib)
    timeit("transition_dissolve 2160p", lambda i: ctx.transition_dissolve(img[i % R], img[(i + 1) % R], 0.3, out_img[i % 2], w, h), 3 * ib)
    timeit("transition_wipe 2160p", lambda i: ctx.transition_wipe(img[i % R], img[(i + 1) % R], img[(i + 2) % R], out_img[i % 2], w, h), 4 * ib)
    # wipe picks ONE of its two inputs per pixel: one image read + one written
    timeit("wipe 2160p", lambda i: ctx.wipe(img[i % R], img[(i + 1) % R], 0.4, out_img[i % 2], w, h), 2 * ib)
    timeit("yadif 2160p", lambda i: ctx.yadif(img[i % R], img[(i + 1) % R], img[(i + 2) % R], out_img[i % 2], w, h, 0, 1), 4 * ib)
    timeit("transform identity 2160p", lambda i: ctx.transform(img[i % R], w, h, m, out_img[i % 2], w, h), 2 * ib)
    timeit("transform 1080->2160", lambda i: ctx.transform(src1080[i % R], 1920, 1080, m, out_img[i % 2], w, h), ib + ib // 4)
    timeit("resize 1080->2160", lambda i: ctx.resize(src1080[i % R], 1920, 1080, 1.0, 0.0, 0.0, flip, out_img[i % 2], w, h), ib + ib // 4)
    src540 = [torch.rand(1920 * 1080 * 4, dtype=torch.float32, device="cuda") for _ in range(4)]
    timeit("yadif 1080p", lambda i: ctx.yadif(src1080[i % R], src1080[(i + 1) % R], src1080[(i + 2) % R], out_img[0], 1920, 1080, 0, 1), 4 * ib // 4)
    out1080 = [torch.empty(1920 * 1080 * 4, dtype=torch.float32, device="cuda") for _ in range(2)]
    # both fields of a frame in one pass: 3 frames read, 2 written
    timeit("yadif_pair 1080p (two fields)", lambda i: ctx.yadif_pair(src1080[i % R], src1080[(i + 1) % R], src1080[(i + 2) % R], out1080[0], out1080[1], 1920, 1080, 1), 5 * ib // 4)
    timeit("yadif_pair 2160p (two fields)", lambda i: ctx.yadif_pair(img[i % R], img[(i + 1) % R], img[(i + 2) % R], out_img[0], out_img[1], w, h, 1), 5 * ib)
    v1080 = [torch.randint(0, 2 ** 30, (capi.v210_pitch_bytes(1920) * 1080 // 4,), dtype=torch.int32, device="cuda") & 0x3FFFFFFF for _ in range(12)]
    o1080 = [torch.empty(1920 * 1080 * 4, dtype=torch.float32, device="cuda") for _ in range(8)]
    # per layer 3 v210 frames in, 2 RGBA frames out
    timeit("v210_yadif_pair 4 x 1080i (unpack + both fields, one launch)",
           lambda i: ctx.v210_yadif_pair([(v1080[(i + 3 * l) % 12], v1080[(i + 3 * l + 1) % 12], v1080[(i + 3 * l + 2) % 12], o1080[2 * l], o1080[2 * l + 1])
                                          for l in range(4)], 1920, 1080, 1, False, *rd),
           4 * (3 * capi.v210_pitch_bytes(1920) * 1080 + 2 * ib // 4))
    timeit("v210_read_batch 4 x 1080p", lambda i: ctx.v210_read_batch([v1080[(i + l) % 12] for l in range(4)], o1080[:4], 1920, 1080, *rd),
           4 * (capi.v210_pitch_bytes(1920) * 1080 + ib // 4))
    timeit("compose_write 4 x (1080p -> 2160p bilinear) -> v210",
           lambda i: ctx.compose_write_v210([(src1080[(i + j) % R], 1920, 1080, m) for j in range(4)], out_v[i % 2], w, h, 0, *wr),
           ib + vb)
    timeit("compose_write 4 x 2160p direct -> v210",
           lambda i: ctx.compose_write_v210([(img[(i + j) % R], w, h, None) for j in range(4)], out_v[i % 2], w, h, 0, *wr),
           4 * ib + vb)
    for n in (1, 2, 4, 8):
        if n <= R:
            timeit("fused_v210_combine_%d 2160p" % n,
                   lambda i, n=n: ctx.fused_v210_combine([v[(i + j) % R] for j in range(n)], out_v[i % 2], w, h, *rd, *wr),
                   (n + 1) * vb)
    # the other pack formats: read and write at 2160p (8-bit matrices for the 8-bit YUV formats)
    rng = {"yuv422p10": (10, 64, 940, 896), "yuv422p8": (8, 16, 235, 224), "yuv420p": (8, 16, 235, 224),
           "nv12": (8, 16, 235, 224), "rgba8": None, "bgra8": None}
    for fmt in only_formats or rng:
        sizes = capi.pack_plane_bytes(fmt, w, h)
        rcm = None if rng[fmt] is None else dev(capi.ycbcr2rgb_matrix("709", *rng[fmt]))
        wcm = None if rng[fmt] is None else dev(capi.rgb2ycbcr_matrix("2020", *rng[fmt]))
        mask = 0x03FF03FF if fmt == "yuv422p10" else -1
        planes = [[torch.randint(0, 2 ** 31, ((n + 3) // 4,), dtype=torch.int32, device="cuda") & mask for n in sizes]
                  for _ in range(R)]
        outp = [[torch.empty((n + 3) // 4, dtype=torch.int32, device="cuda") for n in sizes] for _ in range(2)]
        torch.cuda.synchronize()
        timeit("pack_read %s 2160p" % fmt, lambda i: ctx.pack_read(fmt, planes[i % R], out_img[i % 2], w, h, rcm, rd[1], rd[2]),
               sum(sizes) + ib)
        timeit("pack_write %s 2160p" % fmt, lambda i: ctx.pack_write(fmt, img[i % R], outp[i % 2], w, h, 0, wcm, wr[1]),
               sum(sizes) + ib)
    ctx.close()


if __name__ == "__main__":
    main()
