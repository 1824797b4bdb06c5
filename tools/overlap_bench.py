#!/usr/bin/env python3
"""What two queues would buy a 1080i channel (research for DESIGN.md section 9; nothing in the product does this yet).
BASELINE config 3b's tick - the de-interlacing reader over four windows (packed-RGB fields), then both fields' frames from the 2 x 2-block
compositor - as the product launches it (both on the process queue, in order) and with the reader on the LOAD queue, ordered by events so
that reader(k + 1) runs beside compositor(k): reader(k) | load waits for process's tail (= compositor(k - 1), which read the field set
reader(k + 1) will overwrite) | process waits for load's tail (= reader(k)) | compositor(k).  Two field sets.  The frames of the two ways
are compared.  python tools/overlap_bench.py [ticks=400]"""
import json
import os
import sys

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    import torch
    from phaneron_amd import capi
    ticks = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    ctx = capi.Context(0)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    w, h, L = 1920, 1080, 4
    rd = [dev(capi.ycbcr2rgb_matrix("709")), dev(capi.gamma2linear_lut("709")), dev(np.concatenate([capi.rgb2rgb_matrix("709", "709"), np.zeros(3, np.float32)]))]
    wr = [dev(capi.rgb2ycbcr_matrix("709")), dev(capi.linear2gamma_lut("709"))]
    torch.cuda.synchronize()
    ctx.register_lut(rd[1], capi.gamma2linear_lut("709"))
    ctx.register_lut(wr[1], capi.linear2gamma_lut("709"))
    words = capi.v210_pitch_bytes(w) * h // 4
    R = 5
    src = [[torch.randint(0, 2 ** 30, (words,), dtype=torch.int32, device="cuda") & 0x3FFFFFFF for _ in range(L)] for _ in range(R)]
    fields = [[[torch.empty(w * h * 3, device="cuda") for _ in range(2)] for _ in range(L)] for _ in range(2)]  # [set][layer][field]
    outs = [[torch.zeros(words, dtype=torch.int32, device="cuda") for _ in range(2)] for _ in range(3)]
    mh = capi.transform_matrix(w, h)
    torch.cuda.synchronize()
    P, LD = capi.QUEUE_PROCESS, capi.QUEUE_LOAD

    def win(k, l):
        return (src[k % R][l], src[(k + 1) % R][l], src[(k + 2) % R][l])

    def reader(k, queue):
        s = k & 1
        ctx.v210_yadif_pair([win(k, l) + (fields[s][l][0], fields[s][l][1]) for l in range(L)], w, h, 1, False, *rd, queue=queue, rgb=True)

    def compositor(k, queue):
        s, o = k & 1, outs[k % 3]
        ctx.compose_up_write_v210_pair([(fields[s][l][0], w, h, mh) for l in range(L)], [(fields[s][l][1], w, h, mh) for l in range(L)], o[0], o[1],
                                       w, h, 0, *wr, queue=queue, rgb=True)

    def serial(k):
        reader(k, P)
        compositor(k, P)

    def two_queues(k):
        reader(k, LD)
        ctx.queue_wait_queue(LD, P)   # what load launches NEXT (reader k + 1) waits for compositor(k - 1): it read the set reader(k + 1) writes
        ctx.queue_wait_queue(P, LD)   # compositor(k) waits for reader(k)
        compositor(k, P)

    import time
    res = {"one_queue": [], "two_queues": []}
    frames = {}
    for k in range(6000):  # a second of load first: the chip's clocks settle (bench.py FIXED_WARMUP)
        serial(k)
    for rep in range(4):  # the two ways in turn, four times
        for name, tick in (("one_queue", serial), ("two_queues", two_queues)):
            for k in range(100):
                tick(k)
            ctx.wait(LD), ctx.wait(P)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for k in range(ticks):
                tick(k)
            ctx.wait(LD), ctx.wait(P)
            torch.cuda.synchronize()
            res[name].append(round(1e6 * (time.perf_counter() - t0) / ticks, 2))
            frames[name] = [o.clone() for o in outs[(ticks - 1) % 3]]
    same = all(torch.equal(a, b) for a, b in zip(frames["one_queue"], frames["two_queues"]))
    print(json.dumps({"bench": "overlap", "shape": "4 x 1080i -> both fields (packed RGB) -> own size -> combine_4 -> v210 x 2", "ticks": ticks,
                      "us_per_tick": res, "us_per_field_best": {k: round(min(v) / 2, 2) for k, v in res.items()}, "same_frames": same}), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
