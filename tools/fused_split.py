#!/usr/bin/env python3
"""What would a split of the headline kernel into reader CUs and writer CUs buy?  (option i.)  Measured, not
built: the PH_FUSED_SPLIT variant of the library runs the kernel's two halves as launches of their own - reader half (table,
phase 1, packed indices stored to a hand-over buffer) and writer half (indices loaded, table, phase 2) - on all CUs and on
the CU counts a split would give each role.  A split can be no faster than max(reader half on R CUs, writer half on W CUs)
(both halves here already pay the hand-over traffic; the cross-CU flags, the pipeline fill and drain come on top).
  python tools/fused_split.py            one JSON line per measurement, then the bound for every R + W = 256"""
import json
import os
import subprocess
import sys

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, ROOT)


def child(mode, cus):
    import numpy as np
    import torch
    from phaneron_amd import capi
    sys.path.insert(0, ROOT)
    import bench
    ctx = capi.Context(0)
    stream = ctx.torch_stream()
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    w, h, n, R = 3840, 2160, 4, 8
    rd = [dev(capi.ycbcr2rgb_matrix("709")), dev(capi.gamma2linear_lut("709")), dev(np.concatenate([capi.rgb2rgb_matrix("709", "2020"), np.zeros(3, np.float32)]))]
    wr = [dev(capi.rgb2ycbcr_matrix("2020")), dev(capi.linear2gamma_lut("2020"))]
    torch.cuda.synchronize()
    ctx.register_lut(rd[1], capi.gamma2linear_lut("709"))
    ctx.register_lut(wr[1], capi.linear2gamma_lut("2020"))
    ring = [([bench.synth_v210(torch, w, h, 0x5EED0000 + 16 * r + l, torch.device("cuda", 0)) for l in range(n)],
             torch.empty(capi.v210_pitch_bytes(w) * h // 4, dtype=torch.int32, device="cuda")) for r in range(R)]
    torch.cuda.synchronize()
    for i in range(600):
        ctx.fused_v210_combine(ring[i % R][0], ring[i % R][1], w, h, *rd, *wr)
    ctx.wait()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 600
    e0.record(stream)
    for i in range(reps):
        ctx.fused_v210_combine(ring[i % R][0], ring[i % R][1], w, h, *rd, *wr)
    e1.record(stream)
    ctx.wait()
    print(json.dumps({"mode": {0: "fused (shipped form)", 1: "reader half", 2: "writer half", 3: "both halves, back to back"}[mode],
                      "cus": cus or 256, "us_per_frame": round(1e3 * e0.elapsed_time(e1) / reps, 2)}), flush=True)
    ctx.close()


def main():
    if len(sys.argv) > 2:
        return child(int(sys.argv[1]), int(sys.argv[2]))
    from phaneron_amd import build
    lib = build.build(extra_flags=["-DPH_FUSED_SPLIT=1"], variant="split")  # objects are rebuilt only when a source changed
    res = {}
    runs = [(0, 0), (3, 0), (1, 0), (2, 0)] + [(1, c) for c in (224, 216, 208, 200, 192)] + [(2, c) for c in (32, 40, 48, 56, 64)]
    for mode, cus in runs:
        env = dict(os.environ, PHANERON_HIP_LIB=lib, PH_FUSED_MODE=str(mode))
        if cus:
            env["PH_FUSED_CUS"] = str(cus)
        out = subprocess.run([sys.executable, os.path.abspath(__file__), str(mode), str(cus)], env=env, capture_output=True, text=True).stdout
        line = [l for l in out.splitlines() if l.startswith("{")]
        if line:
            print(line[-1], flush=True)
            d = json.loads(line[-1])
            res[(mode, d["cus"])] = d["us_per_frame"]
    best = None
    for w_cus in (32, 40, 48, 56, 64):
        r_cus = 256 - w_cus
        if (1, r_cus) in res and (2, w_cus) in res:
            bound = max(res[(1, r_cus)], res[(2, w_cus)])
            print(json.dumps({"split": "%d reader CUs + %d writer CUs" % (r_cus, w_cus), "reader_half_us": res[(1, r_cus)],
                              "writer_half_us": res[(2, w_cus)], "lower_bound_us": bound}), flush=True)
            best = bound if best is None else min(best, bound)
    if best is not None and (0, 256) in res:
        print(json.dumps({"fused_us": res[(0, 256)], "best_split_lower_bound_us": best,
                          "verdict": "a split cannot beat the fused kernel" if best >= res[(0, 256)] else
                          "a split could save at most %.1f us before its own protocol costs" % (res[(0, 256)] - best)}), flush=True)


if __name__ == "__main__":
    main()
