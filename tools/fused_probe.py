#!/usr/bin/env python3
"""Where the headline kernel's time goes, measured inside the real kernel: builds lib/libphaneron_hip_probe.so
(-DPH_PROBE=1: wave 0 of every workgroup stamps s_memtime / s_memrealtime at the phase boundaries), runs the
bench workload once per variant and prints the sustained shader clock and the per-phase share.
usage: python tools/fused_probe.py            (on the GPU box; PH_FUSED_PIPE=0|1 selects the variant)"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, ROOT)
from phaneron_amd import build  # noqa: E402

# PH_PROBE_ABLATE=1|8: also build in the timing experiments of ph_ldslut.h (1 = no LDS reads, 8 = conflict-free reads;
# WRONG results, timing only) to see what the VALU stream costs on its own
ablate = os.environ.get("PH_PROBE_ABLATE", "")
variant = "probe" + ablate
lib_path = build.variant_path(variant)
if not os.path.exists(lib_path) or "--build" in sys.argv:
    build.build(variant=variant, extra_flags=["-DPH_PROBE=1"] + (["-DPH_ABLATE=" + ablate] if ablate else []))
if "--build-only" in sys.argv:
    sys.exit(0)
os.environ["PHANERON_HIP_LIB"] = lib_path
import numpy as np  # noqa: E402
import torch  # noqa: E402
from phaneron_amd import capi  # noqa: E402

sys.argv = [sys.argv[0]]
import bench  # noqa: E402

w, h, n = 3840, 2160, 4
ctx = capi.Context(0)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
rd = [dev(capi.ycbcr2rgb_matrix("709")), dev(capi.gamma2linear_lut("709")),
      dev(np.concatenate([capi.rgb2rgb_matrix("709", "2020"), np.zeros(3, np.float32)]))]
wr = [dev(capi.rgb2ycbcr_matrix("2020")), dev(capi.linear2gamma_lut("2020"))]
torch.cuda.synchronize()
ctx.register_lut(rd[1], capi.gamma2linear_lut("709"))
ctx.register_lut(wr[1], capi.linear2gamma_lut("2020"))
ring = [[bench.synth_v210(torch, w, h, 0x5EED0000 + 16 * r + l, torch.device("cuda", 0)) for l in range(n)] for r in range(8)]
out = torch.empty(capi.v210_pitch_bytes(w) * h // 4, dtype=torch.int32, device="cuda")
torch.cuda.synchronize()
for i in range(400):
    ctx.fused_v210_combine(ring[i % 8], out, w, h, *rd, *wr)
ctx.wait()
probe = np.zeros(256 * 12, np.uint64)
rc = capi.lib().__getattr__("ph_debug_fused_probe")(probe.ctypes.data_as(C.c_void_p), probe.size)
assert rc == 0, rc
p = probe.reshape(256, 6, 2).astype(np.float64)[:, :5]
cyc = p[:, 1:, 0] - p[:, :-1, 0]           # per workgroup: table load, phase 1, table swap, phase 2 (shader cycles)
real = p[:, -1, 1] - p[:, 0, 1]            # 100 MHz ticks
total = p[:, -1, 0] - p[:, 0, 0]
ghz = total / real * 0.1
names = ["reader table -> LDS", "phase 1 (unpack, CSC, LUT, gamut, combine)", "barrier + writer table -> LDS", "phase 2 (LUT, CSC, pack, store)"]
res = {"pipe": os.environ.get("PH_FUSED_PIPE", "0"), "ablate": ablate or "0", "workgroups": 256, "shader_ghz_median": round(float(np.median(ghz)), 3),
       "tile_cycles_median": int(np.median(total)), "tile_us_median": round(float(np.median(real)) / 100.0, 2),
       "phases": {nm: {"cycles_median": int(np.median(cyc[:, i])), "share": round(float(np.median(cyc[:, i]) / np.median(total)), 3)}
                  for i, nm in enumerate(names)},
       "start_skew_us": round(float(p[:, 0, 1].max() - p[:, 0, 1].min()) / 100.0, 2),
       "end_skew_us": round(float(p[:, -1, 1].max() - p[:, -1, 1].min()) / 100.0, 2)}
print(json.dumps(res))
ctx.close()
