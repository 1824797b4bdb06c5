#!/usr/bin/env python3
"""Where the fused field kernel's time goes (config 3 shape), from s_memtime stamps inside the kernel
(libphaneron_hip_probe.so, -DPH_PROBE=1).  usage (GPU box): python tools/field_probe.py"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, ROOT)
from phaneron_amd import build  # noqa: E402

lib_path = os.path.join(ROOT, "phaneron_amd", "lib", "libphaneron_hip_probe.so")
if not os.path.exists(lib_path) or "--build" in sys.argv:
    build.build(variant="probe", extra_flags=["-DPH_PROBE=1"])
if "--build-only" in sys.argv:
    sys.exit(0)
os.environ["PHANERON_HIP_LIB"] = lib_path
import numpy as np  # noqa: E402
import torch  # noqa: E402
from phaneron_amd import capi  # noqa: E402

sw, sh, ow, oh = 1920, 1080, 3840, 2160
ctx = capi.Context(0)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
wr = [dev(capi.rgb2ycbcr_matrix("2020")), dev(capi.linear2gamma_lut("2020"))]
torch.cuda.synchronize()
ctx.register_lut(wr[1], capi.linear2gamma_lut("2020"))
mh = capi.transform_matrix(ow, oh)
m = dev(mh)
frames = [[torch.rand(sw * sh * 4, device="cuda") for _ in range(3)] for _ in range(4)]
out = torch.empty(capi.v210_pitch_bytes(ow) * oh // 4, dtype=torch.int32, device="cuda")
torch.cuda.synchronize()
layers = [dict(prev=f[0], cur=f[1], next=f[2], width=sw, height=sh, matrix=m, matrix_host=mh, deinterlace=True, parity=0, tff=1) for f in frames]
for _ in range(20):
    ctx.fused_field_v210(layers, out, ow, oh, *wr)
ctx.wait()
probe = np.zeros(256 * 8, np.uint64)
assert capi.lib().__getattr__("ph_debug_field_probe")(probe.ctypes.data_as(C.c_void_p), probe.size) == 0
p = probe.reshape(256, 8).astype(np.float64)[:240, :3]
d = p[:, 1:] - p[:, :-1]
names = ["slice 0: build the windows (yadif)", "slice 0: sample + combine"]
print(json.dumps({"slice0_cycles_median": int(np.median(p[:, 2] - p[:, 0])),
                  "stages": {n: int(np.median(d[:, i])) for i, n in enumerate(names)}}))
ctx.close()
