"""The reference's own pack kernels ON THE MI355X: oracle/_ref/refgpu/<fmt>.co is the OpenCL C text of src/process/<fmt>.ts
compiled unmodified for gfx950 by the ROCm OpenCL toolchain (oracle/refbuild/build_ref_gpu.sh, build container only); here the
code objects are loaded through the HIP module API and launched with the work-group geometry the reference's Reader / Writer
classes pass to createProgram (recorded in tests/golden/host_trace.json).  Test infrastructure: the checker, never the product."""
import ctypes as C
import json
import os

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
DIR = os.path.join(ROOT, "oracle", "_ref", "refgpu")
FORMATS = ["v210", "yuv422p10", "yuv422p8", "yuv420p", "nv12", "rgba8", "bgra8"]


def available():
    return all(os.path.exists(os.path.join(DIR, f + ".co")) for f in FORMATS)


def geometry():
    """(format, 'read' | 'write') -> [(globalWorkItems, workItemsPerGroup)] at 1920 x 1080, as the reference's classes computed them"""
    trace = json.load(open(os.path.join(ROOT, "tests", "golden", "host_trace.json")))
    sha = json.load(open(os.path.join(ROOT, "tests", "golden", "kernel_text_sha.json")))
    out = {}
    for e in trace:
        if e["op"] == "createProgram" and e["name"] in ("read", "write"):
            g = (e["globalWorkItems"], e["workItemsPerGroup"])
            out.setdefault((sha[e["srcSha"]], e["name"]), [])
            if g not in out[(sha[e["srcSha"]], e["name"])]:
                out[(sha[e["srcSha"]], e["name"])].append(g)
    return out


class RefGpu:
    def __init__(self):
        import torch  # noqa: F401  (brings the HIP runtime in)
        self.hip = C.CDLL("libamdhip64.so")
        self.fn = {}
        for fmt in FORMATS:
            mod = C.c_void_p()
            rc = self.hip.hipModuleLoad(C.byref(mod), os.path.join(DIR, fmt + ".co").encode())
            assert rc == 0, "hipModuleLoad(%s) -> %d" % (fmt, rc)
            for name in ("read", "write"):
                f = C.c_void_p()
                rc = self.hip.hipModuleGetFunction(C.byref(f), mod, name.encode())
                assert rc == 0, "hipModuleGetFunction(%s.%s) -> %d" % (fmt, name, rc)
                self.fn[(fmt, name)] = f

    def launch(self, fmt, name, global_items, group_items, args):
        """args: torch tensors (device pointers) or ints (32-bit by-value)"""
        import torch
        vals = []
        for a in args:
            vals.append(C.c_uint32(a) if isinstance(a, int) else C.c_void_p(a.data_ptr()))
        params = (C.c_void_p * len(vals))(*[C.cast(C.pointer(v), C.c_void_p) for v in vals])
        assert global_items % group_items == 0
        rc = self.hip.hipModuleLaunchKernel(self.fn[(fmt, name)], global_items // group_items, 1, 1, group_items, 1, 1, 0, None, params, None)
        assert rc == 0, "hipModuleLaunchKernel(%s.%s) -> %d" % (fmt, name, rc)
        torch.cuda.synchronize()
