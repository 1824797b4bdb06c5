"""Build libphaneron_hip.so (hand-written gfx950 kernels + the C ABI) in-tree with hipcc.

hipcc cross-compiles for gfx950 without a GPU, so this runs in the build container; the
resulting .so travels to the GPU box with the repository snapshot.
"""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIB_DIR, "libphaneron_hip.so")
# experiment builds (PH_PROBE, PH_FUSED_SPLIT ...: tools/*_probe.py) live outside the package: lib/ holds the product only
VARIANT_DIR = os.path.normpath(os.path.join(HERE, "..", "tools", "_variants"))
ARCH = "gfx950"

SOURCES = ["ph_kernels.hip", "ph_kernels_lds.hip", "ph_kernels_fmt.hip", "ph_kernels_deint.hip", "ph_kernels_chan.hip", "ph_kernels_up.hip", "ph_api.cpp", "ph_program.cpp", "ph_colour.cpp", "ph_lut.cpp"]
HEADERS = ["ph_device.h", "ph_kernels.h", "ph_lut.h", "ph_lut_host.h", "ph_ldslut.h", "ph_program.h", "ph_yadif.h", os.path.join("..", "..", "include", "phaneron_hip.h")]
# -ffp-contract=off: every fused multiply-add in the kernels is explicit (parity with the
# reference's OpenCL arithmetic); no fast-math anywhere.
# -fno-slp-vectorize: the SLP vectoriser pairs independent f32 operations into v_pk_add / v_pk_fma_f32.  On gfx950 a
# packed f32 instruction occupies the issue slot as long as its two halves would (tools/opbench3: 4.2 cycles), and the
# register PAIRS it needs cost v_mov shuffles and spills: the de-interlacing reader lost 16 % to it.
COMMON = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize", "-Wall", "-Wno-unused-result",
          "-Wno-unused-value", "-Wno-int-to-pointer-cast",
          # A kernel's argument struct is a private copy of the kernarg segment that instcombine replaces by direct scalar
          # loads only if it can walk all its users - at most 300 by default.  The channel compositor indexes its op list
          # in several inlined places and crossed that line: the whole 1.7 KiB struct went to scratch (1792 B per lane, 4x slower).
          "-mllvm", "-instcombine-max-copied-from-constant-users=8000"]


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: libphaneron_hip.so cannot be built (there is no CPU fallback)")
    return exe


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def variant_path(variant):
    return os.path.join(VARIANT_DIR, "libphaneron_hip_%s.so" % variant)


def build(force=False, verbose=False, extra_flags=(), variant=None):
    """Compile every HIP source for gfx950 and link libphaneron_hip.so.  Returns its path.
    variant: build tools/_variants/libphaneron_hip_<variant>.so with extra_flags (A/B experiments)."""
    out_dir = LIB_DIR if variant is None else VARIANT_DIR
    lib = LIB if variant is None else variant_path(variant)
    if variant is None and not force and not _stale():
        return LIB
    os.makedirs(out_dir, exist_ok=True)
    cc = hipcc()
    objs, jobs = [], []
    hdr_time = max(os.path.getmtime(os.path.join(CSRC, h)) for h in HEADERS)
    hdr_time = max(hdr_time, os.path.getmtime(os.path.abspath(__file__)))
    for src in SOURCES:
        obj = os.path.join(out_dir, os.path.splitext(src)[0] + ("" if variant is None else "_" + variant) + ".o")
        objs.append(obj)
        # an object is rebuilt when its source, any header or this script is newer (force: always)
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(hdr_time, os.path.getmtime(os.path.join(CSRC, src))):
            continue
        cmd = [cc, "--offload-arch=" + ARCH, "-x", "hip", "-c", os.path.join(CSRC, src), "-o", obj] + COMMON + list(extra_flags)
        if verbose:
            print(" ".join(cmd))
        jobs.append((src, subprocess.Popen(cmd)))  # the sources compile side by side
    for src, proc in jobs:
        if proc.wait() != 0:
            for _, other in jobs:
                other.wait()
            raise subprocess.CalledProcessError(proc.returncode, "hipcc " + src)
    cmd = [cc, "--offload-arch=" + ARCH, "-shared", "-o", lib] + objs + ["-ldl"]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return lib


if __name__ == "__main__":
    print(build(force=True, verbose=True))
