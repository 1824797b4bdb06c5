#!/usr/bin/env python3
"""Extract the reference's OpenCL C kernel text into oracle/_ref/work/cl/ (never committed, never shipped to the GPU box).

TEST INFRASTRUCTURE ONLY.  Runs only in the build container, where the reference
checkout exists at /root/reference.  The kernel text is read from the sources where
they lie and written under oracle/_ref/work/ (git-ignored, gpurun-ignored); nothing derived from it is
committed except golden input/output *data* under tests/golden/.

Static kernels are template strings (`const xxxKernel = `...``):
  src/process/v210.ts:24-196, yuv422p10.ts:24-219, yuv422p8.ts:24-219, yuv420p.ts:24-250,
  nv12.ts:24-245, rgba8.ts:24-102, bgra8.ts:24-102, yadifCl.ts:27-168, transform.ts:24-60, resize.ts:24-60,
  mix.ts:23-46, wipe.ts:23-48
Generated kernels come from two string-builder arrow functions which are evaluated
under node after dropping their TypeScript annotations:
  src/process/combine.ts:24-68 (getCombineKernel), transition.ts:24-81 (getTransitionKernel)
"""
import os
import re
import subprocess
import sys

REF = os.environ.get("PHANERON_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "_ref", "work", "cl")

STATIC = [
    ("v210.ts", "v210Kernel", "v210"),
    ("yuv422p10.ts", "yuv422p10leKernel", "yuv422p10"),
    ("yuv422p8.ts", "yuv422p8Kernel", "yuv422p8"),
    ("yuv420p.ts", "yuv420pKernel", "yuv420p"),
    ("nv12.ts", "nv12Kernel", "nv12"),
    ("rgba8.ts", "rgba8Kernel", "rgba8"),
    ("bgra8.ts", "bgra8Kernel", "bgra8"),
    ("yadifCl.ts", "yadifKernel", "yadif"),
    ("transform.ts", "transformKernel", "transform"),
    ("resize.ts", "resizeKernel", "resize"),
    ("mix.ts", "mixKernel", "mix"),
    ("wipe.ts", "wipeKernel", "wipe"),
]


def read_src(name):
    with open(os.path.join(REF, "src", "process", name)) as f:
        return f.read()


def static_kernel(ts_name, const_name):
    src = read_src(ts_name)
    m = re.search(r"const %s = `(.*?)`" % const_name, src, re.S)
    if not m:
        raise SystemExit("kernel string %s not found in %s" % (const_name, ts_name))
    return m.group(1)


def builder_fn(ts_name, fn_name):
    """Return the JS text of `const fn = (...) => {...}` with TS annotations dropped."""
    src = read_src(ts_name)
    start = src.index("const %s = " % fn_name)
    end = src.index("\nexport default", start)
    js = src[start:end]
    js = re.sub(r"\((\w+): (number|string)\): string =>", r"(\1) =>", js)
    return js


def run_node(js):
    return subprocess.run(["node", "-e", js], check=True, capture_output=True, text=True).stdout


def main():
    os.makedirs(OUT, exist_ok=True)
    for ts_name, const_name, stem in STATIC:
        with open(os.path.join(OUT, stem + ".cl"), "w") as f:
            f.write(static_kernel(ts_name, const_name))
    comb = builder_fn("combine.ts", "getCombineKernel")
    for n in range(2, 9):
        txt = run_node(comb + "\nprocess.stdout.write(getCombineKernel(%d))" % n)
        with open(os.path.join(OUT, "combine_%d.cl" % n), "w") as f:
            f.write(txt)
    trans = builder_fn("transition.ts", "getTransitionKernel")
    for t in ("dissolve", "wipe"):
        txt = run_node(trans + "\nprocess.stdout.write(getTransitionKernel('%s'))" % t)
        with open(os.path.join(OUT, "transition_%s.cl" % t), "w") as f:
            f.write(txt)
    print("extracted kernels into", os.path.normpath(OUT))


if __name__ == "__main__":
    sys.exit(main())
