#!/usr/bin/env python3
"""Take the OpenCL built-ins the reference's kernels call from AMD's OWN device library and
retarget them to x86-64, so that oracle/_ref runs the reference kernel text on the arithmetic
ROCm's OpenCL would give it on the GPU - not on a hand-written stand-in.

TEST INFRASTRUCTURE ONLY (build container; output goes to the git-ignored oracle/_ref/).

  /opt/rocm/amdgcn/bitcode/opencl.bc   dot, fma, fmin/fmax, fabs, round, convert_*_sat*, convert_float4
  /opt/rocm/amdgcn/bitcode/ocml.bc     __ocml_fma_f32, __ocml_round_f32, __ocml_fmin/fmax_f32 ... (their callees)
  oclc_*.bc                            the control constants ocml reads (finite_only off, daz off, ...)

Recipe (ROCm's LLVM ships no llvm-extract, so the selection is done with `opt`):
  1. llvm-dis opencl.bc, give the wanted functions external linkage (they are `linkonce_odr hidden`, which
     llvm-link and globaldce drop when unreferenced), llvm-link that with ocml.bc and the oclc_* constants;
  2. in the text of the result: replace the amdgcn triple / datalayout by x86-64's and delete the
     `target-cpu` / `target-features` / amdgpu-* function attributes.  Function BODIES are untouched;
  3. `opt -passes=internalize,globaldce` keeps the wanted functions and what they call;
  4. the module must then be free of amdgcn intrinsics and address spaces (checked), and is compiled
     with `clang -O1 -mfma -ffp-contract=off`.  -mfma matters: the library's dot() is a chain of
     `llvm.fmuladd.f32`, which gfx950 executes as v_fma_f32 / v_fmac_f32 (fused, verified on the GPU by
     tests/test_builtins_gpu.py); with FMA available the x86 backend fuses it the same way.

Only the work-item id functions, the image functions (no CDNA image hardware, OpenCL 1.2 formula) and
the kernel drivers stay in ocl_shim.cpp.
"""
import os
import re
import subprocess
import sys

LLVM = "/opt/rocm/lib/llvm/bin"
BC = "/opt/rocm/amdgcn/bitcode"
LIBS = ["opencl.bc", "ocml.bc", "oclc_finite_only_off.bc", "oclc_daz_opt_off.bc", "oclc_correctly_rounded_sqrt_on.bc",
        "oclc_unsafe_math_off.bc", "oclc_isa_version_950.bc", "oclc_wavefrontsize64_on.bc", "oclc_abi_version_500.bc"]

# mangled name -> OpenCL spelling (every arithmetic built-in left unresolved by the reference's kernels)
WANTED = {
    "_Z3dotDv3_fS_": "dot(float3, float3)",
    "_Z3dotDv4_fS_": "dot(float4, float4)",
    "_Z3fmafff": "fma(float, float, float)",
    "_Z3fmaDv2_fS_S_": "fma(float2, float2, float2)",
    "_Z3fmaDv4_fS_S_": "fma(float4, float4, float4)",
    "_Z4fabsDv4_f": "fabs(float4)",
    "_Z4fminDv4_fS_": "fmin(float4, float4)",
    "_Z4fmaxDv4_fS_": "fmax(float4, float4)",
    "_Z5roundf": "round(float)",
    "_Z14convert_float4Dv4_t": "convert_float4(ushort4)",
    "_Z14convert_float4Dv4_h": "convert_float4(uchar4)",
    "_Z21convert_uchar_sat_rtef": "convert_uchar_sat_rte(float)",
    "_Z18convert_ushort_satf": "convert_ushort_sat(float)",
    "_Z22convert_ushort_sat_rtef": "convert_ushort_sat_rte(float)",
    "_Z22convert_ushort_sat_rtzf": "convert_ushort_sat_rtz(float)",
}

# (mangled name, wrapper head + argument bitcasts, the call, result bitcast + ret)
ABI_GLUE = [
    ("_Z14convert_float4Dv4_h",
     "define <4 x float> @_Z14convert_float4Dv4_h(i32 %x) {\n  %v = bitcast i32 %x to <4 x i8>\n",
     "  %r = call <4 x float> @IMPL(<4 x i8> %v)\n", "  ret <4 x float> %r\n}"),
    ("_Z14convert_float4Dv4_t",
     "define <4 x float> @_Z14convert_float4Dv4_t(double %x) {\n  %v = bitcast double %x to <4 x i16>\n",
     "  %r = call <4 x float> @IMPL(<4 x i16> %v)\n", "  ret <4 x float> %r\n}"),
    ("_Z3fmaDv2_fS_S_",
     "define double @_Z3fmaDv2_fS_S_(double %a, double %b, double %c) {\n  %va = bitcast double %a to <2 x float>\n"
     "  %vb = bitcast double %b to <2 x float>\n  %vc = bitcast double %c to <2 x float>\n",
     "  %r = call <2 x float> @IMPL(<2 x float> %va, <2 x float> %vb, <2 x float> %vc)\n",
     "  %d = bitcast <2 x float> %r to double\n  ret double %d\n}"),
]

X86_DATALAYOUT = "e-m:e-p270:32:32-p271:32:32-p272:64:64-i64:64-i128:128-f80:128-n8:16:32:64-S128"
X86_TRIPLE = "x86_64-unknown-linux-gnu"


def run(cmd, **kw):
    return subprocess.run(cmd, check=True, **kw)


def build(out_dir, cflags):
    """Writes <out_dir>/devlib_builtins.o (x86-64) and <out_dir>/devlib_builtins.ll (what was compiled)."""
    os.makedirs(out_dir, exist_ok=True)
    # (1) the wanted functions are `linkonce_odr hidden` in opencl.bc: llvm-link (and globaldce) drop such
    # definitions unless something references them, so they get external linkage first - then linking
    # pulls in exactly their callees from ocml.bc and the oclc_* constants
    ocl = subprocess.run([LLVM + "/llvm-dis", os.path.join(BC, "opencl.bc"), "-o", "-"], check=True, capture_output=True,
                         text=True).stdout
    missing = [m for m in WANTED if not re.search(r"^define [^\n]*@%s\(" % re.escape(m), ocl, re.M)]
    if missing:
        raise SystemExit("opencl.bc does not define: %s" % ", ".join(missing))
    for m in WANTED:
        ocl = re.sub(r"^define linkonce_odr hidden ([^\n]*@%s\()" % re.escape(m), r"define \1", ocl, flags=re.M)
    patched = os.path.join(out_dir, "opencl_patched.ll")
    with open(patched, "w") as f:
        f.write(ocl)
    linked = os.path.join(out_dir, "devlib_linked.bc")
    run([LLVM + "/llvm-link", patched] + [os.path.join(BC, l) for l in LIBS[1:]] + ["-o", linked])
    os.remove(patched)
    text = subprocess.run([LLVM + "/llvm-dis", linked, "-o", "-"], check=True, capture_output=True, text=True).stdout
    os.remove(linked)
    # (2) linkage of the wanted functions; target; attributes
    text = re.sub(r'^target datalayout = "[^"]*"', 'target datalayout = "%s"' % X86_DATALAYOUT, text, flags=re.M)
    text = re.sub(r'^target triple = "[^"]*"', 'target triple = "%s"' % X86_TRIPLE, text, flags=re.M)
    text = re.sub(r'\s*"target-cpu"="[^"]*"', "", text)
    text = re.sub(r'\s*"target-features"="[^"]*"', "", text)
    text = re.sub(r'\s*"amdgpu-[a-z0-9-]*"(="[^"]*")?', "", text)
    text = re.sub(r'\s*"uniform-work-group-size"="[^"]*"', "", text)
    # (2b) ABI glue.  The kernels were compiled by clang for the x86-64 C ABI, which coerces small vector
    # arguments: uchar4 -> i32 (a general register), ushort4 and float2 -> double.  The library bodies take
    # the IR vector types.  Those three are renamed and fronted by bitcast-only wrappers with the C signature
    # (no arithmetic in the glue); every other wanted function has the same signature on both sides.
    for name, pre, call, post in ABI_GLUE:
        text = text.replace("@%s(" % name, "@devlib%s(" % name)
        text += "\n" + pre + call.replace("@IMPL", "@devlib" + name) + post + "\n"
    retargeted = os.path.join(out_dir, "devlib_retargeted.ll")
    with open(retargeted, "w") as f:
        f.write(text)
    # (3) keep the wanted functions and their callees
    kept = os.path.join(out_dir, "devlib_builtins.ll")
    run([LLVM + "/opt", "-passes=internalize,globaldce", "-internalize-public-api-list=" + ",".join(WANTED),
         retargeted, "-S", "-o", kept])
    os.remove(retargeted)
    body = open(kept).read()
    # (4) nothing GPU-specific may be left
    bad = sorted(set(re.findall(r"@llvm\.amdgcn\.[A-Za-z0-9_.]+", body)))
    if bad:
        raise SystemExit("amdgcn intrinsics survive in the selection: %s" % ", ".join(bad))
    if re.search(r"addrspace\([1-9]", body):
        raise SystemExit("non-default address spaces survive in the selection")
    obj = os.path.join(out_dir, "devlib_builtins.o")
    run([LLVM + "/clang", "-x", "ir", "-c", kept, "-o", obj, "-fPIC", "-Wno-override-module"] + list(cflags))
    return obj


if __name__ == "__main__":
    out = sys.argv[1]
    print(build(out, sys.argv[2:]))
