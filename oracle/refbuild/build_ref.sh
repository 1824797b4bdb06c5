#!/bin/bash
# Build oracle/_ref/libphaneron_ref.so: the reference's own OpenCL C kernel text, taken from
# /root/reference where it lies, compiled UNMODIFIED for x86-64 and linked with
#   - devlib_builtins.o : the OpenCL arithmetic built-ins the kernels call (dot, fma, fmin/fmax, round,
#                         convert_*_sat*, ...) taken from AMD's own device library
#                         (/opt/rocm/amdgcn/bitcode/opencl.bc + ocml.bc) and retargeted to x86-64
#                         by devlib_builtins.py - function bodies unchanged;
#   - ocl_shim.cpp      : work-item ids, image access (OpenCL 1.2 s8.2) and the NDRange loops.
# TEST INFRASTRUCTURE ONLY; output is git-ignored.  Only runs where the reference checkout exists
# (the build container).
#
# Layout: oracle/_ref/*.so is what travels to the GPU box (bench.py's cpu_baseline "reference" and the
# built-in equivalence tests load it); oracle/_ref/work/ holds everything else derived from reference
# text (kernel .cl, objects, stripped JS, LUT dumps) and is listed in .gpurunignore.
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/../_ref"
WORK="$OUT/work"
LLVM=/opt/rocm/lib/llvm/bin
REF="${PHANERON_REFERENCE:-/root/reference}"
if [ ! -d "$REF/src/process" ]; then
  echo "reference checkout not present at $REF - skipping oracle/_ref build" >&2
  exit 0
fi
mkdir -p "$WORK/cl"
python3 "$HERE/extract_kernels.py"
# Two builds of the same sources:
#   libphaneron_ref.so      -O1: what tests/golden/gen_golden.py runs
#   libphaneron_ref_fast.so -O3 -mavx2: bench.py's cpu_baseline "reference".
#                           -ffp-contract=off keeps its results identical to the first (tests check).
# Both use -mfma for the device-library object: its dot() is a chain of llvm.fmuladd, which gfx950
# executes fused (v_fmac_f32); without FMA the x86 backend would split it into mul + add.
build_one() {  # $1 = output name, $2 = object dir, $3.. = extra flags
  local out="$1" dir="$2"; shift 2
  mkdir -p "$dir"
  local objs=""
  for cl in "$WORK"/cl/*.cl; do
    local o="$dir/$(basename "${cl%.cl}").o"
    "$LLVM/clang" -x cl -cl-std=CL1.2 -Xclang -finclude-default-header -target x86_64-unknown-linux-gnu -fPIC "$@" -c "$cl" -o "$o"
    objs="$objs $o"
  done
  # every pack format names its kernels read/write: rename per format (and away from libc's)
  for fmt in v210 yuv422p10 yuv422p8 yuv420p nv12 rgba8 bgra8; do
    "$LLVM/llvm-objcopy" --redefine-sym read=refk_${fmt}_read --redefine-sym write=refk_${fmt}_write \
      --redefine-sym __clang_ocl_kern_imp_read=refk_imp_${fmt}_read \
      --redefine-sym __clang_ocl_kern_imp_write=refk_imp_${fmt}_write "$dir/$fmt.o"
  done
  python3 "$HERE/devlib_builtins.py" "$dir" "$@" -mfma -ffp-contract=off > /dev/null
  "$LLVM/clang++" -fPIC -ffp-contract=off -std=c++17 "$@" -c "$HERE/ocl_shim.cpp" -o "$dir/ocl_shim.o"
  "$LLVM/clang++" -shared "$@" -o "$OUT/$out" "$dir/ocl_shim.o" "$dir/devlib_builtins.o" $objs -lm -lpthread
  echo "built $OUT/$out"
}
build_one libphaneron_ref.so "$WORK/obj" -O1
build_one libphaneron_ref_fast.so "$WORK/obj_fast" -O3 -mavx2 -mfma -ffp-contract=off
