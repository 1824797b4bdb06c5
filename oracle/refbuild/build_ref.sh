#!/bin/bash
# Build oracle/_ref/libphaneron_ref.so: the reference's own OpenCL C kernel text, taken from
# /root/reference where it lies, compiled UNMODIFIED for x86-64 and linked with ocl_shim.cpp
# (the OpenCL built-ins it calls).  TEST INFRASTRUCTURE ONLY; output is git-ignored.
# Only runs where the reference checkout exists (the build container).
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/../_ref"
LLVM=/opt/rocm/lib/llvm/bin
REF="${PHANERON_REFERENCE:-/root/reference}"
if [ ! -d "$REF/src/process" ]; then
  echo "reference checkout not present at $REF - skipping oracle/_ref build" >&2
  exit 0
fi
mkdir -p "$OUT"
python3 "$HERE/extract_kernels.py"
# Two builds of the same sources:
#   libphaneron_ref.so      -O1, baseline x86-64: what tests/golden/gen_golden.py runs
#   libphaneron_ref_fast.so -O3 -mavx2 -mfma, built as one LTO unit so the built-ins inline into the
#                           kernels like an OpenCL CPU runtime would: bench.py's cpu_baseline "reference".
#                           -ffp-contract=off keeps its results identical to the first (tests check).
build_one() {  # $1 = output name, $2 = object dir, $3.. = extra flags
  local out="$1" dir="$2"; shift 2
  mkdir -p "$dir"
  local objs=""
  for cl in "$OUT"/*.cl; do
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
  "$LLVM/clang++" -fPIC -ffp-contract=off -std=c++17 "$@" -c "$HERE/ocl_shim.cpp" -o "$dir/ocl_shim.o"
  "$LLVM/clang++" -shared "$@" -o "$OUT/$out" "$dir/ocl_shim.o" $objs -lm -lpthread
  echo "built $OUT/$out"
}
build_one libphaneron_ref.so "$OUT" -O1
build_one libphaneron_ref_fast.so "$OUT/fast" -O3 -mavx2 -mfma -ffp-contract=off
