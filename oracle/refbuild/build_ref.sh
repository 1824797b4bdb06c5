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
CLFLAGS="-x cl -cl-std=CL1.2 -Xclang -finclude-default-header -target x86_64-unknown-linux-gnu -O1 -fPIC"
OBJS=""
for cl in "$OUT"/*.cl; do
  o="${cl%.cl}.o"
  "$LLVM/clang" $CLFLAGS -c "$cl" -o "$o"
  OBJS="$OBJS $o"
done
# every pack format names its kernels read/write: rename per format (and away from libc's)
for fmt in v210 yuv422p10 yuv422p8 yuv420p nv12 rgba8 bgra8; do
  "$LLVM/llvm-objcopy" --redefine-sym read=refk_${fmt}_read --redefine-sym write=refk_${fmt}_write \
    --redefine-sym __clang_ocl_kern_imp_read=refk_imp_${fmt}_read \
    --redefine-sym __clang_ocl_kern_imp_write=refk_imp_${fmt}_write "$OUT/$fmt.o"
done
"$LLVM/clang++" -O1 -fPIC -ffp-contract=off -std=c++17 -c "$HERE/ocl_shim.cpp" -o "$OUT/ocl_shim.o"
"$LLVM/clang++" -shared -o "$OUT/libphaneron_ref.so" "$OUT/ocl_shim.o" $OBJS -lm
echo "built $OUT/libphaneron_ref.so"
