#!/bin/bash
# Build oracle/_ref/refgpu/<fmt>.co: the reference's own OpenCL C text of the seven pack formats (the buffer kernels `read` /
# `write` of src/process/{v210,yuv422p10,yuv422p8,yuv420p,nv12,rgba8,bgra8}.ts), taken from /root/reference where it lies and
# compiled UNMODIFIED for gfx950 by the ROCm OpenCL toolchain: clang -x cl, AMD's device library (opencl.bc, ocml.bc, ockl.bc
# and the oclc_* control libraries) - what the OpenCL runtime's own compiler would build from that text for this device.
# The MI355X is an OpenCL device on the GPU box but reports no image support (profiles/r03_opencl_probe.txt): these buffer
# kernels are the part of the reference's path that can run there at all; tests/test_ref_on_gpu.py loads the code objects
# through the HIP module API and compares this repository's kernels with them bit for bit on the same device.
# TEST INFRASTRUCTURE ONLY; the outputs are compiled artefacts (git-ignored, they travel like the other oracle/_ref builds).
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/../_ref/refgpu"
WORK="$HERE/../_ref/work"
LLVM=/opt/rocm/lib/llvm/bin
BC=/opt/rocm/amdgcn/bitcode
[ -d "$WORK/cl" ] || { echo "no extracted kernel text ($WORK/cl): run build_ref.sh first" >&2; exit 0; }
mkdir -p "$OUT"
LIBS=""
for l in opencl ocml ockl oclc_isa_version_950 oclc_abi_version_500 oclc_correctly_rounded_sqrt_off oclc_daz_opt_off oclc_finite_only_off oclc_unsafe_math_off oclc_wavefrontsize64_on; do
  LIBS="$LIBS -Xclang -mlink-builtin-bitcode -Xclang $BC/$l.bc"
done
for fmt in v210 yuv422p10 yuv422p8 yuv420p nv12 rgba8 bgra8; do
  "$LLVM/clang" -x cl -cl-std=CL1.2 -Xclang -finclude-default-header -target amdgcn-amd-amdhsa -mcpu=gfx950 -O3 -nogpulib $LIBS \
    "$WORK/cl/$fmt.cl" -o "$OUT/$fmt.co"
done
echo "built $(ls "$OUT" | wc -l) gfx950 code objects of the reference's pack kernels in $OUT"
