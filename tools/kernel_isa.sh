#!/bin/bash
# Disassemble the gfx950 code of one object / library of phaneron_amd/lib into /tmp/isa/<name>.s and print, per kernel,
# registers, scratch and the instruction classes that matter here.  usage: bash tools/kernel_isa.sh ph_kernels_deint.o
set -e
LLVM=/opt/rocm/lib/llvm/bin
ROOT=$(cd "$(dirname "$0")/.." && pwd)
f=${1:-libphaneron_hip.so}
mkdir -p /tmp/isa && cd /tmp/isa && rm -f "$f".* && cp "$ROOT/phaneron_amd/lib/$f" . && $LLVM/llvm-objdump --offloading "$f" > /dev/null 2>&1
b=$(ls "$f".*gfx950 | head -1)
$LLVM/llvm-objdump -d "$b" > "${f%.*}.s"
$LLVM/llvm-readelf --notes "$b" | grep -E "\.name:|\.vgpr_count|\.agpr_count|\.sgpr_count|private_segment_fixed_size|vgpr_spill" | paste - - - - - - | sed 's/  */ /g'
echo "asm: /tmp/isa/${f%.*}.s  dpp $(grep -c dpp "${f%.*}.s")  ds_bpermute $(grep -c ds_bpermute "${f%.*}.s")  s_nop $(grep -c s_nop "${f%.*}.s")  scratch $(grep -c 'scratch_' "${f%.*}.s")"
