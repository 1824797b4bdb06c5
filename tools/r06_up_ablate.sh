#!/bin/bash
# The block compositor's timing experiments of round 6 (profiles/r06_up_ablate.txt).  The builds are NOT product code: apply
# profiles/r06_up_ablate_experiment.patch to phaneron_amd/csrc/ph_kernels_up.hip (its hunks are the successive experiments; each is
# guarded by a macro), build a variant per macro -
#   python -c "from phaneron_amd import build; build.build(variant='upab6', extra_flags=['-DPH_UP_ABLATE=6'])"
# (PH_UP_ABLATE=1..7, PH_UP_PLAIN_STORE, PH_UP_NO_STORE, PH_UP_PRIO=1|2, PH_UP_PIPE; the late-start experiment was three lines after the
# table's barrier - `if ((wave >> 2) & 1) __builtin_amdgcn_s_sleep(N);` - and is not in the patch) - and run:
#   bash tools/r06_up_ablate.sh "upab1 upab6 pipe" [layers ...]
# prints tools/up_bench.py's compositor legs for the product library and each named variant, by layer count.
VARIANTS=${1:-}
shift
LAYERS=${@:-4}
for n in $LAYERS; do
  for v in "" $VARIANTS; do
    lib=""; [ -n "$v" ] && lib=tools/_variants/libphaneron_hip_$v.so
    [ -n "$v" ] && [ ! -f "$lib" ] && { echo "$v layers=$n: no such build ($lib)"; continue; }
    echo "${v:-product} layers=$n: $(PH_UP_LAYERS=$n PHANERON_HIP_LIB=$lib python tools/up_bench.py 300 up 2>/dev/null | tail -1)"
  done
done
