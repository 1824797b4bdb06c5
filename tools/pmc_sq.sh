#!/bin/bash
# SQ counters of the fused kernel, one --pmc pass per group (kernel-trace only), summaries -> gpurun_out/pmc_sq/
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_sq; mkdir -p $OUT
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-secondary"
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAVES_EQ_64 SQ_INSTS_VALU_TRANS" "GRBM_GUI_ACTIVE SQ_CYCLES SQ_THREAD_CYCLES_VALU SQ_IFETCH"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/g$i -- $CMD > $OUT/g$i.log 2>&1
  f=$(find $OUT/g$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    if 'fused_v210' in r.get('Kernel_Name',''):
        agg[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in agg.items():
    print(k, sum(v)/len(v), len(v))
PY
  else echo "group $i: no csv"; tail -3 $OUT/g$i.log; fi
done
