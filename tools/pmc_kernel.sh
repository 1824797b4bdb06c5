#!/bin/bash
# Counters of ONE kernel of a command, one --pmc group per pass (kernel-trace only).
# usage: tools/pmc_kernel.sh <kernel substring> <command...>; output: mean per dispatch
cd /tmp && export TMPDIR=/tmp
K=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_kernel; rm -rf $OUT; mkdir -p $OUT
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS" "TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" "FETCH_SIZE" "WRITE_SIZE" "TA_BUFFER_LOAD_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "TCP_TA_TCP_STATE_READ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/g$i -- "$@" > $OUT/g$i.log 2>&1
  f=$(find $OUT/g$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python3 - "$f" "$K" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r.get('Kernel_Name', ''):
        agg[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in agg.items():
    print(k, sum(v) / len(v), len(v))
PY
  else echo "group $i ($grp): no csv"; tail -2 $OUT/g$i.log; fi
  rm -rf $OUT/g$i
done
