#!/bin/bash
# Round 5 A/B of the channel compositor, all in one call on one box: the one-job kernel with its wave steps dealt in turn (PH_CHAN_SCHED=0)
# against the kernel that hands them out at run time, dearest first, alone and with 2 / 4 channels per launch.  -> gpurun_out/r05_chan_ab.jsonl
out=${1:-gpurun_out/r05_chan_ab.jsonl}
: > "$out"
run() { timeout 120 env "$@" python tools/chan_bench.py 400 rgba ${VARIANT:-wipe} >> "$out" 2>gpurun_out/r05_chan_ab.err || echo "{\"failed\": \"$*\"}" >> "$out"; }
for pass in 1 2 3; do
  VARIANT=wipe run PH_CHAN_SCHED=0
  VARIANT=wipe run PH_CHAN_SCHED=1
done
for pass in 1 2; do
  VARIANT=wipe run PH_CHAN_BENCH_JOBS=2
  VARIANT=wipe run PH_CHAN_BENCH_JOBS=4
  VARIANT=nowipe run PH_CHAN_SCHED=0
  VARIANT=nowipe run PH_CHAN_SCHED=1
  VARIANT=nowipe run PH_CHAN_BENCH_JOBS=4
  VARIANT=layer0 run PH_CHAN_SCHED=0
  VARIANT=layer0 run PH_CHAN_SCHED=1
  VARIANT=nowipe run PH_CHAN_BENCH_W=1280 PH_CHAN_BENCH_H=720 PH_CHAN_SCHED=0
  VARIANT=nowipe run PH_CHAN_BENCH_W=1280 PH_CHAN_BENCH_H=720 PH_CHAN_SCHED=1
  VARIANT=nowipe run PH_CHAN_BENCH_W=1280 PH_CHAN_BENCH_H=720 PH_CHAN_BENCH_JOBS=4
  VARIANT=nowipe run PH_CHAN_BENCH_W=1280 PH_CHAN_BENCH_H=720 PH_CHAN_BENCH_JOBS=8
done
cat "$out"
