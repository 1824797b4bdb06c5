#!/bin/bash
# round 5: where the time of the channel kernel goes (in-kernel s_memrealtime stamps, tools/chan_probe.py): the one-job kernel and the
# batch kernel with two and four channels to a launch
for v in wipe layer0; do
 for cfg in "PH_CHAN_BENCH_JOBS=1" "PH_CHAN_BENCH_JOBS=2" "PH_CHAN_BENCH_JOBS=4"; do
  echo "== $v $cfg"; env $cfg timeout 120 python tools/chan_probe.py $v 2>&1 | grep -v '"turn"' | grep -v '"turns"' | tail -2
 done
done
