#!/bin/bash
# a longer campaign with fresh seeds at the round's final state (output: profiles/r06_fuzz_long.txt)
for seed in 700001 700002 700003 700004 700005 700006 700007 700008; do
  echo "== PH_FUZZ_SEED=$seed PH_FUZZ_CASES=200"
  PH_FUZZ_SEED=$seed PH_FUZZ_CASES=200 timeout 1500 python -m pytest tests/test_chan_gpu.py tests/test_boundary_gpu.py -q -m gpu -x \
    -k "random_channel_programs or chan_batch_random_calls or random" 2>&1 < /dev/null | grep -E "passed|failed|Error" | tail -3
done
echo "== node/test/channels_fuzz.js first=20000 seeds=1500 ticks=12"
PH_FUZZ_ROUTES=1 timeout 2400 node node/test/channels_fuzz.js 20000 1500 12 2>&1 < /dev/null | tail -1 | cut -c1-1800
echo "== PHANERON_FIELD_BATCH=1 node/test/channels_fuzz.js first=40000 seeds=300 ticks=12"
PHANERON_FIELD_BATCH=1 timeout 1200 node node/test/channels_fuzz.js 40000 300 12 2>&1 < /dev/null | tail -1 | cut -c1-900
echo "== node/test/defer_fuzz.js first=50000 streams=1500 steps=80"
timeout 1500 node node/test/defer_fuzz.js 50000 1500 80 2>&1 < /dev/null | tail -1 | cut -c1-600
echo "== node/test/channels_fuzz.js first=9000 seeds=12 ticks=6 at 1920 x 1080"; PH_FUZZ_ROUTES=1 timeout 1500 node node/test/channels_fuzz.js 9000 12 6 1920 1080 2>&1 < /dev/null | tail -1 | cut -c1-1500
echo "== PHANERON_FUZZ_SIZE=1920x1080 node/test/defer_fuzz.js first=80000 streams=40 steps=60"; PHANERON_FUZZ_SIZE=1920x1080 timeout 1500 node node/test/defer_fuzz.js 80000 40 60 2>&1 < /dev/null | tail -1 | cut -c1-600
