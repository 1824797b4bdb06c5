#!/bin/bash
# The seeded campaigns at >= 10 x their default case counts, once per round; the seeds that ran are in the output, which is
# committed as profiles/r06_fuzz_campaign.txt.  usage (GPU box): bash tools/r06_fuzz_campaign.sh
for seed in 606001 606002 606003; do
  echo "== PH_FUZZ_SEED=$seed PH_FUZZ_CASES=120: test_random_channel_programs, test_chan_batch_random_calls, test_random_channel_programs_with_planar_clips, boundary fuzz"
  PH_FUZZ_SEED=$seed PH_FUZZ_CASES=120 timeout 1500 python -m pytest tests/test_chan_gpu.py tests/test_boundary_gpu.py -q -m gpu -x \
    -k "random_channel_programs or chan_batch_random_calls or random" 2>&1 | tail -3
done
echo "== node/test/channels_fuzz.js first=1000 seeds=200 ticks=12 (with 1080i channels; routes = the recording context's launches by kernel)"
PH_FUZZ_ROUTES=1 timeout 1500 node node/test/channels_fuzz.js 1000 200 12 2>&1 < /dev/null | tail -1 | cut -c1-1800
echo "== the same seeds 1000..1059 with PHANERON_FIELD_BATCH=1 (the channels' Yadif windows and compositor frames in shared launches)"
PHANERON_FIELD_BATCH=1 PH_FUZZ_ROUTES=1 timeout 1500 node node/test/channels_fuzz.js 1000 60 12 2>&1 < /dev/null | tail -1 | cut -c1-1800
echo "== node/test/defer_fuzz.js first=2000 streams=200 steps=60"
timeout 1500 node node/test/defer_fuzz.js 2000 200 60 2>&1 | tail -1 | cut -c1-600
