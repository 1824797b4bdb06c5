#!/bin/bash
# Round 5 A/B of the channel compositor, all in one call on one box: C channels' frames posted as one ph_chan_compose_batch call, run by
# the batch kernel (one launch) against the same call with every job through the one-job kernel (PH_CHAN_BATCH=0).  -> gpurun_out/r05_chan_ab.jsonl
out=${1:-gpurun_out/r05_chan_ab.jsonl}
: > "$out"
run() { timeout 120 env "$@" python tools/chan_bench.py 400 rgba ${VARIANT:-wipe} >> "$out" 2>gpurun_out/r05_chan_ab.err || echo "{\"failed\": \"$*\"}" >> "$out"; }
for pass in 1 2; do
  for c in 1 2 4 8; do
    VARIANT=wipe run PH_CHAN_BENCH_JOBS=$c PH_CHAN_BATCH=0
    VARIANT=wipe run PH_CHAN_BENCH_JOBS=$c
  done
  for c in 1 4; do
    VARIANT=nowipe run PH_CHAN_BENCH_JOBS=$c PH_CHAN_BATCH=0
    VARIANT=nowipe run PH_CHAN_BENCH_JOBS=$c
    VARIANT=layer0 run PH_CHAN_BENCH_JOBS=$c PH_CHAN_BATCH=0
    VARIANT=layer0 run PH_CHAN_BENCH_JOBS=$c
  done
  for c in 1 4 8; do
    VARIANT=nowipe run PH_CHAN_BENCH_W=1280 PH_CHAN_BENCH_H=720 PH_CHAN_BENCH_JOBS=$c PH_CHAN_BATCH=0
    VARIANT=nowipe run PH_CHAN_BENCH_W=1280 PH_CHAN_BENCH_H=720 PH_CHAN_BENCH_JOBS=$c
  done
done
python - "$out" <<'PY'
import json, sys
rows = [json.loads(l) for l in open(sys.argv[1]) if l.startswith("{") and "us_per_frame" in l]
seen = {}
for r in rows:
    seen.setdefault((r["width"], r["variant"], r["jobs_per_launch"], r["batch_kernel"]), []).append(r["us_per_frame"])
for k in sorted(seen):
    print("%4d %-7s C=%d batch_kernel=%-5s us_per_frame %s" % (k + (seen[k],)))
PY
