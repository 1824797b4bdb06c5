# channels with planar sources in the batch kernel (its PLANAR instantiation) against the same call with every job through the one-job kernel (PH_CHAN_BATCH=0)
for spec in "inset yuv422p10" "inset yuv420p" "overlay v210" "nowipe yuv422p10" "layer0 yuv420p"; do set -- $spec
 for b in 1 0; do for c in 1 4; do
  [ $c = 1 ] && [ $b = 0 ] && continue
  echo "$1 $2 jobs=$c batch=$b: $(PH_CHAN_BATCH=$b PH_CHAN_BENCH_JOBS=$c python tools/chan_bench.py 300 rgba $1 $2 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["us_per_frame"])')"
 done; done
done
