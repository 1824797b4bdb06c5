for pass in 1 2 3; do
 for v in wipe nowipe; do
  echo "$v head:  $(PHANERON_HIP_LIB=tools/_variants/libphaneron_hip_head.so PH_CHAN_BENCH_JOBS=4 python tools/chan_bench.py 400 rgba $v | python -c 'import json,sys; print(json.loads(sys.stdin.read())["us_per_frame"])')"
  echo "$v box:   $(PH_CHAN_BENCH_JOBS=4 python tools/chan_bench.py 400 rgba $v | python -c 'import json,sys; print(json.loads(sys.stdin.read())["us_per_frame"])')"
  echo "$v nobox: $(PH_CHAN_NO_BOX=1 PH_CHAN_BENCH_JOBS=4 python tools/chan_bench.py 400 rgba $v | python -c 'import json,sys; print(json.loads(sys.stdin.read())["us_per_frame"])')"
 done
 echo "720 head: $(PHANERON_HIP_LIB=tools/_variants/libphaneron_hip_head.so PH_CHAN_BENCH_W=1280 PH_CHAN_BENCH_H=720 PH_CHAN_BENCH_JOBS=4 python tools/chan_bench.py 400 rgba nowipe | python -c 'import json,sys; print(json.loads(sys.stdin.read())["us_per_frame"])')"
 echo "720 box:  $(PH_CHAN_BENCH_W=1280 PH_CHAN_BENCH_H=720 PH_CHAN_BENCH_JOBS=4 python tools/chan_bench.py 400 rgba nowipe | python -c 'import json,sys; print(json.loads(sys.stdin.read())["us_per_frame"])')"
done
