for i in 1 2; do for h in 1080 1072 1024; do for v in layer0 wipe; do PH_CHAN_BENCH_H=$h python tools/chan_bench.py 300 rgba $v 2>/dev/null | grep "^{" | cut -c30-140; done; done; done
