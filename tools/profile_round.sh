#!/bin/bash
# One gpurun call that regenerates every measured artifact under profiles/ (written to
# gpurun_out/profile/, copied into profiles/ afterwards).  usage: bash tools/profile_round.sh <tag>
TAG=${1:-r06}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/profile; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py"

# 0. the stand-alone microbenchmarks (binaries are not tracked: built here when the snapshot did not bring them)
for t in microbench opbench3 opbench4; do
  [ -x $ROOT/tools/$t ] || hipcc --offload-arch=gfx950 -O2 -o $ROOT/tools/$t $ROOT/tools/$t.hip > /dev/null 2>&1
done

# 1. the bench line exactly as the driver runs it
$BENCH > $OUT/${TAG}_bench.json 2> $OUT/bench.err
tail -1 $OUT/${TAG}_bench.json

# 2. rocprofv3 kernel stats of the same command (shorter K)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $BENCH --cpu-seconds 0 > $OUT/stats.log 2>&1
f=$(find $OUT/stats -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && (head -1 "$f"; grep "ph::" "$f") > $OUT/${TAG}_bench_kernel_stats.csv

# 3. HBM traffic: separate --pmc passes, calibrated on a 1 GiB copy (tools/microbench copy16)
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_bench_$c -- $BENCH --steps 40 --warmup 5 --cpu-seconds 0 --no-secondary > $OUT/pmc_bench_$c.log 2>&1
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_micro_$c -- $ROOT/tools/microbench > $OUT/pmc_micro_$c.log 2>&1
done
python3 - "$OUT" "$TAG" <<'PY'
import csv, glob, json, sys
out, tag = sys.argv[1], sys.argv[2]
def mean(dirname, counter, kernel_sub):
    vals = []
    for f in glob.glob("%s/%s/**/*counter_collection.csv" % (out, dirname), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and kernel_sub in r["Kernel_Name"]:
                vals.append(float(r["Counter_Value"]))
    return (sum(vals) / len(vals), len(vals)) if vals else (None, 0)
res = {}
cal = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    res[c], n = mean("pmc_bench_" + c, c, "fused_v210_combine")
    res["dispatches"] = n
    cal[c], _ = mean("pmc_micro_" + c, c, "copy16")
gib_kb = 1024.0 * 1024.0
fc = gib_kb / cal["FETCH_SIZE"] if cal["FETCH_SIZE"] else None
wc = gib_kb / cal["WRITE_SIZE"] if cal["WRITE_SIZE"] else None
traffic = None
if fc and wc and res["FETCH_SIZE"] is not None:
    traffic = int(round((res["FETCH_SIZE"] * fc + res["WRITE_SIZE"] * wc) * 1024))
doc = {"fused_v210_combine_4_2160p_bytes_per_launch": traffic,
       "detail": {"FETCH_SIZE_KB_mean": res["FETCH_SIZE"], "WRITE_SIZE_KB_mean": res["WRITE_SIZE"], "dispatches": res["dispatches"],
                  "calibration": {"kernel": "tools/microbench copy16 (1 GiB read + 1 GiB written per launch)",
                                  "FETCH_SIZE_KB": cal["FETCH_SIZE"], "WRITE_SIZE_KB": cal["WRITE_SIZE"],
                                  "fetch_correction": fc, "write_correction": wc},
                  "algorithmic_bytes": 110592000,
                  "ratio_traffic_over_algorithmic": traffic / 110592000.0 if traffic else None,
                  "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE (separate passes) --output-format csv -- python bench.py --steps 40 --warmup 5 --cpu-seconds 0 --no-secondary"}}
json.dump(doc, open("%s/pmc_traffic.json" % out, "w"), indent=1)
print("traffic", traffic)
PY

# 4. SQ counters of the fused kernel
bash $ROOT/tools/pmc_sq.sh > $OUT/${TAG}_pmc_sq.txt 2>&1
rm -rf $ROOT/gpurun_out/pmc_sq/*/  # keep the summary only

# 4b. configs 2 and 3: their kernels timed alone, in-kernel phase probe, and counters incl. FETCH_SIZE / WRITE_SIZE (separate passes)
(for v in wipe nowipe layer0; do python $ROOT/tools/chan_bench.py 300 rgba $v; done; python $ROOT/tools/chan_bench.py 300 v210 wipe; for f in yuv422p10 yuv420p nv12; do for v in wipe nowipe layer0 overlay; do python $ROOT/tools/chan_bench.py 300 rgba $v $f; done; done; python $ROOT/tools/chan_bench.py 300 rgba overlay v210; for o in yuv422p8 rgba8; do PH_CHAN_BENCH_OUT=$o python $ROOT/tools/chan_bench.py 300 rgba layer0 v210; PH_CHAN_BENCH_OUT=$o python $ROOT/tools/chan_bench.py 300 rgba wipe v210; PH_CHAN_BENCH_OUT=$o python $ROOT/tools/chan_bench.py 300 rgba layer0 yuv420p; done) 2>/dev/null | grep '^{' > $OUT/${TAG}_chan_bench.jsonl
# round 5: the in-kernel phase probe of the one-job kernel and of the batch kernel (2 / 4 channels per launch), and the A/B of one
# ph_chan_compose_batch call run by the batch kernel against the same call with every job through the one-job kernel
(cd $ROOT && bash tools/r05_probe_run.sh) > $OUT/${TAG}_chan_batch_probe.txt 2>&1
(cd $ROOT && bash tools/r05_chan_ab.sh $OUT/${TAG}_chan_batch_ab.jsonl > $OUT/${TAG}_chan_batch_ab_summary.txt 2>&1)
python $ROOT/tools/up_bench.py 100 2>/dev/null | grep '^{' > $OUT/${TAG}_up_bench.jsonl
bash $ROOT/tools/pmc_kernel.sh chan_compose python $ROOT/tools/chan_bench.py 40 rgba wipe 2>&1 | grep -v '^{\|simple_timer\|^W2\|^find' > $OUT/${TAG}_pmc_chan.txt
PH_CHAN_BENCH_JOBS=4 bash $ROOT/tools/pmc_kernel.sh chan_compose_batch python $ROOT/tools/chan_bench.py 40 rgba wipe 2>&1 | grep -v '^{\|simple_timer\|^W2\|^find' > $OUT/${TAG}_pmc_chan_batch.txt
bash $ROOT/tools/pmc_kernel.sh compose_up python $ROOT/tools/up_bench.py 12 up_single 2>&1 | grep -v '^{\|simple_timer\|^W2\|^find' > $OUT/${TAG}_pmc_up.txt
bash $ROOT/tools/pmc_kernel.sh v210_yadif_pair python $ROOT/tools/up_bench.py 12 deint 2>&1 | grep -v '^{\|simple_timer\|^W2\|^find' > $OUT/${TAG}_pmc_deint.txt
rm -rf $ROOT/gpurun_out/pmc_kernel
python3 - "$OUT" "$TAG" <<'PY'
import json, sys
out, tag = sys.argv[1], sys.argv[2]
def counters(name):
    d = {}
    try:
        for line in open("%s/%s_%s.txt" % (out, tag, name)):
            f = line.split()
            if len(f) == 3:
                try:
                    d[f[0]] = float(f[1])
                except ValueError:
                    pass
    except OSError:
        pass
    return d
doc = {"note": "per launch; FETCH_SIZE / WRITE_SIZE in KiB as rocprofv3 reports them (separate --pmc passes).  MI355X_MICROARCH.md: FETCH_SIZE counts half the "
               "bytes of a wide coalesced streaming read on gfx950 - `fetch_x2` doubles it (an upper figure: these kernels' narrow loads are uncalibrated)"}
for name, algo, unit in (("pmc_chan", 38707200, "config 2: ph_chan_compose_v210, one launch per frame"),
                         ("pmc_up", 4 * 1920 * 1080 * 12 + 22118400, "config 3 compositor: ph_compose_up_write_v210 on packed-RGB fields, one launch per field (its own inputs + v210 out)"),
                         ("pmc_deint", 4 * 3 * 5529600 + 8 * 1920 * 1080 * 12, "config 3 de-interlacing reader: ph_v210_yadif_pair_fmt, one launch per frame = two fields (v210 windows in + packed-RGB fields out)")):
    c = counters(name)
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        raw = (c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024
        x2 = (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024
        doc[name] = {"what": unit, "FETCH_SIZE_KiB": c["FETCH_SIZE"], "WRITE_SIZE_KiB": c["WRITE_SIZE"], "bytes_counter_units": int(raw), "bytes_fetch_x2": int(x2),
                     "kernel_io_bytes": algo, "ratio_counter_units": round(raw / algo, 3), "ratio_fetch_x2": round(x2 / algo, 3)}
json.dump(doc, open("%s/%s_config_traffic.json" % (out, tag), "w"), indent=1)
print(json.dumps(doc))
PY

# 5. per-kernel, per-config, per-instruction and staging measurements
python $ROOT/tools/kernel_bench.py 2>/dev/null | grep '^{' > $OUT/${TAG}_kernel_bench.jsonl
PH_BENCH_CACHED_IMAGES=1 python $ROOT/tools/kernel_bench.py 2>/dev/null | grep '^{' | grep -v '"pack_\|fused_v210\|compose_write\|v210_write' > $OUT/${TAG}_kernel_bench_cached_images.jsonl
python $ROOT/tools/config_bench.py 2>/dev/null | grep '^{' > $OUT/${TAG}_config_bench.jsonl
$ROOT/tools/opbench3 > $OUT/${TAG}_opbench3.jsonl 2>/dev/null
$ROOT/tools/opbench4 > $OUT/${TAG}_opbench4.jsonl 2>/dev/null
python $ROOT/tools/route_bench.py --loopback 2>/dev/null | grep '^{' > $OUT/${TAG}_route_loopback.jsonl
# clips smaller than their channel (read + 2 x 2-block compositor inside ph_chan_compose*), alone and 2 / 4 channels per call; config 3 in the reference's own formats
(for args in "500 1 1280 720 1920 1080" "500 1 720 576 1920 1080" "500 2 1280 720 1920 1080" "500 4 1280 720 1920 1080" "500 1 1920 1080 3840 2160" "500 2 1920 1080 3840 2160" "500 4 1920 1080 3840 2160" "500 1 720 576 1280 720" "500 1 1280 720 3840 2160"; do python $ROOT/tools/enlarge_bench.py $args; done;
 for C in 2 4; do for args in "500 1 1280 720 1920 1080" "500 2 1280 720 1920 1080" "500 1 720 576 1280 720"; do PH_ENLARGE_CHANNELS=$C python $ROOT/tools/enlarge_bench.py $args; done; done;
 for f in yuv420p yuv422p10 nv12; do PH_ENLARGE_FORMAT=$f python $ROOT/tools/enlarge_bench.py 500 1 1280 720 1920 1080; done; PH_ENLARGE_FORMAT=yuv420p python $ROOT/tools/enlarge_bench.py 500 1 720 576 1920 1080;
 PH_ENLARGE_FORMAT=yuv420p python $ROOT/tools/enlarge_bench.py 500 2 1280 720 1920 1080; PH_ENLARGE_FORMAT=yuv422p10 python $ROOT/tools/enlarge_bench.py 500 1 1920 1080 3840 2160;
 PH_ENLARGE_CHANNELS=4 PH_ENLARGE_FORMAT=yuv420p python $ROOT/tools/enlarge_bench.py 500 1 1280 720 1920 1080;
 for C in 1 4; do PH_ENLARGE_CHANNELS=$C PH_ENLARGE_FORMAT=yuv420p python $ROOT/tools/enlarge_bench.py 500 1 1920 1080 1920 1080; done;
 for C in 1 2 4; do PH_ENLARGE_CHANNELS=$C python $ROOT/tools/enlarge_bench.py 500 1 1920 1080 1920 1080; done) 2>/dev/null | grep '^{' > $OUT/${TAG}_enlarge_bench.jsonl
python $ROOT/tools/config3b_bench.py 300 2>/dev/null | grep '^{' > $OUT/${TAG}_config3b.json
python $ROOT/tools/staging_bench.py 60 2>/dev/null | grep '^{' > $OUT/${TAG}_staging_bench.jsonl
$ROOT/tools/microbench 2>/dev/null | grep '^{' > $OUT/${TAG}_microbench.jsonl
# through node: the four modes at 2160p and 1080p (3000 frames: a 600-frame run is a third warm-up), released buffers parked or not,
# and 1 / 4 channels of config 2's shape per tick (their frames in one launch)
(node $ROOT/node/test/bench_node.js 3000; node $ROOT/node/test/bench_node.js 5000 1920 1080 4;
 for r in 1 0; do for size in "3000 3840 2160" "5000 1920 1080"; do PHANERON_RECYCLE=$r PH_NODE_BENCH_MODES=deferred node $ROOT/node/test/bench_node.js $size | sed "s/^{/{\"recycle_buffers\": $r, /"; done;
   for c in 1 4; do PHANERON_RECYCLE=$r PH_NODE_BENCH_CHANNELS=$c PH_NODE_BENCH_MODES=channels node $ROOT/node/test/bench_node.js 3000 1920 1080 | sed "s/^{/{\"recycle_buffers\": $r, /"; done; done;
 PHANERON_EARLY_LAUNCH=1 PH_NODE_BENCH_CHANNELS=4 PH_NODE_BENCH_MODES=channels node $ROOT/node/test/bench_node.js 3000 1920 1080;
 for c in 1 4; do for e in 0 1; do PH_NODE_BENCH_PLAIN=1 PHANERON_EARLY_LAUNCH=$e PH_NODE_BENCH_CHANNELS=$c PH_NODE_BENCH_MODES=channels node $ROOT/node/test/bench_node.js 3000 1920 1080; done; done;
 for f in 1920x1080 1280x720; do for c in 1 4; do PH_NODE_BENCH_FILE=$f PH_NODE_BENCH_CHANNELS=$c PH_NODE_BENCH_MODES=channels node $ROOT/node/test/bench_node.js 3000 1920 1080; done; done) 2>/dev/null | grep '^{' > $OUT/${TAG}_node_bench.jsonl
# round 6: the reference's own channel kind through node (4 x 1080i sources per 1080p channel; fields packed or not; the host's share alone)
(cd $ROOT && bash tools/r06_node_interlaced.sh) 2>/dev/null | grep '^{' > $OUT/${TAG}_node_interlaced.jsonl
# round 6: the matrix pipe priced (tools/mfma_probe.hip) and the one-launch clip kernel's phases (timing builds, if the snapshot brought them)
[ -x $ROOT/tools/mfma_probe ] || hipcc --offload-arch=gfx950 -O3 -o $ROOT/tools/mfma_probe $ROOT/tools/mfma_probe.hip > /dev/null 2>&1
$ROOT/tools/mfma_probe 2>/dev/null | grep -v '"probe": "step"' > $OUT/${TAG}_mfma_probe.jsonl
# (the clip kernel's phase prices, profiles/r06_clip_ablate.txt, come from timing builds: tools/r06_clip_ablate.sh after building the clip1/2/4/6 variants - not part of this run)
node $ROOT/node/test/soak_run.js 100000 2>/dev/null | grep '^{' > $OUT/${TAG}_node_soak.json
(node $ROOT/node/test/napi_costs.js 1920 1080; node $ROOT/node/test/napi_costs.js 3840 2160; node $ROOT/node/test/defer_host_bench.js 20000; node $ROOT/node/test/defer_host_bench.js 20000 --plain; node $ROOT/node/test/defer_host_bench.js 8000 1920 1080 4 --interlaced) 2>/dev/null | grep '^{' > $OUT/${TAG}_node_host_costs.jsonl
# the recording context (node/defer.js) against the launch-as-posted one: scenarios, frames compared byte for byte, launch counters
(node $ROOT/node/test/defer_run.js; node $ROOT/node/test/defer_run.js 1920 64) 2>/dev/null | grep '^{' > $OUT/${TAG}_defer_run.jsonl
node $ROOT/node/test/defer_fuzz.js 100 400 120 2>/dev/null | grep '^{' > $OUT/${TAG}_defer_fuzz.jsonl
(for v in "--no-secondary" "--no-secondary --content picture" "--width 1920 --height 1080" "--width 1920 --height 1080 --channels 2" "--width 1920 --height 1080 --channels 4" "--width 1920 --height 1080 --frames-per-launch 2" "--width 1920 --height 1080 --frames-per-launch 4" "--channels 2 --ring 4"; do $BENCH $v --steps 1500 --cpu-seconds 0; done) 2>/dev/null | grep '^{' > $OUT/${TAG}_bench_variants.jsonl
rm -rf $OUT/stats $OUT/pmc_bench_* $OUT/pmc_micro_*
ls -la $OUT
