TAG=${1:-r06}; ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out/profile_node; rm -rf $OUT; mkdir -p $OUT; exec < /dev/null
# the node sections of tools/profile_round.sh alone (after a change to node/ only)
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
node $ROOT/node/test/soak_run.js 100000 2>/dev/null | grep '^{' > $OUT/${TAG}_node_soak.json
(node $ROOT/node/test/napi_costs.js 1920 1080; node $ROOT/node/test/napi_costs.js 3840 2160; node $ROOT/node/test/defer_host_bench.js 20000; node $ROOT/node/test/defer_host_bench.js 20000 --plain; node $ROOT/node/test/defer_host_bench.js 8000 1920 1080 4 --interlaced) 2>/dev/null | grep '^{' > $OUT/${TAG}_node_host_costs.jsonl
# the recording context (node/defer.js) against the launch-as-posted one: scenarios, frames compared byte for byte, launch counters
(node $ROOT/node/test/defer_run.js; node $ROOT/node/test/defer_run.js 1920 64) 2>/dev/null | grep '^{' > $OUT/${TAG}_defer_run.jsonl
node $ROOT/node/test/defer_fuzz.js 100 400 120 2>/dev/null | grep '^{' > $OUT/${TAG}_defer_fuzz.jsonl
ls -la $OUT
