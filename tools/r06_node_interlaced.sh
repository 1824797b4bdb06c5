# node: C channels of 4 x 1080i sources on 1080p channels through the recording context, de-interlaced fields packed (default) or as RGBA images;
# PHANERON_FIELD_BATCH=1: the channels' windows and compositor frames in shared launches (round 6's first form) instead of channel by channel
for c in 1 4; do for p in 1 0; do
 PHANERON_PACK_FIELDS=$p PH_NODE_BENCH_INTERLACED=1 PH_NODE_BENCH_CHANNELS=$c PH_NODE_BENCH_MODES=channels timeout 300 node node/test/bench_node.js 1500 1920 1080 2>&1 < /dev/null | tail -1
done; done
PHANERON_FIELD_BATCH=1 PH_NODE_BENCH_INTERLACED=1 PH_NODE_BENCH_CHANNELS=4 PH_NODE_BENCH_MODES=channels timeout 300 node node/test/bench_node.js 1500 1920 1080 2>&1 < /dev/null | sed 's/^{/{"field_batch":true,/' | tail -1
# the host's share alone (dry run: nothing enqueued)
for c in 1 4; do PH_NODE_BENCH_DRY=1 PH_NODE_BENCH_INTERLACED=1 PH_NODE_BENCH_CHANNELS=$c PH_NODE_BENCH_MODES=channels timeout 300 node node/test/bench_node.js 1500 1920 1080 2>&1 < /dev/null | tail -1; done
PH_NODE_BENCH_DRY=1 PH_NODE_BENCH_CHANNELS=4 PH_NODE_BENCH_MODES=channels timeout 300 node node/test/bench_node.js 1500 1920 1080 2>&1 < /dev/null | tail -1
