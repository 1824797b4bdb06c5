#!/bin/bash
# rocprofv3 kernel trace of the node bench's 1080i channels (1 and 4 channels; PHANERON_FIELD_BATCH=1 as the third run): kernel stats, the order of
# the launches of one tick, the device's busy share and the gaps between ticks.  Output: profiles/r06_node_kernel_trace.txt
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/r06/nodetrace; rm -rf $OUT; mkdir -p $OUT
for spec in "1 0" "4 0" "4 1"; do set -- $spec
  echo "== channels=$1 PHANERON_FIELD_BATCH=$2"
  PHANERON_FIELD_BATCH=$2 PH_NODE_BENCH_INTERLACED=1 PH_NODE_BENCH_CHANNELS=$1 PH_NODE_BENCH_MODES=channels timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c$1b$2 -o np -- node node/test/bench_node.js 1200 1920 1080 2>&1 < /dev/null | grep '^{' | tail -1 | cut -c1-330
  python3 - $OUT/c$1b$2 <<'PY'
import csv, glob, sys, statistics
d = sys.argv[1]
st = glob.glob(d + '/**/*kernel_stats.csv', recursive=True)[0]
for r in list(csv.DictReader(open(st)))[:2]:
    print('  %-60s calls %s avg %.1f us' % (r['Name'][:60], r['Calls'], float(r['AverageNs']) / 1e3))
tr = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(tr)) if 'yadif' in r['Kernel_Name'] or 'compose_up' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
rows = rows[len(rows) // 2:]  # the second half: warm
names = ['yadif' if 'yadif' in r['Kernel_Name'] else 'compose' for r in rows]
print('  order of launches:', ' '.join(names[:8]))
busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rows)
span = int(rows[-1]['End_Timestamp']) - int(rows[0]['Start_Timestamp'])
gaps = sorted((int(b['Start_Timestamp']) - int(a['End_Timestamp'])) / 1e3 for a, b in zip(rows, rows[1:]))
print('  device busy %.1f %% of %.1f ms; gaps between launches: median %.1f us, p90 %.1f, max %.0f' % (100.0 * busy / span, span / 1e6, statistics.median(gaps), gaps[int(0.9 * len(gaps))], gaps[-1]))
PY
done
