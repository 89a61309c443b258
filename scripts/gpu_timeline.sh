#!/bin/bash
# kernel timeline of ONE LM trial (start offset, duration, gap to the previous kernel, in us).
# usage (via gpurun): bash scripts/gpu_timeline.sh <tag> [extra bench args]
TAG=${1:-t}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/timeline_$TAG
mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -o $TAG -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-lm --no-kernel-table "$@" > $OUT/log.txt 2>&1
python3 - <<PY
import csv
rows = list(csv.DictReader(open('$OUT/${TAG}_kernel_trace.csv')))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# the timed region = the last 20 trials; one trial starts at k_linearize
idx = [i for i, r in enumerate(rows) if 'k_linearize' in r['Kernel_Name']]
a, b = idx[-2], idx[-1]
t0 = int(rows[a]['Start_Timestamp']); prev = None; busy = 0
with open('$OUT/timeline.txt', 'w') as f:
    for r in rows[a:b]:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        gap = 0 if prev is None else s - prev
        line = '%-34s grid %8s  start %8.2f  dur %7.2f  gap %6.2f' % (r['Kernel_Name'].split('(')[0][-34:], r.get('Grid_Size', r.get('Grid_Size_X', '?')), (s - t0) / 1e3, (e - s) / 1e3, gap / 1e3)
        print(line); f.write(line + '\n'); prev = e; busy += e - s
    tot = int(rows[b]['Start_Timestamp']) - t0
    line = 'trial %.2f us, kernels busy %.2f us, %d launches' % (tot / 1e3, busy / 1e3, b - a)
    print(line); f.write(line + '\n')
PY
rm -f $OUT/*kernel_trace.csv
