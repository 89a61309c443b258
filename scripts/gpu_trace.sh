#!/bin/bash
# kernel-trace only: per-kernel durations of bench.py (no PMC).  usage: bash scripts/gpu_trace.sh <tag>
TAG=${1:-t}
OUT=$GRAFT_REPO_ROOT/gpurun_out/trace_$TAG
mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o $TAG -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-lm > $OUT/log.txt 2>&1
python3 - <<PY
import csv
rows=list(csv.DictReader(open('$OUT/${TAG}_kernel_trace.csv')))
# print the per-launch durations of one trial's solver kernels (last trial)
sel=[r for r in rows if 'bcr' in r['Kernel_Name'] or 'band_solve' in r['Kernel_Name']]
last=sel[-40:]
for r in last: print(r['Kernel_Name'].split('(')[0][-28:], r['Grid_Size'], int(r['End_Timestamp'])-int(r['Start_Timestamp']))
PY
rm -f $OUT/*kernel_trace.csv
