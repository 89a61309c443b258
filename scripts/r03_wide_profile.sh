#!/bin/bash
# kernel trace of a wide-band scene (bench.py flags as arguments, e.g. --track-len 32)
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_wide
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o wide -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-lm --windows 0 --no-other-configs --no-live-pmc --no-kernel-table "$@" > $OUT/trace.log 2>&1
python - <<PY
import csv,glob
f=glob.glob('$OUT/trace/*kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:16]:
    print('%-70s %6s %10.1f us %6s%%' % (r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3, r['Percentage']))
PY
rm -f $(find $OUT/trace -name '*kernel_trace.csv')
