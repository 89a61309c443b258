#!/bin/bash
# kernel trace of the long-track scene (bench.py --long-tracks 50,80): which launches the reduction's time goes to
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_lt
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o lt -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-lm --windows 0 --no-other-configs --no-live-pmc --no-kernel-table --long-tracks ${1:-50,80} > $OUT/trace.log 2>&1
find $OUT/trace -name '*kernel_stats.csv' | head -1 | xargs head -25
rm -f $(find $OUT/trace -name '*kernel_trace.csv')
