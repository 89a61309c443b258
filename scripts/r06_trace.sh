#!/bin/bash
# round 6: kernel trace + stats of bench.py with extra flags.  usage (via gpurun): bash scripts/r06_trace.sh <tag> [bench.py flags]
TAG=${1:-t}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/trace_$TAG
mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o $TAG -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-lm --no-other-configs --no-live-pmc --windows 0 --no-kernel-table "$@" > $OUT/log.txt 2>&1
tail -1 $OUT/log.txt | cut -c1-400
python3 $GRAFT_REPO_ROOT/scripts/trim_rocprof_stats.py $OUT/${TAG}_kernel_stats.csv $OUT/${TAG}_kernel_stats_trimmed.csv 24
cut -d, -f1-6 $OUT/${TAG}_kernel_stats_trimmed.csv | sed 's/(.*)//' | cut -c1-150
rm -f $OUT/*kernel_trace.csv $OUT/*agent_info.csv
