#!/bin/bash
# Profile bench.py on the GPU box: kernel trace + stats, then PMC passes (own runs) for HBM traffic.
# usage (via gpurun): bash scripts/gpu_profile.sh <tag>
set -u
TAG=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-lm"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o $TAG -- $CMD > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o $TAG -- $CMD > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o $TAG -- $CMD > $OUT/pmc_write.log 2>&1
rm -f $OUT/trace/*kernel_trace.csv        # tens of MB; the stats file is what we keep
ls -R $OUT | head -30
