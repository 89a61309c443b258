#!/bin/bash
# Profile bench.py on the GPU box: kernel trace + stats, then PMC passes (own runs) for HBM traffic.
# usage (via gpurun): bash scripts/gpu_profile.sh <tag> [bench.py flags]
set -u
TAG=${1:-r03}
shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-lm --windows 0 --no-other-configs --no-live-pmc $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o $TAG -- $CMD > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o $TAG -- $CMD > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o $TAG -- $CMD > $OUT/pmc_write.log 2>&1
rm -f $OUT/trace/*kernel_trace.csv        # tens of MB; the stats file is what we keep
cd $GRAFT_REPO_ROOT
python bench.py --steps 20 --warmup 20 $* > $OUT/bench.json 2> $OUT/bench.err
mkdir -p gpurun_out/profiles_$TAG
python scripts/summarize_profile.py $OUT $TAG gpurun_out/profiles_$TAG | head -30
tail -1 $OUT/bench.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms/step', d['ms_per_step'], d['ms_per_step_windows'], d['roofline'])"
