#!/bin/bash
# SQ counters of the trial's kernels (own rocprofv3 passes, counters only - no trace domains): what `limited_by` in the bench
# line rests on.  Pass A: cycles (busy, wave, waiting, issuing, matrix-core busy).  Pass B: instruction mix (fp64 MFMA ops,
# fp64 FMAs, VALU, LDS), LDS bank conflicts, waves launched.  Pass C: occupancy (SQ_LEVEL_WAVES accumulates, own pass).
# usage (via gpurun): bash scripts/gpu_sq_counters.sh <tag> [bench.py flags]
TAG=${1:-r05}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/sq_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-lm --windows 0 --no-other-configs --no-live-pmc $*"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS --output-format csv -d $OUT/pmcA -o $TAG -- $CMD > $OUT/pmcA.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES --output-format csv -d $OUT/pmcB -o $TAG -- $CMD > $OUT/pmcB.log 2>&1
rocprofv3 --pmc SQ_LEVEL_WAVES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmcC -o $TAG -- $CMD > $OUT/pmcC.log 2>&1
cd $GRAFT_REPO_ROOT
python scripts/summarize_sq_counters.py $OUT $TAG
