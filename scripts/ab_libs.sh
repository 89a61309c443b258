#!/bin/bash
# A/B of several builds of the library on the GPU box: pysfm_amd/libvar_<name>.so (they travel with the snapshot; git-ignored);
# `gpurun -- bash scripts/ab_libs.sh name1 name2 ...`: three alternating runs each of bench.py; prints min / median of five
# 20-step windows and the per-kernel microseconds.  Restore pysfm_amd/libpysfm_ba.so (make) afterwards.
cd $GRAFT_REPO_ROOT
cp pysfm_amd/libpysfm_ba.so /tmp/lib_keep.so
for V in "$@"; do [ -f pysfm_amd/libvar_$V.so ] || { echo "ab_libs: pysfm_amd/libvar_$V.so is missing (did its build fail?)"; exit 1; }; done
for rep in 1 2 3; do for V in "$@"; do cp pysfm_amd/libvar_$V.so pysfm_amd/libpysfm_ba.so; python bench.py --full-line --windows 5 --no-cpu-baseline --no-lm --no-other-configs --no-live-pmc $BENCH_ARGS 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%-12s' % '$V', round(d['ms_per_step_windows']['min'],4), round(d['ms_per_step_windows']['median'],4), {k: round(v*1000,1) for k, v in d['kernel_ms_per_step'].items()})"; done; done
cp /tmp/lib_keep.so pysfm_amd/libpysfm_ba.so
