#!/bin/bash
# A/B of two builds of the library on the GPU box: put them at pysfm_amd/libpysfm_ba_old.so / _new.so (both travel with the
# snapshot), then `gpurun -- bash scripts/ab_compare.sh`: three alternating runs each of bench.py (min / median of five 20-step
# windows, and the cyclic reduction's elimination in us per trial).  Restore pysfm_amd/libpysfm_ba.so (make) afterwards.
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do for V in old new; do cp pysfm_amd/libpysfm_ba_$V.so pysfm_amd/libpysfm_ba.so; python bench.py --full-line --windows 5 --no-cpu-baseline --no-lm 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$V', round(d['ms_per_step_windows']['min'],4), round(d['ms_per_step_windows']['median'],4), round(d['kernel_ms_per_step']['bcr_eliminate']*1000,1), round(d['kernel_ms_per_step']['bcr_backsolve']*1000,1))"; done; done
