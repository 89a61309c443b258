#!/bin/bash
# hb = 12, 13 (track lengths 13, 14): the cyclic reduction with a node over three workgroups (default) against the wide solver's
# four kernels per level (option solver = bcr1 selects it there).  `gpurun -- bash scripts/ab_wide_nodes.sh`
cd $GRAFT_REPO_ROOT
for L in ${LENGTHS:-12 13 14 15}; do for OPT in "" "--option solver=bcr1"; do
python bench.py --full-line --track-len $L --windows 5 --no-cpu-baseline --no-lm --no-other-configs --no-live-pmc $OPT 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('L=$L %-22s' % '$OPT', round(d['ms_per_step_windows']['min'],4), round(d['ms_per_step_windows']['median'],4), {k: round(v*1000,1) for k, v in d['kernel_ms_per_step'].items()})"
done; done
