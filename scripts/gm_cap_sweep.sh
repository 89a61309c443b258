#!/bin/bash
# How the group size of the matrix-core reduction (option gm_cap: points per group) moves the reduction's time when the groups
# do not fit one round of wavefront-pair slots: config 3, a config-5 shard of an 8-GPU run (1250 cameras / 125k points), config 5.
cd $GRAFT_REPO_ROOT
run() { python bench.py --full-line --windows 2 --no-cpu-baseline --no-lm --no-other-configs --no-live-pmc "$@" 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('   ms/step %.4f  schur %.1f us  linearize %.1f  backsub %.1f  groups %s' % (d['ms_per_step_windows']['median'], 1e3*k['schur_pairs'], 1e3*k['linearize'], 1e3*k['backsub'], d['problem_info']['mfma_groups']))"; }
for CAP in 0 24 36 48 60 90; do
  echo "gm_cap=$CAP"
  echo -n " config 3           "; run --option gm_cap=$CAP
  echo -n " 1250 cams x 125k   "; run --cams 1250 --pts-per-gpu 125000 --option gm_cap=$CAP
  echo -n " config 5           "; run --config 5 --option gm_cap=$CAP
done
