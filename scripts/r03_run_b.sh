#!/bin/bash
# round 3, fused elimination: stress, the whole -m gpu suite, A/B bench
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03b
timeout 300 python scripts/bcr_fused_stress.py 1000 100000 300 > gpurun_out/r03b/stress.log 2>&1; echo "stress rc=$?" >> gpurun_out/r03b/stress.log
timeout 300 python scripts/bcr_fused_stress.py 2400 60000 100 >> gpurun_out/r03b/stress.log 2>&1; echo "stress rc=$?" >> gpurun_out/r03b/stress.log
timeout 300 python scripts/bcr_fused_stress.py 333 20000 100 5 >> gpurun_out/r03b/stress.log 2>&1; echo "stress rc=$?" >> gpurun_out/r03b/stress.log
tail -8 gpurun_out/r03b/stress.log
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/r03b/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r03b/pytest.log
tail -15 gpurun_out/r03b/pytest.log
for opt in "fused_eliminate=0" "fused_eliminate=1" "fused_eliminate=0" "fused_eliminate=1"; do
  timeout 600 python bench.py --full-line --no-other-configs --no-live-pmc --no-cpu-baseline --no-lm --option $opt 2> /dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$opt', 'ms/step %.4f' % d['ms_per_step'], d['ms_per_step_windows']['min'], d['ms_per_step_windows']['median'], {k: round(v, 5) for k, v in d['kernel_ms_per_step'].items()})
" | tee -a gpurun_out/r03b/ab.log
done
