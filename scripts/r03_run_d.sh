#!/bin/bash
# fused elimination iteration: stress, solver tests + fuzz, A/B bench, time line
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03d; mkdir -p $O
( timeout 300 python scripts/bcr_fused_stress.py 1000 100000 300; echo "rc=$?"
  timeout 300 python scripts/bcr_fused_stress.py 2400 60000 100; echo "rc=$?"
  timeout 300 python scripts/bcr_fused_stress.py 333 20000 100 5; echo "rc=$?"
  timeout 300 python scripts/bcr_fused_stress.py 700 30000 100 12; echo "rc=$?" ) > $O/stress.log 2>&1
grep -v amdgpu.ids $O/stress.log | tail -8
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q -k "cyclic or reduction or solve or fuzz or random or trial or lm" > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
for opt in "fused_eliminate=0" "fused_eliminate=1" "fused_eliminate=0" "fused_eliminate=1"; do
  timeout 600 python bench.py --full-line --no-other-configs --no-live-pmc --no-cpu-baseline --no-lm --option $opt 2> /dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$opt', 'ms/step %.4f' % d['ms_per_step'], d['ms_per_step_windows']['min'], d['ms_per_step_windows']['median'], {k: round(v, 5) for k, v in d['kernel_ms_per_step'].items()})
" | tee -a $O/ab.log
done
if [ "$1" == "trace" ]; then
  make -C pysfm_amd/csrc PROFILE=1 > $O/make.log 2>&1
  python scripts/bcr_phase_trace.py 1000 100000 2> $O/trace.log | tail -1
  grep "k_bcr_eliminate_fused" $O/trace.log | tail -19
fi
