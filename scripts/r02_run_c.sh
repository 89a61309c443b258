#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "reduction or ragged or factorised or fused or lm_trial_entry or group_packed" > $O/r02c_pytest.log 2>&1; echo "pytest rc=$?" >> $O/r02c_pytest.log
tail -25 $O/r02c_pytest.log
for L in 10 11 12 13 16 20 24; do
  timeout 300 python bench.py --track-len $L --no-cpu-baseline --no-lm --windows 2 > $O/r02c_bench_L$L.json 2> $O/r02c_bench_L$L.err; echo "L$L rc=$?"
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r02c_bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, 'ms/step %.4f'%d['ms_per_step'], 'win', d['ms_per_step_windows']['min'], d["roofline"]["timer"], d['reduced_system']['solve_kind'])
        print('   ', {k: round(v,4) for k,v in d['kernel_ms_per_step'].items()})
    except Exception as e:
        print(f, 'FAILED', e)
PY
