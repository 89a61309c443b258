#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "cyclic or band_solver or lm_trial_entry or config3 or dense_cholesky_equals or non_positive" > $O/r02d_pytest.log 2>&1; echo "pytest rc=$?" >> $O/r02d_pytest.log
tail -25 $O/r02d_pytest.log
timeout 300 python bench.py --no-cpu-baseline --windows 3 > $O/r02d_bench_c3.json 2> $O/r02d_bench_c3.err; echo "rc=$?"
timeout 300 python bench.py --no-cpu-baseline --no-lm --windows 3 --track-len 12 > $O/r02d_bench_L12.json 2> $O/r02d_bench_L12.err; echo "rc=$?"
timeout 300 python bench.py --no-cpu-baseline --no-lm --windows 3 --config 5 > $O/r02d_bench_c5.json 2> $O/r02d_bench_c5.err; echo "rc=$?"
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r02d_bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, 'ms/step %.4f'%d['ms_per_step'], 'win', d['ms_per_step_windows']['min'], d["roofline"]["timer"], d['reduced_system']['solve_kind'], d.get('final_reproj_rmse'))
        print('   ', {k: round(v,4) for k,v in d['kernel_ms_per_step'].items()})
    except Exception as e:
        print(f, 'FAILED', e)
PY
