#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "cyclic or dense or wide_band or lu_fallback or large_system" > $O/r02e_pytest.log 2>&1; echo "pytest rc=$?" >> $O/r02e_pytest.log
tail -8 $O/r02e_pytest.log
for L in 22 23 24; do
timeout 300 python bench.py --no-cpu-baseline --no-lm --windows 2 --track-len $L > $O/r02e_bench_L$L.json 2> $O/r02e_bench_L$L.err; echo "rc=$?"
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r02e_bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, 'ms/step %.4f'%d['ms_per_step'], 'win', d['ms_per_step_windows']['min'], d["roofline"]["timer"], d['reduced_system']['solve_kind'])
        print('   ', {k: round(v,4) for k,v in d['kernel_ms_per_step'].items()})
    except Exception as e:
        print(f, 'FAILED', e)
PY
