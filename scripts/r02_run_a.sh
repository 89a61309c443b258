#!/bin/bash
# first GPU pass of round 2: tests, copy probe, bench lines
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > $O/r02a_pytest.log 2>&1; echo "pytest rc=$?" >> $O/r02a_pytest.log
tail -5 $O/r02a_pytest.log
timeout 120 ./tools/copy_probe > $O/r02a_copy_probe.log 2>&1
cat $O/r02a_copy_probe.log
timeout 600 python bench.py > $O/r02a_bench_c3.json 2> $O/r02a_bench_c3.err; echo "c3 rc=$?"
timeout 300 python bench.py --shuffle-points --no-cpu-baseline > $O/r02a_bench_c3_shuffled.json 2> $O/r02a_bench_c3_shuffled.err; echo "c3s rc=$?"
timeout 300 python bench.py --config 4 --no-cpu-baseline > $O/r02a_bench_c4.json 2> $O/r02a_bench_c4.err; echo "c4 rc=$?"
timeout 600 python bench.py --config 5 --no-cpu-baseline > $O/r02a_bench_c5.json 2> $O/r02a_bench_c5.err; echo "c5 rc=$?"
timeout 300 python bench.py --force-comm --no-cpu-baseline --no-lm > $O/r02a_bench_fc.json 2> $O/r02a_bench_fc.err; echo "fc rc=$?"
for L in 11 12 16; do
  timeout 300 python bench.py --track-len $L --no-cpu-baseline --no-lm > $O/r02a_bench_L$L.json 2> $O/r02a_bench_L$L.err; echo "L$L rc=$?"
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r02a_bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, 'ms/step %.4f'%d['ms_per_step'], 'win', d['ms_per_step_windows']['min'], d['ms_per_step_windows']['median'], 'value %.3e'%d['value'], d["roofline"]["timer"], d['roofline']['bound'], '%.4f'%d['roofline']['frac'], d.get('final_reproj_rmse'), d['reduced_system']['solve_kind'])
        print('   ', {k: round(v,4) for k,v in d['kernel_ms_per_step'].items()})
    except Exception as e:
        print(f, 'FAILED', e)
PY
