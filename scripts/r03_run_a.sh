#!/bin/bash
# round 3, first GPU pass: the whole -m gpu suite, then the default bench line (with other_configs and live PMC traffic)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03a
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r03a/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r03a/pytest.log
tail -5 gpurun_out/r03a/pytest.log
( time timeout 900 python bench.py ) > gpurun_out/r03a/bench.json 2> gpurun_out/r03a/bench.err
tail -c 3000 gpurun_out/r03a/bench.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r03a/bench.json').read().strip().splitlines()[-1])
print('ms/step', d['ms_per_step'], d['ms_per_step_windows'])
print('roofline', {k: d['roofline'][k] for k in ('kernel', 'frac', 'traffic', 'traffic_source', 'traffic_note', 'avg_launch_ms')})
print('pass', d['roofline_linearise_schur_pass'])
print('kernels', d['kernel_ms_per_step'])
print('lm', {k: d[k] for k in d if k.startswith('lm_') or k == 'final_reproj_rmse'})
for k, v in d.get('other_configs', {}).items():
    print(k, v if not isinstance(v, dict) else {q: v.get(q) for q in ('ms_per_step', 'dominant_kernel', 'dominant_kernel_ms_per_step', 'linearise_schur_pass_fraction_of_kernel_time', 'error')})
PY
