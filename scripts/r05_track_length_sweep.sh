#!/bin/bash
# round 5: one trial of the config-3 scene (1000 cameras, 100 000 points) by track length - the cliff past 11 cameras per node
# (profiles/r04b_sweep.json: L12 0.320, L13 0.439, L16 0.688, L20 0.953, L24 1.190 ms).  Writes gpurun_out/r05_track_length_sweep.json
cd $GRAFT_REPO_ROOT
python - <<'PY'
import json, subprocess, sys
out = {}
for L in (8, 10, 11, 12, 13, 14, 15, 16, 20, 24):
    r = subprocess.run([sys.executable, 'bench.py', '--track-len', str(L), '--windows', '5', '--no-cpu-baseline', '--no-lm', '--no-other-configs',
                        '--no-live-pmc'], capture_output=True, text=True)
    d = json.loads(r.stdout.strip().splitlines()[-1])
    out['L%d' % L] = {'ms_per_step': d['ms_per_step'], 'ms_per_step_windows': d['ms_per_step_windows'], 'value': d['value'],
                      'kernel_us_per_step': {k: round(v * 1000, 1) for k, v in d['kernel_ms_per_step'].items()},
                      'half_bandwidth': d['config'].get('half_bandwidth'), 'solver': d.get('trials_by_solver_and_outcome')}
    print('L%d' % L, round(d['ms_per_step_windows']['median'], 4), out['L%d' % L]['kernel_us_per_step'], flush=True)
json.dump(out, open('gpurun_out/r05_track_length_sweep.json', 'w'), indent=1)
PY
