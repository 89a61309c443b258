#!/bin/bash
# wide bands (half-bandwidth > 23) at config-3 size: the big-node cyclic reduction against the dense blocked Cholesky, the
# rectangular-task reduction of long tracks against the pair kernel -> gpurun_out/wide/*.json, summary: profiles/r03d_wide_band.json
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/wide
F="--steps 20 --warmup 5 --no-cpu-baseline --no-lm --windows 2 --no-other-configs --no-live-pmc"
run() { name=$1; shift; python bench.py $F "$@" > gpurun_out/wide/$name.json 2> gpurun_out/wide/$name.err; }
run L25 --track-len 25
run L32 --track-len 32
run L32_dense --track-len 32 --option solver=dense
run L40 --track-len 40
run L40_dense --track-len 40 --option solver=dense
run lt80 --long-tracks 50,80
run lt80_dense --long-tracks 50,80 --option solver=dense
run lt80_pairs --long-tracks 50,80 --option schur=pairs
run lt200 --long-tracks 10,200
run lt200_bcr --long-tracks 10,200 --option solver=bcr
python - <<'PY'
import json, glob, os
out = {}
for f in sorted(glob.glob('gpurun_out/wide/*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        out[os.path.basename(f)[:-5]] = {'error': str(e)}
        continue
    out[os.path.basename(f)[:-5]] = {'ms_per_step': d['ms_per_step'], 'windows': d.get('ms_per_step_windows'), 'kernel_ms_per_step': d['kernel_ms_per_step'],
                                     'kernel_launches_per_step': d.get('kernel_launches_per_step'), 'workload': d['config'], 'options': d.get('options')}
json.dump(out, open('gpurun_out/wide/summary.json', 'w'), indent=1, sort_keys=True)
for k, v in out.items():
    if 'error' in v: print(k, v); continue
    print('%-12s %8.3f ms  %s' % (k, v['ms_per_step'], {a: round(b, 3) for a, b in v['kernel_ms_per_step'].items()}))
PY
