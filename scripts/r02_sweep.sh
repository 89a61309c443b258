#!/bin/bash
# bench lines of round 2: config 3 (sorted / shuffled), configs 2, 4, 5, track lengths, forced one-rank RCCL path
cd $GRAFT_REPO_ROOT
O=gpurun_out/sweep_${1:-r02}
mkdir -p $O
run() { name=$1; shift; timeout 600 python bench.py --windows 3 "$@" > $O/$name.json 2> $O/$name.err; echo "$name rc=$?"; }
run c3
run c3_shuffled --shuffle-points --no-cpu-baseline
run c2 --config 2 --no-cpu-baseline
run c4_huber --config 4 --no-cpu-baseline
run c4_cauchy --config 4 --sensor cauchy --no-cpu-baseline
run c5 --config 5 --no-cpu-baseline
run c3_forcecomm --force-comm --no-cpu-baseline --no-lm
run c3_forcecomm_torch --force-comm --collectives torch --no-cpu-baseline --no-lm
for L in 11 12 13 16 20 24; do run L$L --track-len $L --no-cpu-baseline --no-lm; done
run L16_shuffled --track-len 16 --shuffle-points --no-cpu-baseline --no-lm
run c3_ragged --drop-observations 0.3 --no-cpu-baseline
run c3_ragged_shuffled --drop-observations 0.3 --shuffle-points --no-cpu-baseline --no-lm
run c3_ragged_pairs --drop-observations 0.3 --option schur=pairs --no-cpu-baseline --no-lm
run c5_shuffled --config 5 --shuffle-points --no-cpu-baseline --no-lm
python - $O <<'PY'
import json,glob,sys,os
for f in sorted(glob.glob(sys.argv[1]+'/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        w=d['ms_per_step_windows']
        print('%-22s ms/step %.4f  windows min %.4f med %.4f  %.3e obs/s  dom %s (%s, frac %.4f)  solve %s  rmse %s trials %s' % (os.path.basename(f)[:-5], d['ms_per_step'], w['min'], w['median'], d['value'], d["roofline"]["timer"], d['roofline']['bound'], d['roofline']['frac'], d['reduced_system']['solve_kind'], d.get('final_reproj_rmse'), d.get('lm_trials')))
        print('      ', {k: round(v,4) for k,v in d['kernel_ms_per_step'].items()})
    except Exception as e:
        print(f, 'FAILED', e)
PY
