#!/bin/bash
# after a solver change: the full GPU suite, then the bench lines that move
cd $GRAFT_REPO_ROOT
O=gpurun_out/quick_${1:-x}
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
line() { python - $1 <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], d['ms_per_step_windows']['min'], d['ms_per_step_windows']['median'], {k: round(v,4) for k,v in d['kernel_ms_per_step'].items()})
PY
}
timeout 300 python bench.py --windows 5 --no-cpu-baseline > $O/c3.json 2> $O/c3.err; line $O/c3.json
timeout 300 python bench.py --windows 3 --no-cpu-baseline --no-lm --option solver=bcr1 > $O/c3_bcr1.json 2> $O/c3_bcr1.err; line $O/c3_bcr1.json
timeout 300 python bench.py --windows 3 --no-cpu-baseline --no-lm --config 5 > $O/c5.json 2> $O/c5.err; line $O/c5.json
for L in 13 24; do timeout 300 python bench.py --windows 3 --track-len $L --no-cpu-baseline --no-lm > $O/L$L.json 2> $O/L$L.err; line $O/L$L.json; done
timeout 300 python bench.py --windows 3 --no-cpu-baseline --no-lm --option solver=dense > $O/c3_dense.json 2> $O/c3_dense.err; line $O/c3_dense.json
