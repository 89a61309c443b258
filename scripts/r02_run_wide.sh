#!/bin/bash
# wide cyclic reduction after a kernel change: the tests that exercise it, then the track-length bench lines
cd $GRAFT_REPO_ROOT
O=gpurun_out/wide_${1:-x}
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "wide or track_len or L16 or L20 or L24 or long or fuzz or reduction" 2>&1 | tail -5
for L in 13 16 20 24; do timeout 300 python bench.py --windows 3 --track-len $L --no-cpu-baseline --no-lm > $O/L$L.json 2> $O/L$L.err; python - $O/L$L.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], d['ms_per_step_windows'], {k: round(v,4) for k,v in d['kernel_ms_per_step'].items()})
PY
done
