#!/bin/bash
# per-kernel us of a trial at config 3 and config 5 (bench.py without the side configurations): the quick look after a kernel change
cd $GRAFT_REPO_ROOT
for cfg in 3 5; do
python bench.py --config $cfg --no-other-configs --no-cpu-baseline --no-lm --no-live-pmc --full-line "$@" | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('config $cfg  ms_per_step', round(d['ms_per_step'],4), {k:round(v*1e3,1) for k,v in d.get('kernel_ms_per_step',{}).items()})"
done
