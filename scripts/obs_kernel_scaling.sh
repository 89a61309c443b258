#!/bin/bash
# the observation kernels' time against the number of points (1000 cameras, tracks of 10): what of a kernel is a fixed cost
# (launch, the chain of trips to memory of one wavefront) and what grows with the observations
cd $GRAFT_REPO_ROOT
for pts in 12500 25000 50000 100000 200000 400000; do
python bench.py --config 3 --pts-per-gpu $pts --no-other-configs --no-cpu-baseline --no-lm --no-live-pmc --full-line | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('points $pts  ms_per_step', round(d['ms_per_step'],4), {k:round(v*1e3,1) for k,v in d.get('kernel_ms_per_step',{}).items()})"
done
