#!/bin/bash
# wide bands (half-bandwidth > 23): the big-node cyclic reduction against the dense blocked Cholesky, config-3 sized scenes
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/wide
F="--steps 20 --warmup 5 --no-cpu-baseline --no-lm --windows 0 --no-other-configs --no-live-pmc"
python bench.py $F --track-len 32 > gpurun_out/wide/L32_auto.json 2>&1
python bench.py $F --track-len 32 --option solver=dense > gpurun_out/wide/L32_dense.json 2>&1
python bench.py $F --track-len 40 > gpurun_out/wide/L40_auto.json 2>&1
python bench.py $F --track-len 25 > gpurun_out/wide/L25_auto.json 2>&1
python bench.py $F --long-tracks 50,80 > gpurun_out/wide/lt80_auto.json 2>&1
python bench.py $F --long-tracks 50,80 --option solver=dense > gpurun_out/wide/lt80_dense.json 2>&1
python bench.py $F --long-tracks 10,200 > gpurun_out/wide/lt200_auto.json 2>&1
