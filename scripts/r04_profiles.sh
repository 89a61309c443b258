#!/bin/bash
# round 4: rocprofv3 kernel stats + PMC traffic of the default bench (config 3) and of config 5 on one GPU, the default bench
# line, and the sweep of the other configurations -> gpurun_out/profiles_r04*/ (copy what is to be judged into profiles/)
cd $GRAFT_REPO_ROOT
TAG=${1:-r04a}
bash scripts/gpu_profile.sh $TAG > gpurun_out/profile_$TAG.log 2>&1
tail -25 gpurun_out/profile_$TAG.log
bash scripts/gpu_profile.sh ${TAG}_config5 --config 5 > gpurun_out/profile_${TAG}_config5.log 2>&1
tail -12 gpurun_out/profile_${TAG}_config5.log
( time python bench.py ) > gpurun_out/profiles_$TAG/${TAG}_bench_default.json 2> gpurun_out/profiles_$TAG/${TAG}_bench_default.err
tail -c 300 gpurun_out/profiles_$TAG/${TAG}_bench_default.err
bash scripts/r02_sweep.sh $TAG 2>&1 | tail -60
