#!/bin/bash
# round 5: rocprofv3 kernel stats + PMC traffic of the default bench (config 3) and of config 5 on one GPU, SQ counters of the
# trial's kernels (three counter-only passes), the same for config 3 + 10 loop-closure tracks (band + border), the default
# bench line -> gpurun_out/profiles_<tag>*/ (copy what is to be judged into profiles/)
cd $GRAFT_REPO_ROOT
TAG=${1:-r05a}
bash scripts/gpu_profile.sh $TAG > gpurun_out/profile_$TAG.log 2>&1
tail -25 gpurun_out/profile_$TAG.log
bash scripts/gpu_profile.sh ${TAG}_config5 --config 5 > gpurun_out/profile_${TAG}_config5.log 2>&1
tail -12 gpurun_out/profile_${TAG}_config5.log
bash scripts/gpu_profile.sh ${TAG}_loops --loop-closures 10 > gpurun_out/profile_${TAG}_loops.log 2>&1
tail -12 gpurun_out/profile_${TAG}_loops.log
bash scripts/gpu_sq_counters.sh $TAG > gpurun_out/sq_$TAG.log 2>&1
tail -12 gpurun_out/sq_$TAG.log
( time python bench.py ) > gpurun_out/profiles_$TAG/${TAG}_bench_default.json 2> gpurun_out/profiles_$TAG/${TAG}_bench_default.err
tail -c 300 gpurun_out/profiles_$TAG/${TAG}_bench_default.err
