#!/bin/bash
# round 6: rocprofv3 kernel stats + PMC traffic of the default bench (config 3), config 5 on one GPU, config 4 (Huber + 10 % outliers)
# and config 4 with the Cauchy model; SQ counters of the trial's kernels (three counter-only passes); kernel stats of the
# 5000-camera unordered collection (conjugate gradients); the default bench line and its detail file
#   -> gpurun_out/profiles_<tag>*/ (copy what is to be judged into profiles/)
cd $GRAFT_REPO_ROOT
TAG=${1:-r06a}
bash scripts/gpu_profile.sh $TAG > gpurun_out/profile_$TAG.log 2>&1
tail -22 gpurun_out/profile_$TAG.log
bash scripts/gpu_profile.sh ${TAG}_config5 --config 5 > gpurun_out/profile_${TAG}_config5.log 2>&1
tail -12 gpurun_out/profile_${TAG}_config5.log
bash scripts/gpu_profile.sh ${TAG}_config4_huber --config 4 > gpurun_out/profile_${TAG}_config4_huber.log 2>&1
tail -12 gpurun_out/profile_${TAG}_config4_huber.log
bash scripts/gpu_profile.sh ${TAG}_config4_cauchy --config 4 --sensor cauchy > gpurun_out/profile_${TAG}_config4_cauchy.log 2>&1
tail -12 gpurun_out/profile_${TAG}_config4_cauchy.log
bash scripts/gpu_sq_counters.sh $TAG > gpurun_out/sq_$TAG.log 2>&1
tail -12 gpurun_out/sq_$TAG.log
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_${TAG}_collection
mkdir -p $OUT; (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o coll -- python $GRAFT_REPO_ROOT/scripts/collection_probe.py 5000 200000 > $OUT/log.txt 2>&1)
rm -f $OUT/*kernel_trace.csv $OUT/*agent_info.csv
python scripts/trim_rocprof_stats.py $OUT/coll_kernel_stats.csv gpurun_out/profiles_$TAG/${TAG}_collection5000_kernel_stats.csv 20
cp $OUT/log.txt gpurun_out/profiles_$TAG/${TAG}_collection5000_probe.txt
( time python bench.py --detail-out gpurun_out/profiles_$TAG/${TAG}_bench_detail.json ) > gpurun_out/profiles_$TAG/${TAG}_bench_default.json 2> gpurun_out/profiles_$TAG/${TAG}_bench_default.err
tail -c 300 gpurun_out/profiles_$TAG/${TAG}_bench_default.err
