#!/bin/bash
# The resident loop's profiles (profiles/r04c_*, r04d_*): rocprofv3 kernel statistics of the sliding-window run, the phase trace of a
# trial, the sliding-window timing with either loop.  usage (GPU box): bash scripts/r04_resident_profiles.sh
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/r04d
python scripts/resident_check.py > gpurun_out/r04d/resident_check.txt 2>&1
python scripts/window_slam_profile.py --profile > gpurun_out/r04d/window_slam_profile.txt 2>&1
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/gpurun_out/r04d/prof" -o slam -- python "$OLDPWD/scripts/window_slam_profile.py" ) > gpurun_out/r04d/rocprof.log 2>&1
find gpurun_out/r04d/prof -name "*kernel_stats.csv" -exec cp {} gpurun_out/r04d/kernel_stats.csv \;
ls -la gpurun_out/r04d
