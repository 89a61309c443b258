#!/bin/bash
# round 5, after nodes of 12 / 13 cameras moved into the one-launch cyclic reduction and the window chunks were relaxed: rocprofv3
# kernel stats (+ PMC traffic) of track lengths 13, 14, 16, and the track-length sweep
cd $GRAFT_REPO_ROOT
for spec in "L13:--track-len 13" "L14:--track-len 14" "L16:--track-len 16"; do
  name=${spec%%:*}; flags=${spec#*:}
  bash scripts/gpu_profile.sh r05x_$name $flags > gpurun_out/profile_r05x_$name.log 2>&1
  tail -3 gpurun_out/profile_r05x_$name.log | cut -c1-300
done
bash scripts/r05_track_length_sweep.sh 2>&1 | tail -11 | cut -c1-200
