#!/bin/bash
# round 5: rocprofv3 kernel stats (+ PMC traffic) of the wide-band configurations the cliff of profiles/r04b_sweep.json is made of
cd $GRAFT_REPO_ROOT
for spec in "L13:--track-len 13" "L16:--track-len 16" "L32:--track-len 32" "long80:--long-tracks 50,80"; do
  name=${spec%%:*}; flags=${spec#*:}
  bash scripts/gpu_profile.sh r05w_$name $flags > gpurun_out/profile_r05w_$name.log 2>&1
  tail -3 gpurun_out/profile_r05w_$name.log | cut -c1-300
done
