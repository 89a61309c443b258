#!/bin/bash
# cycle stamps of one wavefront of k_backsub_groups (PROFILE build on the box): header loaded, after each batch, after the cost pass
cd $GRAFT_REPO_ROOT
make -C pysfm_amd/csrc -j8 PROFILE=1 > /dev/null 2>&1
for pts in 12500 100000; do
python - <<PY 2>&1 | grep "k_backsub_groups wg" | head -4
import sys
sys.path.insert(0, '.')
from pysfm_amd import Bundle, BundleAdjuster
from pysfm_amd import synthetic_data as sd
s = sd.generate_banded_scene(1000, $pts)
b = Bundle.FromObservations(s['K'], s['R0'], s['t0'], s['X0'], s['obs_cam'], s['obs_pt'], s['obs_z'])
ba = BundleAdjuster(b, verbose=False)
for _ in range(3):
    ba.backend.lm_trial(10., 1e-5, None)
ba.backend.synchronize()
PY
done
