"""Per-kernel times of one LM trial on a TINY problem (the sliding-window caller's: 10 cameras x 100 tracks), HIP events.
usage (GPU box): python scripts/small_trial_timing.py [cams] [tracks]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pysfm_amd import Bundle, BundleAdjuster, sensor_model      # noqa: E402

g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests', 'golden', 'scene_oleg_100x1000.npz'))
ncam = int(sys.argv[1]) if len(sys.argv) > 1 else 10
ntr = int(sys.argv[2]) if len(sys.argv) > 2 else 100
b = Bundle.FromObservations(g['K'], g['R'].reshape(-1, 3, 3), g['t'], g['X'], g['obs_cam'], g['obs_pt'], g['obs_z'],
                            sensor_model=sensor_model.GaussianModel(1.))
ba = BundleAdjuster(verbose=False)
ba.set_bundle(b, camera_ids=list(range(ncam)), track_ids=list(range(ntr)))
be = ba.backend
print('cameras', be.nc, 'tracks', be.nt, 'obs', be.nobs, 'half bandwidth', be.half_bandwidth, be.problem_info())
for _ in range(5):
    be.lm_trial(10., 1e-5, None)
t0 = time.perf_counter()
n = 200
for _ in range(n):
    be.lm_trial(10., 1e-5, None)
dt = (time.perf_counter() - t0) / n
print('ba_lm_trial: %.1f us per call (solver %s)' % (dt * 1e6, be.last_solve_kind))
be.enable_timing(True)
be.timings(reset=True)
for _ in range(50):
    be.lm_trial(10., 1e-5, None)
tm = be.timings(reset=True)
print({k: (round(v['ms'] / 50 * 1e3, 1), v['launches'] // 50) for k, v in tm.items() if v['launches']}, '(us per trial, launches per trial)')
