"""Timing of LM trials on the reference's own dataset shape (data/oleg_synthetic: 100 cameras x 1000
tracks, every track seen by every camera = 100 000 observations, DENSE reduced system: the band is as
wide as the matrix, so the solve goes through k_flatten + LU)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pysfm_amd import Bundle, BundleAdjuster, sensor_model
g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests', 'golden', 'scene_oleg_100x1000.npz'))
model = sensor_model.GaussianModel(1.) if int(g['sensor_kind']) == 0 else sensor_model.CauchyModel(float(g['sensor_sigma']))
b = Bundle.FromObservations(g['K'], g['R'].reshape(-1, 3, 3), g['t'], g['X'], g['obs_cam'], g['obs_pt'], g['obs_z'], sensor_model=model)
ba = BundleAdjuster(verbose=False)
ba.set_bundle(b)
be = ba.backend
print('cameras', be.nc, 'tracks', be.nt, 'obs', be.nobs, 'half bandwidth', be.half_bandwidth)
ba.optimize(max_steps=2)                      # warm-up (code objects, rocBLAS handle)
ba.set_bundle(b)
t0 = time.perf_counter(); ba.optimize(max_steps=10); t1 = time.perf_counter()
print('optimize: %d steps, %d trials, %.1f ms, cost %.6g -> %.6g, solve path %s' % (ba.num_steps, ba.lm_trials, (t1 - t0) * 1e3, ba.costs[0], ba.costs[-1], be.last_solve_path))
be.enable_timing(True); be.timings(reset=True)
ba.set_bundle(b); ba.optimize(max_steps=5)
tm = be.timings(reset=True)
print({k: round(v['ms'] / max(1, ba.lm_trials), 4) for k, v in tm.items() if v['launches']})
