"""Per-trial wall time, damping, accept flag and solver path over a long run of the bench's trial loop."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pysfm_amd import Bundle, BundleAdjuster, sensor_model
from pysfm_amd import synthetic_data as sd
from pysfm_amd._capi import PARAMS_CUR
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
s = sd.generate_banded_scene(1000, 100000)
b = Bundle.FromObservations(s['K'], s['R0'], s['t0'], s['X0'], s['obs_cam'], s['obs_pt'], s['obs_z'], sensor_model=sensor_model.GaussianModel(1.))
ba = BundleAdjuster(verbose=False); ba.set_bundle(b); be = ba.backend
damping, cur = 10., ba._cost(PARAMS_CUR)
rows = []
for i in range(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    acc, nxt = ba.trial(damping, None, cur)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    rows.append((i, damping, acc, nxt, dt * 1e3, be.last_solve_path))
    if acc: damping *= .1; cur = nxt
    else: damping *= 10.
    if damping >= 1e8 or damping < 1e-12: damping = 10.
slow = [r for r in rows if r[4] > 0.6]
print('trials', n, 'mean ms', np.mean([r[4] for r in rows]), 'slow (>0.6 ms):', len(slow))
for r in slow[:25]: print('  trial %d damping %.1e accepted %s cost %s  %.3f ms  path %s' % r)
print('paths', {p: sum(1 for r in rows if r[5] == p) for p in set(r[5] for r in rows)})
