import time, sys, os
sys.path.insert(0, '/root/repo')
import numpy as np
from pysfm_amd import Bundle, BundleAdjuster, sensor_model
from pysfm_amd import synthetic_data as sd
from pysfm_amd._capi import PARAMS_CUR
s = sd.generate_banded_scene(1000, 100000)
b = Bundle.FromObservations(s['K'], s['R0'], s['t0'], s['X0'], s['obs_cam'], s['obs_pt'], s['obs_z'], sensor_model=sensor_model.GaussianModel(1.))
ba = BundleAdjuster(verbose=False); ba.set_bundle(b); be = ba.backend
cur = ba._cost(PARAMS_CUR)
lam = 10.
t_c = 0.; t_all = 0.
for it in range(60):
    t0 = time.perf_counter()
    # replicate ba.trial inline to time the C call
    t1 = time.perf_counter()
    info, cost = be.lm_trial(lam, 1e-5, None)
    t2 = time.perf_counter()
    if cost < cur:
        be.swap_params(); cur = cost; lam *= .1
    else:
        lam *= 10.
    if lam > 1e8 or lam < 1e-12: lam = 10.
    t3 = time.perf_counter()
    if it >= 10:
        t_c += t2 - t1; t_all += t3 - t0
print('C call %.1f us, python around it %.1f us per trial' % (t_c / 50 * 1e6, (t_all - t_c) / 50 * 1e6))
t0 = time.perf_counter()
for it in range(50):
    acc, nxt = ba.trial(lam, None, cur)
t1 = time.perf_counter()
print('ba.trial: %.1f us per trial' % ((t1 - t0) / 50 * 1e6))
