"""How accurate is the device's reduced solve where the LM walk is sensitive (damping 1e-3 at config 3)?  Walks optimize() to
the first trial at that damping, takes [S | b] of that trial from the device, and compares ||S dC - b|| / ||b|| of the device's
solution with LAPACK's (numpy.linalg.solve = gesv, what the reference calls, bundle_adjuster.py:303) and with Cholesky's."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.linalg as sl
from pysfm_amd import Bundle, BundleAdjuster, sensor_model, synthetic_data as sd
from pysfm_amd._capi import PARAMS_CUR
nc, nt = 1000, 100000
s = sd.generate_banded_scene(nc, nt, init_mode='params')
b = Bundle.FromObservations(s['K'], s['R0'], s['t0'], s['X0'], s['obs_cam'], s['obs_pt'], s['obs_z'], sensor_model=sensor_model.GaussianModel(1.))
ba = BundleAdjuster(b, verbose=False)
ba.optimize(max_steps=int(sys.argv[1]) if len(sys.argv) > 1 else 5)
print('walk so far', [(d, o) for d, o, _ in ba.trial_log], 'next damping', ba._damping)
be = ba.backend
for damping in (ba._damping, 1e-3, 1e-4, 1e-2):
    be.linearize(PARAMS_CUR)
    be.schur(PARAMS_CUR, damping, 1e-5)
    S, rhs = be.get_reduced()
    n = be.nco * 6
    A = S.transpose(0, 2, 1, 3).reshape(n, n)
    rhs = rhs.reshape(n)
    be.solve_reduced(None)
    x_dev = be.get_solution().reshape(n)
    t0 = time.time(); x_lu = np.linalg.solve(A, rhs); t_lu = time.time() - t0
    try:
        x_ch = sl.cho_solve(sl.cho_factor(A), rhs)
    except Exception as e:
        x_ch = None
    res = lambda x: np.linalg.norm(A @ x - rhs) / np.linalg.norm(rhs)
    ev = np.linalg.eigvalsh(A)
    print('damping %g: cond %.2e | residual device %.2e  LAPACK LU %.2e  LAPACK Cholesky %s | |x_dev - x_lu| / |x_lu| %.2e  |x_ch - x_lu| / |x_lu| %s  (%s)'
          % (damping, ev[-1] / ev[0], res(x_dev), res(x_lu), 'n/a' if x_ch is None else '%.2e' % res(x_ch), np.linalg.norm(x_dev - x_lu) / np.linalg.norm(x_lu),
             'n/a' if x_ch is None else '%.2e' % (np.linalg.norm(x_ch - x_lu) / np.linalg.norm(x_lu)), be.last_solve_kind))
