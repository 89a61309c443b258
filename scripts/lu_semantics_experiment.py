"""Reduced systems with more unknowns than the LU fallback takes (backend.LU_FALLBACK_MAX_UNKNOWNS = 2048): does
reporting a failed device Cholesky as ill-conditioned change the LM trajectory against the reference's LU
(numpy.linalg.solve, bundle_adjuster.py:302-305)?  400 cameras (2394 unknowns), run to the noise floor, then restarted
with the damping at 1e-10 where S is numerically singular.  Prints both trial sequences side by side.
usage (GPU box): python scripts/lu_semantics_experiment.py [cams] [points]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ba_oracle as O                                      # noqa: E402
from pysfm_amd import Bundle, BundleAdjuster                           # noqa: E402
from pysfm_amd import synthetic_data as sd                             # noqa: E402

nc = int(sys.argv[1]) if len(sys.argv) > 1 else 400
nt = int(sys.argv[2]) if len(sys.argv) > 2 else 8000
init_mode = sys.argv[3] if len(sys.argv) > 3 else 'params'
s = sd.generate_banded_scene(nc, nt, init_mode=init_mode)
flags = (np.arange(nc, dtype=np.int32) - 1, np.ones(nt, bool))
sen = O.Sensor.gaussian(1.)
for tag, damping0, steps, start in (('from the initial guess', 10., 25, None), ('restart at the floor, damping 1e-10', 1e-10, 6, 'prev')):
    if start is None:
        R, t, X = s['R0'], s['t0'], s['X0']
    b = Bundle.FromObservations(s['K'], R, t, X, s['obs_cam'], s['obs_pt'], s['obs_z'])
    ba = BundleAdjuster(b, verbose=False)
    ba.optimize(max_steps=steps, init_damping=damping0)
    t0 = time.time()
    trace = []
    ref = O.lm_optimize(sen, s['K'], R, t, X, s['obs_cam'], s['obs_pt'], s['obs_z'], *flags, max_steps=steps, init_damping=damping0, trace=trace)
    print('==', tag, '| oracle %.1f s | GPU trials %d, oracle trials %d, Cholesky rejections on the GPU %d' % (
        time.time() - t0, len(ba.trial_log), len(trace), getattr(ba, 'cholesky_rejections', 0)))
    for i in range(max(len(trace), len(ba.trial_log))):
        g = ba.trial_log[i] if i < len(ba.trial_log) else None
        o = trace[i] if i < len(trace) else None
        print('  %2d  GPU %-40s | oracle %s' % (i, '-' if g is None else '%.0e %-15s %s' % (g[0], g[1], 'None' if g[2] is None else '%.9e' % g[2]),
                                               '-' if o is None else '%.0e %-9s %.9e' % (o['damping'], 'accepted' if o['next'] < o['cur'] else 'rejected', o['next'])))
    print('  costs GPU   ', ' '.join('%.9e' % c for c in ba.costs))
    print('  costs oracle', ' '.join('%.9e' % c for c in ref['costs']))
    R, t, X = ref['R'], ref['t'], ref['X']
