"""Debug aid: one stepwise LM trial at a given size, printing after each entry point (AMD_SERIALIZE_KERNEL=3)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pysfm_amd import Bundle, BundleAdjuster, sensor_model
from pysfm_amd import synthetic_data as sd
nc, nt = int(sys.argv[1]), int(sys.argv[2])
s = sd.generate_banded_scene(nc, nt)
b = Bundle.FromObservations(s['K'], s['R0'], s['t0'], s['X0'], s['obs_cam'], s['obs_pt'], s['obs_z'], sensor_model=sensor_model.GaussianModel(1.))
ba = BundleAdjuster(verbose=False); ba.set_bundle(b); be = ba.backend
def step(name, f):
    r = f(); torch.cuda.synchronize(); print(name, 'ok', flush=True); return r
step('linearize', lambda: be.linearize(0))
step('schur', lambda: be.schur(0, 10., 1e-5))
S, bb = be.get_reduced(); print('S', np.abs(S).max(), np.isfinite(S).all(), flush=True)
step('solve', lambda: be.solve_reduced(None))
step('backsub', lambda: be.backsubstitute(0, fetch=False))
print('trial', be.lm_trial(10., 1e-5, None), flush=True)
print('trial', be.lm_trial(1., 1e-5, None), flush=True)
from pysfm_amd._capi import PARAMS_CUR
damping, cur = 10., ba._cost(PARAMS_CUR)
_orig = be.lm_trial
def _wrapped(*a, **k):
    r = _orig(*a, **k); print('   lm_trial ->', r, be.last_solve_path, flush=True); return r
be.lm_trial = _wrapped
for i in range(int(os.environ.get('NTRIALS', '16'))):
    acc, nxt = ba.trial(damping, None, cur)
    torch.cuda.synchronize()
    print(i, 'damping', damping, 'accepted', acc, 'cost', nxt, flush=True)
    if acc: damping *= .1; cur = nxt
    else: damping *= 10.
    if damping >= 1e8 or damping < 1e-12: damping = 10.
