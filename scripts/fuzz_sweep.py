"""A longer run of the randomised parity sweep of tests/test_gpu_fuzz.py (seeds outside the committed range), every problem
starting from poisoned LDS / workspace: one LM trial per scene against the oracle, a table of which reduction kernel /
solver each scene took.  usage (GPU box): python scripts/fuzz_sweep.py [first_seed] [last_seed]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import test_gpu_fuzz as F
from test_gpu_parity import load_problem
from oracle import ba_oracle as O
from pysfm_amd.backend import HipBackend
HipBackend.poison_after_set_problem = True
be = HipBackend(0)
bad = 0
kinds = {}
first = int(sys.argv[1]) if len(sys.argv) > 1 else 60
last = int(sys.argv[2]) if len(sys.argv) > 2 else 460
for seed in range(first, last):
    c = F.make_case(seed)
    a, cp, po, sensor = c['a'], c['cam_opt_pos'], c['pt_opt'], c['sensor']
    cmask = None if c['mask'] is None else c['mask'].astype(bool)
    try:
        mu, su, parts = O.compute_update(sensor, *a, cp, po, damping=c['damping'], cam_param_mask=cmask, return_parts=True)
    except O.NormalEquationsIllconditioned:
        continue
    load_problem(be, *a, cp, po, sensor)
    info, cost = be.lm_trial(c['damping'], 1e-5, c['mask'])
    S, b = be.get_reduced()
    pinfo = be.problem_info()
    key = (pinfo['schur_kernel'], be.last_solve_kind, 'permuted' if pinfo['cameras_permuted'] else 'caller order', 'border %s' % ('yes' if pinfo['border_cameras'] else 'no'))
    kinds[key] = kinds.get(key, 0) + 1
    eS = np.abs(S - parts['S']).max() / np.abs(parts['S']).max(); eb = np.abs(b - parts['b']).max() / max(1e-300, np.abs(parts['b']).max())
    R2, t2, X2 = O.apply_update(a[1], a[2], a[3], mu, su, cp, po)
    Xg = be.get_params(1)[2]
    A = O.flatten_reduced(parts['S'], parts['b'])[0]
    idx = np.nonzero(cmask)[0] if cmask is not None else np.arange(len(A))
    cond = np.linalg.cond(A[np.ix_(idx, idx)])
    tol = max(1e-9, 1e-14 * cond)
    eX = np.abs(Xg - X2).max() / max(1e-300, np.abs(X2).max()) if info == 0 else np.nan
    ok = info == 0 and eS < 1e-11 and eb < 1e-11 and eX < 10 * tol
    if not ok:
        bad += 1
        print('seed', seed, 'L', c['L'], 'nc', c['nc'], key, 'info', info, 'eS %.1e eb %.1e eX %.1e tol %.1e cond %.1e' % (eS, eb, eX, tol, cond))
print('kinds', kinds, 'bad', bad)
