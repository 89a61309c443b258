"""Randomised sweep of the camera layouts ba_set_problem chooses (round 5): sequences of 60 .. 400 cameras with tracks of 3 .. 12,
cameras renumbered at random or not, 0 .. 12 loop-closure tracks of width 1 .. 3 (single points or a few points per tie),
frozen cameras anywhere, masked camera parameters, all sensor models - one LM trial per scene through the C ABI against the
CPU oracle ([S | b] with band, border columns and border block expanded; the solution; the trial's points), every problem
starting from poisoned LDS / workspace.  usage (GPU box): python scripts/layout_fuzz.py [first_seed] [last_seed]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from test_gpu_parity import load_problem, banded
from oracle import ba_oracle as O
from pysfm_amd import synthetic_data as sd
from pysfm_amd.backend import HipBackend
HipBackend.poison_after_set_problem = True
be = HipBackend(0)
first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
last = int(sys.argv[2]) if len(sys.argv) > 2 else 300
bad, kinds = 0, {}
for seed in range(first, last):
    rs = np.random.RandomState(5000 + seed)
    nc = int(rs.randint(60, 400)); L = int(rs.randint(3, 13)); nt = int(rs.randint(5 * nc, 14 * nc))
    s = banded(nc, nt, track_len=L, outlier_frac=float(rs.choice([0., .04])), seed=int(rs.randint(1, 10000)))
    npairs = int(rs.choice([0, 1, 3, 6, 12]))
    if npairs:
        width = int(rs.randint(1, 4))
        near = rs.choice(np.arange(1, nc // 2 - 6), npairs, replace=False)
        far = near + nc // 3 + rs.randint(0, nc // 6, npairs)
        pairs = [(int(a), int(min(b, nc - width - 1))) for a, b in zip(near, far)]
        reps = int(rs.choice([1, 1, 3]))                       # single points, or a few points per tie
        s = sd.add_loop_closure_tracks(s, pairs * reps, width=width, seed=seed)
        nt = len(s['X0'])
    cam, pt, z = s['obs_cam'].copy(), s['obs_pt'], s['obs_z']
    R0, t0 = s['R0'].copy(), s['t0'].copy()
    if rs.rand() < .5:
        perm = rs.permutation(nc)
        R0[perm], t0[perm] = s['R0'], s['t0']
        cam = perm[cam].astype(np.int32)
    keep = rs.rand(len(cam)) >= float(rs.choice([0., .15]))
    keep[np.unique(pt, return_index=True)[1]] = True
    cam, pt, z = cam[keep], pt[keep], z[keep]
    frozen = set(rs.choice(nc, int(rs.randint(1, 4)), replace=False).tolist())
    cp = -np.ones(nc, np.int32)
    opt = [c for c in range(nc) if c not in frozen]
    cp[opt] = np.arange(len(opt))
    po = (rs.rand(nt) >= float(rs.choice([0., .2]))).astype(np.uint8)
    sensor = [O.Sensor.gaussian(1.), O.Sensor.cauchy(.05), O.Sensor.huber(.06)][int(rs.randint(0, 3))]
    mask = (rs.rand(len(opt) * 6) > .05).astype(np.uint8) if rs.rand() < .4 else None
    damping = float(rs.choice([1e-2, .5, 10.]))
    a = (s['K'], R0, t0, s['X0'], cam, pt, z)
    try:
        mu, su, parts = O.compute_update(sensor, *a, cp, po, damping=damping, cam_param_mask=None if mask is None else mask.astype(bool), return_parts=True)
    except O.NormalEquationsIllconditioned:
        continue
    load_problem(be, *a, cp, po, sensor)
    info, cost = be.lm_trial(damping, 1e-5, mask)
    S, b = be.get_reduced()
    pinfo = be.problem_info()
    key = (be.last_solve_kind, 'permuted' if pinfo['cameras_permuted'] else 'caller order', 'border %d' % (1 if pinfo['border_cameras'] else 0))
    kinds[key] = kinds.get(key, 0) + 1
    eS = np.abs(S - parts['S']).max() / np.abs(parts['S']).max(); eb = np.abs(b - parts['b']).max() / max(1e-300, np.abs(parts['b']).max())
    R2, t2, X2 = O.apply_update(R0, t0, s['X0'], mu, su, cp, po)
    Xg = be.get_params(1)[2]
    A = O.flatten_reduced(parts['S'], parts['b'])[0]
    idx = np.nonzero(mask)[0] if mask is not None else np.arange(len(A))
    tol = max(1e-9, 1e-14 * np.linalg.cond(A[np.ix_(idx, idx)]))
    eX = np.abs(Xg - X2).max() / max(1e-300, np.abs(X2).max()) if info == 0 else np.nan
    ok = info == 0 and eS < 1e-11 and eb < 1e-11 and eX < 10 * tol
    if not ok:
        bad += 1
        print('seed', seed, 'nc', nc, 'L', L, 'pairs', npairs, key, pinfo['half_bandwidth'], pinfo['border_cameras'], 'info', info, 'eS %.1e eb %.1e eX %.1e tol %.1e' % (eS, eb, eX, tol))
print('scenes by (solver, camera order, border):', kinds, '| bad', bad)
