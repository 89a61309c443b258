"""Does ba_set_problem pick the fastest kernels?  One LM trial (ba_lm_trial at damping 10) on scenes of several shapes with the
library's own choice and with each alternative forced.  usage (GPU box): python scripts/kernel_choice_probe.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pysfm_amd import Bundle, BundleAdjuster, sensor_model, synthetic_data      # noqa: E402


def trial_ms(b, opts, kw={}):
    ba = BundleAdjuster(verbose=False)
    ba.resident = False
    for k, v in opts.items():
        ba.backend.set_option(k, v)
    ba.set_bundle(b, **kw)
    be = ba.backend
    for _ in range(3):
        be.lm_trial(10., ba.SCHUR_COMPLIMENT_PINV_THRESHOLD)
    t0 = time.perf_counter()
    n = 20
    for _ in range(n):
        info, cost = be.lm_trial(10., ba.SCHUR_COMPLIMENT_PINV_THRESHOLD)
    dt = (time.perf_counter() - t0) / n
    pi = be.problem_info()
    return dt * 1e3, cost, pi


def scene(nc, nt, L, drop=0., seed=1, mix=None):
    s = synthetic_data.generate_banded_scene(nc, nt, track_len=L, seed=seed, msm_noise=.01, init_perturbation=.01)
    cam, pt, z = s['obs_cam'], s['obs_pt'], s['obs_z']
    rs = np.random.RandomState(5)
    keep = np.ones(len(cam), bool)
    if drop:
        keep = rs.rand(len(cam)) >= drop
    if mix:                                   # tracks of mixed lengths: track k keeps its first mix[k % len(mix)] observations
        first = np.concatenate(([0], np.flatnonzero(pt[1:] != pt[:-1]) + 1))
        idx_in_track = np.arange(len(cam)) - np.repeat(first, np.diff(np.concatenate((first, [len(cam)]))))
        keep &= idx_in_track < np.asarray(mix)[pt % len(mix)]
    first = np.concatenate(([True], pt[1:] != pt[:-1]))
    keep |= first | np.concatenate(([False], first[:-1]))
    return Bundle.FromObservations(s['K'], s['R0'], s['t0'], s['X0'], cam[keep], pt[keep], z[keep], sensor_model=sensor_model.GaussianModel(1.))


g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests', 'golden', 'scene_oleg_100x1000.npz'))
scenes = [('oleg 100 x 1000 (real tracks)', Bundle.FromObservations(g['K'], g['R'].reshape(-1, 3, 3), g['t'], g['X'], g['obs_cam'], g['obs_pt'], g['obs_z'], sensor_model=sensor_model.GaussianModel(1.))),
          ('300 x 30000 L=6', scene(300, 30000, 6)),
          ('300 x 30000 L=6, 10 % missing', scene(300, 30000, 6, .1)),
          ('300 x 30000 L=10, lengths 3..10 mixed', scene(300, 30000, 10, mix=(3, 5, 7, 10, 4, 10, 6, 8))),
          ('300 x 20000 L=16', scene(300, 20000, 16)),
          ('300 x 20000 L=16, 10 % missing', scene(300, 20000, 16, .1)),
          ('200 x 10000 L=24, 5 % missing', scene(200, 10000, 24, .05)),
          ('300 x 20000 L=12, 10 % missing', scene(300, 20000, 12, .1)),
          ('300 x 20000 L=13', scene(300, 20000, 13)),
          ('1000 x 20000 L=10 (20 points a camera step)', scene(1000, 20000, 10)),
          ('1000 x 5000 L=10 (5 points a camera step)', scene(1000, 5000, 10))]
alts = [('auto', {}), ('schur=mfma2', {'schur': 'mfma2'}), ('schur=mfma', {'schur': 'mfma'}), ('schur=groups', {'schur': 'groups'}), ('schur=pairs', {'schur': 'pairs'}),
        ('point_kernels=v1', {'point_kernels': 'v1'}), ('fuse_cam=0', {'fuse_cam': '0'})]
for name, b in scenes:
    out = []
    ref = None
    for an, o in alts:
        try:
            ms, cost, pi = trial_ms(b, o)
        except Exception as e:
            out.append('%s: %s' % (an, str(e)[:40]))
            continue
        if ref is None:
            ref = cost
        out.append('%s %.3f ms (kernel %d%s)%s' % (an, ms, pi['schur_kernel'], ', runs' if pi['point_groups'] else '', '' if abs(cost - ref) <= 1e-9 * abs(ref) else ' COST DIFFERS'))
    print('%-45s %s' % (name, ' | '.join(out)))
