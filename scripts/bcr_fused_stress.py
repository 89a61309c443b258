#!/usr/bin/env python3
"""Stress of k_bcr_eliminate_fused on the GPU box: the same reduced system solved `reps` times through the one-launch
elimination and once through the per-level launches; every solution must agree with the per-level one to 1e-11 of its
largest entry (the order of the fp64 atomics differs, nothing else may), and no solve may time out.
usage: bcr_fused_stress.py [cams] [points] [reps] [track_len] [refine: 0 | 1 (round 6: the refinement step behind every solve)]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pysfm_amd import synthetic_data as sd          # noqa: E402
from pysfm_amd.backend import HipBackend            # noqa: E402

nc = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
nt = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 300
L = int(sys.argv[4]) if len(sys.argv) > 4 else 10
refine = sys.argv[5] if len(sys.argv) > 5 else '0'
s = sd.generate_banded_scene(nc, nt, track_len=L)
be = HipBackend(0)
be.set_problem(nc, nt, s['obs_cam'], s['obs_pt'], s['obs_z'], s['K'], np.arange(nc, dtype=np.int32) - 1, np.ones(nt, np.uint8))
be.set_sensor(0, np.eye(2).reshape(4))
be.set_params(0, s['R0'], s['t0'], s['X0'])
be.linearize(0)
be.schur(0, 10., 1e-5)
be.set_option('fused_eliminate', 0)
be.set_option('fused_backsolve', 0)            # (nodes of 14 .. 21 cameras: the one-launch back-substitution of ba_bcr_wide.h is what is stressed)
be.solve_reduced(None)
ref = be.get_solution()
kind = be.last_solve_kind
be.set_option('fused_eliminate', 1)
be.set_option('fused_backsolve', 1)
be.set_option('refine', refine)
worst, t0 = 0., time.time()
for r in range(reps):
    if r % 3 == 2:
        be.debug_poison()
        be.linearize(0)
        be.schur(0, 10., 1e-5)
    be.solve_reduced(None)
    x = be.get_solution()
    assert be.last_solve_kind == kind and kind in ('bcr', 'bcr_wide') and be.last_solve_path == 'band', (be.last_solve_kind, be.last_solve_path)
    d = np.max(np.abs(x - ref)) / np.max(np.abs(ref))
    assert np.all(np.isfinite(x)) and d <= 1e-11, (r, d)
    worst = max(worst, d)
print('bcr_fused_stress (refine = %s, %d solves refined): %d cameras, hb %d, %d solves, worst relative difference to the per-level solve %.2e, %.1f s'
      % (refine, be.problem_info()['solves_refined'], nc, be.half_bandwidth, reps, worst, time.time() - t0))
