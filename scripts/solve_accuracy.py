"""How accurate is the device's reduced solve where the LM walk is sensitive (config 3 after five LM steps, damping 1e-3, condition
number ~1e13)?  Uploads the ORACLE's [S | b] of that trial (tests/golden/config3_reduced_damping1e-3.npz, oracle/gen_golden_reduced.py:
the same bits every run) into the device's band and solves it REPEAT times without and with the step of iterative refinement
(csrc/ba_bcr_refine.h; option refine), against LAPACK's LU (numpy.linalg.solve = gesv, what the reference calls,
bundle_adjuster.py:302-305) and LAPACK's Cholesky of the same numbers.

    python scripts/solve_accuracy.py [REPEAT=20]      ->  the table kept as profiles/r06_solve_accuracy.txt

Columns: relative residual ||S x - b|| / ||b||, backward error ||S x - b|| / || |S| |x| + |b| || (the measure of Oettli and Prager that the
tests bound: with ||S||_2 ||x|| in its place every row reads 0.000) in units of
eps = 2^-52, distance to LAPACK's Cholesky solution relative to LAPACK LU's distance to it."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from pysfm_amd import synthetic_data as sd
from pysfm_amd.backend import HipBackend
from pysfm_amd._capi import PARAMS_CUR, SENSOR_GAUSS

REPEAT = int(sys.argv[1]) if len(sys.argv) > 1 else 20
EPS = 2. ** -52
g = np.load(os.path.join(ROOT, 'tests', 'golden', 'config3_reduced_damping1e-3.npz'))
band, rhs, x_lu, x_ch, norm2 = g['band'], g['b'].reshape(-1), g['x_lu'], g['x_chol'], float(g['norm2'])
nco, hb = band.shape[0], band.shape[1] - 1
n = 6 * nco
A = np.zeros((nco, nco, 6, 6))
for d in range(hb + 1):
    i = np.arange(nco - d)
    A[i, i + d] = band[i, d]
    A[i + d, i] = band[i, d].transpose(0, 2, 1)
A = A.transpose(0, 2, 1, 3).reshape(n, n)
res = lambda x: np.linalg.norm(A @ x - rhs) / np.linalg.norm(rhs)
absA = np.abs(A)
bwd = lambda x: np.linalg.norm(A @ x - rhs) / np.linalg.norm(absA @ np.abs(x) + np.abs(rhs)) / EPS
dist = lambda x: np.linalg.norm(x - x_ch) / np.linalg.norm(x_lu - x_ch)
print('system: %d unknowns, half-bandwidth %d cameras, cond %.2e, ||S||_2 ||x|| / ||b|| = %.2e' % (n, hb, float(g['cond']), norm2 * np.linalg.norm(x_ch) / np.linalg.norm(rhs)))
print('%-28s %12s %14s %22s' % ('solver', 'residual', 'backward / eps', '|x - x_chol| / |x_lu - x_chol|'))
print('%-28s %12.3e %14.3f %22.3f' % ('LAPACK LU (gesv)', res(x_lu), bwd(x_lu), 1.))
print('%-28s %12.3e %14.3f %22.3f' % ('LAPACK Cholesky (potrf)', res(x_ch), bwd(x_ch), 0.))

s = sd.generate_banded_scene(1000, 100000, init_mode='params')
be = HipBackend(0)
be.set_problem(1000, 100000, s['obs_cam'], s['obs_pt'], s['obs_z'], s['K'], np.arange(1000, dtype=np.int32) - 1, np.ones(100000, np.uint8))
be.set_sensor(SENSOR_GAUSS, [1., 0., 0., 1.])
be.set_params(PARAMS_CUR, s['R0'], s['t0'], s['X0'])
assert be.nco == nco and be.half_bandwidth == hb
for refine in ('0', 'auto', '1'):
    be.set_option('refine', refine)
    rows = []
    for rep in range(REPEAT):
        be.linearize(PARAMS_CUR)
        be.schur(PARAMS_CUR, float(g['damping']), 1e-5)          # (state only: the golden system replaces what it formed)
        be.synchronize()                                          # (the handle works on a stream of its own: its kernels first, then the copies)
        S_t, b_t = be.reduced_tensors()
        S_t.copy_(torch.from_numpy(band.reshape(-1)))
        b_t.copy_(torch.from_numpy(rhs))
        torch.cuda.synchronize()
        be.solve_reduced(None)
        x = be.get_solution().reshape(n)
        rows.append((res(x), bwd(x), dist(x)))
    rows = np.array(rows)
    print('%-28s %12.3e %14.3f %22.3f   (worst of %d; best %.3e %.3f %.3f; %d distinct answers; solver %s, %d solves refined)'
          % ('device, refine = %s' % refine, rows[:, 0].max(), rows[:, 1].max(), rows[:, 2].max(), REPEAT, rows[:, 0].min(), rows[:, 1].min(), rows[:, 2].min(),
             len(set(map(tuple, rows))), be.last_solve_kind, be.problem_info()['solves_refined']))
# what the step costs: HIP events around the solve's kernels, 50 solves each
for refine in ('0', '1'):
    be.set_option('refine', refine)
    be.enable_timing(True)
    be.timings(reset=True)
    for rep in range(50):
        be.solve_reduced(None)
    tm = be.timings(reset=True)
    be.enable_timing(False)
    print('refine = %s: %s' % (refine, {k: '%.1f us' % (1e3 * v['ms'] / 50) for k, v in tm.items() if v['launches']}))
