"""How much the device's solution of the golden reduced system (tests/golden/config3_reduced_damping1e-3.npz) varies from run to run
(the cyclic reduction adds its Schur complements with fp64 atomics: the order is not fixed), refined and not, whole and with the masked
parameters of test_golden_reduced_system_refinement_with_masked_parameters.  usage (GPU box): python scripts/golden_solve_spread.py [runs]"""
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch                                                                  # noqa: E402
from pysfm_amd.backend import HipBackend                                      # noqa: E402
from test_gpu_parity import banded, load_problem, default_flags, load_golden, O   # noqa: E402

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 40
g = load_golden('config3_reduced_damping1e-3')
band, rhs = g['band'], g['b'].reshape(-1)
nco, hb = band.shape[0], band.shape[1] - 1
A = np.zeros((nco, nco, 6, 6))
for d in range(hb + 1):
    i = np.arange(nco - d)
    A[i, i + d] = band[i, d]
    A[i + d, i] = band[i, d].transpose(0, 2, 1)
A = A.transpose(0, 2, 1, 3).reshape(6 * nco, 6 * nco)
s = banded(1000, 100000)
be = HipBackend(0)
load_problem(be, s['K'], s['R0'], s['t0'], s['X0'], s['obs_cam'], s['obs_pt'], s['obs_z'], *default_flags(1000, 100000), O.Sensor.gaussian(1.))
n = 6 * nco
norm2 = float(g['norm2'])
res = lambda M, r, v: np.linalg.norm(M @ v - r) / np.linalg.norm(r)
bwd = lambda M, r, v: np.linalg.norm(M @ v - r) / np.linalg.norm(np.abs(M) @ np.abs(v) + np.abs(r))      # (normwise, in the measure of Oettli and Prager: |S| |x| + |b|)
masks = {'whole': None}
m = (np.arange(n) % 7 != 2).astype(np.uint8); masks['every 7th parameter'] = m.copy()
m[6 * 400:6 * 403] = 0; masks['every 7th + three cameras (the test)'] = m.copy()
m = np.ones(n, np.uint8); m[6 * 400:6 * 403] = 0; masks['three cameras'] = m
m = np.ones(n, np.uint8); m[3::6] = 0; masks['one translation of every camera'] = m
for name, m in masks.items():
    keep = np.arange(n) if m is None else np.nonzero(m)[0]
    Ak, rk = A[np.ix_(keep, keep)], rhs[keep]
    ref = np.linalg.solve(Ak, rk)
    print('%s: LAPACK residual %.3e backward error %.2f eps' % (name, res(Ak, rk, ref), bwd(Ak, rk, ref) / 2. ** -52))
    for refine in ('0', '1'):
        out, bw = [], []
        for _ in range(runs):
            be.set_option('refine', refine)
            be.linearize(0); be.schur(0, float(g['damping']), 1e-5); be.synchronize()
            S_t, b_t = be.reduced_tensors()
            S_t.copy_(torch.from_numpy(np.ascontiguousarray(band).reshape(-1))); b_t.copy_(torch.from_numpy(rhs)); torch.cuda.synchronize()
            be.solve_reduced(m)
            x = be.get_solution().reshape(-1)
            out.append(res(Ak, rk, x[keep])); bw.append(bwd(Ak, rk, x[keep]) / 2. ** -52)
        out = np.array(out)
        print('    refine %s: residual min %.3e median %.3e max %.3e | backward error max %.2f eps | distinct %d of %d' % (refine, out.min(), np.median(out), out.max(), max(bw), len(set(out.tolist())), runs))

# what one step of refinement SHOULD give on the masked system: the unrefined device solution corrected on the host (residual in
# long double, correction by LAPACK)
m = masks['every 7th parameter']
keep = np.nonzero(m)[0]
Ak, rk = A[np.ix_(keep, keep)], rhs[keep]
be.set_option('refine', '0')
be.linearize(0); be.schur(0, float(g['damping']), 1e-5); be.synchronize()
S_t, b_t = be.reduced_tensors()
S_t.copy_(torch.from_numpy(np.ascontiguousarray(band).reshape(-1))); b_t.copy_(torch.from_numpy(rhs)); torch.cuda.synchronize()
be.solve_reduced(m)
x0 = be.get_solution().reshape(-1)[keep]
r = (rk.astype(np.longdouble) - Ak.astype(np.longdouble) @ x0.astype(np.longdouble)).astype(float)
x1 = x0 + np.linalg.solve(Ak, r)
print('every 7th parameter, host refinement of the device\'s unrefined solution: %.3e -> %.3e' % (res(Ak, rk, x0), res(Ak, rk, x1)))

# distance to LAPACK's Cholesky solution in units of LAPACK's own LU - Cholesky distance (test_golden_reduced_system_solution_vs_lapack)
d_lu = np.linalg.norm(g['x_lu'] - g['x_chol'])
for refine in ('1', '0'):
    rc, rl = [], []
    for _ in range(runs):
        be.set_option('refine', refine)
        be.linearize(0); be.schur(0, float(g['damping']), 1e-5); be.synchronize()
        S_t, b_t = be.reduced_tensors()
        S_t.copy_(torch.from_numpy(np.ascontiguousarray(band).reshape(-1))); b_t.copy_(torch.from_numpy(rhs)); torch.cuda.synchronize()
        be.solve_reduced(None)
        x = be.get_solution().reshape(-1)
        rc.append(np.linalg.norm(x - g['x_chol']) / d_lu); rl.append(np.linalg.norm(x - g['x_lu']) / d_lu)
    print('refine %s: |x - x_chol| / |x_lu - x_chol| min %.2f median %.2f max %.2f ; |x - x_lu| / same: max %.2f' % (refine, min(rc), np.median(rc), max(rc), max(rl)))
