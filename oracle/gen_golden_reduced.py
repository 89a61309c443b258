#!/usr/bin/env python3
"""Golden reduced camera system where the LM walk is sensitive to the last digits of its solve: config 3 (1000 cameras /
100 000 points / 1M observations, seed 654) after five LM steps of the CPU oracle, damping 1e-3 - condition number ~1e13, the
free scale of a monocular reconstruction barely damped.

    python oracle/gen_golden_reduced.py         # writes tests/golden/config3_reduced_damping1e-3.npz  (~1.5 minutes on 8 cores)

Stored: the band of S (block (i, i + d) for d = 0 .. 9; S is symmetric and nothing lies outside: asserted), b, LAPACK's LU
solution (numpy.linalg.solve = gesv: what the reference calls, bundle_adjuster.py:302-305) and LAPACK's Cholesky solution of
exactly these numbers, ||S||_2, and the walk that led there.  The GPU tests upload [S | b] as it is (no device arithmetic before the
solve: the input is the same bits every run) and check the device's solve against these.  Test infrastructure: only tests/ and
scripts/solve_accuracy.py read the file."""
import os
import sys
import time

import numpy as np
import scipy.linalg as sl

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ba_oracle as O                      # noqa: E402
from pysfm_amd import synthetic_data as sd             # noqa: E402

NC, NT, STEPS, DAMPING, HB = 1000, 100000, 5, 1e-3, 9


def main():
    t0 = time.time()
    s = sd.generate_banded_scene(NC, NT, track_len=10, outlier_frac=0., init_mode='params')
    sen = O.Sensor.gaussian(1.)
    flags = (np.arange(NC, dtype=np.int32) - 1, np.ones(NT, bool))
    obs = (s['obs_cam'], s['obs_pt'], s['obs_z'])
    trace = []
    ref = O.lm_optimize(sen, s['K'], s['R0'], s['t0'], s['X0'], *obs, *flags, max_steps=STEPS, init_damping=10., trace=trace)
    print('walk: %s, next damping %g (%.0f s)' % ([(tr['damping'], tr['next'] < tr['cur']) for tr in trace], ref['damping'], time.time() - t0), flush=True)
    _, _, parts = O.compute_update(sen, s['K'], ref['R'], ref['t'], ref['X'], *obs, *flags, damping=DAMPING, return_parts=True)
    S, b = parts['S'], parts['b']
    nco = S.shape[0]
    band = np.zeros((nco, HB + 1, 6, 6))
    inside = np.zeros((nco, nco), bool)
    for d in range(HB + 1):
        i = np.arange(nco - d)
        band[i, d] = S[i, i + d]
        inside[i, i + d] = inside[i + d, i] = True
    assert not np.any(S[~inside]), 'the reduced system of this scene is a band of half-width 9'
    # The device keeps the upper triangle of S (blocks (i, i + d), the diagonal blocks by their upper triangle): the system the
    # goldens are solved for is THAT one, mirrored - the oracle's own S is symmetric only to round-off (its lower blocks are summed
    # in another order), ~1e-16 relative, and LAPACK's LU would otherwise solve a slightly different matrix than the device sees.
    iu = np.triu_indices(6, 1)
    band[:, 0][:, iu[1], iu[0]] = band[:, 0][:, iu[0], iu[1]]
    Sm = np.zeros_like(S)
    for d in range(HB + 1):
        i = np.arange(nco - d)
        Sm[i, i + d] = band[i, d]
        Sm[i + d, i] = band[i, d].transpose(0, 2, 1)
    print('asymmetry of the oracle\'s S: %.2e of its largest entry' % (np.max(np.abs(Sm - S)) / np.max(np.abs(S))))
    A, rhs = O.flatten_reduced(Sm, b)
    assert np.array_equal(A, A.T)
    x_lu = np.linalg.solve(A, rhs)
    x_ch = sl.cho_solve(sl.cho_factor(A), rhs)
    ev = np.linalg.eigvalsh(A)
    res = lambda x: float(np.linalg.norm(A @ x - rhs) / np.linalg.norm(rhs))
    path = os.path.join(ROOT, 'tests', 'golden', 'config3_reduced_damping1e-3.npz')
    np.savez_compressed(path, band=band, b=b, x_lu=x_lu, x_chol=x_ch, norm2=ev[-1], cond=ev[-1] / ev[0], damping=DAMPING, lm_steps=STEPS,
                        walk_damping=np.array([tr['damping'] for tr in trace]), walk_next=np.array([tr['next'] for tr in trace]),
                        residual_lu=res(x_lu), residual_chol=res(x_ch))
    print('nco %d, cond %.3e, ||S||_2 %.3e, ||b|| %.3e, ||x|| %.3e | residual LU %.2e Cholesky %.2e | |x_lu - x_ch| / |x_ch| %.2e -> %s (%.0f s)'
          % (nco, ev[-1] / ev[0], ev[-1], np.linalg.norm(rhs), np.linalg.norm(x_ch), res(x_lu), res(x_ch),
             np.linalg.norm(x_lu - x_ch) / np.linalg.norm(x_ch), path, time.time() - t0))


if __name__ == '__main__':
    main()
