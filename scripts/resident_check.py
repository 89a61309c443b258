"""The one-workgroup LM loop (csrc/ba_resident.h) against the Python loop over ba_lm_trial on the same problems: decisions,
costs, final parameters - and what each costs in wall-clock.  usage (GPU box): python scripts/resident_check.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pysfm_amd import Bundle, BundleAdjuster, sensor_model, synthetic_data, window_slam      # noqa: E402


def run(bundle, resident, **kw):
    ba = BundleAdjuster(verbose=False)
    ba.resident = resident
    ba.set_bundle(bundle, **kw)
    ba.optimize()                   # warm-up
    ba.set_bundle(bundle, **kw)
    t0 = time.perf_counter()
    ba.optimize()
    dt = time.perf_counter() - t0
    b = ba.bundle
    return ba, dt, np.array([c.R for c in b.cameras]), np.array([c.t for c in b.cameras]), np.asarray(b.reconstruction)


def compare(name, bundle, **kw):
    a, ta, Ra, ta_, Xa = run(bundle, False, **kw)
    r, tr, Rr, tr_, Xr = run(bundle, True, **kw)
    dec_a = [(d, o) for d, o, _ in a.trial_log]
    dec_r = [(d, o) for d, o, _ in r.trial_log]
    same = dec_a == dec_r
    nc = min(len(a.costs), len(r.costs))
    cd = max(abs(x - y) / max(1e-300, abs(x)) for x, y in zip(a.costs[:nc], r.costs[:nc]))
    print('%-28s trials %3d / %3d  steps %2d / %2d  conv %d / %d  decisions %s  costs rel %.2e  dR %.2e dt %.2e dX %.2e   %.2f ms -> %.2f ms (%.1f us per trial)'
          % (name, a.lm_trials, r.lm_trials, a.num_steps, r.num_steps, a.converged, r.converged, 'same' if same else 'DIFFER', cd,
             np.abs(Ra - Rr).max(), np.abs(ta_ - tr_).max(), np.abs(Xa - Xr).max(), ta * 1e3, tr * 1e3, tr * 1e6 / max(1, r.lm_trials)))
    if not same:
        for x, y in zip(a.trial_log, r.trial_log):
            print('   ', x, y)
    return same and cd < 1e-6


ok = True
rng = np.random.default_rng(5)
for model in (sensor_model.GaussianModel(1.), sensor_model.CauchyModel(.05), sensor_model.GaussianModel(.3)):
    for nc, nt, L in ((5, 50, 5), (10, 100, 10), (8, 200, 6), (10, 37, 4), (4, 300, 4), (14, 120, 12), (17, 256, 16), (12, 90, 5)):
        sc = synthetic_data.generate_banded_scene(nc, nt, track_len=L, seed=int(rng.integers(1 << 30)), msm_noise=.01, init_perturbation=.03,
                                                  outlier_frac=0. if isinstance(model, sensor_model.GaussianModel) else .05)
        b = Bundle.FromObservations(sc['K'], sc['R0'], sc['t0'], sc['X0'], sc['obs_cam'], sc['obs_pt'], sc['obs_z'], sensor_model=model)
        ok &= compare('%s %dx%d L=%d' % (type(model).__name__[:6], nc, nt, L), b)

g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests', 'golden', 'scene_oleg_100x1000.npz'))
b = Bundle.FromObservations(g['K'], g['R'].reshape(-1, 3, 3), g['t'], g['X'], g['obs_cam'], g['obs_pt'], g['obs_z'], sensor_model=sensor_model.GaussianModel(1.))
ok &= compare('oleg window 0', b, camera_ids=list(range(10)), track_ids=list(range(100)))
for resident in (False, True):
    BundleAdjuster.resident = resident
    window_slam.run(b, 10, num_tracks=100, max_steps=3, verbose=False)
    t0 = time.perf_counter()
    out, hist = window_slam.run(b, 10, num_tracks=100, verbose=False)
    dt = time.perf_counter() - t0
    print('window_slam.run resident=%d: %d windows: %.1f ms, final costs first / last %.9g / %.9g' % (resident, len(hist), dt * 1e3, hist[0][-1], hist[-1][-1]))

# where the time of a trial goes (option solve_trace: clock stamps at the phase boundaries)
import ctypes as C                                                          # noqa: E402
ba = BundleAdjuster(verbose=False)
ba.resident = True
ba.backend.set_option('solve_trace', '1')
ba.set_bundle(b, camera_ids=list(range(10)), track_ids=list(range(100)))
ba.optimize()
tr = np.zeros(64 * 16, np.int64)
ba.backend._check(ba.backend._lib.ba_lm_resident_trace(ba.backend._h, tr.ctypes.data_as(C.POINTER(C.c_int64))))
tr = tr.reshape(64, 16)
ba.set_bundle(b, camera_ids=list(range(10)), track_ids=list(range(100)))
lg = ba.backend.lm_resident(1, 0, False, False, 10., 1e-4, ba.SCHUR_COMPLIMENT_PINV_THRESHOLD, None)
print('one step from the start: trials %d, exit reason %d (info %d), cost0 %.9g, trial costs %s' % (lg.ntrials, lg.exit_reason, lg.exit_info, lg.cost0, [lg.trial_cost[k] for k in range(lg.ntrials)]))
S, bb, dC = ba.backend.lm_resident_debug()
ref = BundleAdjuster(verbose=False)
ref.resident = False
ref.set_bundle(b, camera_ids=list(range(10)), track_ids=list(range(100)))
be = ref.backend
be.linearize(0)
be.schur(0, 10., ref.SCHUR_COMPLIMENT_PINV_THRESHOLD)
S4, b4 = be.get_reduced()
n = 6 * be.nco
Sr = S4.transpose(0, 2, 1, 3).reshape(n, n)
br = b4.reshape(-1)
be.solve_reduced()
dCr = be.get_solution().reshape(-1)
print('first trial: S max rel %.2e (diag blocks %.2e), b %.2e, dC %.2e' % (
    np.abs(S - Sr).max() / np.abs(Sr).max(), max(np.abs(S[6 * i:6 * i + 6, 6 * i:6 * i + 6] - Sr[6 * i:6 * i + 6, 6 * i:6 * i + 6]).max() for i in range(be.nco)) / np.abs(Sr).max(),
    np.abs(bb - br).max() / np.abs(br).max(), np.abs(dC - dCr).max() / np.abs(dCr).max()))
bad = np.argwhere(np.abs(S - Sr) > 1e-9 * np.abs(Sr).max())
print('entries of S that differ: %d of %d; (row, cols) upper: %s' % (len(bad), n * n, {int(r): sorted(int(c) for rr, c in bad if rr == r and c >= r) for r in sorted(set(bad[:, 0]))}))
names = ('linearise / damp / stage', 'camera blocks + matrix cores', 'publish + wait', 'sum partials, S', 'Cholesky', 'back-substitution of the solve', 'trial set + cost', 'publish + wait + decide')
for k in range(min(ba.lm_trials, 10)):
    d = np.diff(tr[k, :9]) * 0.01
    print('trial %d (linearised %d): total %.1f us: ' % (k, tr[k, 15], (tr[k, 8] - tr[k, 0]) * 0.01) + ', '.join('%s %.1f' % (n, x) for n, x in zip(names, d)))
print('inside "sum partials, S" (two-stage exchange), us: slices added up and stored %s, published + everybody seen %s, sums fetched %s, S built %s' % tuple(np.round((tr[1:6, b] - tr[1:6, a]) * 0.01, 2) for a, b in ((3, 9), (9, 12), (12, 13), (13, 4))))
print('shader clock during the trials: %s MHz' % np.round((tr[:8, 11] - tr[:8, 10]) / ((tr[:8, 8] - tr[:8, 0]) * 0.01), 0))
for J in range(5):
    print('  step %d (wavefront 0, cycles): diagonal block %d, barrier + first panel tile + its update of the next diagonal block %d, barrier %d' % (J, tr[32 + J, 1] - tr[32 + J, 0], tr[32 + J, 2] - tr[32 + J, 1], tr[32 + J, 3] - tr[32 + J, 2]))
print('between trials: %s us' % np.round((tr[1:ba.lm_trials, 0] - tr[:ba.lm_trials - 1, 8]) * 0.01, 1)[:8])
print('ALL OK' if ok else 'MISMATCH')
