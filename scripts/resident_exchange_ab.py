"""Exchange 1 of the resident loop (csrc/ba_resident.h): every workgroup reading every record against the two-stage form
(slices added up by one workgroup each, one record of sums fetched by all).  Same sums in the same order: the two runs must
end on the same bits.  usage (GPU box): python scripts/resident_exchange_ab.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pysfm_amd import Bundle, BundleAdjuster, sensor_model, synthetic_data      # noqa: E402


def run(bundle, scatter_min, reps=7):
    ba = BundleAdjuster(verbose=False)
    ba.resident = True
    ba.backend.set_option('resident_scatter_min', str(scatter_min))
    best = 1e9
    for _ in range(reps):
        ba.set_bundle(bundle)
        t0 = time.perf_counter()
        ba.optimize()
        best = min(best, time.perf_counter() - t0)
    b = ba.bundle
    return ba, best, np.array([c.R for c in b.cameras]), np.array([c.t for c in b.cameras]), np.asarray(b.reconstruction)


ok = True
for nc, nt, L in ((10, 32, 10), (10, 48, 10), (10, 64, 10), (10, 100, 10), (5, 50, 5), (10, 256, 10), (17, 256, 16), (10, 1024, 4), (17, 1024, 8)):
    sc = synthetic_data.generate_banded_scene(nc, nt, track_len=L, seed=11 + nc + nt, msm_noise=.01, init_perturbation=.03)
    b = Bundle.FromObservations(sc['K'], sc['R0'], sc['t0'], sc['X0'], sc['obs_cam'], sc['obs_pt'], sc['obs_z'], sensor_model=sensor_model.GaussianModel(1.))
    a, ta, Ra, tta, Xa = run(b, 1000)
    r, tr, Rr, ttr, Xr = run(b, 2)
    same = a.trial_log == r.trial_log and np.array_equal(Ra, Rr) and np.array_equal(tta, ttr) and np.array_equal(Xa, Xr)
    ok &= same
    print('%2d x %4d L=%2d (%2d workgroups): %3d trials, all read all %.1f us per trial, two stages %.1f us per trial, results %s'
          % (nc, nt, L, (nt + 15) // 16, a.lm_trials, ta * 1e6 / a.lm_trials, tr * 1e6 / r.lm_trials, 'identical' if same else 'DIFFER'))
print('ALL OK' if ok else 'MISMATCH')
