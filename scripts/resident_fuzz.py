"""Random small scenes through both LM loops (the resident launch and the Python loop over ba_lm_trial): how often do the walks
differ, and by how much?  usage (GPU box): python scripts/resident_fuzz.py [scenes]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pysfm_amd import Bundle, BundleAdjuster, sensor_model, synthetic_data as sd      # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rs = np.random.RandomState(12345)
differ = worst = 0
for it in range(n):
    nc = int(rs.randint(2, 18))
    L = int(rs.randint(2, min(nc, 16) + 1))
    nt = int(rs.randint(1, 400))
    kind = int(rs.randint(3))
    model = (sensor_model.GaussianModel(1.), sensor_model.CauchyModel(.05), sensor_model.HuberModel(.06))[kind]
    s = sd.generate_banded_scene(nc, nt, track_len=L, seed=int(rs.randint(1 << 30)), msm_noise=.01, init_perturbation=float(rs.choice([.003, .03])),
                                 outlier_frac=0. if kind == 0 else .05)
    cam, pt, z = s['obs_cam'], s['obs_pt'], s['obs_z']
    if rs.rand() < .5:
        keep = rs.rand(len(cam)) > .3
        first = np.concatenate(([True], pt[1:] != pt[:-1]))
        keep |= first | np.concatenate(([False], first[:-1]))
        cam, pt, z = cam[keep], pt[keep], z[keep]
    if rs.rand() < .5:
        o = rs.permutation(len(cam))
        cam, pt, z = cam[o], pt[o], z[o]
    b = Bundle.FromObservations(s['K'], s['R0'], s['t0'], s['X0'], cam, pt, z, sensor_model=model)
    out = []
    for resident in (False, True):
        ba = BundleAdjuster(verbose=False)
        ba.resident = resident
        ba.set_bundle(b)
        if resident and not ba._resident_applies(None):
            out.append(None)
            break
        ba.optimize(max_steps=int(rs.randint(2, 12)) if False else 10)
        out.append(ba)
        ba.backend.close()
    if out[-1] is None:
        continue
    a, r = out
    same = [(d, o) for d, o, _ in a.trial_log] == [(d, o) for d, o, _ in r.trial_log]
    k = min(len(a.costs), len(r.costs))
    rel = max(abs(x - y) / max(abs(x), 1e-300) for x, y in zip(a.costs[:k], r.costs[:k]))
    if not same:
        differ += 1
        print('scene %d (%d cameras, %d tracks, L %d, model %d): decisions differ after %d common trials; costs rel %.2e' % (
            it, nc, nt, L, kind, next((i for i, (x, y) in enumerate(zip(a.trial_log, r.trial_log)) if x[:2] != y[:2]), -1), rel), flush=True)
    else:
        worst = max(worst, rel)
print('%d scenes: %d walks differ; worst relative cost difference among the equal walks %.2e' % (n, differ, worst))
