"""Is the resident loop deterministic to the bit - alone, and with another handle setting up a problem while it runs?
usage (GPU box): python scripts/resident_determinism.py [scatter_min]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pysfm_amd import Bundle, BundleAdjuster, sensor_model      # noqa: E402

g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests', 'golden', 'scene_oleg_100x1000.npz'))
b = Bundle.FromObservations(g['K'], g['R'].reshape(-1, 3, 3), g['t'], g['X'], g['obs_cam'], g['obs_pt'], g['obs_z'], sensor_model=sensor_model.GaussianModel(1.))
ids = dict(camera_ids=list(range(10)), track_ids=list(range(100)))
a = BundleAdjuster(verbose=False)
o = BundleAdjuster(verbose=False)
if len(sys.argv) > 1:
    a.backend.set_option('resident_scatter_min', sys.argv[1])
for mode in ('alone', 'other handle busy'):
    seen = {}
    for it in range(300):
        a.set_bundle(b, **ids)
        a.optimize_begin(max_steps=5)
        if mode != 'alone':
            o.set_bundle(b, camera_ids=list(range(1, 11)), track_ids=list(range(100)), upload=False)
        a.optimize_end()
        key = tuple(a.costs)
        seen[key] = seen.get(key, 0) + 1
    print('%s: %d distinct cost histories in 300 runs: %s' % (mode, len(seen), sorted(seen.values(), reverse=True)))
    if len(seen) > 1:
        ks = list(seen)
        for k in ks[:3]:
            print('   ', ['%.17g' % c for c in k])
