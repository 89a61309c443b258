"""Host-side time of window_slam.run by function, with wrappers instead of cProfile (whose own cost is of the order of what is
measured here).  usage (GPU box): python scripts/window_slam_host_breakdown.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pysfm_amd import Bundle, BundleAdjuster, sensor_model, window_slam      # noqa: E402
from pysfm_amd import backend as be_mod, bundle as bundle_mod                # noqa: E402

acc = {}


def wrap(cls, name, label=None):
    f = getattr(cls, name)
    label = label or '%s.%s' % (cls.__name__, name)

    def g(*a, **k):
        t0 = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            d = acc.setdefault(label, [0, 0.])
            d[0] += 1
            d[1] += time.perf_counter() - t0
    setattr(cls, name, g)


wrap(BundleAdjuster, 'set_bundle')
wrap(BundleAdjuster, 'optimize')
wrap(BundleAdjuster, '_upload')
wrap(BundleAdjuster, '_resident_steps')
wrap(bundle_mod.Bundle, 'select_observations')
wrap(bundle_mod.Bundle, 'clone_params')
wrap(bundle_mod.Bundle, 'check_consistency')
wrap(be_mod.HipBackend, 'set_problem')
wrap(be_mod.HipBackend, 'set_params')
wrap(be_mod.HipBackend, 'get_params')
wrap(be_mod.HipBackend, 'lm_resident')
bprop = BundleAdjuster.bundle
if isinstance(bprop, property):
    def timed_bundle(self):
        t0 = time.perf_counter()
        try:
            return bprop.fget(self)
        finally:
            d = acc.setdefault('BundleAdjuster.bundle', [0, 0.])
            d[0] += 1
            d[1] += time.perf_counter() - t0
    BundleAdjuster.bundle = property(timed_bundle, bprop.fset)

g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests', 'golden', 'scene_oleg_100x1000.npz'))
b = Bundle.FromObservations(g['K'], g['R'].reshape(-1, 3, 3), g['t'], g['X'], g['obs_cam'], g['obs_pt'], g['obs_z'], sensor_model=sensor_model.GaussianModel(1.))
window_slam.run(b, 10, num_tracks=100, max_steps=3, verbose=False)
best = None
for _ in range(5):
    acc.clear()
    t0 = time.perf_counter()
    out, hist = window_slam.run(b, 10, num_tracks=100, verbose=False)
    dt = time.perf_counter() - t0
    if best is None or dt < best[0]:
        best = (dt, {k: tuple(v) for k, v in acc.items()})
dt, a = best
print('window_slam.run: %d windows, %.1f ms' % (len(hist), dt * 1e3))
for k, (n, t) in sorted(a.items(), key=lambda kv: -kv[1][1]):
    print('  %-34s %4d calls  %7.2f ms  %6.1f us per call' % (k, n, t * 1e3, t * 1e6 / n))
