"""How much of a trial's wall-clock is the host: ba.trial (Python accept / reject bookkeeping) against HipBackend.lm_trial
(the ctypes call alone) against the sum of the kernels (HIP events).  usage (GPU box): python scripts/host_gap.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pysfm_amd import Bundle, BundleAdjuster          # noqa: E402
from pysfm_amd import synthetic_data as sd            # noqa: E402

s = sd.generate_banded_scene(1000, 100000)
b = Bundle.FromObservations(s['K'], s['R0'], s['t0'], s['X0'], s['obs_cam'], s['obs_pt'], s['obs_z'])
ba = BundleAdjuster(verbose=False)
ba.set_bundle(b)
be = ba.backend
for _ in range(30):
    be.lm_trial(10., 1e-5, None)
n = 400
for name, f in (('HipBackend.lm_trial (rejected every time: damping 1e9)', lambda: be.lm_trial(1e9, 1e-5, None)),
                ('HipBackend.lm_trial (damping 10)', lambda: be.lm_trial(10., 1e-5, None)),
                ('BundleAdjuster.trial (damping 10, never accepted: cur_cost 0)', lambda: ba.trial(10., None, 0.))):
    be.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    be.synchronize()
    print('%-70s %.2f us per call' % (name, (time.perf_counter() - t0) / n * 1e6))
be.enable_timing(True)
be.timings(reset=True)
for _ in range(50):
    be.lm_trial(10., 1e-5, None)
tm = be.timings(reset=True)
print('kernels (HIP events, each bracket adds ~2.5 us): %.1f us per trial' % (sum(v['ms'] for v in tm.values()) / 50 * 1e3))
