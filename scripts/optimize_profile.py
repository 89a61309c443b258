"""Where does the wall-clock of BundleAdjuster.optimize go beyond the kernels (config 3)?  cProfile of the LM loop.
usage (GPU box): python scripts/optimize_profile.py [cams] [points]"""
import cProfile
import os
import pstats
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                                            # noqa: E402
from pysfm_amd import Bundle, BundleAdjuster                           # noqa: E402
from pysfm_amd import synthetic_data as sd                             # noqa: E402
nc, nt = int(sys.argv[1]) if len(sys.argv) > 1 else 1000, int(sys.argv[2]) if len(sys.argv) > 2 else 100000
s = sd.generate_banded_scene(nc, nt, init_mode='params')
b = Bundle.FromObservations(s['K'], s['R0'], s['t0'], s['X0'], s['obs_cam'], s['obs_pt'], s['obs_z'])
ba = BundleAdjuster(b, verbose=False)
ba.optimize(max_steps=25)
for rep in range(3):
    ba.set_bundle(b)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ba.optimize(max_steps=25)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print('optimize(25 steps): %.2f ms, %d trials = %.1f us per trial' % (dt * 1e3, ba.lm_trials, dt * 1e6 / ba.lm_trials))
ba.set_bundle(b)
pr = cProfile.Profile()
pr.enable()
ba.optimize(max_steps=25)
pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(14)
t0 = time.perf_counter(); out = ba.bundle; t1 = time.perf_counter()
print('ba.bundle (device -> host, clone): %.2f ms' % ((t1 - t0) * 1e3))
