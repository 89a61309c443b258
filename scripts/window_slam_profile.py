"""Where does the wall-clock of the sliding-window caller go (window_slam.py:17-48: ~90 windows of 10 cameras x 100
tracks, one after the other)?  cProfile of window_slam.run on the reference's own scene shape (100 cameras x 1000 tracks).
usage (GPU box): python scripts/window_slam_profile.py [window] [tracks] [--profile]"""
import cProfile
import os
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pysfm_amd import Bundle, sensor_model, window_slam      # noqa: E402

g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests', 'golden', 'scene_oleg_100x1000.npz'))
window = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 10
ntr = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 100
model = sensor_model.GaussianModel(1.)
b = Bundle.FromObservations(g['K'], g['R'].reshape(-1, 3, 3), g['t'], g['X'], g['obs_cam'], g['obs_pt'], g['obs_z'], sensor_model=model)
window_slam.run(b, window, num_tracks=ntr, max_steps=3, verbose=False)           # warm-up: code objects, handle
t0 = time.perf_counter()
out, hist = window_slam.run(b, window, num_tracks=ntr, verbose=False)
dt = time.perf_counter() - t0
trials = sum(len(h) for h in hist)
print('window_slam.run: %d windows of %d cameras x %d tracks: %.1f ms (%.2f ms per window, %d accepted steps)' % (len(hist), window, ntr, dt * 1e3, dt * 1e3 / len(hist), trials - len(hist)))
print('final costs of the first / last window: %.6g / %.6g' % (hist[0][-1], hist[-1][-1]))
if '--profile' in sys.argv:
    pr = cProfile.Profile()
    pr.enable()
    window_slam.run(b, window, num_tracks=ntr, verbose=False)
    pr.disable()
    pstats.Stats(pr).sort_stats('cumulative').print_stats(28)
