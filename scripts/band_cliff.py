"""Reduced-solve time against the block half-bandwidth (track length L = hb + 1) at 1000 cameras:
cyclic reduction (hb <= 11), single-workgroup band Cholesky (hb <= 21), dense fallback beyond."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pysfm_amd import Bundle, BundleAdjuster, sensor_model
from pysfm_amd import synthetic_data as sd
import os as _os
LS = tuple(int(v) for v in _os.environ['BAND_L'].split(',')) if 'BAND_L' in _os.environ else (6, 10, 12, 13, 16, 22, 23)
for L in LS:
    s = sd.generate_banded_scene(1000, 20000, track_len=L)
    b = Bundle.FromObservations(s['K'], s['R0'], s['t0'], s['X0'], s['obs_cam'], s['obs_pt'], s['obs_z'], sensor_model=sensor_model.GaussianModel(1.))
    ba = BundleAdjuster(verbose=False); ba.set_bundle(b); be = ba.backend
    be.linearize(0); be.schur(0, 10., 1e-5)
    be.solve_reduced(None)
    t0 = time.perf_counter()
    for _ in range(5): be.solve_reduced(None)
    dt = (time.perf_counter() - t0) / 5
    print('L=%2d hb=%2d: solve %.3f ms (%s)' % (L, be.half_bandwidth, dt * 1e3, be.last_solve_path))
