"""Per-phase shader-cycle counts of k_bcr_eliminate (library built with `make PROFILE=1`).
usage (GPU box): python scripts/bcr_phase_trace.py [cams] [points] [solver: bcr | bcr1] [track length]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pysfm_amd import Bundle, BundleAdjuster, sensor_model          # noqa: E402
from pysfm_amd import synthetic_data as sd                          # noqa: E402

nc = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
nt = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
s = sd.generate_banded_scene(nc, nt, track_len=int(sys.argv[4]) if len(sys.argv) > 4 else 10)
b = Bundle.FromObservations(s['K'], s['R0'], s['t0'], s['X0'], s['obs_cam'], s['obs_pt'], s['obs_z'],
                            sensor_model=sensor_model.GaussianModel(1.))
ba = BundleAdjuster(verbose=False)
ba.set_bundle(b)
be = ba.backend
be.set_option('solve_trace', 1)
be.set_option('solver', sys.argv[3] if len(sys.argv) > 3 else 'bcr')
be.linearize(0)
be.schur(0, 10., 1e-5)
for _ in range(3):
    be.solve_reduced(None)
print('solve path', be.last_solve_path, 'hb', be.half_bandwidth, '|dC|', float(np.linalg.norm(be.get_solution())))
