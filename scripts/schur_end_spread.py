"""When do the workgroups of the Schur reduction end (library built with `make PROFILE=1`)?  The rows of [S | b] are final when
the last workgroup that adds to them has ended; a consumer (the reduced solve) could only start early on rows whose
workgroups end early.  Prints the spread of the end times at config 3.
usage (GPU box, PROFILE library in place of pysfm_amd/libpysfm_ba.so): python scripts/schur_end_spread.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pysfm_amd import Bundle, BundleAdjuster                           # noqa: E402
from pysfm_amd import synthetic_data as sd                             # noqa: E402
s = sd.generate_banded_scene(1000, 100000)
b = Bundle.FromObservations(s['K'], s['R0'], s['t0'], s['X0'], s['obs_cam'], s['obs_pt'], s['obs_z'])
ba = BundleAdjuster(b, verbose=False)
be = ba.backend
for _ in range(3):
    be.lm_trial(10., 1e-5, None)
be.set_option('solve_trace', 1)
for _ in range(3):
    be.linearize(0)
    be.schur(0, 10., 1e-5)          # (stepwise: the camera blocks come from k_camera_blocks here, the reduction is the same kernel)
be.synchronize()
