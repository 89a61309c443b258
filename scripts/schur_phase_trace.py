"""Cycle counts of the producer / consumer Schur reduction (library built with `make PROFILE=1`): one LM trial at
config 3; the kernel prints the numbers of workgroup 100.  usage (GPU box): python scripts/schur_phase_trace.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pysfm_amd import Bundle, BundleAdjuster                           # noqa: E402
from pysfm_amd import synthetic_data as sd                             # noqa: E402
s = sd.generate_banded_scene(1000, 100000)
b = Bundle.FromObservations(s['K'], s['R0'], s['t0'], s['X0'], s['obs_cam'], s['obs_pt'], s['obs_z'])
ba = BundleAdjuster(b, verbose=False)
for _ in range(2):
    print(ba.backend.lm_trial(10., 1e-5, None))
ba.backend.synchronize()
