"""Host-side setup cost: Bundle.FromObservations + BundleAdjuster.set_bundle (sorting, CSR, Schur work lists, upload)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pysfm_amd import Bundle, BundleAdjuster, sensor_model
from pysfm_amd import synthetic_data as sd
nc, nt = int(sys.argv[1]) if len(sys.argv) > 1 else 1000, int(sys.argv[2]) if len(sys.argv) > 2 else 100000
t0 = time.perf_counter(); s = sd.generate_banded_scene(nc, nt); t1 = time.perf_counter()
b = Bundle.FromObservations(s['K'], s['R0'], s['t0'], s['X0'], s['obs_cam'], s['obs_pt'], s['obs_z'], sensor_model=sensor_model.GaussianModel(1.))
t2 = time.perf_counter()
ba = BundleAdjuster(verbose=False); t3 = time.perf_counter()
ba.set_bundle(b); torch.cuda.synchronize(); t4 = time.perf_counter()
ba.set_bundle(b); torch.cuda.synchronize(); t5 = time.perf_counter()
print('generate %.3f s, FromObservations %.3f s, BundleAdjuster() %.3f s, set_bundle %.3f s (again: %.3f s)' % (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); ba.set_bundle(b); pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(12)
