"""What the device's LU form of band + border costs where it is needed (a system that is not positive definite: negative damping)
at config-3 size with ten loop-closure tracks: the solve round 5 did on the host (288 MB over PCIe, numpy.linalg.solve)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import with_loop_closures
from pysfm_amd import synthetic_data as sd
from pysfm_amd.backend import HipBackend
from pysfm_amd._capi import PARAMS_CUR, SENSOR_GAUSS
s = with_loop_closures(sd.generate_banded_scene(1000, 100000, init_mode='params'), 10)
nt = len(s['X0'])
be = HipBackend(0)
be.set_problem(1000, nt, s['obs_cam'], s['obs_pt'], s['obs_z'], s['K'], np.arange(1000, dtype=np.int32) - 1, np.ones(nt, np.uint8))
be.set_sensor(SENSOR_GAUSS, [1., 0., 0., 1.])
be.set_params(PARAMS_CUR, s['R0'], s['t0'], s['X0'])
print('border cameras', be.problem_info()['border_cameras'], 'half-bandwidth', be.half_bandwidth)
be.linearize(PARAMS_CUR)
be.schur(PARAMS_CUR, -.6, 1e-5)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.time()
    be.solve_reduced(None)
    torch.cuda.synchronize(); dt = time.time() - t0
    print('solve_reduced of the indefinite bordered system: %.1f ms, solver %s' % (1e3 * dt, be.last_solve_kind))
x = be.get_solution().reshape(-1)
t0 = time.time(); S, b = be.get_reduced(); t1 = time.time()
A = S.transpose(0, 2, 1, 3).reshape(len(x), len(x))
ref = np.linalg.solve(A, b.reshape(-1)); t2 = time.time()
print('host: get_reduced %.0f ms + numpy.linalg.solve %.0f ms; max |x_dev - x_ref| / max |x_ref| = %.2e' % (1e3 * (t1 - t0), 1e3 * (t2 - t1), np.max(np.abs(x - ref)) / np.max(np.abs(ref))))
