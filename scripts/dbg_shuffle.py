import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..')); sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tests'))
import numpy as np
from oracle import ba_oracle as O
from pysfm_amd.backend import HipBackend
from pysfm_amd import synthetic_data as sd
be = HipBackend(0)
nc, nt, L = 40, 1500, 10
s0 = sd.generate_banded_scene(nc, nt, track_len=L, outlier_frac=.03)
rs = np.random.RandomState(7)
new_id = rs.permutation(nt); X0 = np.empty_like(s0['X0']); X0[new_id] = s0['X0']; o = rs.permutation(len(s0['obs_cam']))
s1 = dict(s0); s1.update(X0=X0, obs_cam=s0['obs_cam'][o], obs_pt=new_id[s0['obs_pt'][o]].astype(np.int32), obs_z=s0['obs_z'][o])
for frozen in (False, True):
    cam_opt_pos = np.arange(nc, dtype=np.int32) - 1
    if frozen:
        cam_opt_pos[17] = -1; cam_opt_pos[18:] -= 1
    pt_opt = np.ones(nt, np.uint8)
    for name, s in (('sorted', s0), ('shuffled', s1)):
        for sort in (1, 0):
            if name == 'shuffled' and sort == 0: continue
            be.set_option('sort_points', sort)
            a = (s['K'], s['R0'], s['t0'], s['X0'], s['obs_cam'], s['obs_pt'], s['obs_z'])
            be.set_problem(nc, nt, s['obs_cam'], s['obs_pt'], s['obs_z'], s['K'], cam_opt_pos, pt_opt)
            be.set_sensor(0, np.eye(2).reshape(4)); be.set_params(0, s['R0'], s['t0'], s['X0'])
            mu, su, parts = O.compute_update(O.Sensor.gaussian(1.), *a, cam_opt_pos, pt_opt, damping=3., return_parts=True)
            for kern in ('pairs', 'groups', 'mfma1', 'mfma'):
                be.set_option('schur', kern)
                be.linearize(0); be.schur(0, 3., 1e-5)
                S, b = be.get_reduced()
                eS = np.abs(S - parts['S']).max() / np.abs(parts['S']).max(); eb = np.abs(b - parts['b']).max() / np.abs(parts['b']).max()
                bad = np.argwhere(np.abs(S - parts['S']).max(axis=(2, 3)) > 1e-8 * np.abs(parts['S']).max())
                print('frozen', frozen, name, 'sort', sort, kern, 'errS %.2e errb %.2e' % (eS, eb), 'bad blocks', len(bad), bad[:6].tolist(), be.problem_info())
            be.set_option('schur', 'auto')
