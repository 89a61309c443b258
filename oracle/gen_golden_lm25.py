#!/usr/bin/env python3
"""Golden LM runs of the CPU oracle at BASELINE's full sizes: the 25-step optimize() the bench line reports, walked by
oracle/ba_oracle.py (the pinned NumPy restatement of bundle_adjuster.py:117-162) in the build container.

    python oracle/gen_golden_lm25.py            # writes tests/golden/config3_lm25.npz, config4_huber_lm25.npz
    python oracle/gen_golden_lm25.py config3    # one of them

Stored per run: every trial (damping, cost of the current set, cost of the trial set), the accepted costs, steps,
converged, the final damping, the raw reprojection RMSE of the start and of the end ("final reprojection RMSE
... next to the CPU restatement's value on the same scene and seed"), and a few rows of the final parameters.  Inputs are not
stored: the scene is pysfm_amd.synthetic_data.generate_banded_scene with the arguments recorded in `scene_args` (pure NumPy,
seeded).  Test infrastructure: only tests/ and bench.py read these files.  ~10 minutes per run on 8 cores (a dense
5994 x 5994 LU per trial, as the reference does it)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ba_oracle as O                      # noqa: E402
from pysfm_amd import synthetic_data as sd             # noqa: E402

RUNS = {
    'config3': dict(scene=dict(ncams=1000, npts=100000, track_len=10, outlier_frac=0., init_mode='params'), sensor=('gaussian', 1.)),
    'config4_huber': dict(scene=dict(ncams=1000, npts=100000, track_len=10, outlier_frac=.1, init_mode='params'), sensor=('huber', .06)),
    'config3_pose': dict(scene=dict(ncams=1000, npts=100000, track_len=10, outlier_frac=0., init_mode='pose'), sensor=('gaussian', 1.)),
}


def rmse(K, R, t, X, obs):
    e = O.reproj_error(K, R, t, X, *obs)
    return float(np.sqrt(np.sum(e * e) / len(e)))


def run(name, out_dir):
    spec = RUNS[name]
    s = sd.generate_banded_scene(**spec['scene'])
    nc, nt = spec['scene']['ncams'], spec['scene']['npts']
    sen = O.Sensor.gaussian(spec['sensor'][1]) if spec['sensor'][0] == 'gaussian' else O.Sensor.huber(spec['sensor'][1])
    flags = (np.arange(nc, dtype=np.int32) - 1, np.ones(nt, bool))
    obs = (s['obs_cam'], s['obs_pt'], s['obs_z'])
    trace = []
    t0 = time.time()
    ref = O.lm_optimize(sen, s['K'], s['R0'], s['t0'], s['X0'], *obs, *flags, max_steps=25, init_damping=10., trace=trace)
    wall = time.time() - t0
    d = dict(scene_args=json.dumps(spec['scene'], sort_keys=True), sensor_kind=spec['sensor'][0], sensor_param=spec['sensor'][1],
             trial_damping=np.array([tr['damping'] for tr in trace]), trial_cur=np.array([tr['cur'] for tr in trace]),
             trial_next=np.array([tr['next'] for tr in trace]), trial_step=np.array([tr['step'] for tr in trace]),
             costs=np.array(ref['costs']), num_steps=ref['num_steps'], converged=bool(ref['converged']), damping=ref['damping'],
             rmse_initial=rmse(s['K'], s['R0'], s['t0'], s['X0'], obs), rmse_final=rmse(s['K'], ref['R'], ref['t'], ref['X'], obs),
             t_final_head=ref['t'][:32], X_final_head=ref['X'][:64], oracle_wall_s=wall)
    path = os.path.join(out_dir, name + '_lm25.npz')
    np.savez_compressed(path, **d)
    print('%s: %d trials, %d steps, converged %s, cost %.6f -> %.6f, rmse %.6f -> %.6f, %.0f s -> %s'
          % (name, len(trace), ref['num_steps'], ref['converged'], ref['costs'][0], ref['costs'][-1], d['rmse_initial'], d['rmse_final'], wall, path), flush=True)


if __name__ == '__main__':
    out_dir = os.path.join(ROOT, 'tests', 'golden')
    for name in (sys.argv[1:] or ['config3', 'config4_huber']):
        run(name, out_dir)
