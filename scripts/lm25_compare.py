"""The 25-step LM run of the bench (config 3 / config 4 Huber, `params` start) on the GPU beside the oracle's golden walk
(tests/golden/*_lm25.npz, oracle/gen_golden_lm25.py): damping, decision and trial cost, trial by trial."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pysfm_amd import Bundle, BundleAdjuster, sensor_model, synthetic_data as sd
import json
for name in sys.argv[1:] or ['config3', 'config4_huber']:
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests', 'golden', name + '_lm25.npz'))
    a = json.loads(str(g['scene_args']))
    s = sd.generate_banded_scene(**a)
    model = sensor_model.GaussianModel(1.) if str(g['sensor_kind']) == 'gaussian' else sensor_model.HuberModel(float(g['sensor_param']))
    b = Bundle.FromObservations(s['K'], s['R0'], s['t0'], s['X0'], s['obs_cam'], s['obs_pt'], s['obs_z'], sensor_model=model)
    ba = BundleAdjuster(b, verbose=False)
    ba.optimize(max_steps=25)
    n = max(len(ba.trial_log), len(g['trial_damping']))
    print(name, 'gpu trials', len(ba.trial_log), 'oracle trials', len(g['trial_damping']), 'steps', ba.num_steps, int(g['num_steps']))
    for i in range(n):
        gt = ba.trial_log[i] if i < len(ba.trial_log) else None
        ot = (g['trial_damping'][i], g['trial_next'][i] < g['trial_cur'][i], g['trial_next'][i]) if i < len(g['trial_damping']) else None
        rel = abs(gt[2] - ot[2]) / ot[2] if gt and ot and gt[2] is not None else float('nan')
        print(i, gt, ot, '%.2e' % rel)
    e = ba.bundle.reproj_errors() if hasattr(ba.bundle, 'reproj_errors') else None
    print('final cost gpu %.6f oracle %.6f' % (ba.costs[-1], g['costs'][-1]), 'rmse oracle', float(g['rmse_final']))
