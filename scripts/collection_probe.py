"""An unordered photo collection (synthetic_data.generate_collection_scene: every camera shares tracks with cameras drawn at random
from all the others) through the library: what ba_set_problem makes of it, which kernels the trial takes, what a trial costs.
    python scripts/collection_probe.py [ncams npts [option=value ...]]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pysfm_amd import Bundle, BundleAdjuster, synthetic_data as sd
from pysfm_amd._capi import PARAMS_CUR

sizes = [(int(sys.argv[1]), int(sys.argv[2]))] if len(sys.argv) > 2 else [(300, 6000), (2000, 80000), (5000, 200000)]
opts = [a for a in sys.argv[3:] if '=' in a]
for nc, nt in sizes:
    s = sd.generate_collection_scene(nc, nt, partners=int(os.environ.get('PARTNERS', 8)), track_len=int(os.environ.get('TRACK_LEN', 3)))
    b = Bundle.FromObservations(s['K'], s['R0'], s['t0'], s['X0'], s['obs_cam'], s['obs_pt'], s['obs_z'])
    ba = BundleAdjuster(verbose=False)
    for kv in opts:
        ba.backend.set_option(*kv.split('='))
    t0 = time.time(); ba.set_bundle(b); t1 = time.time(); ba.set_bundle(b); torch.cuda.synchronize(); t2 = time.time()
    be = ba.backend
    info = be.problem_info()
    print('%d cameras / %d points / %d observations: set_bundle %.1f ms (first %.1f), half-bandwidth %d (caller\'s %d), schur kernel %d, border %d, S %.1f MB'
          % (nc, nt, be.nobs, 1e3 * (t2 - t1), 1e3 * (t1 - t0), be.half_bandwidth, info['caller_half_bandwidth'], info['schur_kernel'], info['border_cameras'], 8e-6 * be.S_doubles), flush=True)
    cur = ba._cost(PARAMS_CUR)
    for damping in (10., 10., 1., .1):
        torch.cuda.synchronize(); t0 = time.time()
        acc, nxt = ba.trial(damping, None, cur)
        torch.cuda.synchronize(); dt = time.time() - t0
        print('   trial at damping %g: %.2f ms, %s, cost %.6f -> %.6f, solver %s / %s %s' % (damping, 1e3 * dt, acc, cur, nxt, be.last_solve_kind, be.last_solve_path,
                                                                                          be.pcg_info() if be.last_solve_kind == 'pcg' else ''), flush=True)
    be.enable_timing(True); be.timings(reset=True)
    for _ in range(3):
        ba.trial(10., None, cur)
    tm = be.timings(reset=True); be.enable_timing(False)
    print('   per kernel (us, launches per trial):', {k: (round(1e3 * v['ms'] / 3, 1), v['launches'] / 3) for k, v in tm.items() if v['launches']}, flush=True)
    ba.backend.close()
