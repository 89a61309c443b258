"""Set-up cost: Bundle.FromObservations + BundleAdjuster.set_bundle (id bookkeeping on the host; validation, internal order,
CSR, tables on the device; work lists from per-point summaries on the host) for a scene in generator order and shuffled."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pysfm_amd import Bundle, BundleAdjuster, sensor_model
from pysfm_amd import synthetic_data as sd
nc, nt = int(sys.argv[1]) if len(sys.argv) > 1 else 1000, int(sys.argv[2]) if len(sys.argv) > 2 else 100000
t0 = time.perf_counter(); s = sd.generate_banded_scene(nc, nt); t1 = time.perf_counter()
print('generate %.3f s' % (t1 - t0))
rs = np.random.RandomState(3)
new_id = rs.permutation(nt); o = rs.permutation(len(s['obs_cam']))
X1 = np.empty_like(s['X0']); X1[new_id] = s['X0']
variants = {'generator order': (s['obs_cam'], s['obs_pt'], s['obs_z'], s['X0']),
            'tracks and observations shuffled': (s['obs_cam'][o], new_id[s['obs_pt'][o]], s['obs_z'][o], X1)}
ba = BundleAdjuster(verbose=False)
for name, (oc, op, oz, X0) in variants.items():
    t1 = time.perf_counter()
    b = Bundle.FromObservations(s['K'], s['R0'], s['t0'], X0, oc, op, oz, sensor_model=sensor_model.GaussianModel(1.))
    t2 = time.perf_counter()
    times = []
    for rep in range(4):
        t3 = time.perf_counter(); ba.set_bundle(b); torch.cuda.synchronize(); times.append(time.perf_counter() - t3)
    print('%-34s FromObservations %.3f s, set_bundle %s s, problem %s' % (name, t2 - t1, ' '.join('%.4f' % t for t in times), ba.backend.problem_info()))
    t3 = time.perf_counter(); ba.optimize(max_steps=25); torch.cuda.synchronize(); t4 = time.perf_counter()
    print('    optimize(25 steps): %.4f s, %d trials, cost %.6g -> %.6g' % (t4 - t3, ba.lm_trials, ba.costs[0], ba.costs[-1]))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); ba.set_bundle(b); pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(14)
# the C side alone, by stage (host clock around ba_set_problem)
be = ba.backend
oc, op, oz = b.select_observations(range(nc), range(nt))
cp = np.arange(nc, dtype=np.int32) - 1; po = np.ones(nt, np.uint8)
for rep in range(3):
    t5 = time.perf_counter(); be.set_problem(nc, nt, oc, op, oz, s['K'], cp, po); t6 = time.perf_counter()
    print('ba_set_problem (+ layout query, buffer binding): %.4f s' % (t6 - t5))
