"""What would keeping the NEXT damping's trial in flight buy (round-4 review, item 3)?  The upper bound is what the GPU does with
two complete LM trials of the same scene side by side: two handles (own streams, own workspaces), one thread each, against the
same trials one after the other on one handle.  If two concurrent trials take about as long as one, speculation is nearly free;
if they take about twice as long, the trial already fills the machine and a speculative trial only delays the real one."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pysfm_amd import Bundle, BundleAdjuster, sensor_model, synthetic_data as sd
from pysfm_amd._capi import PARAMS_CUR

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
nc, nt = (1000, 100000) if cfg == 3 else (10000, 1000000)
s = sd.generate_banded_scene(nc, nt, init_mode='params' if cfg == 3 else 'pose')
b = Bundle.FromObservations(s['K'], s['R0'], s['t0'], s['X0'], s['obs_cam'], s['obs_pt'], s['obs_z'], sensor_model=sensor_model.GaussianModel(1.))
bas = [BundleAdjuster(b, verbose=False) for _ in range(2)]
for ba in bas:
    ba.backend.set_option('reuse_linearization', 0)
    for _ in range(5):
        ba.backend.lm_trial(10., 1e-5, None)
N = 200

def run(ba, damping, n=N):
    for _ in range(n):
        ba.backend.lm_trial(damping, 1e-5, None)

torch.cuda.synchronize()
t0 = time.time(); run(bas[0], 1.); torch.cuda.synchronize(); t_one = (time.time() - t0) / N
t0 = time.time()
th = [threading.Thread(target=run, args=(ba, d)) for ba, d in zip(bas, (1., 10.))]
[t.start() for t in th]; [t.join() for t in th]
torch.cuda.synchronize(); t_two = (time.time() - t0) / N
print('config %d: one trial %.1f us; two trials side by side (two handles, two streams, two threads) %.1f us per pair = %.2f x one trial'
      % (cfg, 1e6 * t_one, 1e6 * t_two, t_two / t_one))
# the LM walk's arithmetic: a launch of (lambda, 10 lambda) decides two trials when lambda is rejected, one when it is accepted
for name, log in (('config 3 bench walk', 'AAAARAARARARRRARAAAARRRRAARRAARARRAARARAARARRAA'),):
    launches, i = 0, 0
    while i < len(log):
        launches += 1
        i += 2 if (log[i] == 'R' and i + 1 < len(log)) else 1
    print('%s: %d trials -> %d paired launches: %.2f of the sequential time at the measured pair cost' % (name, len(log), launches, launches * t_two / (len(log) * t_one)))
