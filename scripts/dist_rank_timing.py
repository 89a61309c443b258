"""What ONE rank of a sharded config-5 run does per trial when the reduced solve is spread over the ranks (csrc/ba_dist.h):
rank r of `world` on this GPU, its shard of the scene, every stage of the trial timed with HIP events - the three sums over the
ranks are NOT done (there is one GPU here), so the separator phase works on incomplete blocks and its time is only indicative;
the shard kernels, the local elimination and the back-substitution are exact.
usage (GPU box): python scripts/dist_rank_timing.py [world] [rank] [cams] [points]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pysfm_amd import Bundle, BundleAdjuster          # noqa: E402
from pysfm_amd import synthetic_data as sd            # noqa: E402
from pysfm_amd.distributed import shard_tracks        # noqa: E402

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rank = int(sys.argv[2]) if len(sys.argv) > 2 else 3
nc = int(sys.argv[3]) if len(sys.argv) > 3 else 10000
nt = int(sys.argv[4]) if len(sys.argv) > 4 else 1000000
s = sd.generate_banded_scene(nc, nt)
b = Bundle.FromObservations(s['K'], s['R0'], s['t0'], s['X0'], s['obs_cam'], s['obs_pt'], s['obs_z'])
ba = BundleAdjuster(verbose=False)
be = ba.backend
ids = shard_tracks(b, rank, world, plan=be.dist_plan)
be.set_min_half_bandwidth(9)
ba.set_bundle(b, track_ids=ids)
info = be.dist_enable(rank, world)
print('rank %d of %d: %d tracks, %d observations; plan %s' % (rank, world, len(ids), be.nobs, info))
assert info['on']


def trial():
    be.lm_trial_begin(10., 1e-5)
    for stage in (1, 2, 3):
        be.dist_stage(stage)
    be.dist_stage(4)
    be.lm_trial_finish()


for _ in range(5):
    trial()
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 20
for _ in range(n):
    trial()
be.synchronize()
dt = (time.perf_counter() - t0) / n
be.enable_timing(True)
be.timings(reset=True)
for _ in range(10):
    trial()
tm = be.timings(reset=True)
print('one trial of this rank without the sums over the ranks: %.3f ms' % (dt * 1e3))
print({k: round(v['ms'] / 10 * 1e3, 1) for k, v in tm.items() if v['launches']}, 'us per trial')
print('exchange bytes per trial: %d + %d + %d (band [S | b]: %d)' % (8 * info['exchange1_doubles'], 8 * info['exchange2_doubles'],
                                                                      8 * info['exchange3_doubles'], 8 * (be.S_doubles + 6 * be.nco)))
