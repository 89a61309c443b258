#!/usr/bin/env python3
"""Golden vectors FROM THE REFERENCE for a scene whose cameras come in no particular order and whose tracks include loop closures.

TEST INFRASTRUCTURE ONLY - runs in the build container (it imports /root/reference through oracle/gen_golden.py's out-of-tree
lib2to3 recipe), never on the GPU box.  The reference's reduced system is dense (bundle_adjuster.py:259-312): camera order and
long-range tracks do not exist for it.  The library orders the cameras itself and turns the far ends of loop closures into a
border of the band (csrc/ba_order.hip, ba_border.h): this fixture pins that path to the reference's own numbers, not only to the
oracle's.  Scene: 60 cameras in a row (tracks of 5 consecutive cameras, 420 points), 4 points seen by cameras {i, i+1, i+30, i+31},
the cameras renumbered at random; unit Gaussian sensor model.

    python oracle/gen_golden_layout.py [--ref /root/reference] [--out tests/golden]

writes tests/golden/scene_loop_closure_60x424.npz: inputs, S / b / dC / dP / cost at damping 2, compute_update(2.), and the
trials of optimize(max_steps=4).
"""
import argparse
import importlib
import os
import shutil
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import gen_golden as gg                                    # noqa: E402
from pysfm_amd import synthetic_data as sd                 # noqa: E402  (pure-numpy scene builders)


def build_scene():
    nc, nt, L = 60, 420, 5
    s = sd.generate_banded_scene(nc, nt, track_len=L, seed=321, init_seed=99)
    s = sd.add_loop_closure_tracks(s, [(3, 33), (11, 41), (17, 47), (24, 54)], width=2, seed=5)
    perm = np.random.RandomState(8).permutation(nc)
    out = dict(s)
    for k in ('R0', 't0', 'R', 't'):
        a = np.empty_like(s[k])
        a[perm] = s[k]
        out[k] = a
    out['obs_cam'] = perm[s['obs_cam']].astype(np.int32)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--ref', default='/root/reference')
    ap.add_argument('--out', default=os.path.join(HERE, '..', 'tests', 'golden'))
    args = ap.parse_args()
    s = build_scene()
    nc, nt = len(s['R0']), len(s['X0'])
    msm = np.zeros((nc, nt, 2))
    mask = np.zeros((nc, nt), bool)
    msm[s['obs_cam'], s['obs_pt']] = s['obs_z']
    mask[s['obs_cam'], s['obs_pt']] = True
    tmp = gg.import_reference(args.ref)
    try:
        with gg.quiet():
            ref = {n: importlib.import_module(n) for n in ('bundle', 'bundle_adjuster', 'sensor_model', 'lie', 'schur', 'optimize')}
            b = ref['bundle'].Bundle.FromArrays(s['K'], s['R0'], s['t0'], s['X0'], msm, mask)
        d = gg.bundle_arrays(b)
        d.update(gg.sensor_arrays(b.sensor_model))
        d['complete_cost'] = b.complete_cost()
        gg.adjuster_blocks(ref, b, 2., d, 'l2_')
        for k in ('l2_W', 'l2_HPP_inv'):                  # (large, and pinned elsewhere)
            d.pop(k)
        with gg.quiet():
            ba = ref['bundle_adjuster'].BundleAdjuster(b)
            mu, su = ba.compute_update(2.)
        d['update_l2_motion'], d['update_l2_structure'] = np.array(mu), np.array(su)
        ba, trials = gg.traced_optimize(ref, b, max_steps=4)
        d['lm_costs'], d['lm_trials'] = np.array(ba.costs), trials
        d['lm_num_steps'], d['lm_converged'] = ba.num_steps, ba.converged
        fin = gg.bundle_arrays(ba.bundle)
        d['lm_R'], d['lm_t'], d['lm_X'] = fin['R'], fin['t'], fin['X']
        gg.save(args.out, 'scene_loop_closure_60x424', d)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == '__main__':
    main()
