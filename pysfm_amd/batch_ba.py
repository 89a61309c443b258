"""Batch bundle adjustment from files (batch_ba.py:12-42):

    python -m pysfm_amd.batch_ba tracks.txt poses.txt [out_dir] [--cameras N] [--tracks M] [--max-steps K]

Loads the two text files, triangulates every track, keeps the first N cameras / M tracks
(default 100 / 100 as in the reference), freezes the first camera and the x-translation
of the second one (the reference's `param_mask[:6] = param_mask[9] = False`, which fixes
the gauge including scale), runs optimize(param_mask, max_steps=10) and writes
`adjusted_poses.txt`."""
import argparse
import os

import numpy as np

from . import bundle_io
from .bundle import Bundle
from .bundle_adjuster import BundleAdjuster


def adjust(bundle, num_cameras=100, num_tracks=100, max_steps=10, verbose=True, backend=None):
    bundle.triangulate_all()
    nc = min(num_cameras, len(bundle.cameras))
    nt = min(num_tracks, len(bundle.tracks))
    cam, trk, z = bundle.observation_table()
    keep = (cam < nc) & (trk < nt)
    sub = Bundle.FromObservations(bundle.K, bundle.Rs()[:nc], bundle.ts()[:nc], bundle.reconstruction[:nt],
                                  cam[keep], trk[keep], z[keep], sensor_model=bundle.sensor_model)
    ba = BundleAdjuster(sub, verbose=verbose, backend=backend)
    # the adjuster's parameter vector starts at the second camera (the first is frozen by
    # default), so the reference's param_mask[9] is index 3 here: t_x of camera 1
    param_mask = np.ones(6 * len(ba.optim_camera_ids) + 3 * len(ba.optim_track_ids), bool)
    param_mask[3] = False
    ba.optimize(param_mask, max_steps=max_steps)
    return ba


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('tracks_path')
    ap.add_argument('cameras_path')
    ap.add_argument('out_dir', nargs='?', default='out')
    ap.add_argument('--cameras', type=int, default=100)
    ap.add_argument('--tracks', type=int, default=100)
    ap.add_argument('--max-steps', type=int, default=10)
    args = ap.parse_args(argv)
    bundle = bundle_io.load(args.tracks_path, args.cameras_path)
    ba = adjust(bundle, args.cameras, args.tracks, args.max_steps)
    os.makedirs(args.out_dir, exist_ok=True)
    out = os.path.join(args.out_dir, 'adjusted_poses.txt')
    bundle_io.save_poses(out, ba.bundle)
    print('wrote', out)
    return ba


if __name__ == '__main__':
    main()
