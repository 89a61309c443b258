"""Sliding-window bundle adjustment over a camera sequence (window_slam.py:17-48, without
the plotting): every window selects `camera_ids = [i, i+window)` and the first
`num_tracks` tracks through `BundleAdjuster.set_bundle`, optimises, and carries the
result into the next window.

    python -m pysfm_amd.window_slam tracks.txt poses.txt WINDOW_SIZE
"""
import sys
from copy import deepcopy

import numpy as np

from . import bundle_io, geometry
from .bundle_adjuster import BundleAdjuster


def run(complete_bundle, window_size, num_tracks=100, max_steps=25, verbose=True, on_window=None, backend=None):
    """Returns the bundle after the last window and the per-window cost histories."""
    track_ids = list(range(min(num_tracks, len(complete_bundle.tracks))))
    cur_bundle = complete_bundle
    histories = []
    ba = BundleAdjuster(verbose=verbose, backend=backend)  # one device handle for all windows
    for i in range(0, len(complete_bundle.cameras) - window_size + 1):
        if verbose:
            print('\n\n==============\nWINDOW: [%d..%d]\n' % (i, i + window_size))
        prev_bundle = deepcopy(cur_bundle) if not hasattr(cur_bundle, 'clone_params') else cur_bundle.clone_params()
        camera_ids = list(range(i, i + window_size))
        ba.set_bundle(cur_bundle, camera_ids=camera_ids, track_ids=track_ids)
        ba.optimize(max_steps=max_steps)
        cur_bundle = ba.bundle
        histories.append(list(ba.costs))
        # propagate the update to the next camera (call shape of window_slam.py:43-48)
        next_camera_id = i + window_size
        if next_camera_id < len(cur_bundle.cameras):
            geometry.propagate_pose_update_inplace(prev_bundle.cameras[i], cur_bundle.cameras[i],
                                                   cur_bundle.cameras[i])
        if on_window is not None:
            on_window(i, ba)
    return cur_bundle, histories


def main(argv):
    window_size = int(argv[3])
    print('Loading bundle...')
    bundle = bundle_io.load(argv[1], argv[2])
    print('Triangulating initial points...')
    bundle.triangulate_all()
    print('Cameras:', len(bundle.cameras))
    print('Tracks:', len(bundle.tracks))
    print('Window Size:', window_size)
    out, _ = run(bundle, window_size)
    return out


if __name__ == '__main__':
    np.seterr(all='raise')
    main(sys.argv)
