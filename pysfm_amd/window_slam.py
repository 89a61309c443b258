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


def run(complete_bundle, window_size, num_tracks=100, max_steps=25, verbose=True, on_window=None, backend=None, overlap=True):
    """Returns the bundle after the last window and the per-window cost histories.

    overlap: which cameras, tracks and observations make up window i + 1 does not depend on what window i computes - only the
    parameter values do.  With two adjusters (two device handles) taking turns, window i + 1's problem is set up on the host
    WHILE window i's loop runs on the device (BundleAdjuster.optimize_begin / optimize_end); its values follow when window i is
    back.  Same numbers either way (tests/test_gpu_resident.py); off with a caller-supplied backend."""
    track_ids = list(range(min(num_tracks, len(complete_bundle.tracks))))
    cur_bundle = complete_bundle
    histories = []
    nwin = len(complete_bundle.cameras) - window_size + 1
    overlap = overlap and backend is None and nwin > 1
    # one device handle for all windows, or two taking turns
    bas = [BundleAdjuster(verbose=verbose, backend=backend) for _ in range(2 if overlap else 1)]
    prepared = False                                  # the adjuster of this window has its problem already, not yet its values
    for i in range(0, nwin):
        if verbose:
            print('\n\n==============\nWINDOW: [%d..%d]\n' % (i, i + window_size))
        prev_bundle = deepcopy(cur_bundle) if not hasattr(cur_bundle, 'clone_params') else cur_bundle.clone_params()
        ba = bas[i % len(bas)]
        if prepared:
            ba.bundle = cur_bundle
        else:
            ba.set_bundle(cur_bundle, camera_ids=list(range(i, i + window_size)), track_ids=track_ids)
        if overlap:
            ba.optimize_begin(max_steps=max_steps)
            prepared = i + 1 < nwin
            if prepared:
                bas[(i + 1) % 2].set_bundle(cur_bundle, camera_ids=list(range(i + 1, i + 1 + window_size)), track_ids=track_ids, upload=False)
            ba.optimize_end()
        else:
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
