"""On-disk formats of pysfm's batch drivers (bundle_io.py:10-27, batch_ba.py:38-42).

tracks file : one track per line, triplets `camera_id u v`
poses file  : one camera per line, 12 floats = the 3x4 matrix [R | t] row-major
The calibration is the reference's hard-coded one (f = 1500, 1480 x 1360 image)."""
import numpy as np

from .bundle import Bundle

width = 1480
height = 1360
K = np.array([1500, 0, width / 2, 0, 1500, height / 2, 0, 0, 1], float).reshape((3, 3))


def load(tracks_path, cameras_path):
    """Array-native loader: parses both files with NumPy and builds a
    Bundle.FromObservations (tracks materialise lazily).  reconstruction is zero until
    triangulate_all() - as in the reference."""
    P = np.loadtxt(cameras_path, ndmin=2).reshape(-1, 3, 4)
    cam, trk, z = [], [], []
    with open(tracks_path) as fd:
        for j, line in enumerate(fd):
            v = np.array(line.split(), float)
            assert len(v) % 3 == 0, 'Error at line %d:\n %s' % (j, line)
            v = v.reshape(-1, 3)
            cam.append(v[:, 0].astype(np.int64))
            trk.append(np.full(len(v), j, np.int64))
            z.append(v[:, 1:])
    ntracks = len(cam)
    cam = np.concatenate(cam) if cam else np.zeros(0, np.int64)
    trk = np.concatenate(trk) if trk else np.zeros(0, np.int64)
    z = np.concatenate(z) if z else np.zeros((0, 2))
    bundle = Bundle.FromObservations(K, P[:, :, :3], P[:, :, 3], np.zeros((ntracks, 3)), cam, trk, z)
    return bundle


def save_poses(path, bundle):
    """One `%f `-formatted 3x4 projection matrix per line (batch_ba.py:38-42)."""
    with open(path, 'w') as fd:
        for camera in bundle.cameras:
            fd.write(''.join(['%f ' % v for v in camera.projection_matrix().flatten()]))
            fd.write('\n')


def save_tracks(path, bundle):
    """Inverse of load() for the tracks file (`%d %f %f ` per measurement, sequence.py:81-86)."""
    cam, trk, z = bundle.observation_table()
    off = np.searchsorted(trk, np.arange(len(bundle.tracks) + 1))
    with open(path, 'w') as fd:
        for j in range(len(bundle.tracks)):
            s, e = off[j], off[j + 1]
            fd.write(''.join('%d %f %f ' % (cam[n], z[n, 0], z[n, 1]) for n in range(s, e)))
            fd.write('\n')
