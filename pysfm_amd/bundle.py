"""Scene data model with pysfm's names: Camera, Track, Bundle (bundle.py:54-505).

Host-side containers only.  Everything numeric that the reference's ``Bundle``
computes per observation - ``predict``, ``reproj_error``, ``residual``,
``Jresidual`` and the dense ``residuals`` / ``Jresiduals`` builders - is evaluated
by the HIP kernel ``k_eval`` through ``ba_eval_observations``; there is no NumPy
restatement of that arithmetic in this package.

Besides the reference's list-of-``Track``-dicts form (``FromArrays``,
``add_track``) a bundle can be array-native (``FromObservations``): 1M-observation
scenes keep their observations as three flat arrays and materialise ``Track``
objects lazily.
"""
import numpy as np

from . import lie
from ._capi import PARAMS_CUR


############################################################################
class Camera(object):
    """Pinhole camera pose (bundle.py:54-91)."""

    def __init__(self, R=None, t=None, idx=None):
        if R is None:
            R = np.eye(3)
        if t is None:
            t = np.zeros(3)
        assert np.shape(R) == (3, 3)
        assert np.shape(t) == (3,)
        self.idx = idx
        self.R = np.asarray(R, float)
        self.t = np.asarray(t, float)

    @property
    def Rt(self):
        return (self.R, self.t)

    def projection_matrix(self):
        return np.hstack((self.R, self.t[:, np.newaxis]))

    def perturb(self, delta):
        """R <- R exp(delta[:3]), t <- t + delta[3:] (bundle.py:76-80)."""
        assert np.shape(delta) == (6,)
        self.R = np.dot(self.R, lie.SO3.exp(delta[:3]))
        self.t = self.t + np.asarray(delta[3:], float)
        return self

    def transform(self, R, t):
        self.R = np.dot(R, self.R)
        self.t = np.dot(R, self.t) + t

    def __repr__(self):
        return 'Camera(%s)' % str(self.projection_matrix()).replace('\n', '\n       ')

    __str__ = __repr__


############################################################################
class Track(object):
    """Measurements of one 3D point: dict camera_id -> 2-vector (bundle.py:95-126)."""

    def __init__(self, camera_ids=(), measurements=()):
        camera_ids = [int(c) for c in camera_ids]
        assert len(camera_ids) == len(measurements)
        self.measurements = dict(zip(camera_ids, [np.asarray(m, float) for m in measurements]))

    def add_measurement(self, camera_id, measurement):
        assert np.shape(measurement) == (2,)
        self.measurements[int(camera_id)] = np.asarray(measurement, float)

    def has_measurement(self, camera_id):
        return camera_id in self.measurements

    def get_measurement(self, camera_id):
        return self.measurements[camera_id]

    def camera_ids(self):
        return self.measurements.keys()

    def intersect_camera_ids(self, camera_ids):
        return self.measurements.keys() & set(camera_ids)

    def __repr__(self):
        return 'Track(%s)' % '\n      '.join('%-2d ->  [%10f, %10f]' % (i, m[0], m[1])
                                             for i, m in self.measurements.items())

    __str__ = __repr__


class _Cameras(list):
    """`Bundle.cameras` over stacked pose arrays: a list whose Camera objects are made on first access.  clone_params of a
    hundred cameras is two array copies instead of two hundred, and the sliding-window caller, which clones the bundle
    twice per frame and touches ten cameras of it, pays for those ten.  A Camera that has been handed out is the truth
    from then on (its R / t may be replaced or edited in place): stacked() reads it back."""

    def __init__(self, R, t):
        list.__init__(self, [None] * len(R))
        self._R, self._t = R, t            # poses of the entries that are still None

    def _make(self, i):
        k = Camera.__new__(Camera)        # (no validation: rows of arrays that were validated when they came in)
        k.idx, k.R, k.t = i, self._R[i].copy(), self._t[i].copy()
        list.__setitem__(self, i, k)
        return k

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(len(self)))]
        c = list.__getitem__(self, i)
        if c is None:
            c = self._make(i if i >= 0 else i + len(self))
        return c

    def __iter__(self):
        for i in range(len(self)):
            yield self[i]

    def __reversed__(self):
        for i in range(len(self) - 1, -1, -1):
            yield self[i]

    def stacked(self):
        """(R [n,3,3], t [n,3]) of all cameras, new arrays."""
        n = len(self)
        R, t = np.empty((n, 3, 3)), np.empty((n, 3))
        m = min(n, len(self._R))
        R[:m], t[:m] = self._R[:m], self._t[:m]           # (rows of entries that are still None; objects override below)
        for i, c in enumerate(list.__iter__(self)):
            if c is not None:
                R[i], t[i] = c.R, c.t
        return R, t

    def poses(self, ids):
        """(R, t) of the cameras `ids` without making objects for them."""
        get = list.__getitem__
        if len(self) > len(self._R):                      # (cameras appended since: they have no row)
            R, t = self.stacked()
            return R[ids], t[ids]
        R, t = self._R[ids], self._t[ids]                 # (fancy indexing: copies)
        for pos, i in enumerate(ids):
            c = get(self, int(i))
            if c is not None:
                R[pos], t[pos] = c.R, c.t
        return R, t

    def set_poses(self, ids, R, t):
        """Camera ids[k] gets the pose (R[k], t[k]): the rows of the stacked arrays, or the object if one has been handed out."""
        get = list.__getitem__
        if len(self) > len(self._R):                      # (cameras appended since: they have no row)
            for pos, i in enumerate(ids):
                c = self[int(i)]
                c.R, c.t = R[pos].copy(), t[pos].copy()
            return
        self._R[ids], self._t[ids] = R, t
        for pos, i in enumerate(ids):
            c = get(self, int(i))
            if c is not None:
                c.R, c.t = R[pos].copy(), t[pos].copy()

    def __reduce_ex__(self, protocol):
        return (list, (list(self),))                      # copies and pickles are plain lists of Camera objects

    # whatever moves entries about first makes every camera a real object (the rows of the stacked arrays are tied to positions)
    def _all(self):
        for i in range(len(self)):
            self[i]
        return self

    def __delitem__(self, i): list.__delitem__(self._all(), i)
    def insert(self, i, c): list.insert(self._all(), i, c)
    def pop(self, *a): return list.pop(self._all(), *a)
    def remove(self, c): list.remove(self._all(), c)
    def reverse(self): list.reverse(self._all())
    def sort(self, *a, **k): list.sort(self._all(), *a, **k)
    def index(self, *a): return list.index(self._all(), *a)
    def count(self, c): return list.count(self._all(), c)
    def __contains__(self, c): return list.__contains__(self._all(), c)
    def __eq__(self, other): return list(self) == list(other) if isinstance(other, list) else NotImplemented
    def __ne__(self, other): return not self == other
    __hash__ = None

    def __setitem__(self, i, c):
        if isinstance(i, slice):
            self._all()
        list.__setitem__(self, i, c)

    # what reads the raw storage at C level (concatenation, repetition, copies) would hand out the None placeholders
    def __add__(self, other): return list(self) + list(other)
    def __radd__(self, other): return list(other) + list(self)
    def __mul__(self, n): return list(self) * n
    __rmul__ = __mul__
    def copy(self): return list(self)
    def __iadd__(self, other):
        list.extend(self, other)
        return self
    def __imul__(self, n):
        raise TypeError('a camera list over stacked poses cannot be repeated in place')


class _LazyTracks(object):
    """Sequence of Track views over a CSR observation table (array-native bundles)."""

    def __init__(self, cam, z, off):
        self._cam, self._z, self._off = cam, z, off
        self._cache = {}

    def __len__(self):
        return len(self._off) - 1

    def __getitem__(self, j):
        if isinstance(j, slice):
            return [self[k] for k in range(*j.indices(len(self)))]
        j = int(j)
        if j < 0:
            j += len(self)
        tr = self._cache.get(j)
        if tr is None:
            s, e = self._off[j], self._off[j + 1]
            tr = Track(self._cam[s:e].tolist(), self._z[s:e])
            self._cache[j] = tr
        return tr

    def __iter__(self):
        return (self[j] for j in range(len(self)))


############################################################################
class Bundle(object):
    NumCamParams = 6
    NumPointParams = 3

    def __init__(self, ncameras=0, ntracks=0):
        self.cameras = []
        self.tracks = []
        self.reconstruction = np.zeros((ntracks, 3))
        self.K = np.eye(3)
        from . import sensor_model
        self.sensor_model = sensor_model.GaussianModel(1.)
        self._table = None          # (cam_id, track_id, z) sorted by (track, cam) for array-native bundles
        for i in range(ncameras):
            self.add_camera()
        for i in range(ntracks):
            self.tracks.append(Track())

    # ------------------------------------------------------------------ checks
    def check_consistency(self):
        """bundle.py:148-166."""
        assert self.sensor_model is not None
        assert np.shape(self.K) == (3, 3), 'shape was ' + str(np.shape(self.K))
        assert np.shape(self.reconstruction) == (len(self.tracks), 3), \
            'shape was ' + str(np.shape(self.reconstruction))
        assert np.sum(np.square(self.reconstruction)) > 1e-8, 'reconstruction must be initialized'
        cam, trk, z = self.observation_table()
        checked = self._table is not None and self.__dict__.get('_table_cameras_checked') == len(self.cameras)
        if len(cam) and not checked:                         # (the table of an array-native bundle is immutable: checked once)
            assert cam.min() >= 0 and cam.max() < len(self.cameras), \
                'There are %d cameras but a track has a measurement for camera %d' % \
                (len(self.cameras), int(cam.max() if cam.max() >= len(self.cameras) else cam.min()))
            if self._table is not None:
                self._table_cameras_checked = len(self.cameras)
        for camera in (list.__iter__(self.cameras) if isinstance(self.cameras, _Cameras) else self.cameras):
            if camera is not None:                           # (cameras nobody has touched are rows of validated arrays)
                assert camera.R.shape == (3, 3)
                assert camera.t.shape == (3,)

    # ------------------------------------------------------------------ building
    def add_camera(self, camera=None):
        if camera is None:
            camera = Camera(np.eye(3), np.zeros(3))
        camera.idx = len(self.cameras)
        self.cameras.append(camera)
        return camera

    def add_track(self, track=None):
        if self._table is not None:
            raise TypeError('array-native bundles are immutable; build with FromObservations')
        if track is None:
            track = Track()
        else:
            for i, idx in enumerate(track.camera_ids()):
                if idx < 0 or idx >= len(self.cameras):
                    raise Exception('Invalid camera ID=%d in new track at camera_ids[%d])' % (int(idx), i))
        self.tracks.append(track)
        self.reconstruction = np.vstack((self.reconstruction, np.zeros(3)))
        return track

    @classmethod
    def FromArrays(cls, K, Rs, ts, pts, measurements, measurement_mask=None):
        """Dense N-cameras x M-tracks x 2 measurement array + bool mask (bundle.py:333-364)."""
        K, Rs, ts = np.asarray(K, float), np.asarray(Rs, float), np.asarray(ts, float)
        measurements = np.asarray(measurements, float)
        if measurement_mask is None:
            measurement_mask = np.ones(measurements.shape[:-1], bool)
        measurement_mask = np.asarray(measurement_mask, bool)
        assert len(Rs) == len(ts)
        assert np.shape(Rs)[1:] == (3, 3)
        assert np.shape(ts)[1:] == (3,)
        assert measurements.shape[0] == len(Rs)
        assert measurements.shape[2] == 2
        assert measurement_mask.shape == measurements.shape[:-1]
        b = cls()
        b.K = K.copy()
        b.cameras = _Cameras(Rs.copy(), ts.copy())
        for j in range(measurements.shape[1]):
            camera_ids = np.nonzero(measurement_mask[:, j])[0].tolist()
            b.tracks.append(Track(camera_ids, measurements[camera_ids, j]))
        b.reconstruction = np.array(pts, float)
        return b

    @classmethod
    def FromObservations(cls, K, Rs, ts, pts, obs_cam, obs_track, obs_z, sensor_model=None):
        """Array-native bundle: observation n is measurement obs_z[n] of track
        obs_track[n] in camera obs_cam[n].  Scales to millions of observations."""
        b = cls()
        b.K = np.array(K, float)
        Rs, ts = np.asarray(Rs, float), np.asarray(ts, float)
        assert Rs.shape[1:] == (3, 3) and ts.shape == (len(Rs), 3)
        b.cameras = _Cameras(Rs.copy(), ts.copy())
        b.reconstruction = np.array(pts, float)
        nt = len(b.reconstruction)
        obs_cam = np.asarray(obs_cam, np.int64)
        obs_track = np.asarray(obs_track, np.int64)
        obs_z = np.asarray(obs_z, float).reshape(-1, 2)
        assert len(obs_cam) == len(obs_track) == len(obs_z)
        if len(obs_cam):
            assert obs_track.min() >= 0 and obs_track.max() < nt
        # (one combined key: a (camera, track) pair occurs at most once, so no stable two-key sort is needed - 4x faster at 1e7)
        key = obs_track * max(len(b.cameras), 1) + obs_cam
        if len(key) < 2 or bool(np.all(key[1:] > key[:-1])):                  # already in (track, camera) order: nothing to sort
            cam, trk, z = obs_cam.astype(np.int32), obs_track.astype(np.int32), np.array(obs_z)
        else:
            order = np.argsort(key)
            cam, trk, z = obs_cam[order].astype(np.int32), obs_track[order].astype(np.int32), obs_z[order]
        if len(cam) > 1:
            dup = (cam[1:] == cam[:-1]) & (trk[1:] == trk[:-1])
            assert not dup.any(), 'a (camera, track) pair may be observed at most once'
        b._table = (cam, trk, z)
        off = np.zeros(nt + 1, np.int64)
        np.cumsum(np.bincount(trk, minlength=nt), out=off[1:])
        b.tracks = _LazyTracks(cam, z, off)
        if sensor_model is not None:
            b.sensor_model = sensor_model
        return b

    # ------------------------------------------------------------------ array views
    def observation_table(self):
        """(camera_id[N], track_id[N], z[N,2]) sorted by (track, camera)."""
        if self._table is not None:
            return self._table
        cam, trk, z = [], [], []
        for j, tr in enumerate(self.tracks):
            for i in sorted(tr.measurements.keys()):
                cam.append(i)
                trk.append(j)
                z.append(tr.measurements[i])
        return (np.array(cam, np.int32), np.array(trk, np.int32),
                np.array(z, float).reshape(-1, 2))

    def select_observations(self, camera_ids, track_ids):
        """Observations seen by camera_ids x track_ids as POSITIONS into those lists,
        in the order the reference's loops visit them (tracks outer, cameras inner;
        bundle_adjuster.py:222-226)."""
        cam, trk, z = self.observation_table()
        cids = np.asarray(camera_ids, np.int64).reshape(-1)
        tids = np.asarray(track_ids, np.int64).reshape(-1)
        ascending = lambda a: len(a) < 2 or bool(np.all(a[1:] > a[:-1]))
        everything = lambda a, n: len(a) == n and (n == 0 or (a[0] == 0 and a[-1] == n - 1 and ascending(a)))
        if everything(cids, len(self.cameras)) and everything(tids, len(self.tracks)):
            # all cameras x all tracks in their own order (BundleAdjuster(bundle) without ids): the table as it is
            return np.ascontiguousarray(cam, np.int32), np.ascontiguousarray(trk, np.int32), np.ascontiguousarray(z, float)
        if (self._table is not None and len(cids) and len(tids) and len(trk) and int(cids[-1]) - int(cids[0]) + 1 == len(cids)
                and ascending(cids)):
            # a window of consecutive cameras over an array-native bundle (window_slam.py:17-48: frame after frame): the table
            # is sorted by (track, camera), so what a track contributes is ONE stretch of its rows - two binary searches per
            # track instead of a pass over every row of the selected tracks
            # (the multiplier is cached WITH the key: add_camera() on an array-native bundle changes len(cameras), and a key built
            #  with another multiplier than `base` finds the wrong rows without any error)
            cached = getattr(self, '_table_key', None)
            if cached is None or len(cached[1]) != len(trk):
                nc_key = max(len(self.cameras), int(cam.max()) + 1 if len(cam) else 1, 1)
                cached = self._table_key = (nc_key, trk.astype(np.int64) * nc_key + cam)
            nc_all, key = cached
            base = tids * nc_all
            lo = np.searchsorted(key, base + int(cids[0]), 'left')
            hi = np.searchsorted(key, base + int(cids[-1]), 'right')
            cnt = hi - lo
            total = int(cnt.sum())
            rows = np.repeat(lo - (np.cumsum(cnt) - cnt), cnt) + np.arange(total)
            return ((cam[rows] - int(cids[0])).astype(np.int32), np.repeat(np.arange(len(tids), dtype=np.int32), cnt),
                    np.ascontiguousarray(z[rows]))
        cpos = -np.ones(max(len(self.cameras), 1), np.int64)
        cpos[cids] = np.arange(len(cids))
        if len(tids) * 4 < len(self.tracks) and len(trk):
            # few of many tracks (the sliding-window caller: 100 of 1000, window after window): only their rows of the
            # table (sorted by track) instead of a pass over all of it
            # (the offsets are cached only beside the immutable table of an array-native bundle: a bundle of Track objects
            # can be edited between calls - a measurement moved to another track keeps every count this cache could check)
            off = getattr(self, '_track_offsets', None) if self._table is not None else None
            if off is None or len(off) != len(self.tracks) + 1 or off[-1] != len(trk):
                off = np.zeros(len(self.tracks) + 1, np.int64)
                np.cumsum(np.bincount(trk, minlength=len(self.tracks)), out=off[1:])
                if self._table is not None:
                    self._track_offsets = off
            if len(tids) and int(tids[-1]) - int(tids[0]) + 1 == len(tids) and ascending(tids):
                # a consecutive stretch of tracks (window_slam.py:19: the first `num_tracks` of them): one slice of the table
                rows = slice(int(off[tids[0]]), int(off[tids[-1] + 1]))
                ci, ti, zz = cpos[cam[rows]], trk[rows].astype(np.int64) - int(tids[0]), z[rows]
            else:
                cnt = off[tids + 1] - off[tids]
                rows = np.repeat(off[tids] - np.concatenate(([0], np.cumsum(cnt)[:-1])), cnt) + np.arange(int(cnt.sum()))
                ci, ti, zz = cpos[cam[rows]], np.repeat(np.arange(len(tids)), cnt), z[rows]
            keep = ci >= 0
            ci, ti, zz = ci[keep], ti[keep], zz[keep]
            in_order = ascending(cids)                      # (rows come track position by track position, cameras ascending by id)
        else:
            tpos = -np.ones(max(len(self.tracks), 1), np.int64)
            tpos[tids] = np.arange(len(tids))
            ci, ti = cpos[cam], tpos[trk]
            keep = (ci >= 0) & (ti >= 0)
            ci, ti, zz = ci[keep], ti[keep], z[keep]
            in_order = ascending(cids) and ascending(tids)  # (the table is sorted by (track id, camera id): so is what is left of it)
        if in_order:
            return ci.astype(np.int32), ti.astype(np.int32), np.ascontiguousarray(zz)
        order = np.argsort(ti * max(len(cids), 1) + ci)
        return ci[order].astype(np.int32), ti[order].astype(np.int32), np.ascontiguousarray(zz[order])

    def Rs(self):
        if isinstance(self.cameras, _Cameras):
            return self.cameras.stacked()[0]
        return np.array([cam.R for cam in self.cameras])

    def ts(self):
        if isinstance(self.cameras, _Cameras):
            return self.cameras.stacked()[1]
        return np.array([cam.t for cam in self.cameras])

    def projection_matrices(self):
        return np.array([cam.projection_matrix() for cam in self.cameras])

    def points(self):
        return self.reconstruction

    def measurement(self, i, j):
        return self.tracks[j].get_measurement(i)

    def measurement_ids(self, track_indices=None):
        if track_indices is None:
            track_indices = range(len(self.tracks))
        return ((i, j) for j in track_indices for i in self.tracks[j].camera_ids())

    def measurement_ids_for_cameras(self, cameras_to_include, track_indices=None):
        if track_indices is None:
            track_indices = range(len(self.tracks))
        return ((i, j) for j in track_indices
                for i in self.tracks[j].intersect_camera_ids(cameras_to_include))

    def num_params(self):
        return len(self.cameras) * Bundle.NumCamParams + len(self.tracks) * Bundle.NumPointParams

    # ------------------------------------------------------------------ device evaluation
    def _device_eval(self, cam_ids, track_ids, z, **want):
        """Evaluate observation (cam_ids[n], track_ids[n], z[n]) on the GPU; outputs
        come back in the caller's order."""
        from .backend import default_backend
        from .sensor_model import device_params_of
        cam_ids = np.asarray(cam_ids, np.int32)
        track_ids = np.asarray(track_ids, np.int32)
        z = np.asarray(z, float).reshape(-1, 2)
        order = np.argsort(track_ids, kind='stable')
        be = default_backend()
        nc, nt = len(self.cameras), len(self.tracks)
        be.set_problem(nc, nt, cam_ids[order], track_ids[order], z[order], self.K,
                       np.arange(nc, dtype=np.int32), np.ones(nt, np.uint8))
        be.set_sensor(*device_params_of(self.sensor_model))
        be.set_params(PARAMS_CUR, self.Rs(), self.ts(), self.reconstruction)
        out = be.eval_observations(PARAMS_CUR, **want)
        inv = np.empty_like(order)
        inv[order] = np.arange(len(order))
        return {k: (v[inv] if v is not None else None) for k, v in out.items()}

    def _all_pairs(self):
        ids = list(self.measurement_ids())
        cam = np.array([i for i, j in ids], np.int32)
        trk = np.array([j for i, j in ids], np.int32)
        z = np.array([self.measurement(i, j) for i, j in ids], float).reshape(-1, 2)
        return cam, trk, z

    def predict(self, i, j):
        """pr(K(R_i x_j + t_i)) (bundle.py:243-244)."""
        return self._device_eval([i], [j], np.zeros((1, 2)), e=True, r=False, Jc=False, Jp=False)['e'][0]

    def reproj_error(self, i, j):
        """bundle.py:247-248."""
        return self._device_eval([i], [j], [self.measurement(i, j)], e=True, r=False, Jc=False, Jp=False)['e'][0]

    def residual(self, i, j):
        """bundle.py:251-252."""
        return self._device_eval([i], [j], [self.measurement(i, j)], e=False, r=True, Jc=False, Jp=False)['r'][0]

    def Jresidual(self, i, j):
        """(2x6 camera block, 2x3 point block) (bundle.py:255-277)."""
        out = self._device_eval([i], [j], [self.measurement(i, j)], e=False, r=False, Jc=True, Jp=True)
        return out['Jc'][0], out['Jp'][0]

    def predictions(self):
        cam, trk, z = self._all_pairs()
        return self._device_eval(cam, trk, np.zeros_like(z), e=True, r=False, Jc=False, Jp=False)['e']

    def reproj_errors(self):
        cam, trk, z = self._all_pairs()
        return self._device_eval(cam, trk, z, e=True, r=False, Jc=False, Jp=False)['e']

    def residuals(self):
        """Complete residual vector in measurement_ids() order (bundle.py:290-291)."""
        cam, trk, z = self._all_pairs()
        return self._device_eval(cam, trk, z, e=False, r=True, Jc=False, Jp=False)['r'].reshape(-1)

    def complete_cost(self):
        """bundle.py:293-295."""
        return float(np.sum(np.square(self.residuals())))

    # ------------------------------------------------------------------ copies / transforms
    def clone_params(self):
        """Deep-copy cameras and points, share tracks and sensor model (bundle.py:301-310)."""
        b = Bundle.__new__(Bundle)                        # (not __init__: it builds a default sensor model only to drop it)
        b.K = self.K.copy()
        if isinstance(self.cameras, _Cameras):
            b.cameras = _Cameras(*self.cameras.stacked())
        else:
            b.cameras = _Cameras(np.array([c.R for c in self.cameras], float).reshape(-1, 3, 3),
                                 np.array([c.t for c in self.cameras], float).reshape(-1, 3))
        b.reconstruction = self.reconstruction.copy()
        b.tracks = self.tracks
        b._table = self._table
        for name in ('_table_key', '_track_offsets', '_table_cameras_checked'):      # (what was derived from the immutable table travels with it)
            if name in self.__dict__:
                b.__dict__[name] = self.__dict__[name]
        b.sensor_model = self.sensor_model
        return b

    def triangulate(self, track):
        """Linear least-squares triangulation of one track (bundle.py:313-317)."""
        from . import triangulate
        ids = list(track.camera_ids())
        return triangulate.algebraic_lsq(self.K, [self.cameras[i].R for i in ids],
                                         [self.cameras[i].t for i in ids],
                                         [track.measurements[i] for i in ids])

    def triangulate_all(self):
        """bundle.py:320-321, batched on the GPU."""
        from . import triangulate
        self.reconstruction = triangulate.triangulate_bundle(self)

    def make_relative_to_first_camera(self):
        R, t = self.cameras[0].Rt
        self.transform(R, t)

    def transform(self, R, t):
        """x -> R x + t for points, inverse for cameras (bundle.py:383-396)."""
        assert np.shape(R) == (3, 3), 'shape was ' + str(np.shape(R))
        assert np.shape(t) == (3,), 'shape was ' + str(np.shape(t))
        self.reconstruction = np.dot(self.reconstruction, np.asarray(R).T) + t
        RR = np.asarray(R).T
        tt = -np.dot(RR, t)
        for camera in self.cameras:
            camera.transform(RR, tt)

    def perturb(self, delta, param_mask=None):
        """Linear update of every camera and point (bundle.py:404-428; test helper)."""
        delta = np.asarray(delta, float)
        nparams = self.num_params()
        if param_mask is not None:
            assert np.shape(param_mask) == (nparams,), 'shape was ' + str(np.shape(param_mask))
            full = np.zeros(nparams)
            full[np.asarray(param_mask)] = delta
            delta = full
        assert delta.shape == (nparams,), 'shape was ' + str(np.shape(delta))
        for i, cam in enumerate(self.cameras):
            cam.perturb(delta[i * 6:(i + 1) * 6])
        offs = len(self.cameras) * 6
        self.reconstruction = self.reconstruction + delta[offs:].reshape(-1, 3)
        return self

    # ------------------------------------------------------------------ dense test builders
    def residuals_partial(self, camera_ids, track_ids):
        """bundle.py:437-450."""
        ci, ti, z = self.select_observations(camera_ids, track_ids)
        cam = np.asarray(camera_ids)[ci]
        trk = np.asarray(track_ids)[ti]
        return self._device_eval(cam, trk, z, e=False, r=True, Jc=False, Jp=False)['r'].reshape(-1)

    def Jresiduals(self):
        return self.Jresiduals_extended()[0]

    def Jresiduals_extended(self):
        """Dense Jacobian with row / column labels (bundle.py:456-480)."""
        cam, trk, z = self._all_pairs()
        out = self._device_eval(cam, trk, z, e=False, r=False, Jc=True, Jp=True)
        n = len(cam)
        J = np.zeros((n * 2, self.num_params()))
        row_labels = np.empty((n * 2, 2), int)
        col_labels = np.empty((self.num_params(), 2), int)
        col_offs = len(self.cameras) * 6
        for m in range(n):
            i, j = int(cam[m]), int(trk[m])
            J[2 * m:2 * m + 2, i * 6:i * 6 + 6] = out['Jc'][m]
            J[2 * m:2 * m + 2, col_offs + j * 3:col_offs + j * 3 + 3] = out['Jp'][m]
            row_labels[2 * m:2 * m + 2] = (i, j)
            col_labels[i * 6:i * 6 + 6] = (i, -1)
            col_labels[col_offs + j * 3:col_offs + j * 3 + 3] = (-1, j)
        return J, row_labels, col_labels

    def Jresiduals_partial(self, camera_ids=None, track_ids=None):
        """bundle.py:483-505."""
        ci, ti, z = self.select_observations(camera_ids, track_ids)
        cam = np.asarray(camera_ids)[ci]
        trk = np.asarray(track_ids)[ti]
        out = self._device_eval(cam, trk, z, e=False, r=False, Jc=True, Jp=True)
        nc, nt = len(camera_ids), len(track_ids)
        J = np.zeros((2 * len(ci), nc * 6 + nt * 3))
        for m in range(len(ci)):
            J[2 * m:2 * m + 2, ci[m] * 6:ci[m] * 6 + 6] = out['Jc'][m]
            c0 = nc * 6 + ti[m] * 3
            J[2 * m:2 * m + 2, c0:c0 + 3] = out['Jp'][m]
        return J


# module-level helpers with the reference's names (bundle.py:8-19) ------------------
def Jpr(x):
    """Jacobian of pr() at a homogeneous 3-vector (bundle.py:8-11).  Shape helper for
    callers; the kernels inline this."""
    x = np.asarray(x, float)
    return np.array([[1. / x[2], 0, -x[0] / (x[2] * x[2])],
                     [0, 1. / x[2], -x[1] / (x[2] * x[2])]])


def project(K, R, t, x):
    """Pinhole projection of a single point through the GPU evaluator (bundle.py:14-19)."""
    return _one_observation(K, R, t, x).predict(0, 0)


def _one_observation(K, R, t, x):
    """A one-camera, one-point bundle with the unit Gaussian sensor model: residual = projection - measurement, so the
    evaluator's residual Jacobians ARE the Jacobians of the projection."""
    b = Bundle()
    b.K = np.asarray(K, float)
    b.add_camera(Camera(np.asarray(R, float), np.asarray(t, float)))
    b.tracks.append(Track())
    b.reconstruction = np.asarray(x, float).reshape(1, 3)
    return b


def project2(K, R0, m, t, x):
    """Projection through the camera turned by exp(m) on the right, R = R0 exp(m) (bundle.py:22-24: the function whose
    derivative at m = 0 is Jproject_R)."""
    return project(K, np.dot(np.asarray(R0, float), lie.SO3.exp(m)), t, x)


def Jproject_all(K, R, t, x):
    """(2x6 Jacobian w.r.t. the camera [rotation | translation], 2x3 Jacobian w.r.t. the point) of project() - evaluated
    by the GPU evaluator (k_eval), as Bundle.Jresidual is (bundle.py:45-50)."""
    out = _one_observation(K, R, t, x)._device_eval([0], [0], np.zeros((1, 2)), e=False, r=False, Jc=True, Jp=True)
    return out['Jc'][0], out['Jp'][0]


def Jproject_cam(K, R, t, x):
    """2x6 Jacobian of project() w.r.t. the camera parameters (bundle.py:40-42)."""
    return Jproject_all(K, R, t, x)[0]


def Jproject_R(K, R, t, x):
    """2x3 Jacobian of project() w.r.t. the rotation update m of R exp(m) (bundle.py:27-29)."""
    return Jproject_all(K, R, t, x)[0][:, :3]


def Jproject_t(K, R, t, x):
    """2x3 Jacobian of project() w.r.t. the translation (bundle.py:32-33)."""
    return Jproject_all(K, R, t, x)[0][:, 3:]


def Jproject_x(K, R, t, x):
    """2x3 Jacobian of project() w.r.t. the point (bundle.py:36-37)."""
    return Jproject_all(K, R, t, x)[1]
