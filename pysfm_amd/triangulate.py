"""Point initialisation by linear triangulation (triangulate.py:6-18), on the GPU.

The reference solves one `numpy.linalg.lstsq` per track in Python; here every track of
a bundle is triangulated by one launch of the HIP kernel `k_triangulate`
(`ba_triangulate`)."""
import numpy as np

from ._capi import PARAMS_CUR


def triangulate_bundle(bundle, device=0):
    """reconstruction[nt,3] for every track of `bundle` from its current cameras
    (Bundle.triangulate_all, bundle.py:320-321)."""
    from .backend import default_backend
    from .sensor_model import device_params_of
    be = default_backend(device)
    nc, nt = len(bundle.cameras), len(bundle.tracks)
    cam, trk, z = bundle.observation_table()
    be.set_problem(nc, nt, cam, trk, z, bundle.K, np.arange(nc, dtype=np.int32), np.ones(nt, np.uint8))
    be.set_sensor(*device_params_of(bundle.sensor_model))
    be.set_params(PARAMS_CUR, bundle.Rs(), bundle.ts(), np.zeros((nt, 3)))
    return be.triangulate(PARAMS_CUR)


def algebraic_lsq(K, Rs, ts, msms):
    """Triangulate ONE point from observations by cameras with fixed parameters
    (triangulate.py:6-18) - a one-track call of the same kernel."""
    from .bundle import Bundle
    Rs, ts = np.asarray(Rs, float), np.asarray(ts, float)
    msms = np.asarray(list(msms), float).reshape(-1, 2)
    n = len(Rs)
    b = Bundle.FromObservations(K, Rs, ts, np.ones((1, 3)), np.arange(n), np.zeros(n, int), msms)
    return triangulate_bundle(b)[0]
