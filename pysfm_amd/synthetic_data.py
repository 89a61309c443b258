"""Synthetic scenes.

``generate_sequence`` is the counterpart of the reference generator
(synthetic_data.py:30-69): same seed, same draw order, so it reproduces the
reference's 5-camera / 50-point scene (BASELINE config 1) number for number.

``generate_banded_scene`` is the scalable generator the reference lacks (its
cameras start inside the point cloud and see every point): cameras along a track,
each point seen by the L nearest cameras, so the reduced camera system is block
banded.  It produces BASELINE configs 2-5.  Scene generation is plain NumPy on the
host - it is input preparation, not part of the adjuster.
"""
import numpy as np

from .lie import SO3


def _project(K, R, t, X):
    p = (X @ R.T + t) @ K.T
    return p[:, :2] / p[:, 2:3]


def generate_sequence(nframes, npts, msm_noise=.02):
    """K, Rs[nframes], ts[nframes], pts[npts], measurements[nframes,npts,2]
    drawn exactly as synthetic_data.py:30-69 does (np.random.seed(654); points
    U[-1,1]^3; per frame npts x randn(2) then randn(3) rotation step, randn(3)
    translation step; R_pert = t_pert = .02; K = I)."""
    rs = np.random.RandomState(654)
    pts = (rs.rand(npts, 3) * 2 - 1) * 1.
    K = np.eye(3)
    R, t = np.eye(3), np.zeros(3)
    Rs, ts, msm = [], [], []
    for i in range(nframes):
        Rs.append(R)
        ts.append(t)
        noise = rs.randn(npts, 2) * msm_noise           # same stream as npts successive randn(2)
        msm.append(_project(K, R, t, pts) + noise)
        R = np.dot(R, SO3.exp(rs.randn(3) * .02))
        t = t + rs.randn(3) * .02
    return K, np.array(Rs), np.array(ts), pts, np.array(msm)


def generate_banded_scene(ncams, npts, track_len=10, seed=654, spacing=.2, msm_noise=.02,
                          init_perturbation=.01, init_seed=1888, outlier_frac=0., outlier_range=10.,
                          outlier_seed=101, init_mode='pose'):
    """Cameras k = 0..ncams-1 at (k*spacing, 0, 0) + N(0,.02^2) jitter, rotations
    exp(.02 randn(3)); points uniform in x in [0, spacing*ncams], y in [-1,1],
    z in [4,8], sorted by x; each point observed by the `track_len` consecutive
    cameras nearest in x (N = track_len * npts exactly); K = I; measurement noise
    N(0, msm_noise^2); optionally a fraction of observations replaced by
    uniform(-outlier_range, outlier_range)^2 (pattern of bundle_unittest.py:59-67).

    Returns a dict: K, true (R,t,X), initial guess (R0,t0,X0) = truth perturbed by
    N(0, init_perturbation^2) on all 6 camera + 3 point parameters (cf.
    test_bundle.py:165-167; camera 0 is left at its true pose since it is the frozen
    gauge camera), obs_cam, obs_pt (sorted by point), obs_z, outlier mask.

    init_mode: how the 6 camera numbers are applied.  'pose' (default): the camera is turned about its OWN
    centre and the centre moved, R0 = R exp(dw), c0 = c + dc, t0 = -R0 c0 - the size of the initial error does
    not depend on where the world origin is.  'params': Camera.perturb on the raw parameters, R0 = R exp(dw),
    t0 = t + dt (bundle.py:76-80), which turns the camera about the WORLD origin: a camera 200 units down the
    track (config 3) is thrown 2 units off by 0.01 rad, one 2000 units away (config 5) by 20 units, with a
    quarter of the points behind the cameras - a start no LM run recovers from.  Round 1 used 'params'."""
    assert init_mode in ('pose', 'params')
    assert ncams >= track_len
    rs = np.random.RandomState(seed)
    X = np.empty((npts, 3))
    X[:, 0] = np.sort(rs.rand(npts)) * spacing * ncams
    X[:, 1] = rs.rand(npts) * 2 - 1
    X[:, 2] = 4 + 4 * rs.rand(npts)
    centers = np.zeros((ncams, 3))
    centers[:, 0] = np.arange(ncams) * spacing
    centers += rs.randn(ncams, 3) * .02
    w = rs.randn(ncams, 3) * .02
    R = _so3_exp_batch(w)
    t = -np.einsum('nij,nj->ni', R, centers)

    c0 = np.rint(X[:, 0] / spacing).astype(np.int64) - track_len // 2
    c0 = np.clip(c0, 0, ncams - track_len)
    obs_cam = (c0[:, None] + np.arange(track_len)[None, :]).reshape(-1).astype(np.int32)
    obs_pt = np.repeat(np.arange(npts, dtype=np.int32), track_len)
    p = np.einsum('nij,nj->ni', R[obs_cam], X[obs_pt]) + t[obs_cam]
    z = p[:, :2] / p[:, 2:3] + rs.randn(len(obs_cam), 2) * msm_noise

    outliers = np.zeros(len(obs_cam), bool)
    if outlier_frac > 0:
        ro = np.random.RandomState(outlier_seed)
        idx = ro.permutation(len(obs_cam))[:int(outlier_frac * len(obs_cam))]
        outliers[idx] = True
        z[idx] = ro.uniform(-outlier_range, outlier_range, (len(idx), 2))

    ri = np.random.RandomState(init_seed)
    dcam = ri.randn(ncams, 6) * init_perturbation
    dcam[0] = 0.
    R0 = np.einsum('nij,njk->nik', R, _so3_exp_batch(dcam[:, :3]))
    if init_mode == 'pose':
        t0 = -np.einsum('nij,nj->ni', R0, centers + dcam[:, 3:])
    else:
        t0 = t + dcam[:, 3:]
    X0 = X + ri.randn(npts, 3) * init_perturbation
    return dict(K=np.eye(3), R=R, t=t, X=X, R0=R0, t0=t0, X0=X0,
                obs_cam=obs_cam, obs_pt=obs_pt, obs_z=z, outliers=outliers)


def generate_collection_scene(ncams, npts, partners=20, track_len=4, seed=654, extent=3., msm_noise=.02, init_perturbation=.01,
                              init_seed=1888):
    """An unordered photo collection: cameras scattered over a (extent x extent) patch at z = 0, all looking roughly along +z
    at points in the slab z in [4, 8] above it; camera c shares tracks with `partners` cameras drawn at random from ALL the
    others (no locality, no order), and every point is seen by one camera and `track_len` - 1 of that camera's partners.  The
    co-visibility graph is a random graph of degree ~2 * partners: no camera order makes its reduced system a narrow band
    (the reference's dense S does not care, bundle_adjuster.py:259-312).  K = I, measurement noise N(0, msm_noise^2); the
    initial guess is the truth perturbed about each camera's own centre ('pose' mode of generate_banded_scene); camera 0 is
    the gauge camera.  Same dict as generate_banded_scene."""
    assert ncams > partners >= track_len - 1 >= 1
    rs = np.random.RandomState(seed)
    centers = np.c_[rs.rand(ncams, 2) * extent, rs.randn(ncams) * .02]
    R = _so3_exp_batch(rs.randn(ncams, 3) * .02)
    t = -np.einsum('nij,nj->ni', R, centers)
    part = np.empty((ncams, partners), np.int64)
    for c in range(ncams):
        o = rs.choice(ncams - 1, partners, replace=False)
        part[c] = o + (o >= c)                                   # (anyone but c itself)
    first = rs.randint(0, ncams, npts)
    order = np.argsort(first, kind='stable')                     # (points ordered by their first camera: as a track database would list them)
    first = first[order]
    pick = np.argsort(rs.rand(npts, partners), axis=1)[:, :track_len - 1]
    cams = np.c_[first, part[first[:, None], pick]]
    cams.sort(axis=1)
    X = np.c_[rs.rand(npts, 2) * extent, 4 + 4 * rs.rand(npts)]
    obs_cam = cams.reshape(-1).astype(np.int32)
    obs_pt = np.repeat(np.arange(npts, dtype=np.int32), track_len)
    p = np.einsum('nij,nj->ni', R[obs_cam], X[obs_pt]) + t[obs_cam]
    z = p[:, :2] / p[:, 2:3] + rs.randn(len(obs_cam), 2) * msm_noise
    ri = np.random.RandomState(init_seed)
    dcam = ri.randn(ncams, 6) * init_perturbation
    dcam[0] = 0.
    R0 = np.einsum('nij,njk->nik', R, _so3_exp_batch(dcam[:, :3]))
    t0 = -np.einsum('nij,nj->ni', R0, centers + dcam[:, 3:])
    X0 = X + ri.randn(npts, 3) * init_perturbation
    return dict(K=np.eye(3), R=R, t=t, X=X, R0=R0, t0=t0, X0=X0, obs_cam=obs_cam, obs_pt=obs_pt, obs_z=z,
                outliers=np.zeros(len(obs_cam), bool))


def add_loop_closure_tracks(s, pairs, width=1, seed=77, msm_noise=.02, init_perturbation=.01):
    """The scene with extra tracks that tie far-apart cameras together (a loop closure): for every (i, j) of `pairs` one new
    point, between the two cameras in x, observed by cameras i .. i + width - 1 and j .. j + width - 1 (measurements from the
    true parameters + noise; K = I, so a camera "sees" whatever is in front of it).  A block-banded reduced system stops being
    one: the reference's dense S does not care (bundle_adjuster.py:259-278), a band does."""
    rs = np.random.RandomState(seed)
    nt = len(s['X'])
    cams, pts, Xn = [], [], []
    for k, (i, j) in enumerate(pairs):
        ci = np.r_[np.arange(i, i + width), np.arange(j, j + width)]
        centres = -np.einsum('nji,nj->ni', s['R'][ci], s['t'][ci])
        x = np.array([centres[:, 0].mean(), rs.rand() * 2 - 1, 4 + 4 * rs.rand()])
        Xn.append(x)
        cams.append(ci)
        pts.append(np.full(len(ci), nt + k))
    cams, pts, Xn = np.concatenate(cams).astype(np.int32), np.concatenate(pts).astype(np.int32), np.array(Xn)
    p = np.einsum('nij,nj->ni', s['R'][cams], Xn[pts - nt]) + s['t'][cams]
    z = p[:, :2] / p[:, 2:3] + rs.randn(len(cams), 2) * msm_noise
    out = dict(s)
    out.update(X=np.vstack((s['X'], Xn)), X0=np.vstack((s['X0'], Xn + rs.randn(*Xn.shape) * init_perturbation)),
               obs_cam=np.concatenate((s['obs_cam'], cams)), obs_pt=np.concatenate((s['obs_pt'], pts)),
               obs_z=np.vstack((s['obs_z'], z)), outliers=np.concatenate((s['outliers'], np.zeros(len(cams), bool))))
    return out


def _so3_exp_batch(w):
    th = np.sqrt(np.sum(w * w, axis=1))
    small = th < 1e-8
    th = np.where(small, 1., th)
    A = np.sin(th) / th
    B = (1 - np.cos(th)) / (th * th)
    Kx = np.zeros((len(w), 3, 3))
    Kx[:, 0, 1], Kx[:, 0, 2] = -w[:, 2], w[:, 1]
    Kx[:, 1, 0], Kx[:, 1, 2] = w[:, 2], -w[:, 0]
    Kx[:, 2, 0], Kx[:, 2, 1] = -w[:, 1], w[:, 0]
    out = np.eye(3) + A[:, None, None] * Kx + B[:, None, None] * (Kx @ Kx)
    out[small] = np.eye(3)
    return out


def reprojection_rmse(bundle_or_errors):
    """sqrt(sum |pr(K(Rx+t)) - z|^2 / N) over all observations (raw, un-robustified)."""
    e = bundle_or_errors.reproj_errors() if hasattr(bundle_or_errors, 'reproj_errors') \
        else np.asarray(bundle_or_errors)
    return float(np.sqrt(np.sum(e * e) / max(1, len(e))))
