"""CPU oracle for the pysfm bundle-adjustment inner loop.

TEST INFRASTRUCTURE ONLY.  Nothing under ``pysfm_amd/`` may import this
module; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` use it, as the checker - never as the thing measured or
shipped.

This is an array-native NumPy (fp64) *restatement* of the reference algorithm
(alexflint/pysfm).  Each function cites the reference ``file:line`` it
follows.  Parity pin: ``tests/test_oracle_golden.py`` checks every function
here against golden vectors that ``oracle/gen_golden.py`` captured from the
reference itself (imported in the build container after an out-of-tree
``lib2to3`` translation; see that script).  The one exception is the Huber
model, which does not exist in the reference: **Huber parity is unpinned** and
is checked only through the reference's own ``sensor_model.validate`` criteria.

Data model (all fp64 / int32, row-major):
  K[3,3]; R[nc,3,3]; t[nc,3]; X[nt,3]
  obs_cam[N], obs_pt[N]  positions into the *selected* camera / track lists
  obs_z[N,2]             measurements
  observations are ordered by track position, then camera position - the
  order the reference's loops visit them (bundle_adjuster.py:222-226).
  cam_opt_pos[nc]        position in the optimised-camera list, or -1
  pt_opt[nt]             bool: track is in optim_track_ids
"""
import numpy as np

GAUSS, CAUCHY, HUBER, MODEL = 0, 1, 2, 3      # MODEL: any object with the four-method protocol (Sensor.from_model)
CAUCHY_LINEAR_WINDOW = 1e-5       # sensor_model.py:39
SO3_EXP_EPS = 1e-8                # lie.py:26


# --------------------------------------------------------------------------
# sensor models (sensor_model.py:7-72)
# --------------------------------------------------------------------------
class Sensor(object):
    """kind + parameters.  GAUSS: L (2x2 lower Cholesky factor of cov^-1,
    sensor_model.py:16-17).  CAUCHY: sigma (sensor_model.py:41-43).
    HUBER: k (not in the reference)."""

    def __init__(self, kind, L=None, sigma=1.0, k=1.0):
        self.kind = kind
        self.L = np.eye(2) if L is None else np.asarray(L, float)
        self.sigma = float(sigma)
        self.k = float(k)

    @classmethod
    def gaussian(cls, cov=1.0):
        # sensor_model.py:8-17
        if np.isscalar(cov):
            cov = cov * np.eye(2)
        elif np.shape(cov) == (2,):
            cov = np.diag(cov)
        cov = np.asarray(cov, float)
        return cls(GAUSS, L=np.linalg.cholesky(np.linalg.inv(cov)))

    @classmethod
    def cauchy(cls, sigma):
        return cls(CAUCHY, sigma=sigma)

    @classmethod
    def huber(cls, k):
        return cls(HUBER, k=k)

    @classmethod
    def from_model(cls, model):
        """Any object with the reference's protocol (sensor_model.py:19-32), called once per observation exactly as
        Bundle.residual / Bundle.Jresidual call it (bundle.py:251-252, 269-273).  Small scenes only: a Python loop."""
        s = cls(MODEL)
        s.model = model
        return s


def sensor_residual(sensor, e):
    """r = residual_from_error(e) for e[N,2].
    Gaussian sensor_model.py:23-25; Cauchy sensor_model.py:48-57."""
    e = np.asarray(e, float)
    if sensor.kind == GAUSS:
        return e @ sensor.L.T
    if sensor.kind == MODEL:
        return np.array([sensor.model.residual_from_error(x) for x in e], float).reshape(-1, 2)
    rho = np.sqrt(np.sum(e * e, axis=1))
    if sensor.kind == CAUCHY:
        s = sensor.sigma
        small = rho < CAUCHY_LINEAR_WINDOW
        safe = np.where(small, 1.0, rho)
        g = np.sqrt(np.log(1.0 + safe * safe / (s * s)))
        out = e * (g / safe)[:, None]
        out[small] = e[small] / s
        return out
    if sensor.kind == HUBER:
        k = sensor.k
        inside = rho <= k
        safe = np.where(inside, 1.0, rho)
        g = np.sqrt(np.maximum(2.0 * k * safe - k * k, 0.0))
        out = e * (g / safe)[:, None]
        out[inside] = e[inside]
        return out
    raise ValueError(sensor.kind)


def sensor_jacobian(sensor, e):
    """Jr[N,2,2] = Jresidual_from_error(e).
    Gaussian sensor_model.py:27-29; Cauchy sensor_model.py:59-69."""
    e = np.asarray(e, float)
    n = len(e)
    I = np.eye(2)
    if sensor.kind == GAUSS:
        return np.broadcast_to(sensor.L, (n, 2, 2)).copy()
    if sensor.kind == MODEL:
        return np.array([sensor.model.Jresidual_from_error(x) for x in e], float).reshape(-1, 2, 2)
    rho = np.sqrt(np.sum(e * e, axis=1))
    ee = e[:, :, None] * e[:, None, :]
    if sensor.kind == CAUCHY:
        s2 = sensor.sigma * sensor.sigma
        small = rho < CAUCHY_LINEAR_WINDOW
        safe = np.where(small, 1.0, rho)
        g = np.sqrt(np.log(1.0 + safe * safe / s2))
        g = np.where(small, 1.0, g)
        a = 1.0 / (safe * g * (safe * safe + s2))
        J = ee * a[:, None, None] + \
            (safe[:, None, None] * I - ee / safe[:, None, None]) * (g / (safe * safe))[:, None, None]
        J[small] = I / sensor.sigma
        return J
    if sensor.kind == HUBER:
        k = sensor.k
        inside = rho <= k
        safe = np.where(inside, 1.0, rho)
        rho_h = np.where(inside, 1.0, 2.0 * k * safe - k * k)
        g = np.sqrt(rho_h)
        gp = k / g                       # d sqrt(2ks-k^2)/ds
        J = (g / safe)[:, None, None] * I + ee * ((gp * safe - g) / safe ** 3)[:, None, None]
        J[inside] = I
        return J
    raise ValueError(sensor.kind)


def sensor_cost(sensor, e):
    """cost_from_error, sensor_model.py:19-21 / 45-46 (only used by validate)."""
    e = np.asarray(e, float)
    if sensor.kind == GAUSS:
        covinv = sensor.L @ sensor.L.T               # L = chol(cov^-1), sensor_model.py:16-17
        return np.einsum('ni,ij,nj->n', e, covinv, e)
    s2 = np.sum(e * e, axis=1)
    if sensor.kind == CAUCHY:
        return np.log(1.0 + s2 / sensor.sigma ** 2)
    rho = np.sqrt(s2)
    return np.where(rho <= sensor.k, s2, 2 * sensor.k * rho - sensor.k ** 2)


# --------------------------------------------------------------------------
# projection + per-observation Jacobian blocks
# --------------------------------------------------------------------------
def skew(m):
    """algebra.py:51-56, batched over the leading axis."""
    m = np.asarray(m, float)
    out = np.zeros(m.shape[:-1] + (3, 3))
    out[..., 0, 1] = -m[..., 2]
    out[..., 0, 2] = m[..., 1]
    out[..., 1, 0] = m[..., 2]
    out[..., 1, 2] = -m[..., 0]
    out[..., 2, 0] = -m[..., 1]
    out[..., 2, 1] = m[..., 0]
    return out


def homogeneous_prediction(K, R, t, X, obs_cam, obs_pt):
    """p = K (R x + t) per observation (bundle.py:260)."""
    Rx = np.einsum('nij,nj->ni', R[obs_cam], X[obs_pt])
    return (Rx + t[obs_cam]) @ K.T


def reproj_error(K, R, t, X, obs_cam, obs_pt, obs_z):
    """e = pr(K(Rx+t)) - z.  algebra.py:5-12, bundle.py:14-19, 243-248."""
    p = homogeneous_prediction(K, R, t, X, obs_cam, obs_pt)
    return p[:, :2] / p[:, 2:3] - obs_z


def residuals(sensor, K, R, t, X, obs_cam, obs_pt, obs_z):
    """bundle.py:251-252."""
    return sensor_residual(sensor, reproj_error(K, R, t, X, obs_cam, obs_pt, obs_z))


def jacobians(sensor, K, R, t, X, obs_cam, obs_pt, obs_z):
    """(r[N,2], Jc[N,2,6], Jp[N,2,3]) per bundle.py:255-277:
    Jpr (bundle.py:8-11), J_t = Jpr K, J_x = J_t R, J_R = J_x skew(-x)
    (lie.py:38-40), chained through the sensor model (bundle.py:269-273)."""
    p = homogeneous_prediction(K, R, t, X, obs_cam, obs_pt)
    n = len(p)
    Jpr = np.zeros((n, 2, 3))
    Jpr[:, 0, 0] = 1.0 / p[:, 2]
    Jpr[:, 1, 1] = 1.0 / p[:, 2]
    Jpr[:, 0, 2] = -p[:, 0] / (p[:, 2] * p[:, 2])
    Jpr[:, 1, 2] = -p[:, 1] / (p[:, 2] * p[:, 2])
    J_t = Jpr @ K
    J_x = J_t @ R[obs_cam]
    J_R = J_x @ skew(-X[obs_pt])
    e = p[:, :2] / p[:, 2:3] - obs_z
    Jr = sensor_jacobian(sensor, e)
    Jc = Jr @ np.concatenate((J_R, J_t), axis=2)
    Jp = Jr @ J_x
    return sensor_residual(sensor, e), Jc, Jp


# --------------------------------------------------------------------------
# cost (bundle_adjuster.py:165-171)
# --------------------------------------------------------------------------
def cost(sensor, K, R, t, X, obs_cam, obs_pt, obs_z, cam_opt_pos, pt_opt):
    """sum ||r||^2 over observations whose camera AND track are optimised."""
    r = residuals(sensor, K, R, t, X, obs_cam, obs_pt, obs_z)
    m = (np.asarray(cam_opt_pos)[obs_cam] >= 0) & np.asarray(pt_opt, bool)[obs_pt]
    return float(np.sum(r[m] * r[m]))


def complete_cost(sensor, K, R, t, X, obs_cam, obs_pt, obs_z):
    """bundle.py:293-295."""
    r = residuals(sensor, K, R, t, X, obs_cam, obs_pt, obs_z)
    return float(np.sum(r * r))


# --------------------------------------------------------------------------
# normal-equation blocks (bundle_adjuster.py:211-234)
# --------------------------------------------------------------------------
def normal_blocks(sensor, K, R, t, X, obs_cam, obs_pt, obs_z, nc, nt):
    """HCC[nc,6,6], HPP[nt,3,3], W[N,6,3] (the nonzero HCP blocks, one per
    observation), bC[nc,6], bP[nt,3]."""
    r, Jc, Jp = jacobians(sensor, K, R, t, X, obs_cam, obs_pt, obs_z)
    HCC = np.zeros((nc, 6, 6))
    HPP = np.zeros((nt, 3, 3))
    bC = np.zeros((nc, 6))
    bP = np.zeros((nt, 3))
    JcT = np.transpose(Jc, (0, 2, 1))
    JpT = np.transpose(Jp, (0, 2, 1))
    np.add.at(HCC, obs_cam, JcT @ Jc)
    np.add.at(HPP, obs_pt, JpT @ Jp)
    np.add.at(bC, obs_cam, np.einsum('nij,nj->ni', JcT, r))
    np.add.at(bP, obs_pt, np.einsum('nij,nj->ni', JpT, r))
    W = JcT @ Jp
    return HCC, HPP, W, bC, bP


def damp_blocks(H, damping):
    """optimize.py:7-9 applied per block (bundle_adjuster.py:238-242)."""
    H = H.copy()
    n = H.shape[-1]
    idx = np.arange(n)
    H[:, idx, idx] *= (1.0 + damping)
    return H


def invert_point_blocks(HPP_damped, rcond):
    """bundle_adjuster.py:252-256: pinv(HPP, rcond), or inv if rcond is None."""
    if rcond is None:
        return np.linalg.inv(HPP_damped)
    if not len(HPP_damped):
        return HPP_damped.copy()
    return np.linalg.pinv(HPP_damped, rcond)          # stacked: one LAPACK SVD per 3x3 block


def schur_complement(HCC_d, HPP_inv, W, bC, bP, obs_cam, obs_pt, cam_opt_pos,
                     chunk_points=20000):
    """(S[nco,nco,6,6], b[nco,6]) per bundle_adjuster.py:259-278, visiting only
    the nonzero HCP blocks.  HCC_d is already damped (bundle_adjuster.py:199-201)."""
    cam_opt_pos = np.asarray(cam_opt_pos)
    obs_cam, obs_pt = np.asarray(obs_cam), np.asarray(obs_pt)
    if len(obs_pt) > 1 and np.any(np.diff(obs_pt) < 0):     # the pair enumeration below walks the observations point by
        order = np.argsort(obs_pt, kind='stable')             # point: bring them into that order (the sums do not care)
        obs_cam, obs_pt, W = obs_cam[order], obs_pt[order], W[order]
    nco = int(np.sum(cam_opt_pos >= 0))
    S = np.zeros((nco, nco, 6, 6))
    b = np.zeros((nco, 6))
    opt = np.nonzero(cam_opt_pos >= 0)[0]
    S[cam_opt_pos[opt], cam_opt_pos[opt]] = HCC_d[opt]      # :263-265
    b[cam_opt_pos[opt]] = bC[opt]

    pos = cam_opt_pos[obs_cam]
    keep = pos >= 0
    T = W @ HPP_inv[obs_pt]                                   # W_i HPPinv_k
    np.subtract.at(b, pos[keep], np.einsum('nij,nj->ni', T[keep], bP[obs_pt[keep]]))  # :269

    # all ordered pairs of observations that share a point (:270-276)
    N = len(obs_pt)
    if N == 0:
        return S, b
    nt = int(obs_pt.max()) + 1
    counts = np.bincount(obs_pt, minlength=nt)
    starts = np.concatenate(([0], np.cumsum(counts)))
    for p0 in range(0, nt, chunk_points):
        p1 = min(nt, p0 + chunk_points)
        L = counts[p0:p1]
        if L.sum() == 0:
            continue
        # for every observation a in the chunk, pair it with all obs b of its point
        a_idx = np.arange(starts[p0], starts[p1])
        La = L[obs_pt[a_idx] - p0]
        A = np.repeat(a_idx, La)
        first = starts[obs_pt[a_idx]]
        off = np.arange(len(A)) - np.repeat(np.cumsum(La) - La, La)
        B = np.repeat(first, La) + off
        ok = (pos[A] >= 0) & (pos[B] >= 0)
        A, B = A[ok], B[ok]
        contrib = T[A] @ np.transpose(W[B], (0, 2, 1))
        np.subtract.at(S, (pos[A], pos[B]), contrib)
    return S, b


def flatten_reduced(S, b):
    """bundle_adjuster.py:290-291."""
    nco = S.shape[0]
    return S.transpose((0, 2, 1, 3)).reshape((nco * 6, nco * 6)), b.reshape(-1)


class NormalEquationsIllconditioned(Exception):
    """bundle_adjuster.py:27-30."""


def solve_reduced(S, b, cam_param_mask):
    """bundle_adjuster.py:281-312 (LAPACK gesv through numpy.linalg.solve)."""
    nco = S.shape[0]
    AC, bCf = flatten_reduced(S, b)
    m = np.asarray(cam_param_mask, bool)
    A = AC[m].T[m].T
    try:
        d = np.linalg.solve(A, bCf[m])
    except np.linalg.LinAlgError:
        raise NormalEquationsIllconditioned
    dC = np.zeros(nco * 6)
    dC[m] = d
    return dC.reshape(nco, 6)


def backsubstitute(dC, HPP_inv, W, bP, obs_cam, obs_pt, cam_opt_pos, nt):
    """dP[nt,3] for every track position (bundle_adjuster.py:316-331); the
    caller selects optim_track_indices."""
    pos = np.asarray(cam_opt_pos)[obs_cam]
    keep = pos >= 0
    acc = bP.copy()
    WT = np.transpose(W[keep], (0, 2, 1))
    np.subtract.at(acc, obs_pt[keep], np.einsum('nij,nj->ni', WT, dC[pos[keep]]))
    return np.einsum('nij,nj->ni', HPP_inv, acc)


def compute_update(sensor, K, R, t, X, obs_cam, obs_pt, obs_z, cam_opt_pos, pt_opt,
                   damping, cam_param_mask=None, rcond=1e-5, return_parts=False):
    """bundle_adjuster.py:176-208.  Returns (-dC[nco,6], -dP[nto,3])."""
    nc, nt = len(R), len(X)
    cam_opt_pos = np.asarray(cam_opt_pos)
    nco = int(np.sum(cam_opt_pos >= 0))
    if cam_param_mask is None:
        cam_param_mask = np.ones(nco * 6, bool)
    HCC, HPP, W, bC, bP = normal_blocks(sensor, K, R, t, X, obs_cam, obs_pt, obs_z, nc, nt)
    HCC_d = damp_blocks(HCC, damping)
    HPP_d = damp_blocks(HPP, damping)
    HPP_inv = invert_point_blocks(HPP_d, rcond)
    S, b = schur_complement(HCC_d, HPP_inv, W, bC, bP, obs_cam, obs_pt, cam_opt_pos)
    dC = solve_reduced(S, b, cam_param_mask)
    dP_all = backsubstitute(dC, HPP_inv, W, bP, obs_cam, obs_pt, cam_opt_pos, nt)
    dP = dP_all[np.asarray(pt_opt, bool)]
    if return_parts:
        return -dC, -dP, dict(HCC=HCC, HPP=HPP, W=W, bC=bC, bP=bP, HPP_inv=HPP_inv, S=S, b=b)
    return -dC, -dP


# --------------------------------------------------------------------------
# parameter update (bundle_adjuster.py:334-343, bundle.py:76-80, lie.py:21-34)
# --------------------------------------------------------------------------
def so3_exp(m):
    """Rodrigues, identity when |m| < 1e-8 (lie.py:21-34); batched."""
    m = np.asarray(m, float)
    th = np.sqrt(np.sum(m * m, axis=-1))
    small = th < SO3_EXP_EPS
    safe = np.where(small, 1.0, th)
    A = np.sin(safe) / safe
    B = (1.0 - np.cos(safe)) / (safe * safe)
    Kx = skew(m)
    out = np.eye(3) + A[..., None, None] * Kx + B[..., None, None] * (Kx @ Kx)
    out[small] = np.eye(3)
    return out


def apply_update(R, t, X, motion_update, structure_update, cam_opt_pos, pt_opt):
    """Trial parameters on copies: R <- R exp(d[:3]), t <- t + d[3:], x <- x + dx."""
    cam_opt_pos = np.asarray(cam_opt_pos)
    R2, t2, X2 = R.copy(), t.copy(), X.copy()
    opt = np.nonzero(cam_opt_pos >= 0)[0]
    d = motion_update[cam_opt_pos[opt]]
    R2[opt] = R[opt] @ so3_exp(d[:, :3])
    t2[opt] = t[opt] + d[:, 3:]
    X2[np.asarray(pt_opt, bool)] += structure_update
    return R2, t2, X2


# --------------------------------------------------------------------------
# LM loop (bundle_adjuster.py:117-162)
# --------------------------------------------------------------------------
def lm_optimize(sensor, K, R, t, X, obs_cam, obs_pt, obs_z, cam_opt_pos, pt_opt,
                cam_param_mask=None, max_steps=25, init_damping=10.0,
                improvement_threshold=1e-4, rcond=1e-5, trace=None):
    """Returns dict(R,t,X,costs,num_steps,converged,damping)."""
    args = (obs_cam, obs_pt, obs_z, cam_opt_pos, pt_opt)
    damping = init_damping
    num_steps = 0
    converged = False
    costs = [cost(sensor, K, R, t, X, *args)]
    while not converged and num_steps < max_steps:
        num_steps += 1
        cur_cost = cost(sensor, K, R, t, X, *args)
        while not converged and damping < 1e8:
            try:
                mu, su = compute_update(sensor, K, R, t, X, *args, damping=damping,
                                        cam_param_mask=cam_param_mask, rcond=rcond)
            except NormalEquationsIllconditioned:
                damping *= 10.0
                converged = damping > 1e8
                continue
            R2, t2, X2 = apply_update(R, t, X, mu, su, cam_opt_pos, pt_opt)
            next_cost = cost(sensor, K, R2, t2, X2, *args)
            if trace is not None:
                trace.append(dict(step=num_steps, damping=damping, cur=cur_cost, next=next_cost))
            if next_cost < cur_cost:
                damping *= 0.1
                R, t, X = R2, t2, X2
                costs.append(next_cost)
                converged = abs(cur_cost - next_cost) < improvement_threshold
                break
            else:
                damping *= 10.0
                converged = damping > 1e8
    return dict(R=R, t=t, X=X, costs=costs, num_steps=num_steps,
                converged=converged, damping=damping)


# --------------------------------------------------------------------------
# triangulation (triangulate.py:6-18, bundle.py:313-321)
# --------------------------------------------------------------------------
def triangulate_all(K, R, t, obs_cam, obs_pt, obs_z, nt):
    """Per point: numpy.linalg.lstsq on the 2L x 3 algebraic system."""
    X = np.zeros((nt, 3))
    order = np.argsort(obs_pt, kind='stable')
    cam, pt, z = obs_cam[order], obs_pt[order], obs_z[order]
    bounds = np.searchsorted(pt, np.arange(nt + 1))
    for k in range(nt):
        s, e = bounds[k], bounds[k + 1]
        if e == s:
            continue
        A = np.empty((2 * (e - s), 3))
        b = np.empty(2 * (e - s))
        for i, n in enumerate(range(s, e)):
            Ri, ti = R[cam[n]], t[cam[n]]
            b[2 * i] = (z[n, 0] * K[2] - K[0]) @ ti
            b[2 * i + 1] = (z[n, 1] * K[2] - K[1]) @ ti
            A[2 * i] = (K[0] - z[n, 0] * K[2]) @ Ri
            A[2 * i + 1] = (K[1] - z[n, 1] * K[2]) @ Ri
        X[k] = np.linalg.lstsq(A, b, rcond=None)[0]
    return X


# --------------------------------------------------------------------------
# dense-matrix oracle (schur.py:4-44, bundle.py:452-505) - small scenes only
# --------------------------------------------------------------------------
def dense_jacobian(sensor, K, R, t, X, obs_cam, obs_pt, obs_z):
    """(r[2N], J[2N, 6nc+3nt]) in observation order (bundle.py:482-505)."""
    nc, nt = len(R), len(X)
    r, Jc, Jp = jacobians(sensor, K, R, t, X, obs_cam, obs_pt, obs_z)
    N = len(obs_cam)
    J = np.zeros((2 * N, 6 * nc + 3 * nt))
    for n in range(N):
        J[2 * n:2 * n + 2, 6 * obs_cam[n]:6 * obs_cam[n] + 6] = Jc[n]
        c0 = 6 * nc + 3 * obs_pt[n]
        J[2 * n:2 * n + 2, c0:c0 + 3] = Jp[n]
    return r.reshape(-1), J


def dense_schur_complement(A, b, n):
    """schur.py:4-11: A11 - A12 A22^-1 A21 with a plain inverse."""
    m = A.shape[0] - n
    V = A[:n, -m:] @ np.linalg.inv(A[-m:, -m:])
    return A[:n, :n] - V @ A[-m:, :n], b[:n] - V @ b[-m:]


# --------------------------------------------------------------------------
# literal per-observation loop (the shape of the reference's own hot loop;
# used as the 1-core CPU baseline "port" in bench.py and on tiny cases)
# --------------------------------------------------------------------------
def normal_blocks_loop(sensor, K, R, t, X, obs_cam, obs_pt, obs_z, nc, nt):
    """bundle_adjuster.py:211-234 one observation at a time."""
    HCC = np.zeros((nc, 6, 6))
    HPP = np.zeros((nt, 3, 3))
    bC = np.zeros((nc, 6))
    bP = np.zeros((nt, 3))
    W = np.zeros((len(obs_cam), 6, 3))
    one = np.arange(1)
    for n in range(len(obs_cam)):
        i, j = int(obs_cam[n]), int(obs_pt[n])
        r, Jc, Jp = jacobians(sensor, K, R[i:i + 1], t[i:i + 1], X[j:j + 1],
                              one * 0, one * 0, obs_z[n:n + 1])
        r, Jc, Jp = r[0], Jc[0], Jp[0]
        HCC[i] += Jc.T @ Jc
        HPP[j] += Jp.T @ Jp
        W[n] = Jc.T @ Jp
        bC[i] += Jc.T @ r
        bP[j] += Jp.T @ r
    return HCC, HPP, W, bC, bP
