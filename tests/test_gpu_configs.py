"""BASELINE configurations 4 and 5 at full size, scenes handed over in arbitrary track order, and the sharded path
on a config-5-shaped scene - through the C ABI, against the CPU oracle where it finishes in seconds and through
size-independent properties elsewhere.  Needs a real MI355X: run with ``-m gpu``.

Tolerances as in test_gpu_parity.py (fp64 everywhere): blocks / S / b 1e-11 ... 1e-10 relative to the largest entry,
solves 1e-8 ... 1e-9, LM trajectories 1e-6 (the north-star tolerance).
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from oracle import ba_oracle as O
from test_gpu_parity import DEFAULT_OPTIONS, banded, close, default_flags, load_problem, sensor_params  # noqa: F401

pytestmark = pytest.mark.gpu

TIGHT = 1e-11


@pytest.fixture(scope='module')
def be():
    from pysfm_amd.backend import HipBackend
    b = HipBackend(0)
    yield b
    b.close()


@pytest.fixture(autouse=True)
def _default_options(request):
    yield
    if 'be' in request.fixturenames:
        b = request.getfixturevalue('be')
        for k, v in DEFAULT_OPTIONS.items():
            b.set_option(k, v)


def shuffled(s, seed=7):
    """The same scene with the tracks renumbered at random and the observations in random order."""
    rs = np.random.RandomState(seed)
    nt = len(s['X0'])
    new_id = rs.permutation(nt)
    X0 = np.empty_like(s['X0'])
    X0[new_id] = s['X0']
    o = rs.permutation(len(s['obs_cam']))
    out = dict(s)
    out.update(X0=X0, obs_cam=s['obs_cam'][o], obs_pt=new_id[s['obs_pt'][o]].astype(np.int32), obs_z=s['obs_z'][o])
    return out, new_id, o


# ------------------------------------------------------------------ any track order (bundle_adjuster.py:222-226)
@pytest.mark.parametrize('sensor,L', [(O.Sensor.gaussian(1.), 10), (O.Sensor.cauchy(.05), 7), (O.Sensor.huber(.06), 12)])
def test_shuffled_tracks_full_step_vs_oracle(be, sensor, L):
    """Tracks and observations in random order, a frozen camera in the middle, tracks that are not optimised:
    every host-facing array must come back in the CALLER's order and agree with the oracle on the same arrays."""
    nc, nt = 40, 1500
    s, new_id, o = shuffled(banded(nc, nt, track_len=L, outlier_frac=.03))
    cam_opt_pos = np.arange(nc, dtype=np.int32) - 1
    cam_opt_pos[17] = -1
    cam_opt_pos[18:] -= 1
    pt_opt = np.ones(nt, np.uint8)
    pt_opt[::7] = 0
    a = (s['K'], s['R0'], s['t0'], s['X0'], s['obs_cam'], s['obs_pt'], s['obs_z'])
    load_problem(be, *a, cam_opt_pos, pt_opt, sensor)
    info = be.problem_info()
    assert info['points_permuted'] == 1 and info['obs_permuted'] == 1
    assert info['point_groups'] == 1 and info['schur_groups'] == 1         # the grouped kernels, despite the order
    # per-observation values in the caller's observation order
    ev = be.eval_observations(0)
    r0, Jc0, Jp0 = O.jacobians(sensor, *a)
    close(ev['e'], O.reproj_error(*a), 1e-13)
    close(ev['r'], r0, 1e-13)
    close(ev['Jc'], Jc0, 1e-12)
    close(ev['Jp'], Jp0, 1e-12)
    close(be.cost(0), O.cost(sensor, *a, cam_opt_pos, pt_opt), 1e-12)
    mu, su, parts = O.compute_update(sensor, *a, cam_opt_pos, pt_opt, damping=3., return_parts=True)
    be.linearize(0, store_W=True)
    blk = be.get_blocks(W=True)
    for k in ('HCC', 'bC', 'HPP', 'bP', 'W'):
        close(blk[k], parts[k], TIGHT)
    be.schur(0, 3., 1e-5)
    S, b = be.get_reduced()
    close(S, parts['S'], TIGHT)
    close(b, parts['b'], TIGHT)
    close(be.get_point_inverses(), parts['HPP_inv'], 1e-10)
    be.solve_reduced(None)
    dC = be.get_solution()
    dP = be.backsubstitute(0)
    close(-dC, mu, 1e-8)
    close(-dP[pt_opt.astype(bool)], su, 1e-8)
    # the whole trial in one call, and the trial parameter set in the caller's order
    infoT, cost = be.lm_trial(3., 1e-5, None)
    assert infoT == 0
    R2, t2, X2 = O.apply_update(s['R0'], s['t0'], s['X0'], mu, su, cam_opt_pos, pt_opt)
    Rg, tg, Xg = be.get_params(1)
    close(Xg, X2, 1e-9)
    close(tg, t2, 1e-9)
    close(cost, O.cost(sensor, s['K'], R2, t2, X2, *a[4:], cam_opt_pos, pt_opt), 1e-8)
    # caller-supplied structure update goes through the permutation too
    delta = np.random.RandomState(1).randn(nt, 3) * 1e-3
    be.apply_update(0, 1, np.zeros((be.nco, 6)), delta)
    close(be.get_params(1)[2], s['X0'] + delta * pt_opt[:, None], 1e-14)
    # triangulation writes X in the caller's order
    Xt = be.triangulate(1)
    close(Xt, O.triangulate_all(s['K'], s['R0'], s['t0'], s['obs_cam'], s['obs_pt'], s['obs_z'], nt), 1e-6)


def test_internal_order_is_invisible(be):
    """The same scene sorted and shuffled, with and without the internal sort: identical S, b (as sets of blocks the
    band layout is the same) and updates mapped through the renumbering."""
    nc, nt = 60, 3000
    s0 = banded(nc, nt, track_len=8)
    s1, new_id, o = shuffled(s0)
    flags = default_flags(nc, nt)
    res = []
    for s, sort in ((s0, 1), (s0, 0), (s1, 1)):
        be.set_option('sort_points', sort)
        load_problem(be, s['K'], s['R0'], s['t0'], s['X0'], s['obs_cam'], s['obs_pt'], s['obs_z'], *flags, O.Sensor.gaussian(1.))
        be.linearize(0)
        be.schur(0, 2., 1e-5)
        S, b = be.get_reduced()
        be.solve_reduced(None)
        res.append((S, b, be.get_solution(), be.backsubstitute(0), be.problem_info()))
    assert res[0][4]['points_permuted'] == 0 and res[1][4]['points_permuted'] == 0 and res[2][4]['points_permuted'] == 1
    for r in res[1:]:
        close(r[0], res[0][0], 1e-12)
        close(r[1], res[0][1], 1e-12)
        close(r[2], res[0][2], 1e-9)
    close(res[1][3], res[0][3], 1e-9)
    close(res[2][3][new_id], res[0][3], 1e-9)


def test_shuffled_config3_takes_the_matrix_core_path(be):
    """BASELINE config 3 with its tracks in random order: same kernels as the sorted scene (the verdict's cliff), same
    reduced system, same LM trajectory."""
    from pysfm_amd import Bundle, BundleAdjuster
    nc, nt = 1000, 100000
    s0 = banded(nc, nt)
    s1, new_id, o = shuffled(s0)
    flags = default_flags(nc, nt)
    bands = []
    for s in (s0, s1):
        load_problem(be, s['K'], s['R0'], s['t0'], s['X0'], s['obs_cam'], s['obs_pt'], s['obs_z'], *flags, O.Sensor.gaussian(1.))
        info = be.problem_info()
        assert info['schur_mfma'] == 1 and info['point_groups'] == 1 and info['half_bandwidth'] == 9
        i2, c2 = be.lm_trial(10., 1e-5, None)
        assert i2 == 0
        St, bt = be.reduced_tensors()
        bands.append((St.cpu().numpy().copy(), bt.cpu().numpy().copy(), c2, info))
    assert bands[0][3]['mfma_groups'] == bands[1][3]['mfma_groups']
    close(bands[1][0], bands[0][0], 1e-11)
    close(bands[1][1], bands[0][1], 1e-11)
    assert abs(bands[1][2] - bands[0][2]) <= 1e-10 * bands[0][2]
    costs = []
    for s in (s0, s1):
        b0 = Bundle.FromObservations(s['K'], s['R0'], s['t0'], s['X0'], s['obs_cam'], s['obs_pt'], s['obs_z'])
        ba = BundleAdjuster(b0, verbose=False)
        ba.optimize(max_steps=6)
        costs.append(np.array(ba.costs))
        ba.backend.close()
    assert len(costs[0]) == len(costs[1])
    close(costs[1], costs[0], 1e-6)


# ------------------------------------------------------------------ ba_set_problem on the device (ba_setup_kernels.h)
def test_device_side_setup_edge_cases(be):
    """The device validates and orders whatever arrives: a repeated (camera, track) pair hidden in a shuffled input, an index
    out of range in the middle of a long input, tracks without observations and tracks seen by frozen cameras only (they go
    last, add nothing to S), a camera / track count whose sort key does not fit 32 bits, and the caller's order kept when it is
    already good.  Every accepted scene must give the oracle's trial whatever order it came in."""
    sensor = O.Sensor.gaussian(1.)
    s = banded(90, 2700, track_len=7)
    cam, pt, z = s['obs_cam'].copy(), s['obs_pt'].copy(), s['obs_z'].copy()
    flags = default_flags(90, 2700)
    K = s['K']
    o = np.random.RandomState(5).permutation(len(cam))
    # (1) duplicates and range errors, sorted and shuffled
    for order in (np.arange(len(cam)), o):
        c2, p2 = np.r_[cam[order], cam[order][777]], np.r_[pt[order], pt[order][777]]
        with pytest.raises(ValueError, match='two observations'):
            be.set_problem(90, 2700, c2, p2, np.r_[z[order], z[order][777:778]], K, *flags)
        c3 = cam[order].copy()
        c3[5000] = 90
        with pytest.raises(ValueError, match='out of range'):
            be.set_problem(90, 2700, c3, pt[order], z[order], K, *flags)
        p3 = pt[order].copy()
        p3[123] = -1
        with pytest.raises(ValueError, match='out of range'):
            be.set_problem(90, 2700, cam[order], p3, z[order], K, *flags)
    # (2) empty tracks, tracks of frozen cameras only, a frozen camera in the middle; shuffled
    cam_opt_pos = np.arange(90, dtype=np.int32) - 1
    cam_opt_pos[40] = -1
    cam_opt_pos[41:] -= 1
    keep = ~np.isin(pt, [3, 4, 2699])                                       # three tracks lose all their observations
    only_frozen = (pt == 10) & ~np.isin(cam, [0, 40])                        # ... and one keeps those of the frozen cameras only (if any)
    keep &= ~only_frozen
    a = (K, s['R0'], s['t0'], s['X0'], cam[keep], pt[keep], z[keep])
    oo = np.random.RandomState(6).permutation(int(keep.sum()))
    b = (K, s['R0'], s['t0'], s['X0'], cam[keep][oo], pt[keep][oo], z[keep][oo])
    mu, su, parts = O.compute_update(sensor, *a, cam_opt_pos, flags[1], damping=2., return_parts=True)
    for arrays in (a, b):
        load_problem(be, *arrays, cam_opt_pos, flags[1], sensor)
        be.linearize(0)
        be.schur(0, 2., 1e-5)
        S, bb = be.get_reduced()
        close(S, parts['S'], TIGHT)
        close(bb, parts['b'], TIGHT)
        info, cost = be.lm_trial(2., 1e-5, None)
        assert info == 0
        close(-be.get_solution(), mu, 1e-8)
    # (3) the caller's order is kept when it is as good as the sorted one, and only then
    load_problem(be, *a, cam_opt_pos, flags[1], sensor)
    assert be.problem_info()['points_permuted'] == 1                          # (the empty tracks move to the end)
    load_problem(be, K, s['R0'], s['t0'], s['X0'], cam, pt, z, *flags, sensor)
    assert be.problem_info()['points_permuted'] == 0 and be.problem_info()['obs_permuted'] == 0
    # (4) many cameras x many tracks: (track, rank) keys beyond 32 bits; a handful of observations, shuffled
    nc, nt = 70000, 40000
    rs = np.random.RandomState(8)
    pts = rs.choice(nt, 300, replace=False)
    c4 = (rs.randint(1, nc - 3, len(pts))[:, None] + np.arange(3)[None, :]).reshape(-1).astype(np.int32)      # (three consecutive cameras each: hb = 2)
    p4 = np.repeat(pts, 3).astype(np.int32)
    q = rs.permutation(len(c4))
    be.set_problem(nc, nt, c4[q], p4[q], rs.randn(len(c4), 2), K, *default_flags(nc, nt))
    info = be.problem_info()
    assert info['max_track_len'] == 3 and info['points_permuted'] == 1
    dup = np.r_[q, q[17]]
    with pytest.raises(ValueError, match='two observations'):
        be.set_problem(nc, nt, c4[dup], p4[dup], rs.randn(len(dup), 2), K, *default_flags(nc, nt))


# ------------------------------------------------------------------ config 4 at full size
@pytest.mark.parametrize('sensor', [O.Sensor.huber(.06), O.Sensor.cauchy(.05)], ids=['huber', 'cauchy'])
def test_config4_properties_1000x100k_outliers(be, sensor):
    """BASELINE config 4 (1000 cams / 100k pts / 1M obs, 10 % gross outliers, Huber k = .06; Cauchy sigma = .05 is the
    reference's own robustifier): cost, blocks and b against the oracle (O(N)), rows of S against the oracle on the
    tracks that touch them, symmetry / band structure, linearity over point shards."""
    nc, nt = 1000, 100000
    s = banded(nc, nt, outlier_frac=.1)
    assert abs(s['outliers'].mean() - .1) < 1e-3
    flags = default_flags(nc, nt)
    a = (s['K'], s['R0'], s['t0'], s['X0'], s['obs_cam'], s['obs_pt'], s['obs_z'])
    load_problem(be, *a, *flags, sensor)
    close(be.cost(0), O.cost(sensor, *a, *flags), 1e-12)
    HCC, HPP, W, bC, bP = O.normal_blocks(sensor, *a, nc, nt)
    be.linearize(0)
    blk = be.get_blocks()
    close(blk['HCC'], HCC, TIGHT)
    close(blk['bC'], bC, TIGHT)
    close(blk['HPP'], HPP, TIGHT)
    close(blk['bP'], bP, TIGHT)
    be.schur(0, 10., 1e-5)
    S, b = be.get_reduced()
    close(S, S.transpose(1, 0, 3, 2), 1e-13)
    i, j = np.nonzero(np.abs(S).sum(axis=(2, 3)))
    assert np.max(np.abs(i - j)) == 9
    HPPi = O.invert_point_blocks(O.damp_blocks(HPP, 10.), 1e-5)
    T = W @ HPPi[s['obs_pt']]
    pos = flags[0][s['obs_cam']]
    kk = pos >= 0
    b0 = bC[1:].copy()
    np.subtract.at(b0, pos[kk], np.einsum('nij,nj->ni', T[kk], bP[s['obs_pt'][kk]]))
    close(b, b0, 1e-10)
    rows = [0, 317, 998]
    touching = np.zeros(nt, bool)
    for r in rows:
        touching[s['obs_pt'][pos == r]] = True
    m = touching[s['obs_pt']]
    S0, _ = O.schur_complement(O.damp_blocks(HCC, 10.), HPPi, W[m], bC, bP, s['obs_cam'][m], s['obs_pt'][m], flags[0])
    for r in rows:
        close(S[r], S0[r], 1e-10)
    half = nt // 2
    parts = []
    for lo, hi in ((0, half), (half, nt)):
        m = (s['obs_pt'] >= lo) & (s['obs_pt'] < hi)
        be.set_problem(nc, hi - lo, s['obs_cam'][m], s['obs_pt'][m] - lo, s['obs_z'][m], s['K'], flags[0], np.ones(hi - lo, np.uint8))
        be.set_sensor(*sensor_params(sensor))
        be.set_params(0, s['R0'], s['t0'], s['X0'][lo:hi])
        be.linearize(0)
        be.schur(0, 10., 1e-5)
        parts.append(be.get_reduced())
    close(parts[0][0] + parts[1][0], S, 1e-11)
    close(parts[0][1] + parts[1][1], b, 1e-11)


@pytest.mark.parametrize('model', ['huber', 'cauchy'])
def test_config4_full_lm_rejects_the_outliers(model):
    """Full LM on config 4: the robust cost falls monotonically and the INLIERS end at the measurement noise although
    10 % of the observations are gross outliers (an L2 cost would be dragged far off)."""
    from pysfm_amd import Bundle, BundleAdjuster, sensor_model
    s = banded(1000, 100000, outlier_frac=.1)
    sm = sensor_model.HuberModel(.06) if model == 'huber' else sensor_model.CauchyModel(.05)
    b0 = Bundle.FromObservations(s['K'], s['R0'], s['t0'], s['X0'], s['obs_cam'], s['obs_pt'], s['obs_z'], sensor_model=sm)
    ba = BundleAdjuster(b0, verbose=False)
    ba.optimize(max_steps=25)
    assert all(c1 < c0 for c0, c1 in zip(ba.costs, ba.costs[1:]))
    e = ba.backend.eval_observations(0, e=True, r=False, Jc=False, Jp=False)['e']
    inl = ~s['outliers']
    rmse_in = float(np.sqrt(np.sum(e[inl] ** 2) / inl.sum()))
    assert rmse_in < 1.25 * .02 * np.sqrt(2), rmse_in
    ba.backend.close()


# ------------------------------------------------------------------ config 5 at full size on ONE GPU
@pytest.fixture(scope='module')
def config5():
    return banded(10000, 1000000)


def _chunked_cost(sensor, s, flags, chunk=2000000):
    tot = 0.
    for lo in range(0, len(s['obs_cam']), chunk):
        sl = slice(lo, lo + chunk)
        tot += O.cost(sensor, s['K'], s['R0'], s['t0'], s['X0'], s['obs_cam'][sl], s['obs_pt'][sl], s['obs_z'][sl], *flags)
    return tot


def test_config5_properties_10000x1M(be, config5):
    """BASELINE config 5 size (10 000 cams / 1M pts / 10M obs) unsharded on one GPU: cost vs the oracle; blocks, b and
    rows of S vs the oracle on the window of tracks that touch them; band structure; linearity over 8 point shards
    (what the 8 ranks would add up); the 1112-node cyclic reduction vs the sequential band Cholesky and vs LAPACK on a
    masked sub-block."""
    s = config5
    nc, nt = 10000, 1000000
    flags = default_flags(nc, nt)
    sensor = O.Sensor.gaussian(1.)
    load_problem(be, s['K'], s['R0'], s['t0'], s['X0'], s['obs_cam'], s['obs_pt'], s['obs_z'], *flags, sensor)
    info = be.problem_info()
    assert info['schur_mfma'] == 1 and info['point_groups'] == 1 and info['half_bandwidth'] == 9
    assert abs(be.cost(0) - _chunked_cost(sensor, s, flags)) <= 1e-11 * be.cost(0)
    be.linearize(0)
    be.schur(0, 10., 1e-5)
    be.synchronize()                                                  # (the kernels run on the backend's own stream, .cpu() on torch's)
    St, bt = be.reduced_tensors()
    band = St.cpu().numpy().reshape(nc - 1, 10, 6, 6).copy()
    bfull = bt.cpu().numpy().reshape(nc - 1, 6).copy()
    assert np.abs(band[:, 9]).max() > 0                               # the band is as wide as the tracks are long
    assert np.abs(band[-9:][np.arange(9)[:, None] + np.arange(10)[None, :] >= 9]).max() == 0      # nothing beyond the last camera
    close(band[:, 0], band[:, 0].transpose(0, 2, 1), 1e-13)         # diagonal blocks stored in full, symmetric
    # window: the tracks seen by cameras 5000 .. 5040 -> oracle on that sub-scene gives their rows of S and b exactly
    lo_c, hi_c = 5000, 5040
    touching = np.zeros(nt, bool)
    touching[s['obs_pt'][(s['obs_cam'] >= lo_c) & (s['obs_cam'] <= hi_c)]] = True
    m = touching[s['obs_pt']]
    ids = np.nonzero(touching)[0]
    renum = -np.ones(nt, np.int64)
    renum[ids] = np.arange(len(ids))
    sub = (s['K'], s['R0'], s['t0'], s['X0'][ids], s['obs_cam'][m], renum[s['obs_pt'][m]].astype(np.int32), s['obs_z'][m])
    HCC, HPP, W, bC, bP = O.normal_blocks(sensor, *sub, nc, len(ids))
    blk = be.get_blocks()
    close(blk['HPP'][ids], HPP, TIGHT)
    close(blk['bP'][ids], bP, TIGHT)
    close(blk['HCC'][lo_c:hi_c + 1], HCC[lo_c:hi_c + 1], TIGHT)
    close(blk['bC'][lo_c:hi_c + 1], bC[lo_c:hi_c + 1], TIGHT)
    # reduced rows of cameras lo_c .. hi_c: restrict the oracle to a camera window so that its dense S stays small
    w0, w1 = lo_c - 12, hi_c + 12
    cpos = -np.ones(nc, np.int32)
    cpos[w0:w1 + 1] = np.arange(w1 - w0 + 1)
    HPPi = O.invert_point_blocks(O.damp_blocks(HPP, 10.), 1e-5)
    S0, b0 = O.schur_complement(O.damp_blocks(HCC, 10.), HPPi, W, bC, bP, sub[4], sub[5], cpos)
    for cam in range(lo_c, hi_c + 1):
        p, q = cam - 1, cam - w0                                       # optimised position in the full / windowed system
        for d in range(10):
            close(band[p, d], S0[q, q + d], 1e-10, atol=1e-9 * np.abs(band[p, 0]).max())
        close(bfull[p], b0[q], 1e-10)
    # linearity over 8 contiguous point shards (the 8 ranks of config 5)
    from pysfm_amd.distributed import shard_bounds
    bounds = shard_bounds(np.full(nt, 10), 8)
    acc_S, acc_b = np.zeros_like(band), np.zeros_like(bfull)
    be.set_min_half_bandwidth(9)
    for g in range(8):
        lo, hi = bounds[g], bounds[g + 1]
        mm = (s['obs_pt'] >= lo) & (s['obs_pt'] < hi)
        be.set_problem(nc, hi - lo, s['obs_cam'][mm], s['obs_pt'][mm] - lo, s['obs_z'][mm], s['K'], flags[0], np.ones(hi - lo, np.uint8))
        be.set_sensor(0, np.eye(2).reshape(4))
        be.set_params(0, s['R0'], s['t0'], s['X0'][lo:hi])
        assert be.half_bandwidth == 9
        be.linearize(0)
        be.schur(0, 10., 1e-5)
        be.synchronize()
        St, bt = be.reduced_tensors()
        acc_S += St.cpu().numpy().reshape(band.shape)
        acc_b += bt.cpu().numpy().reshape(bfull.shape)
    be.set_min_half_bandwidth(0)
    close(acc_S, band, 1e-11)
    close(acc_b, bfull, 1e-11)


def test_config5_reduced_solve_1112_nodes(be, config5):
    """The 11-level cyclic reduction over 1112 super-blocks (59 994 unknowns) against the single-workgroup band
    Cholesky on the same device-resident system, and against LAPACK on a 300-camera sub-block (all other camera
    parameters masked out: the reference's row / column deletion, bundle_adjuster.py:290-299)."""
    s = config5
    nc, nt = 10000, 1000000
    flags = default_flags(nc, nt)
    load_problem(be, s['K'], s['R0'], s['t0'], s['X0'], s['obs_cam'], s['obs_pt'], s['obs_z'], *flags, O.Sensor.gaussian(1.))
    be.linearize(0)
    be.schur(0, 10., 1e-5)
    be.solve_reduced(None)
    assert be.last_solve_kind == 'bcr' and be.last_solve_path == 'band'
    x = be.get_solution()
    be.set_option('solver', 'band')
    be.solve_reduced(None)
    assert be.last_solve_kind == 'band'
    close(x, be.get_solution(), 1e-9)
    be.set_option('solver', 'auto')
    # residual of the full solve through the band itself:  S x = b
    St, bt = be.reduced_tensors()
    band = St.cpu().numpy().reshape(nc - 1, 10, 6, 6)
    bb = bt.cpu().numpy().reshape(nc - 1, 6)
    y = np.einsum('nab,nb->na', band[:, 0], x)
    for d in range(1, 10):
        y[:-d] += np.einsum('nab,nb->na', band[:-d, d], x[d:])
        y[d:] += np.einsum('nba,nb->na', band[:-d, d], x[:-d])
    assert np.abs(y - bb).max() <= 1e-9 * np.abs(bb).max()
    # masked sub-block vs LAPACK
    p0, p1 = 4000, 4300
    mask = np.zeros((nc - 1) * 6, np.uint8)
    mask[6 * p0:6 * p1] = 1
    mask[6 * p0 + 3::97] = 0                                           # and some single parameters inside it
    be.solve_reduced(mask)
    xm = be.get_solution().reshape(-1)
    assert np.all(xm[mask == 0] == 0)
    n = 6 * (p1 - p0)
    A = np.zeros((n, n))
    for i in range(p0, p1):
        for d in range(10):
            if i + d < p1:
                blk = band[i, d]
                A[6 * (i - p0):6 * (i - p0) + 6, 6 * (i + d - p0):6 * (i + d - p0) + 6] = blk
                if d:
                    A[6 * (i + d - p0):6 * (i + d - p0) + 6, 6 * (i - p0):6 * (i - p0) + 6] = blk.T
    keep = np.nonzero(mask[6 * p0:6 * p1])[0]
    ref = np.linalg.solve(A[np.ix_(keep, keep)], bb.reshape(-1)[6 * p0:6 * p1][keep])
    close(xm[6 * p0:6 * p1][keep], ref, 1e-9)


def test_config5_full_lm_converges(config5):
    from pysfm_amd import Bundle, BundleAdjuster
    s = config5
    b0 = Bundle.FromObservations(s['K'], s['R0'], s['t0'], s['X0'], s['obs_cam'], s['obs_pt'], s['obs_z'])
    ba = BundleAdjuster(b0, verbose=False)
    ba.optimize(max_steps=12)
    assert all(c1 < c0 for c0, c1 in zip(ba.costs, ba.costs[1:]))
    e = ba.backend.eval_observations(0, e=True, r=False, Jc=False, Jp=False)['e']
    rmse = float(np.sqrt(np.sum(e * e) / len(e)))
    assert rmse < 1.1 * .02 * np.sqrt(2), rmse
    assert ba.backend.last_solve_kind == 'bcr'
    ba.backend.close()


# ------------------------------------------------------------------ ONE full trial at full size against the oracle (bundle_adjuster.py:176-208)
@pytest.mark.parametrize('sensor,outliers', [(O.Sensor.gaussian(1.), 0.), (O.Sensor.huber(.06), .1), (O.Sensor.cauchy(.05), .1)],
                         ids=['config3-gaussian', 'config4-huber', 'config4-cauchy'])
def test_full_size_trial_end_to_end_vs_oracle(be, sensor, outliers):
    """BASELINE configs 3 and 4 at full size (1000 cameras / 100 000 points / 1 000 000 observations): ONE ba_lm_trial(damping = 10)
    - linearise, damp, invert, reduce, solve, back-substitute, update, trial cost, the whole batch of launches the bench times -
    against the oracle's compute_update / apply_update / cost on the full scene: the camera update dC and the point update dP to
    1e-8 of their largest entry (the damped system's condition number is ~1e5: 1e-14 x cond stays far below), the trial
    parameter set, and the trial cost to 1e-9."""
    nc, nt = 1000, 100000
    s = banded(nc, nt, outlier_frac=outliers)
    flags = default_flags(nc, nt)
    a = (s['K'], s['R0'], s['t0'], s['X0'], s['obs_cam'], s['obs_pt'], s['obs_z'])
    load_problem(be, *a, *flags, sensor)
    cur = be.cost(0)
    info, cost = be.lm_trial(10., 1e-5, None)
    assert info == 0 and be.last_solve_kind == 'bcr'
    dC = be.get_solution()
    Rg, tg, Xg = be.get_params(1)
    mu, su = O.compute_update(sensor, *a, *flags, damping=10.)
    close(-dC, mu, 1e-8)
    R2, t2, X2 = O.apply_update(s['R0'], s['t0'], s['X0'], mu, su, *flags)
    close(Xg - s['X0'], su, 1e-8)                                     # dP itself, not x + dP (x is 1e3 times larger)
    close(tg - s['t0'], t2 - s['t0'], 1e-8)
    close(Rg, R2, 1e-10)
    ref_cost = O.cost(sensor, s['K'], R2, t2, X2, *a[4:], *flags)
    assert abs(cost - ref_cost) <= 1e-9 * ref_cost, (cost, ref_cost)
    assert abs(cur - O.cost(sensor, *a, *flags)) <= 1e-12 * cur
    assert cost < cur                                                  # (the step is a descent step at this damping)


def test_config5_trial_end_to_end_on_a_masked_sub_block(be, config5):
    """BASELINE config 5 size through ba_lm_trial: all camera parameters masked out except those of cameras 4000..4299 (the
    reference's row / column deletion, bundle_adjuster.py:290-299).  The oracle solves the same masked system on the sub-scene of
    the tracks that touch those cameras (its dense S stays small); dC on the sub-block, dP and the new points of those tracks,
    and the trial cost (oracle evaluated on the library's own trial parameter set, all 10M observations) must agree."""
    s = config5
    nc, nt = 10000, 1000000
    flags = default_flags(nc, nt)
    sensor = O.Sensor.gaussian(1.)
    load_problem(be, s['K'], s['R0'], s['t0'], s['X0'], s['obs_cam'], s['obs_pt'], s['obs_z'], *flags, sensor)
    c0, c1 = 4001, 4301                                                # cameras (index) whose parameters stay: positions 4000..4299
    mask = np.zeros((nc - 1) * 6, np.uint8)
    mask[6 * (c0 - 1):6 * (c1 - 1)] = 1
    info, cost = be.lm_trial(10., 1e-5, mask)
    assert info == 0 and be.last_solve_kind == 'bcr'
    dC = be.get_solution()
    assert np.all(dC.reshape(-1)[mask == 0] == 0)
    Rg, tg, Xg = be.get_params(1)
    # the sub-scene: every track that touches a kept camera, with all its observations; cameras of a window around them optimised,
    # the parameters outside [c0, c1) masked
    touching = np.zeros(nt, bool)
    touching[s['obs_pt'][(s['obs_cam'] >= c0) & (s['obs_cam'] < c1)]] = True
    m = touching[s['obs_pt']]
    ids = np.nonzero(touching)[0]
    renum = -np.ones(nt, np.int64)
    renum[ids] = np.arange(len(ids))
    w0, w1 = c0 - 12, c1 + 12
    cpos = -np.ones(nc, np.int32)
    cpos[w0:w1] = np.arange(w1 - w0)
    sub_mask = np.zeros((w1 - w0) * 6, bool)
    sub_mask[6 * (c0 - w0):6 * (c1 - w0)] = True
    sub = (s['K'], s['R0'], s['t0'], s['X0'][ids], s['obs_cam'][m], renum[s['obs_pt'][m]].astype(np.int32), s['obs_z'][m])
    mu, su = O.compute_update(sensor, *sub, cpos, np.ones(len(ids), bool), damping=10., cam_param_mask=sub_mask)
    close(-dC[c0 - 1:c1 - 1], mu[c0 - w0:c1 - w0], 1e-8)
    close(Xg[ids] - s['X0'][ids], su, 1e-8)
    # points that no kept camera sees move by HPPinv bP alone; spot-check a window of them against the oracle
    far = np.arange(200000, 200500)
    mf = np.isin(s['obs_pt'], far)
    subf = (s['K'], s['R0'], s['t0'], s['X0'][far], s['obs_cam'][mf], (s['obs_pt'][mf] - far[0]).astype(np.int32), s['obs_z'][mf])
    _, HPPf, _, _, bPf = O.normal_blocks(sensor, *subf, nc, len(far))
    dPf = np.einsum('nij,nj->ni', O.invert_point_blocks(O.damp_blocks(HPPf, 10.), 1e-5), bPf)
    close(Xg[far] - s['X0'][far], -dPf, 1e-8)
    # the trial cost: the oracle on the trial set the library produced
    strial = dict(s, R0=Rg, t0=tg, X0=Xg)
    ref_cost = _chunked_cost(sensor, strial, flags)
    assert abs(cost - ref_cost) <= 1e-9 * ref_cost, (cost, ref_cost)


# ------------------------------------------------------------------ sharded, config-5-shaped
def _c5_rank_worker(rank, world, port, out_dir, nc, nt, shuffle, dist_solve=True, mask_some=False, renumber=False):
    import sys
    for p in (ROOT, os.path.join(ROOT, 'tests')):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0')
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from pysfm_amd import Bundle, BundleAdjuster
    from pysfm_amd import synthetic_data as sd
    from pysfm_amd.distributed import ShardComm, shard_tracks
    s = sd.generate_banded_scene(nc, nt)
    if shuffle:
        s = shuffled(s)[0]
    if renumber:
        s = cameras_renumbered(s)[0]
    b = Bundle.FromObservations(s['K'], s['R0'], s['t0'], s['X0'], s['obs_cam'], s['obs_pt'], s['obs_z'])
    comm = ShardComm()
    ba = BundleAdjuster(device=0, comm=comm, verbose=False)          # all ranks on GPU 0
    ba.distributed_solve = bool(dist_solve)
    ids = shard_tracks(b, rank, world, plan=ba.backend.dist_plan if dist_solve else None)
    ba.set_bundle(b, track_ids=ids)
    mask = None
    if mask_some:
        mask = np.ones(6 * (nc - 1) + 3 * len(ids), bool)
        mask[6 * 700 + 2:6 * 760:7] = False                         # some camera parameters deleted from the system
    ba.optimize(param_mask=mask, max_steps=5)
    X = comm.gather_points(ba)
    R, t, _ = ba.backend.get_params(0)
    # the shard's cameras: a contiguous stretch of the sequence even when the tracks came shuffled
    cams = np.unique(s['obs_cam'][np.isin(s['obs_pt'], ids)])
    np.savez(os.path.join(out_dir, 'rank%d.npz' % rank), costs=np.array(ba.costs), X=X, R=R, t=t, trials=ba.lm_trials,
             nbytes=comm.bytes_reduced, hb=ba.backend.half_bandwidth, kind=ba.backend.last_solve_kind, cam_lo=cams.min(),
             cam_hi=cams.max(), ntracks=len(ids), dist=int(getattr(ba, '_dist', False)), band_bytes=8 * ba.backend.S_doubles,
             permuted=ba.backend.problem_info()['cameras_permuted'])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('world,shuffle,dist_solve,mask_some', [(2, False, True, False), (4, True, True, True), (8, False, True, False),
                                                               (2, False, False, False), (4, True, False, True)])
def test_ranks_on_one_gpu_config5_shape(tmp_path, world, shuffle, dist_solve, mask_some):
    """A config-5-shaped scene (2400 cameras: 240 cyclic-reduction nodes of 10 cameras, 8 levels; 60 000 points) split by
    points over `world` processes on the one GPU (gloo group staging the collectives through the host): the sharded LM
    trajectory, cameras and points must reproduce the unsharded run; each shard covers one stretch of the camera sequence.
    dist_solve: the reduced solve spread over the ranks (csrc/ba_dist.h: three small sums per trial) - or, off, the whole
    band summed and solved by every rank.  mask_some: camera parameters deleted from the system (bundle_adjuster.py:290-299)."""
    import socket
    import torch.multiprocessing as mp
    from pysfm_amd import Bundle, BundleAdjuster
    from pysfm_amd import synthetic_data as sd
    nc, nt = 2400, 60000
    sk = socket.socket(); sk.bind(('127.0.0.1', 0)); port = sk.getsockname()[1]; sk.close()
    mp.spawn(_c5_rank_worker, args=(world, port, str(tmp_path), nc, nt, shuffle, dist_solve, mask_some), nprocs=world, join=True)
    s = sd.generate_banded_scene(nc, nt)
    if shuffle:
        s = shuffled(s)[0]
    b = Bundle.FromObservations(s['K'], s['R0'], s['t0'], s['X0'], s['obs_cam'], s['obs_pt'], s['obs_z'])
    ba = BundleAdjuster(verbose=False)
    ba.set_bundle(b)
    mask = None
    if mask_some:
        mask = np.ones(6 * (nc - 1) + 3 * nt, bool)
        mask[6 * 700 + 2:6 * 760:7] = False
    ba.optimize(param_mask=mask, max_steps=5)
    R1, t1, X1 = ba.backend.get_params(0)
    total = 0
    for r in range(world):
        d = np.load(os.path.join(str(tmp_path), 'rank%d.npz' % r))
        assert int(d['trials']) == ba.lm_trials and int(d['nbytes']) > 0
        assert int(d['dist']) == int(dist_solve)
        per_trial = int(d['nbytes']) / int(d['trials'])
        if dist_solve:        # three small sums per trial: a fraction of the band [S | b] (2 ranks: one separator; more ranks: more of them)
            assert per_trial <= int(d['band_bytes']) / (5 if world <= 4 else 2.5), (per_trial, int(d['band_bytes']))
        else:
            assert per_trial >= int(d['band_bytes'])
        assert int(d['hb']) == 9 and str(d['kind']) == 'bcr'
        close(d['costs'], np.array(ba.costs), 1e-9)
        close(d['t'], t1, 1e-8)
        close(d['X'], X1, 1e-8)
        assert int(d['cam_hi']) - int(d['cam_lo']) <= (nc // world + 40 if not dist_solve else 2 * nc // world + 40)
        total += int(d['ntracks'])
    assert total == nt
    ba.backend.close()


# ------------------------------------------------------------------ wide bands through the whole LM loop
@pytest.mark.parametrize('nc,nt,L,long_every,long_len,sensor', [
    (200, 1500, 30, 0, 0, 'gaussian'),           # windows of 30 cameras (k_schur_wide_mfma), 7 nodes of 30 cameras (ba_bcr_big.h)
    (300, 4200, 10, 25, 70, 'cauchy'),           # 4 % of the tracks seen by 70 cameras (k_schur_rect_mfma beside the window groups), hb = 69: 5 nodes
])
def test_lm_on_wide_band_scenes_follows_the_oracle(nc, nt, L, long_every, long_len, sensor):
    """BundleAdjuster.optimize on scenes whose tracks span 30 / 70 cameras - reduction through the wide-window and the
    segment-pair kernels, reduced solve through the cyclic reduction with nodes in device memory - against the oracle's LM
    loop: the same accept / reject sequence, accepted costs to 1e-6, the same final parameters."""
    from pysfm_amd import Bundle, BundleAdjuster, sensor_model
    from test_gpu_parity import _with_long_tracks
    s = banded(nc, nt, track_len=L, init_mode='params')
    cam, pt, z = (s['obs_cam'], s['obs_pt'], s['obs_z']) if not long_every else _with_long_tracks(s, nc, nt, long_every, long_len, holes=.15)
    flags = (np.arange(nc, dtype=np.int32) - 1, np.ones(nt, bool))
    sen = O.Sensor.gaussian(1.) if sensor == 'gaussian' else O.Sensor.cauchy(.05)
    model = sensor_model.GaussianModel(1.) if sensor == 'gaussian' else sensor_model.CauchyModel(.05)
    b = Bundle.FromObservations(s['K'], s['R0'], s['t0'], s['X0'], cam, pt, z, sensor_model=model)
    ba = BundleAdjuster(b, verbose=False)
    steps = 6
    ba.optimize(max_steps=steps, init_damping=10.)
    info = ba.backend.problem_info()
    assert info['schur_kernel'] == 4 and ba.backend.last_solve_kind == 'bcr_big'
    assert ba.backend.half_bandwidth == (L - 1 if not long_every else long_len - 1)
    trace = []
    ref = O.lm_optimize(sen, s['K'], s['R0'], s['t0'], s['X0'], cam, pt, z, *flags, max_steps=steps, init_damping=10., trace=trace)
    got = [(d, 'accepted' if o == 'accepted' else 'rejected') for d, o, c in ba.trial_log]
    want = [(tr['damping'], 'accepted' if tr['next'] < tr['cur'] else 'rejected') for tr in trace]
    assert len(got) == len(want)
    for g, w in zip(got, want):
        assert g[1] == w[1] and abs(g[0] - w[0]) <= 1e-12 * w[0], (got, want)
    assert ba.num_steps == ref['num_steps'] and ba.converged == ref['converged']
    close(np.array(ba.costs), np.array(ref['costs']), 1e-6)
    out = ba.bundle
    close(out.ts(), ref['t'], 1e-6, 1e-8)
    close(out.reconstruction, ref['X'], 1e-6, 1e-8)
    ba.backend.close()


# ------------------------------------------------------------------ the LM loop at full size, from the hard start (bundle_adjuster.py:117-162)
@pytest.mark.parametrize('sensor,outliers', [('gaussian', 0.), ('huber', .1)], ids=['config3-gaussian', 'config4-huber'])
def test_full_size_lm_trajectory_from_the_hard_start_vs_oracle(sensor, outliers):
    """BASELINE configs 3 and 4 at full size (1000 cameras / 100 000 points / 1 000 000 observations), from the generator's
    `params` start (every camera and point perturbed, SURVEY 8d - the start the bench's 46-trial run leaves from), the first
    six outer steps of the LM loop (bundle_adjuster.py:127-157), two ways:
    (1) trial by trial FROM IDENTICAL INPUTS ("parity must be asserted per step from identical inputs"): the
        oracle walks the reference's loop; before every trial the device gets the oracle's current parameters, runs ONE
        ba_lm_trial at the oracle's damping and must take the oracle's decision with the oracle's trial cost - to 1e-6 while
        the damping is at least 0.01; below, the damped system's condition number (the free scale of a monocular
        reconstruction: ~1e10 at damping 1e-3) turns the last digits of any solve into 1e-5 .. 1e-2 of the trial cost, for
        the reference's LU as for this Cholesky, and the tolerance says so;
    (2) free running: BundleAdjuster.optimize(max_steps=6) must take the same sequence of dampings and decisions and end
        within 2 % of the oracle's final cost."""
    from pysfm_amd import Bundle, BundleAdjuster, sensor_model
    from pysfm_amd._capi import PARAMS_CUR
    nc, nt, steps = 1000, 100000, 6
    s = banded(nc, nt, outlier_frac=outliers, init_mode='params')
    sen = O.Sensor.gaussian(1.) if sensor == 'gaussian' else O.Sensor.huber(.06)
    model = sensor_model.GaussianModel(1.) if sensor == 'gaussian' else sensor_model.HuberModel(.06)
    flags = (np.arange(nc, dtype=np.int32) - 1, np.ones(nt, bool))
    obs = (s['obs_cam'], s['obs_pt'], s['obs_z'])
    b = Bundle.FromObservations(s['K'], s['R0'], s['t0'], s['X0'], *obs, sensor_model=model)
    ba = BundleAdjuster(b, verbose=False)
    be = ba.backend
    # (1) the reference's loop on the oracle, every trial mirrored on the device from the oracle's state
    R, t, X = s['R0'], s['t0'], s['X0']
    damping, converged, num_steps, log = 10., False, 0, []
    costs = [O.cost(sen, s['K'], R, t, X, *obs, *flags)]
    while not converged and num_steps < steps:
        num_steps += 1
        cur = O.cost(sen, s['K'], R, t, X, *obs, *flags)
        while not converged and damping < 1e8:
            mu, su = O.compute_update(sen, s['K'], R, t, X, *obs, *flags, damping=damping)
            R2, t2, X2 = O.apply_update(R, t, X, mu, su, *flags)
            nxt = O.cost(sen, s['K'], R2, t2, X2, *obs, *flags)
            be.set_params(PARAMS_CUR, R, t, X)
            close(be.cost(PARAMS_CUR), cur, 1e-11)
            info, got = be.lm_trial(damping, ba.SCHUR_COMPLIMENT_PINV_THRESHOLD, None)
            assert info == 0 and be.last_solve_kind == 'bcr'
            tol = 1e-6 if damping >= 1e-2 else 3e-2
            assert abs(got - nxt) <= tol * nxt, (num_steps, damping, got, nxt)
            assert (got < cur) == (nxt < cur), (num_steps, damping, got, nxt, cur)
            log.append((damping, nxt < cur))
            if nxt < cur:
                damping *= .1
                R, t, X = R2, t2, X2
                costs.append(nxt)
                converged = abs(cur - nxt) < 1e-4
                break
            damping *= 10.
            converged = damping > 1e8
    assert len(log) >= steps and sum(1 for d, a in log if d >= 1e-2) >= 4
    # (2) the adjuster's own loop
    ba.set_bundle(b)
    ba.optimize(max_steps=steps)
    assert be.problem_info()['schur_mfma'] == 1
    got = [(d, o == 'accepted') for d, o, c in ba.trial_log]
    assert len(got) == len(log), (got, log)
    for g, w in zip(got, log):
        assert g[1] == w[1] and abs(g[0] - w[0]) <= 1e-12 * w[0], (got, log)
    assert ba.num_steps == num_steps and ba.converged == converged
    close(np.array(ba.costs[:4]), np.array(costs[:4]), 1e-6)
    assert abs(ba.costs[-1] - costs[-1]) <= 2e-2 * costs[-1], (ba.costs, costs)
    assert ba.costs[-1] < ba.costs[0]
    ba.backend.close()


# ------------------------------------------------------------------ LU semantics beyond the fallback size
def test_large_reduced_systems_follow_the_reference_lu_trajectory():
    """400 cameras, 2394 unknowns: the reference solves
    the reduced system by LU (numpy.linalg.solve, bundle_adjuster.py:302-305), which also 'succeeds' on a numerically singular
    matrix; the GPU path solves by Cholesky and, where that fails, again by LU on the device.  The two must walk the same LM trajectory: identical accept / reject sequence
    (ill-conditioned counting as rejected), accepted costs to 1e-6 - from round 1's hard start to the noise floor, and
    restarted at the floor with the damping at 1e-13, where S is singular along the scale gauge to working precision.
    (scripts/lu_semantics_experiment.py prints the two sequences side by side.)"""
    from pysfm_amd import Bundle, BundleAdjuster
    nc, nt = 400, 6000
    s = banded(nc, nt, init_mode='params')
    flags = (np.arange(nc, dtype=np.int32) - 1, np.ones(nt, bool))
    sen = O.Sensor.gaussian(1.)
    R, t, X = s['R0'], s['t0'], s['X0']
    for damping0, steps in ((10., 14), (1e-13, 3)):
        b = Bundle.FromObservations(s['K'], R, t, X, s['obs_cam'], s['obs_pt'], s['obs_z'])
        ba = BundleAdjuster(b, verbose=False)
        ba.optimize(max_steps=steps, init_damping=damping0)
        trace = []
        ref = O.lm_optimize(sen, s['K'], R, t, X, s['obs_cam'], s['obs_pt'], s['obs_z'], *flags, max_steps=steps,
                            init_damping=damping0, trace=trace)
        got = [(d, 'accepted' if o == 'accepted' else 'rejected') for d, o, c in ba.trial_log]
        want = [(tr['damping'], 'accepted' if tr['next'] < tr['cur'] else 'rejected') for tr in trace]
        assert len(got) == len(want)
        for g, w in zip(got, want):
            assert g[1] == w[1] and abs(g[0] - w[0]) <= 1e-12 * w[0], (got, want)
        assert ba.num_steps == ref['num_steps'] and ba.converged == ref['converged']
        close(np.array(ba.costs), np.array(ref['costs']), 1e-6)
        R, t, X = ref['R'], ref['t'], ref['X']
        ba.backend.close()


def test_lm_on_systems_that_are_not_positive_definite_follows_the_reference_lu():
    """LM trials on reduced systems that are indefinite on purpose (NEGATIVE damping): the reference's LU solves every one of
    them (bundle_adjuster.py:302-305) and the loop accepts or rejects the step by its cost.  2394 unknowns: every trial here goes through the cyclic reduction with LU nodes - same decision, same trial cost,
    same trial parameters as the oracle's LU; none reported ill-conditioned."""
    from pysfm_amd import Bundle, BundleAdjuster
    nc, nt = 400, 6000
    s = banded(nc, nt)
    flags = (np.arange(nc, dtype=np.int32) - 1, np.ones(nt, bool))
    sen = O.Sensor.gaussian(1.)
    a = (s['K'], s['R0'], s['t0'], s['X0'], s['obs_cam'], s['obs_pt'], s['obs_z'])
    b = Bundle.FromObservations(*a)
    ba = BundleAdjuster(b, verbose=False)
    R, t, X = s['R0'], s['t0'], s['X0']
    cur = O.cost(sen, s['K'], R, t, X, *a[4:], *flags)
    for n, damping in enumerate((-.6, -.3, -.85, -.5)):
        accepted, nxt = ba.trial(damping, None, cur)
        assert accepted is not None and ba.lu_solves == n + 1 and ba.backend.last_solve_kind == 'bcr_lu'
        mu, su = O.compute_update(sen, s['K'], R, t, X, *a[4:], *flags, damping=damping)
        R2, t2, X2 = O.apply_update(R, t, X, mu, su, *flags)
        ref_next = O.cost(sen, s['K'], R2, t2, X2, *a[4:], *flags)
        # (a step that overshoots by orders of magnitude - these indefinite systems produce them - amplifies the last digits of
        # the solve into its cost: 2e-4 relative at a cost 1e9 times the current one)
        assert abs(nxt - ref_next) <= (1e-7 if ref_next < 100 * cur else 1e-2) * ref_next, (damping, nxt, ref_next)
        assert bool(accepted) == bool(ref_next < cur)
        if accepted:
            R, t, X, cur = R2, t2, X2, ref_next
    ba.backend.close()


def test_lu_semantics_at_1000_cameras_on_the_device():
    """BASELINE config-3 size at the noise floor, the damping restarted at 1e-13: the reduced system (5994 unknowns) is
    singular along the scale gauge to working precision and a Cholesky factorisation fails; the reference's LU
    (numpy.linalg.solve, bundle_adjuster.py:302-305) returns a step all the same, the LM loop judges it by its cost.  The device
    path now does the same - the cyclic reduction with LU nodes, no trial declared ill-conditioned - and must walk the oracle's
    accept / reject sequence with the oracle's accepted costs."""
    from pysfm_amd import Bundle, BundleAdjuster
    nc, nt = 1000, 100000
    s = banded(nc, nt)
    flags = (np.arange(nc, dtype=np.int32) - 1, np.ones(nt, bool))
    sen = O.Sensor.gaussian(1.)
    b = Bundle.FromObservations(s['K'], s['R0'], s['t0'], s['X0'], s['obs_cam'], s['obs_pt'], s['obs_z'])
    ba = BundleAdjuster(b, verbose=False)
    ba.optimize(max_steps=25)                                            # to the floor (every system on the way is positive definite)
    R, t, X = ba.backend.get_params(0)
    ba.backend.close()
    b = Bundle.FromObservations(s['K'], R, t, X, s['obs_cam'], s['obs_pt'], s['obs_z'])
    ba = BundleAdjuster(b, verbose=False)
    ba.optimize(max_steps=2, init_damping=1e-13)
    trace = []
    ref = O.lm_optimize(sen, s['K'], R, t, X, s['obs_cam'], s['obs_pt'], s['obs_z'], *flags, max_steps=2, init_damping=1e-13, trace=trace)
    got = [(d, o) for d, o, c in ba.trial_log]
    want = [(tr['damping'], 'accepted' if tr['next'] < tr['cur'] else 'rejected') for tr in trace]
    assert all(o != 'ill-conditioned' for d, o in got), got             # LU semantics: every trial returns a step
    assert getattr(ba, 'cholesky_rejections', 0) == 0                    # (whether the LU nodes were needed depends on the round-off: ba.lu_solves)
    assert len(got) == len(want)
    for g, w in zip(got, want):
        assert g[1] == w[1] and abs(g[0] - w[0]) <= 1e-12 * w[0], (got, want)
    assert ba.num_steps == ref['num_steps'] and ba.converged == ref['converged']
    close(np.array(ba.costs), np.array(ref['costs']), 1e-6)
    ba.backend.close()


@pytest.mark.gpu
def test_bench_line_keeps_the_drivers_contract():
    """`python bench.py` (small K / W) prints ONE JSON line as its last line of stdout with the keys the driver and the judge read:
    the metric of BASELINE.json, whole-job value, the timing fields, `roofline` (with a live per-launch average and the committed
    counter traffic) and `cpu_baseline` (the oracle timed on this box)."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--steps', '6', '--warmup', '3', '--windows', '1'],
                         capture_output=True, text=True, timeout=560, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    last = out.stdout.strip().splitlines()[-1]
    # the driver keeps a few KB of tail and parses the LAST line: it stays short (round 5's 31 KB line did not parse)
    assert len(last) <= 4096, len(last)
    d = json.loads(last)
    full = json.load(open(os.path.join(root, d['detail'])))         # everything else the run measured
    assert full['value'] == d['value'] and full['ms_per_step'] == d['ms_per_step']
    base = json.load(open(os.path.join(root, 'BASELINE.json')))
    assert d['metric'] == base['metric'] and d['unit'] == (base.get('unit') or d['unit'])
    assert d['n_gpus'] == 1 and d['steps'] == 6 and d['warmup'] == 3 and d['higher_is_better'] is True
    assert d['scaling'] in ('weak', 'strong') and d['vs_baseline'] is None and d['dtype'] == 'f64' and 'synthetic' in d['data']
    assert isinstance(d['config'].get('workload'), str) and 'model' not in d['config']
    assert d['value'] > 1e8 and abs(d['value'] - 1e6 * d['steps'] / (d['ms_per_step'] * 1e-3 * d['steps'])) / d['value'] < 1e-6
    r = d['roofline']
    assert r['bound'] in ('hbm', 'mfma') and r['limited_by'] in ('hbm', 'mfma', 'latency') and r['unit'] in ('GB/s', 'TFLOP/s') and r['peak'] > 0
    assert abs(r['frac'] - r['achieved'] / r['peak']) < 1e-9 and r['avg_launch_ms'] > 0 and r['launches'] > 0
    assert r['traffic'] is None or r['traffic'] > 0
    assert isinstance(r['traffic_stale_possible'], bool) and (r['traffic_source'] == 'live') != r['traffic_stale_possible'] or r['traffic'] is None
    assert isinstance(r['kernel'], str) and 0 < len(r['kernel']) <= 60 and r['algorithmic_bytes_per_launch'] > 0
    assert d['final_reproj_rmse'] > 0 and d['linearise_schur_pass']['ms'] > 0
    # the other BASELINE configurations and scene shapes: a short run each, in the detail file
    oc = full['other_configs']
    for name in ('config2', 'config4_huber', 'config4_cauchy', 'config3_shuffled', 'config3_30pct_dropped', 'config3_2pct_tracks_of_80_cameras', 'config5_one_gpu', 'config3_track_length_32'):
        assert 'error' not in oc[name], oc[name]
        assert oc[name]['ms_per_step'] > 0 and oc[name]['dominant_kernel'] and 0 < oc[name]['linearise_schur_pass_fraction_of_kernel_time'] < 1
    assert oc['config5_one_gpu']['ms_per_step'] < 3.0 and oc['config2']['ms_per_step'] < oc['config4_huber']['ms_per_step']
    # wide bands: the matrix cores (not the pair kernel) and the cyclic reduction with nodes in device memory (not the dense Cholesky)
    assert oc['config3_track_length_32']['ms_per_step'] < 3.0 and oc['config3_track_length_32']['schur_kernel'] == 4 and oc['config3_track_length_32']['solve_kind'] == 'bcr_big'
    lt = oc['config3_2pct_tracks_of_80_cameras']
    assert lt['ms_per_step'] < 4.0 and lt['schur_kernel'] == 4 and lt['half_bandwidth'] == 79 and lt['solve_kind'] == 'bcr_big'
    # a scene with no band at all (round 6): conjugate gradients over the blocks the tracks define - 34.5 s a trial in round 5
    uc = oc['unordered_collection_5000_cameras']
    assert 'error' not in uc, uc
    assert all(t['solver'] == 'pcg' and t['rel_residual'] <= 1e-12 for t in uc['trials'].values()), uc['trials']
    assert uc['trials']['damping_10']['ms_per_trial'] < 5. and uc['band_fill'] < .05
    assert d['config']['init_mode'] in ('params', 'pose') and full['lm_other_start']['init_mode'] != d['config']['init_mode']
    c = d['cpu_baseline']
    assert c['kind'] in ('port', 'reference') and c['value'] > 0 and c['cores'] >= 1 and isinstance(c['sample'], str) and c['unit'] == d['unit']


# ------------------------------------------------------------------ cameras in any order (bundle_adjuster.py:259-312: the reference's dense S does not care)
def cameras_renumbered(s, seed=5):
    """The same scene with its cameras renumbered at random: camera c becomes camera perm[c]."""
    rs = np.random.RandomState(seed)
    nc = len(s['R0'])
    perm = rs.permutation(nc)
    out = dict(s)
    for k in ('R0', 't0', 'R', 't'):
        a = np.empty_like(s[k])
        a[perm] = s[k]
        out[k] = a
    out['obs_cam'] = perm[s['obs_cam']].astype(np.int32)
    return out, perm


@pytest.mark.parametrize('sensor,L,drop', [(O.Sensor.gaussian(1.), 10, 0.), (O.Sensor.cauchy(.05), 6, .25), (O.Sensor.huber(.06), 14, 0.)])
def test_cameras_in_any_order_full_step_vs_oracle(be, sensor, L, drop):
    """The optimised cameras handed over in random order (ids as a database assigns them): the caller's order would make the band
    as wide as the matrix; the library orders the cameras itself (csrc/ba_order.hip) - and everything that crosses the C ABI by
    optimised position (S, b, dC, masks, motion updates, flat indices) is still in the CALLER's positions and agrees with the
    oracle on the caller's arrays."""
    nc, nt = 150, 5000
    s, perm = cameras_renumbered(banded(nc, nt, track_len=L, outlier_frac=.02))
    rs = np.random.RandomState(2)
    keep = rs.rand(len(s['obs_cam'])) >= drop
    keep[::L] = True
    cam, pt, z = s['obs_cam'][keep], s['obs_pt'][keep], s['obs_z'][keep]
    cam_opt_pos = -np.ones(nc, np.int32)
    frozen = {int(perm[0]), int(perm[70])}
    opt = [c for c in range(nc) if c not in frozen]
    cam_opt_pos[opt] = np.arange(len(opt))
    pt_opt = np.ones(nt, np.uint8)
    pt_opt[::9] = 0
    a = (s['K'], s['R0'], s['t0'], s['X0'], cam, pt, z)
    load_problem(be, *a, cam_opt_pos, pt_opt, sensor)
    info = be.problem_info()
    assert info['cameras_permuted'] == 1 and info['caller_half_bandwidth'] > nc // 2
    assert info['half_bandwidth'] <= L + 1, info
    assert be.half_bandwidth == info['half_bandwidth']
    nco = be.nco
    close(be.cost(0), O.cost(sensor, *a, cam_opt_pos, pt_opt), 1e-12)
    mask = (rs.rand(nco * 6) > .05).astype(np.uint8)
    for m in (None, mask):
        mu, su, parts = O.compute_update(sensor, *a, cam_opt_pos, pt_opt, damping=2., cam_param_mask=None if m is None else m.astype(bool), return_parts=True)
        be.linearize(0)
        blk = be.get_blocks()
        for k in ('HCC', 'bC', 'HPP', 'bP'):
            close(blk[k], parts[k], TIGHT)
        be.schur(0, 2., 1e-5)
        S, b = be.get_reduced()
        close(S, parts['S'], TIGHT)
        close(b, parts['b'], TIGHT)
        be.solve_reduced(m)
        assert be.last_solve_kind == ('bcr' if info['half_bandwidth'] <= 13 else 'bcr_wide')
        dC = be.get_solution()
        close(-dC, mu, 1e-8)
        if m is not None:
            assert np.all(dC.reshape(-1)[m == 0] == 0.)
        dP = be.backsubstitute(0)
        close(-dP[pt_opt.astype(bool)], su, 1e-8)
        # the camera update handed back in by the caller (bundle_adjuster.py:316-331 backsubstitute(dC))
        close(be.backsubstitute(0, dC=-mu)[pt_opt.astype(bool)], -su, 1e-8)
        # the whole trial as one batch
        infoT, cost = be.lm_trial(2., 1e-5, m)
        assert infoT == 0
        R2, t2, X2 = O.apply_update(s['R0'], s['t0'], s['X0'], mu, su, cam_opt_pos, pt_opt)
        Rg, tg, Xg = be.get_params(1)
        close(Xg, X2, 1e-9)
        close(tg, t2, 1e-9)
        close(Rg, R2, 1e-9)
        close(cost, O.cost(sensor, s['K'], R2, t2, X2, cam, pt, z, cam_opt_pos, pt_opt), 1e-8)
    # a caller's motion update (update_motion, bundle_adjuster.py:334-337): rows by the caller's optimised positions
    delta = rs.randn(nco, 6) * 1e-3
    be.apply_update(0, 1, delta, np.zeros((nt, 3)))
    R3, t3, X3 = O.apply_update(s['R0'], s['t0'], s['X0'], delta, np.zeros((int(pt_opt.sum()), 3)), cam_opt_pos, pt_opt)
    Rg, tg, Xg = be.get_params(1)
    close(tg, t3, 1e-13)
    close(Rg, R3, 1e-13)
    # the flat matrix as the reference forms it before its solve (ba_flatten_reduced): kept indices by the caller's positions
    import ctypes as C
    import torch
    from pysfm_amd import _capi as capi
    be.linearize(0)
    be.schur(0, 2., 1e-5)
    keepi = np.nonzero(mask)[0].astype(np.int32)
    A = torch.zeros(len(keepi), len(keepi), dtype=torch.float64, device='cuda')
    rhs = torch.zeros(len(keepi), dtype=torch.float64, device='cuda')
    be._check(be._lib.ba_flatten_reduced(be._h, capi.iptr(keepi), len(keepi), C.c_void_p(A.data_ptr()), C.c_void_p(rhs.data_ptr())))
    Af, bf = O.flatten_reduced(parts['S'], parts['b'])
    close(A.cpu().numpy(), Af[np.ix_(keepi, keepi)], TIGHT)
    close(rhs.cpu().numpy(), bf[keepi], TIGHT)
    # the same scene with the library's ordering switched off: the same numbers through the wide band
    be.set_option('camera_order', 'off')
    load_problem(be, *a, cam_opt_pos, pt_opt, sensor)
    assert be.problem_info()['cameras_permuted'] == 0 and be.half_bandwidth == info['caller_half_bandwidth']
    be.linearize(0)
    be.schur(0, 2., 1e-5)
    S0, b0 = be.get_reduced()
    close(S0, parts['S'], TIGHT)
    be.solve_reduced(mask)
    close(-be.get_solution(), mu, 1e-8)


def test_cameras_in_any_order_lm_run_and_band_at_config3_shape():
    """BundleAdjuster on a scene whose cameras come in random order walks the LM trajectory of the same scene in sequence order
    (same decisions; costs to 1e-9: only the order of the sums inside S differs), and the band the library finds at config-3
    shape (1000 cameras, tracks of 10) is the sequence's own."""
    from pysfm_amd import Bundle, BundleAdjuster, sensor_model
    nc, nt = 1000, 20000
    s0 = banded(nc, nt, track_len=10)
    s1, perm = cameras_renumbered(s0)
    runs = []
    for s in (s0, s1):
        b = Bundle.FromObservations(s['K'], s['R0'], s['t0'], s['X0'], s['obs_cam'], s['obs_pt'], s['obs_z'], sensor_model=sensor_model.GaussianModel(1.))
        # the gauge camera is the same physical camera in both runs
        frozen = 0 if s is s0 else int(perm[0])
        ids = [frozen] + [c for c in range(nc) if c != frozen]
        ba = BundleAdjuster(verbose=False)
        ba.set_bundle(b, camera_ids=ids)
        info = ba.backend.problem_info()
        assert info['half_bandwidth'] == 9, info
        assert info['cameras_permuted'] == (0 if s is s0 else 1)
        ba.optimize(max_steps=6)
        runs.append((ba.trial_log, list(ba.costs), ba.bundle.ts(), info))
        ba.backend.close()
    (log0, costs0, ts0, _), (log1, costs1, ts1, info1) = runs
    assert [(d, o) for d, o, _ in log0] == [(d, o) for d, o, _ in log1]
    close(np.array(costs1), np.array(costs0), 1e-9)
    close(ts1[perm], ts0, 1e-6, 1e-9)
    assert info1['schur_mfma'] == 1


# ------------------------------------------------------------------ band + border (csrc/ba_border.h): a sequence with a few long-range tracks
def _loop_scene(nc, nt, L, pairs, width, **kw):
    from pysfm_amd import synthetic_data as sd
    return sd.add_loop_closure_tracks(banded(nc, nt, track_len=L, **kw), pairs, width=width)


@pytest.mark.parametrize('sensor,L,width,npairs,renumber', [(O.Sensor.gaussian(1.), 10, 1, 6, False), (O.Sensor.cauchy(.05), 8, 3, 4, False),
                                                            (O.Sensor.huber(.06), 6, 2, 9, False), (O.Sensor.gaussian(1.), 9, 1, 7, True),
                                                            (O.Sensor.gaussian(1.), 10, 1, 16, False),      # (16 ties: more border cameras than a node of the cyclic reduction holds - k_border_solve)
                                                            (O.Sensor.gaussian(1.), 13, 1, 5, False), (O.Sensor.cauchy(.05), 14, 2, 3, True)])      # (bands of 12 / 13 cameras beside the border: nodes with three matrices in LDS)
def test_loop_closure_tracks_band_plus_border_vs_oracle(be, sensor, L, width, npairs, renumber):
    """A camera sequence plus a few tracks that tie far-apart cameras together: the reference's dense S takes them like any
    other track (bundle_adjuster.py:259-312); here the cameras at their far end become a border of the band (csrc/ba_border.h).
    Everything that crosses the C ABI - S (band, border columns and border block expanded), b, the solution with and without
    masked camera parameters, the point updates, the whole trial, the flat matrix - against the oracle."""
    nc, nt = 160, 4000
    rs = np.random.RandomState(4)
    pairs = [(int(i), int(i) + 60 + int(rs.randint(0, 30))) for i in rs.choice(60, npairs, replace=False) + 2]
    s = _loop_scene(nc, nt, L, pairs, width, outlier_frac=.02)
    nt = len(s['X0'])
    perm = np.arange(nc)
    if renumber:                                         # ... and the cameras in no particular order (camera 0 stays the gauge camera)
        s, perm = cameras_renumbered(s)
        first = int(perm[0])
        swap = perm.copy(); swap[perm == 0] = first; swap[0] = 0
        for k in ('R0', 't0', 'R', 't'):
            s[k][[0, first]] = s[k][[first, 0]]
        oc = s['obs_cam'].copy(); s['obs_cam'][oc == 0] = first; s['obs_cam'][oc == first] = 0
        perm = swap
        pairs = [(int(perm[a]), int(perm[b])) for a, b in pairs]
    cam_opt_pos = np.arange(nc, dtype=np.int32) - 1
    pt_opt = np.ones(nt, np.uint8)
    pt_opt[::11] = 0
    a = (s['K'], s['R0'], s['t0'], s['X0'], s['obs_cam'], s['obs_pt'], s['obs_z'])
    load_problem(be, *a, cam_opt_pos, pt_opt, sensor)
    info = be.problem_info()
    assert 0 < info['border_cameras'] <= npairs * width and info['half_bandwidth'] <= max(11, L - 1) and info['caller_half_bandwidth'] >= 60, info
    assert info['cameras_permuted'] == 1
    if npairs >= 16:
        assert info['border_cameras'] > 11, info
    nco = be.nco
    close(be.cost(0), O.cost(sensor, *a, cam_opt_pos, pt_opt), 1e-12)
    mask = (rs.rand(nco * 6) > .04).astype(np.uint8)
    # mask a parameter of a border camera and one of a camera it shares a track with, whatever the draw
    far = max(pairs[0][1] - 1, 0)
    mask[6 * far + 2] = 0
    mask[6 * max(pairs[0][0] - 1, 0) + 4] = 0
    for m in (None, mask):
        mu, su, parts = O.compute_update(sensor, *a, cam_opt_pos, pt_opt, damping=1.5, cam_param_mask=None if m is None else m.astype(bool), return_parts=True)
        be.linearize(0)
        be.schur(0, 1.5, 1e-5)
        S, b = be.get_reduced()
        close(S, parts['S'], TIGHT)
        close(b, parts['b'], TIGHT)
        be.solve_reduced(m)
        assert be.last_solve_kind == 'bcr'
        dC = be.get_solution()
        close(-dC, mu, 1e-8)
        if m is not None:
            assert np.all(dC.reshape(-1)[m == 0] == 0.)
        dP = be.backsubstitute(0)
        close(-dP[pt_opt.astype(bool)], su, 1e-8)
        infoT, cost = be.lm_trial(1.5, 1e-5, m)
        assert infoT == 0
        S2, b2 = be.get_reduced()                       # (the trial's fused kernels: camera blocks inside the reduction)
        close(S2, parts['S'], TIGHT)
        close(b2, parts['b'], TIGHT)
        R2, t2, X2 = O.apply_update(s['R0'], s['t0'], s['X0'], mu, su, cam_opt_pos, pt_opt)
        Rg, tg, Xg = be.get_params(1)
        close(Xg, X2, 1e-9)
        close(tg, t2, 1e-9)
        close(cost, O.cost(sensor, s['K'], R2, t2, X2, *a[4:], cam_opt_pos, pt_opt), 1e-8)
    import ctypes as C
    import torch
    from pysfm_amd import _capi as capi
    be.linearize(0)
    be.schur(0, 1.5, 1e-5)
    keepi = np.nonzero(mask)[0].astype(np.int32)
    A = torch.zeros(len(keepi), len(keepi), dtype=torch.float64, device='cuda')
    rhs = torch.zeros(len(keepi), dtype=torch.float64, device='cuda')
    be._check(be._lib.ba_flatten_reduced(be._h, capi.iptr(keepi), len(keepi), C.c_void_p(A.data_ptr()), C.c_void_p(rhs.data_ptr())))
    Af, bf = O.flatten_reduced(parts['S'], parts['b'])
    close(A.cpu().numpy(), Af[np.ix_(keepi, keepi)], TIGHT)
    close(rhs.cpu().numpy(), bf[keepi], TIGHT)
    # without the border: the same numbers through the wide band
    be.set_option('border', 0)
    load_problem(be, *a, cam_opt_pos, pt_opt, sensor)
    assert be.problem_info()['border_cameras'] == 0
    be.linearize(0)
    be.schur(0, 1.5, 1e-5)
    be.solve_reduced(mask)
    close(-be.get_solution(), mu, 1e-8)


def test_loop_closure_lm_run_matches_the_oracle_walk():
    """optimize() on a sequence with loop-closure tracks (band + border inside) against the oracle's walk of the reference's loop:
    same decisions, accepted costs to 1e-6."""
    from pysfm_amd import Bundle, BundleAdjuster, sensor_model
    nc, nt = 120, 2500
    pairs = [(5, 80), (9, 100), (30, 111), (31, 112), (40, 95)]
    s = _loop_scene(nc, nt, 9, pairs, 2)
    nt = len(s['X0'])
    b = Bundle.FromObservations(s['K'], s['R0'], s['t0'], s['X0'], s['obs_cam'], s['obs_pt'], s['obs_z'], sensor_model=sensor_model.GaussianModel(1.))
    ba = BundleAdjuster(b, verbose=False)
    assert ba.backend.problem_info()['border_cameras'] > 0
    ba.optimize(max_steps=8)
    flags = (np.arange(nc, dtype=np.int32) - 1, np.ones(nt, bool))
    trace = []
    ref = O.lm_optimize(O.Sensor.gaussian(1.), s['K'], s['R0'], s['t0'], s['X0'], s['obs_cam'], s['obs_pt'], s['obs_z'], *flags, max_steps=8, trace=trace)
    got = [(d, o == 'accepted') for d, o, _ in ba.trial_log]
    want = [(tr['damping'], tr['next'] < tr['cur']) for tr in trace]
    assert got == want, (got, want)
    close(np.array(ba.costs), np.array(ref['costs']), 1e-6)
    close(ba.bundle.ts(), ref['t'], 1e-6, 1e-8)
    ba.backend.close()


# ------------------------------------------------------------------ the 25-step LM run the bench reports, against the oracle's golden walk
@pytest.mark.parametrize('name', ['config3', 'config4_huber', 'config3_pose'])
def test_full_lm25_run_against_the_oracle_golden_walk(name):
    """BASELINE configs 3 / 4 at full size, optimize(max_steps=25) from the generator's start - the run whose final RMSE the bench
    line reports - against the walk of the CPU oracle stored in tests/golden/<name>_lm25.npz (oracle/gen_golden_lm25.py, ~25
    minutes of NumPy per run in the build container).
    Phase A - up to the first ACCEPTED trial at a damping below 1e-2: the device takes the oracle's decision at the oracle's
    damping in every trial, with the oracle's trial cost to 1e-6 while the damping is >= 1e-2 (3e-2 below).
    Phase B - after it: the reduced system's condition number there is ~1e13 (scripts/solve_accuracy.py; the free scale of a
    monocular reconstruction), the accepted step is defined to ~1e-3 whatever solves it (LAPACK's LU and LAPACK's Cholesky differ
    from each other as much as the device differs from either, test below), and the two walks are two samples of a chaotic
    sequence: asserted are the end state - steps taken, final cost and raw reprojection RMSE within 1 % of the oracle's."""
    import json
    from pysfm_amd import Bundle, BundleAdjuster, sensor_model, synthetic_data as sd
    from pysfm_amd._capi import PARAMS_CUR
    g = dict(np.load(os.path.join(ROOT, 'tests', 'golden', name + '_lm25.npz')))
    s = sd.generate_banded_scene(**json.loads(str(g['scene_args'])))
    model = sensor_model.GaussianModel(1.) if str(g['sensor_kind']) == 'gaussian' else sensor_model.HuberModel(float(g['sensor_param']))
    b = Bundle.FromObservations(s['K'], s['R0'], s['t0'], s['X0'], s['obs_cam'], s['obs_pt'], s['obs_z'], sensor_model=model)
    ba = BundleAdjuster(b, verbose=False)
    ba.optimize(max_steps=25)
    want = list(zip(g['trial_damping'], g['trial_next'] < g['trial_cur'], g['trial_next']))
    phase_a = 0
    for i, (got, w) in enumerate(zip(ba.trial_log, want)):
        d, outcome, cost = got
        assert abs(d - w[0]) <= 1e-12 * w[0] and (outcome == 'accepted') == bool(w[1]), (i, got, w)
        assert abs(cost - w[2]) <= (1e-6 if d >= 1e-2 else 3e-2) * w[2], (i, got, w)
        phase_a = i + 1
        if w[1] and d < 1e-2:
            break
    assert phase_a >= 4, phase_a
    assert ba.num_steps == int(g['num_steps']) and ba.converged == bool(g['converged'])
    assert abs(ba.costs[-1] - g['costs'][-1]) <= 1e-2 * g['costs'][-1], (ba.costs[-1], g['costs'][-1])
    e = ba.backend.eval_observations(PARAMS_CUR, e=True, r=False, Jc=False, Jp=False)['e']
    rmse = float(np.sqrt(np.sum(e * e) / len(e)))
    assert abs(rmse - float(g['rmse_final'])) <= 1e-2 * float(g['rmse_final']), (rmse, float(g['rmse_final']))
    ba.backend.close()


@pytest.mark.parametrize('L,kind', [(8, 'bcr_lu'), (13, 'band_lu')])      # (nodes of up to 11 cameras: LU nodes; a band of 12 beside the border: LU down the band)
def test_band_plus_border_system_that_is_not_positive_definite_is_solved_like_the_reference_lu(be, L, kind):
    """The reference solves its reduced system by LU (numpy.linalg.solve, bundle_adjuster.py:302-305) - also one that is not positive
    definite.  A NEGATIVE damping makes such a system (indefinite, far from singular) on a scene with loop closures: the Cholesky
    form of band + border reports it, and the device solves it again with LU in every place of the block elimination (round 6,
    csrc/ba_border.hip border_solve_lu: the columns of C and b1 through the band's LU solver one by one, the border's Schur
    complement by Gaussian elimination with partial pivoting) - rounds 5's host LU (the whole matrix over PCIe to
    numpy.linalg.solve) is gone.  Option solver = lu goes there directly (what the time-out fallback of solve_reduced asks for)."""
    nc, nt = 120, 3000
    s = _loop_scene(nc, nt, L, [(5, 70), (20, 95), (33, 101)], 2)
    nt = len(s['X0'])
    flags = default_flags(nc, nt)
    a = (s['K'], s['R0'], s['t0'], s['X0'], s['obs_cam'], s['obs_pt'], s['obs_z'])
    load_problem(be, *a, *flags, O.Sensor.gaussian(1.))
    assert be.problem_info()['border_cameras'] > 0 and be.half_bandwidth == L - 1
    be.linearize(0)
    be.schur(0, -.6, 1e-5)
    S, b = be.get_reduced()
    n = be.nco * 6
    A = S.transpose(0, 2, 1, 3).reshape(n, n)
    w = np.linalg.eigvalsh(A)
    assert w[0] < 0 < w[-1] and np.min(np.abs(w)) > 1e-9 * np.max(np.abs(w))      # indefinite, not singular
    for mask in (None, (np.arange(n) % 11 != 3).astype(np.uint8)):
        be.set_option('device_lu', 0)
        with pytest.raises(Exception):                                                # (without the LU semantics: reported, not solved)
            be.solve_reduced(mask)
        be.set_option('device_lu', 1)
        be.solve_reduced(mask)
        assert be.last_solve_kind == kind and be.last_solve_path == 'lu'
        x = be.get_solution().reshape(-1)
        keep = np.arange(n) if mask is None else np.nonzero(mask)[0]
        ref = np.linalg.solve(A[np.ix_(keep, keep)], b.reshape(-1)[keep])
        close(x[keep], ref, 1e-10)
        be.set_option('solver', 'lu')                                                # straight to the LU form (the time-out fallback's request)
        be.solve_reduced(mask)
        assert be.last_solve_kind == kind
        close(be.get_solution().reshape(-1)[keep], ref, 1e-10)
        be.set_option('solver', 'auto')
        assert mask is None or np.all(x[mask == 0] == 0)
        dP = be.backsubstitute(0)
        assert np.all(np.isfinite(dP))
        mu, su = O.compute_update(O.Sensor.gaussian(1.), *a, *flags, damping=-.6, cam_param_mask=None if mask is None else mask.astype(bool))
        close(-x.reshape(-1, 6), mu, 1e-7)
        close(-dP, su, 1e-7)


@pytest.fixture(scope='module')
def sensitive_solve():
    """Config 3 after five LM steps of the DEVICE's own walk, damping 1e-3 (the first place where the device's walk and the oracle's
    part): [S | b] as the device formed it (fp64 atomics upstream: not the same bits every run - the deterministic version of these
    checks is tests/test_gpu_parity.py::test_golden_reduced_system_*), the device's solution, LAPACK's LU and Cholesky solutions."""
    import scipy.linalg as sl
    from pysfm_amd import Bundle, BundleAdjuster, sensor_model
    from pysfm_amd._capi import PARAMS_CUR
    nc, nt = 1000, 100000
    s = banded(nc, nt, init_mode='params')
    b = Bundle.FromObservations(s['K'], s['R0'], s['t0'], s['X0'], s['obs_cam'], s['obs_pt'], s['obs_z'], sensor_model=sensor_model.GaussianModel(1.))
    ba = BundleAdjuster(b, verbose=False)
    ba.optimize(max_steps=5)
    be = ba.backend
    be.linearize(PARAMS_CUR)
    be.schur(PARAMS_CUR, 1e-3, 1e-5)
    S, rhs = be.get_reduced()
    n = be.nco * 6
    A = S.transpose(0, 2, 1, 3).reshape(n, n)
    rhs = rhs.reshape(n)
    before = be.problem_info()['solves_refined']
    be.solve_reduced(None)
    assert be.last_solve_kind == 'bcr' and be.problem_info()['solves_refined'] == before + 1      # (damping below 1e-2: the refinement step is on by default)
    x_dev = be.get_solution().reshape(n)
    be.set_option('refine', '0')
    be.solve_reduced(None)
    x_raw = be.get_solution().reshape(n)
    be.set_option('refine', 'auto')
    out = dict(A=A, rhs=rhs, x_dev=x_dev, x_raw=x_raw, x_lu=np.linalg.solve(A, rhs), x_ch=sl.cho_solve(sl.cho_factor(A), rhs))
    ba.backend.close()
    return out


def test_device_solve_leaves_a_residual_no_larger_than_lapack_where_the_walk_is_sensitive(sensitive_solve):
    """On the SAME [S | b] the device's solution leaves a residual ||S dC - b|| / ||b|| no larger than 1.5 times the larger of LAPACK's
    two (numpy.linalg.solve = gesv, what the reference calls: bundle_adjuster.py:303; Cholesky).  Round 5 met this bound only with a
    floor of five units of round-off under it (the cyclic reduction alone: 2.4 - 2.6 times LAPACK); since round 6 the solve takes one
    step of iterative refinement through its kept factors below damping 1e-2 (csrc/ba_bcr_refine.h) - no floor."""
    d = sensitive_solve
    res = lambda x: np.linalg.norm(d['A'] @ x - d['rhs']) / np.linalg.norm(d['rhs'])
    lapack = max(res(d['x_lu']), res(d['x_ch']))
    assert res(d['x_dev']) <= 1.5 * lapack, (res(d['x_dev']), res(d['x_lu']), res(d['x_ch']))
    assert res(d['x_dev']) <= max(1.05 * res(d['x_raw']), lapack) and res(d['x_raw']) <= 6. * lapack, (res(d['x_dev']), res(d['x_raw']), lapack)
    # ... and in the measure that does not depend on LAPACK's luck of the day: ||S x - b|| / || |S| |x| + |b| || (Oettli and Prager) at
    # most 2 units of round-off refined, 4 unrefined (profiles/r06_golden_solve_spread.txt: 0.46 and 1.48 at most over 40 runs, LAPACK 0.58)
    bwd = lambda x: np.linalg.norm(d['A'] @ x - d['rhs']) / np.linalg.norm(np.abs(d['A']) @ np.abs(x) + np.abs(d['rhs']))
    eps = 2. ** -52
    assert bwd(d['x_dev']) <= 2 * eps and bwd(d['x_raw']) <= 4 * eps, (bwd(d['x_dev']) / eps, bwd(d['x_raw']) / eps, bwd(d['x_lu']) / eps)


def test_device_solve_agrees_with_lapack_cholesky_as_closely_as_lapack_lu_does(sensitive_solve):
    """... and it agrees with LAPACK's Cholesky solution as closely as LAPACK's LU does - the sensitivity is the system's (condition
    number > 1e12), not the solver's.  Its own test: the residual bound above cannot hide it."""
    d = sensitive_solve
    assert np.linalg.norm(d['x_dev'] - d['x_ch']) <= 2. * np.linalg.norm(d['x_lu'] - d['x_ch']) + 1e-14 * np.linalg.norm(d['x_ch'])
    assert np.linalg.norm(d['x_raw'] - d['x_ch']) <= 2. * np.linalg.norm(d['x_lu'] - d['x_ch']) + 1e-14 * np.linalg.norm(d['x_ch'])


def test_rejected_trials_reuse_the_linearisation():
    """After a rejected trial the current set has not changed: ba_lm_trial keeps its point blocks instead of forming identical
    ones again (SURVEY 3.1 on bundle_adjuster.py:132-140).  The walk - dampings, decisions, trial costs - must be the one the
    adjuster takes with the reuse switched off, and every write to the current set must end the reuse."""
    from pysfm_amd import Bundle, BundleAdjuster, sensor_model
    nc, nt = 120, 6000
    s = banded(nc, nt, track_len=10, outlier_frac=.05, init_mode='params')
    runs = []
    for reuse in (1, 0):
        b = Bundle.FromObservations(s['K'], s['R0'], s['t0'], s['X0'], s['obs_cam'], s['obs_pt'], s['obs_z'], sensor_model=sensor_model.CauchyModel(.05))
        ba = BundleAdjuster(verbose=False)
        ba.backend.set_option('reuse_linearization', reuse)
        ba.set_bundle(b)
        ba.optimize(max_steps=12)
        info = ba.backend.problem_info()
        rejected = sum(1 for _, o, _ in ba.trial_log if o == 'rejected')
        assert rejected >= 2
        # every trial that follows a rejected one reuses (when allowed), no other does
        follows = sum(1 for i in range(1, len(ba.trial_log)) if ba.trial_log[i - 1][1] == 'rejected')
        assert info['linearizations_reused'] == (follows if reuse else 0), (info, follows)
        runs.append((list(ba.trial_log), list(ba.costs), ba.bundle.reconstruction.copy()))
        if reuse:
            # a caller that edits the current bundle between trials gets a fresh linearisation
            be = ba.backend
            cur = ba._cur_cost
            ba.trial(1e6, None, -1.)                       # a trial that is rejected (cost can not go below -1)
            before = be.problem_info()['linearizations_reused']
            R, t, X = be.get_params(0)
            be.set_params(0, R, t, X + 1e-3)
            ba.trial(1e6, None, -1.)
            assert be.problem_info()['linearizations_reused'] == before
            ba.trial(1e6, None, -1.)
            assert be.problem_info()['linearizations_reused'] == before + 1
        ba.backend.close()
    # (the same walk; its last digits are free either way - the reduction's fp64 atomics land in a different order every run)
    assert [(d, o) for d, o, _ in runs[0][0]] == [(d, o) for d, o, _ in runs[1][0]]
    close(np.array([c for _, _, c in runs[0][0]]), np.array([c for _, _, c in runs[1][0]]), 1e-9)
    close(np.array(runs[0][1]), np.array(runs[1][1]), 1e-9)
    close(runs[0][2], runs[1][2], 1e-7, 1e-10)



@pytest.mark.parametrize('world', [2, 4])
def test_ranks_on_one_gpu_cameras_in_no_order(tmp_path, world):
    """The sharded adjuster on a scene whose cameras come in no particular order: the ranks add their [S | b] buffers element by
    element, so they must all use ONE layout - every rank plans the order of the optimised cameras from the whole bundle
    (BundleAdjuster._shared_camera_layout -> ba_plan_camera_layout -> ba_set_camera_layout) and they agree on the band of THAT order
    (9, not 590).  The sharded walk must be the unsharded one."""
    import socket
    import torch.multiprocessing as mp
    from pysfm_amd import Bundle, BundleAdjuster
    from pysfm_amd import synthetic_data as sd
    nc, nt = 600, 15000
    sk = socket.socket(); sk.bind(('127.0.0.1', 0)); port = sk.getsockname()[1]; sk.close()
    mp.spawn(_c5_rank_worker, args=(world, port, str(tmp_path), nc, nt, False, False, False, True), nprocs=world, join=True)
    s = cameras_renumbered(sd.generate_banded_scene(nc, nt))[0]
    b = Bundle.FromObservations(s['K'], s['R0'], s['t0'], s['X0'], s['obs_cam'], s['obs_pt'], s['obs_z'])
    ba = BundleAdjuster(verbose=False)
    ba.set_bundle(b)
    assert ba.backend.half_bandwidth == 9
    ba.optimize(max_steps=5)
    R1, t1, X1 = ba.backend.get_params(0)
    for r in range(world):
        d = np.load(os.path.join(str(tmp_path), 'rank%d.npz' % r))
        assert int(d['hb']) == 9 and str(d['kind']) == 'bcr' and int(d['permuted']) == 1
        assert int(d['trials']) == ba.lm_trials
        close(d['costs'], np.array(ba.costs), 1e-9)
        close(d['t'], t1, 1e-8)
        close(d['X'], X1, 1e-8)
    ba.backend.close()


def test_camera_layout_of_the_device_views(be):
    """Round-5 ADVICE: ba_reduced_device_ptrs / ba_bind_reduced_buffers expose the band in the library's INTERNAL camera order; since
    round 6 ba_get_camera_layout hands out that order.  On a scene with renumbered cameras and loop closures: new_pos is a
    permutation, the border cameras sit behind the band, and block (new_pos[p], new_pos[q]) of the device's band is block (p, q) of
    the system ba_get_reduced returns in the caller's positions."""
    nc, nt = 90, 2500
    s, perm = cameras_renumbered(_loop_scene(nc, nt, 7, [(4, 60), (15, 77)], 1))
    nt = len(s['X0'])
    flags = default_flags(nc, nt)
    load_problem(be, s['K'], s['R0'], s['t0'], s['X0'], s['obs_cam'], s['obs_pt'], s['obs_z'], *flags, O.Sensor.gaussian(1.))
    info = be.problem_info()
    new_pos, band = be.camera_layout()
    assert info['cameras_permuted'] == 1 and sorted(new_pos.tolist()) == list(range(be.nco))
    assert band == be.nco - info['border_cameras'] and info['border_cameras'] > 0
    be.linearize(0)
    be.schur(0, 3., 1e-5)
    S, b = be.get_reduced()
    be.synchronize()
    S_t, b_t = be.reduced_tensors()
    hb1 = be.half_bandwidth + 1
    Sb = S_t.cpu().numpy().reshape(be.nco, hb1, 6, 6)
    bb = b_t.cpu().numpy().reshape(be.nco, 6)
    inv = np.empty(be.nco, int)
    inv[new_pos] = np.arange(be.nco)                      # internal position -> the caller's
    for i in range(band):
        np.testing.assert_array_equal(bb[i], b[inv[i]])
        for d in range(hb1):
            if i + d < band:
                blk = S[inv[i], inv[i + d]]
                got = Sb[i, d]
                if d == 0:
                    np.testing.assert_array_equal(np.triu(got), np.triu(blk))
                else:
                    np.testing.assert_array_equal(got, blk)


def _collection_rank_worker(rank, world, port, out_dir, nc, nt):
    import sys
    for p in (ROOT, os.path.join(ROOT, 'tests')):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0')
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from pysfm_amd import Bundle, BundleAdjuster
    from pysfm_amd import synthetic_data as sd
    from pysfm_amd.distributed import ShardComm, shard_tracks
    s = sd.generate_collection_scene(nc, nt, partners=8, track_len=3)
    b = Bundle.FromObservations(s['K'], s['R0'], s['t0'], s['X0'], s['obs_cam'], s['obs_pt'], s['obs_z'])
    comm = ShardComm()
    ba = BundleAdjuster(device=0, comm=comm, verbose=False)          # all ranks on GPU 0
    ba.sparse = True                                                 # (the library's own rule starts at 1500 cameras)
    ids = shard_tracks(b, rank, world)
    ba.set_bundle(b, track_ids=ids)
    info = ba.backend.problem_info()
    ba.optimize(max_steps=5)
    X = comm.gather_points(ba)
    R, t, _ = ba.backend.get_params(0)
    np.savez(os.path.join(out_dir, 'rank%d.npz' % rank), costs=np.array(ba.costs), X=X, R=R, t=t, trials=ba.lm_trials, packed=info['packed_store'],
             kind=ba.backend.last_solve_kind, S_doubles=ba.backend.S_doubles, blocks=ba.backend.pcg_info()['blocks'], ntracks=len(ids))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 4])
def test_ranks_on_one_gpu_unordered_collection(tmp_path, world):
    """A sharded scene WITHOUT a band (round 6): an unordered photo collection split by points over `world` processes on the one GPU.
    The ranks add their [S | b] buffers element by element, so they need ONE layout: the list of the blocks that the tracks of the
    WHOLE scene define (every rank derives it from the whole bundle: ba_set_pattern_lists), stored packed; a rank's own tracks supply
    its blocks' observation pairs; every rank then solves the summed system by conjugate gradients.  The sharded walk must be the
    unsharded one."""
    import socket
    import torch.multiprocessing as mp
    from pysfm_amd import Bundle, BundleAdjuster
    from pysfm_amd import synthetic_data as sd
    nc, nt = 300, 6000
    sk = socket.socket(); sk.bind(('127.0.0.1', 0)); port = sk.getsockname()[1]; sk.close()
    mp.spawn(_collection_rank_worker, args=(world, port, str(tmp_path), nc, nt), nprocs=world, join=True)
    s = sd.generate_collection_scene(nc, nt, partners=8, track_len=3)
    b = Bundle.FromObservations(s['K'], s['R0'], s['t0'], s['X0'], s['obs_cam'], s['obs_pt'], s['obs_z'])
    ba = BundleAdjuster(verbose=False)
    ba.backend.set_option('solver', 'pcg')
    ba.set_bundle(b)
    ba.optimize(max_steps=5)
    R1, t1, X1 = ba.backend.get_params(0)
    blocks = ba.backend.pcg_info()['blocks']
    for r in range(world):
        d = np.load(os.path.join(str(tmp_path), 'rank%d.npz' % r))
        assert int(d['packed']) == 1 and str(d['kind']) == 'pcg' and int(d['blocks']) == blocks and int(d['S_doubles']) == 36 * blocks
        assert int(d['trials']) == ba.lm_trials and 0 < int(d['ntracks']) < nt
        close(d['costs'], np.array(ba.costs), 1e-9)
        close(d['X'], X1, 1e-8, 1e-11)
        close(d['t'], t1, 1e-8, 1e-11)
    ba.backend.close()
