"""Parity of the HIP path (through the C ABI) against the CPU oracle and the
reference's golden vectors.  Needs a real MI355X: run with ``-m gpu``.

Tolerances (fp64 everywhere; BASELINE north star: each step within 1e-6 relative):
  * per-observation values, blocks, S, b: 1e-11 relative to the largest entry
    (pure fp64 arithmetic; differences are summation order + fma contraction)
  * solves / updates: 1e-8 (conditioning of the reduced system)
  * LM cost trajectories and final parameters: 1e-6 (the north-star tolerance)
"""
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from conftest import load_golden
from oracle import ba_oracle as O


pytestmark = pytest.mark.gpu

TIGHT = 1e-11
SOLVE = 1e-8
LM = 1e-6


@pytest.fixture(scope='module')
def be():
    from pysfm_amd.backend import HipBackend
    b = HipBackend(0)
    yield b
    b.close()


DEFAULT_OPTIONS = dict(schur='auto', solver='auto', point_kernels='auto', fuse_cost=1, fuse_cam=1,
                       sort_points=1, gm_cap=0, gm_chunk=0, lds_window=1, fused_backsolve=1, fused_eliminate=1, device_lu=1, fast_paths=1, camera_order='auto', border=1, reuse_linearization=1, refine='auto', pcg_max_iter=0, packed_store=1)


@pytest.fixture(autouse=True)
def _default_options(request):
    """Tests pick kernels through HipBackend.set_option (ba_set_option); the shared backend goes back to the
    product path after every test."""
    yield
    if 'be' in request.fixturenames:
        b = request.getfixturevalue('be')
        for k, v in DEFAULT_OPTIONS.items():
            b.set_option(k, v)


def close(a, b, rtol, atol=0.):
    a, b = np.asarray(a, float), np.asarray(b, float)
    assert a.shape == b.shape, (a.shape, b.shape)
    if not b.size:
        return
    scale = np.max(np.abs(b))
    err = np.max(np.abs(a - b))
    assert err <= rtol * scale + atol, 'max abs err %.3e vs scale %.3e (rtol %.1e)' % (err, scale, rtol)


def sensor_of(g):
    return O.Sensor(int(g['sensor_kind']), L=g['sensor_L'], sigma=float(g['sensor_sigma']))


def sensor_params(s):
    if s.kind == O.GAUSS:
        return 0, s.L.reshape(4)
    if s.kind == O.CAUCHY:
        return 1, [s.sigma]
    return 2, [s.k]


def load_problem(be, K, R, t, X, obs_cam, obs_pt, obs_z, cam_opt_pos, pt_opt, sensor):
    be.set_problem(len(R), len(X), obs_cam, obs_pt, obs_z, K, cam_opt_pos, pt_opt)
    be.set_sensor(*sensor_params(sensor))
    be.set_params(0, R, t, X)
    # every test starts from NaNs in all LDS and all workspace buffers of the handle: a kernel that reads something it
    # (or an earlier kernel of the same computation) did not write shows up as a NaN instead of depending on what ran before
    be.debug_poison()


def default_flags(nc, nt):
    return np.arange(nc, dtype=np.int32) - 1, np.ones(nt, np.uint8)


def scene(g):
    return (g['K'], g['R'], g['t'], g['X'], g['obs_cam'], g['obs_pt'], g['obs_z'])


def hip_update(be, damping, rcond=1e-5, keep=None):
    be.linearize(0)
    be.schur(0, damping, rcond)
    n = be.nco * 6
    mask = None
    if keep is not None:
        mask = np.zeros(n, np.uint8)
        mask[keep] = 1
    be.solve_reduced(mask)
    dC = be.get_solution()
    dP = be.backsubstitute(0)
    return dC, dP


# ------------------------------------------------------------------ golden: functions
@pytest.mark.parametrize('tag', ['gauss_iso', 'gauss_diag', 'gauss_full', 'cauchy', 'cauchy2'])
def test_sensor_models_vs_reference(be, tag):
    g = load_golden('functions')
    s = O.Sensor(int(g[tag + '_kind']), L=g[tag + '_L'], sigma=float(g[tag + '_sigma']))
    be.set_sensor(*sensor_params(s))
    r, J = be.eval_sensor(g['sens_e'])
    close(r, g[tag + '_r'], 1e-13)
    close(J, g[tag + '_J'], 1e-12)


def test_huber_vs_oracle(be):
    s = O.Sensor.huber(.06)
    rs = np.random.RandomState(3)
    e = np.concatenate((rs.randn(200, 2) * .05, rs.uniform(-10, 10, (50, 2)), [[0, 0], [.06, 0], [0, .0600001]]))
    be.set_sensor(2, [.06])
    r, J = be.eval_sensor(e)
    close(r, O.sensor_residual(s, e), 1e-13)
    close(J, O.sensor_jacobian(s, e), 1e-12)


def test_fast_paths_equal_the_general_formulas(be):
    """K = I and the unit Gaussian sensor model take short cuts in ba_math.h (no K products, no 2 x 2 sensor Jacobian):
    same per-observation values, blocks, reduced system and trial as the general code (option fast_paths = 0)."""
    s = banded(40, 2000, track_len=9)
    flags = default_flags(40, 2000)
    out = {}
    for fast in (1, 0):
        be.set_option('fast_paths', fast)
        load_problem(be, s['K'], s['R0'], s['t0'], s['X0'], s['obs_cam'], s['obs_pt'], s['obs_z'], *flags, O.Sensor.gaussian(1.))
        ev = be.eval_observations(0)
        be.linearize(0)
        blk = be.get_blocks()
        info, cost = be.lm_trial(2., 1e-5, None)
        assert info == 0
        out[fast] = [ev['e'], ev['r'], ev['Jc'], ev['Jp'], blk['HCC'], blk['HPP'], blk['bC'], blk['bP'], *be.get_reduced(), be.get_params(1)[2],
                     np.array([cost, be.cost(0)])]
    for x, y in zip(out[1], out[0]):
        close(x, y, 1e-13)
    a = (s['K'], s['R0'], s['t0'], s['X0'], s['obs_cam'], s['obs_pt'], s['obs_z'])
    r0, Jc0, Jp0 = O.jacobians(O.Sensor.gaussian(1.), *a)
    close(out[1][2], Jc0, 1e-13)
    close(out[1][3], Jp0, 1e-13)


# ------------------------------------------------------------------ golden: 4x10 Cauchy scene
def test_per_observation_vs_reference(be):
    g = load_golden('scene_4x10_cauchy')
    load_problem(be, *scene(g), g['l0_cam_opt_pos'], g['l0_pt_opt'], sensor_of(g))
    out = be.eval_observations(0)
    close(out['e'], g['e'], 1e-13)
    close(out['r'], g['r'], 1e-13)
    close(out['Jc'], g['Jc'], 1e-12)
    close(out['Jp'], g['Jp'], 1e-12)
    close(be.cost(0), g['l0_cost'], 1e-13)


@pytest.mark.parametrize('lam,tag', [(0., 'l0_'), (2., 'l2_')])
def test_blocks_and_schur_vs_reference(be, lam, tag):
    g = load_golden('scene_4x10_cauchy')
    load_problem(be, *scene(g), g[tag + 'cam_opt_pos'], g[tag + 'pt_opt'], sensor_of(g))
    be.linearize(0, store_W=True)
    blk = be.get_blocks(W=True)
    for k in ('HCC', 'bC', 'HPP', 'bP', 'W'):
        close(blk[k], g[tag + k], TIGHT)
    be.schur(0, lam, 1e-5)
    close(be.get_point_inverses(), g[tag + 'HPP_inv'], 1e-10)
    S, b = be.get_reduced()
    close(S, g[tag + 'S'], TIGHT)
    close(b, g[tag + 'b'], TIGHT)
    close(S, S.transpose(1, 0, 3, 2), 1e-13)                      # off-diagonal blocks are mirrored exactly
    if lam > 0:
        dC, dP = hip_update(be, lam)
        close(dC, g[tag + 'dC'], SOLVE)
        close(dP, g[tag + 'dP'], SOLVE)
        Sflat = S.transpose(0, 2, 1, 3).reshape(18, 18)
        assert np.sum((np.concatenate((-dC.reshape(-1), -dP.reshape(-1))) - g['dense_delta_l2']) ** 2) <= 1e-7
        assert np.linalg.norm(Sflat @ dC.reshape(-1) - b.reshape(-1)) < 1e-9 * np.linalg.norm(b)
    else:
        Sflat = S.transpose(0, 2, 1, 3).reshape(18, 18)
        assert np.sum((Sflat - g['dense_S_l0']) ** 2) <= 1e-7      # numpy_test.py:112-118 criterion
        assert np.sum((b.reshape(-1) - g['dense_b_l0']) ** 2) <= 1e-7


def test_subset_and_masks_vs_reference(be):
    g = load_golden('scene_subset')
    load_problem(be, *scene(g), g['l2_cam_opt_pos'], g['l2_pt_opt'], sensor_of(g))
    be.linearize(0)
    be.schur(0, 2., 1e-5)
    S, b = be.get_reduced()
    assert S.shape == (1, 1, 6, 6)
    close(S, g['l2_S'], TIGHT)
    close(b, g['l2_b'], TIGHT)
    close(be.cost(0), g['l2_cost'], 1e-13)
    dC, dP = hip_update(be, 2.)
    close(-dC, g['update_l2_motion'], SOLVE)
    close(-dP[g['l2_pt_opt'].astype(bool)], g['update_l2_structure'], SOLVE)


@pytest.mark.parametrize('name', ['scene_oleg_10x50', 'scene_oleg_40x100'])
def test_pixel_unit_scenes_vs_reference(be, name):
    """tracks of 10..40 observations (more than one kTile for 40), K with f=1500."""
    g = load_golden(name)
    load_problem(be, *scene(g), g['l10_cam_opt_pos'], g['l10_pt_opt'], sensor_of(g))
    be.linearize(0)
    be.schur(0, 10., 1e-5)
    S, b = be.get_reduced()
    close(b, g['l10_b'], TIGHT)
    close(be.get_point_inverses(), g['l10_HPP_inv'], 1e-9)
    if 'l10_S' in g:
        close(S, g['l10_S'], TIGHT)
    else:
        assert abs(np.linalg.norm(S) / g['l10_S_fro'] - 1) < 1e-11
    dC, dP = hip_update(be, 10.)
    close(-dC, g['update_l10_motion'], 1e-7)
    close(-dP, g['update_l10_structure'], 1e-7)


def test_oleg_full_100x1000_vs_reference(be):
    """100 000 integer-pixel observations, dense tracks of 100 (4 tiles per track, 10 units each)."""
    g = load_golden('scene_oleg_100x1000')
    z = g['obs_z'].astype(float)
    nc, nt = 100, 1000
    load_problem(be, g['K'], g['R'], g['t'], g['X'], g['obs_cam'], g['obs_pt'], z, *default_flags(nc, nt), sensor_of(g))
    close(be.cost(0), g['l10_cost'], 1e-12)
    be.linearize(0)
    be.schur(0, 10., 1e-5)
    S, b = be.get_reduced()
    close(b, g['l10_b'], 1e-10)
    assert abs(np.linalg.norm(S) / g['l10_S_fro'] - 1) < 1e-11
    dC, dP = hip_update(be, 10.)
    close(dC, g['l10_dC'], 1e-6)
    close(dP[:20], g['l10_dP_head'], 1e-6)
    assert abs(np.linalg.norm(dP) / g['l10_dP_norm'] - 1) < 1e-6


# ------------------------------------------------------------------ LM through the public API
@pytest.mark.parametrize('name,steps', [('scene_4x10_cauchy', 10), ('scene_5x50_gauss', 5),
                                        ('scene_5x50_cauchy_masked', 8), ('scene_planar_lm', 50)])
def test_bundle_adjuster_optimize_vs_reference(name, steps):
    from pysfm_amd import Bundle, BundleAdjuster, sensor_model
    g = load_golden(name)
    if int(g['sensor_kind']) == 0:
        m = sensor_model.GaussianModel(1.)
        m.L = g['sensor_L']
    else:
        m = sensor_model.CauchyModel(float(g['sensor_sigma']))
    b0 = Bundle.FromObservations(*scene(g), sensor_model=m)
    ba = BundleAdjuster(b0, verbose=False)
    ba.optimize(max_steps=steps)
    assert ba.num_steps == int(g['lm_num_steps'])
    assert ba.converged == bool(g['lm_converged'])
    close(ba.costs, g['lm_costs'], LM)
    out = ba.bundle
    assert out is not b0 and np.array_equal(b0.Rs(), g['R'])
    close(out.Rs(), g['lm_R'], LM)
    close(out.ts(), g['lm_t'], LM, 1e-9)
    close(out.reconstruction, g['lm_X'], LM)
    # the reference's unit-test call shapes (bundle_adjuster_unittest.py:47-67)
    ba2 = BundleAdjuster(b0, verbose=False)
    mu, su = ba2.compute_update(2.)
    omu, osu = O.compute_update(sensor_of(g), *scene(g), *default_flags(len(g['R']), len(g['X'])), damping=2.)
    close(mu, omu, SOLVE)
    close(su, osu, SOLVE)


def test_renumbered_cameras_and_loop_closures_vs_reference(be):
    """tests/golden/scene_loop_closure_60x424.npz (oracle/gen_golden_layout.py): numbers of the REFERENCE on a scene whose 60 cameras are
    numbered at random and four of whose points tie cameras 30 apart along the sequence.  The reference's reduced system is dense
    (bundle_adjuster.py:259-312); the library orders the cameras itself and makes the far ends of the loop closures a border of the
    band (ba_order.hip, ba_border.h) - S, b, dC, dP in the CALLER's positions, the update and the LM walk must be the reference's."""
    from pysfm_amd import Bundle, BundleAdjuster, sensor_model
    g = load_golden('scene_loop_closure_60x424')
    load_problem(be, *scene(g), g['l2_cam_opt_pos'], g['l2_pt_opt'].astype(np.uint8), sensor_of(g))
    info = be.problem_info()
    assert info['cameras_permuted'] == 1 and info['border_cameras'] > 0 and info['half_bandwidth'] <= 13 and info['caller_half_bandwidth'] > 30, info
    be.linearize(0)
    be.schur(0, 2., 1e-5)
    S, b = be.get_reduced()
    close(S, g['l2_S'], TIGHT)
    close(b, g['l2_b'], TIGHT)
    dC, dP = hip_update(be, 2.)
    assert be.last_solve_kind == 'bcr'
    close(dC, g['l2_dC'], SOLVE)
    close(dP, g['l2_dP'], SOLVE)
    close(be.cost(0), g['l2_cost'], 1e-12)
    b0 = Bundle.FromObservations(*scene(g), sensor_model=sensor_model.GaussianModel(1.))
    ba = BundleAdjuster(b0, verbose=False)
    mu, su = ba.compute_update(2.)
    close(mu, g['update_l2_motion'], SOLVE)
    close(su, g['update_l2_structure'], SOLVE)
    ba = BundleAdjuster(b0, verbose=False)
    ba.optimize(max_steps=4)
    assert ba.num_steps == int(g['lm_num_steps']) and ba.converged == bool(g['lm_converged'])
    close(ba.costs, g['lm_costs'], LM)
    close(ba.bundle.Rs(), g['lm_R'], LM)
    close(ba.bundle.ts(), g['lm_t'], LM, 1e-9)
    close(ba.bundle.reconstruction, g['lm_X'], LM)


def test_bundle_api_on_device():
    from pysfm_amd import Bundle, sensor_model
    g = load_golden('scene_4x10_cauchy')
    b = Bundle.FromObservations(*scene(g), sensor_model=sensor_model.CauchyModel(.05))
    n = 11
    i, j = int(g['obs_cam'][n]), int(g['obs_pt'][n])
    close(b.reproj_error(i, j), g['e'][n], 1e-13)
    close(b.residual(i, j), g['r'][n], 1e-13)
    Jc, Jp = b.Jresidual(i, j)
    close(Jc, g['Jc'][n], 1e-12)
    close(Jp, g['Jp'][n], 1e-12)
    close(b.complete_cost(), g['complete_cost'], 1e-13)
    r, J = O.dense_jacobian(sensor_of(g), *scene(g))
    close(b.Jresiduals(), J, 1e-12)
    for m in (sensor_model.GaussianModel([2., 3.]), sensor_model.CauchyModel(2.), sensor_model.HuberModel(.7)):
        assert sensor_model.validate(m)                      # sensor_model.py:76-99 on the device path


# ------------------------------------------------------------------ seeded scenes vs the oracle
def banded(nc, nt, **kw):
    from pysfm_amd import synthetic_data as sd
    return sd.generate_banded_scene(nc, nt, **kw)


@pytest.mark.parametrize('sensor,outliers', [(O.Sensor.gaussian(1.), 0.), (O.Sensor.cauchy(.05), .1),
                                             (O.Sensor.huber(.06), .1)])
def test_banded_scene_full_step_vs_oracle(be, sensor, outliers):
    s = banded(60, 3000, outlier_frac=outliers)
    nc, nt = 60, 3000
    flags = default_flags(nc, nt)
    a = (s['K'], s['R0'], s['t0'], s['X0'], s['obs_cam'], s['obs_pt'], s['obs_z'])
    load_problem(be, *a, *flags, sensor)
    close(be.cost(0), O.cost(sensor, *a, *flags), 1e-12)
    mu, su, parts = O.compute_update(sensor, *a, *flags, damping=10., return_parts=True)
    be.linearize(0, store_W=True)
    blk = be.get_blocks(W=True)
    for k in ('HCC', 'bC', 'HPP', 'bP', 'W'):
        close(blk[k], parts[k], TIGHT)
    be.schur(0, 10., 1e-5)
    S, b = be.get_reduced()
    close(S, parts['S'], TIGHT)
    close(b, parts['b'], TIGHT)
    close(be.get_point_inverses(), parts['HPP_inv'], 1e-9)
    dC, dP = hip_update(be, 10.)
    close(-dC, mu, SOLVE)
    close(-dP, su, SOLVE)
    be.apply_update(0, 1)
    R2, t2, X2 = O.apply_update(s['R0'], s['t0'], s['X0'], mu, su, flags[0], flags[1])
    Rh, th, Xh = be.get_params(1)
    close(Rh, R2, 1e-9)
    close(th, t2, 1e-9)
    close(Xh, X2, 1e-9)
    close(be.cost(1), O.cost(sensor, s['K'], R2, t2, X2, *a[4:], *flags), 1e-8)


def test_config2_gauss_newton_100x10k(be):
    """BASELINE config 2: 100 cameras / 10k points / 100k obs, lambda = 0, Jacobian + Schur only."""
    s = banded(100, 10000)
    flags = default_flags(100, 10000)
    a = (s['K'], s['R0'], s['t0'], s['X0'], s['obs_cam'], s['obs_pt'], s['obs_z'])
    sensor = O.Sensor.gaussian(1.)
    load_problem(be, *a, *flags, sensor)
    HCC, HPP, W, bC, bP = O.normal_blocks(sensor, *a, 100, 10000)
    HPPi = O.invert_point_blocks(HPP, 1e-5)
    S0, b0 = O.schur_complement(HCC, HPPi, W, bC, bP, s['obs_cam'], s['obs_pt'], flags[0])
    be.linearize(0)
    be.schur(0, 0., 1e-5)
    S, b = be.get_reduced()
    close(S, S0, 1e-10)
    close(b, b0, 1e-10)


def test_rank_deficient_point_blocks_use_pinv_cutoff(be):
    """Tracks seen by a single camera have a rank-2 HPP: numpy.linalg.pinv(., 1e-5)
    drops the null direction (bundle_adjuster.py:256).  Plain inverse must raise."""
    s = banded(12, 40, track_len=10)
    keep = np.ones(len(s['obs_cam']), bool)
    for k in (3, 17, 29):                       # cut these tracks down to one observation
        idx = np.nonzero(s['obs_pt'] == k)[0]
        keep[idx[1:]] = False
    keep[s['obs_pt'] == 8] = False              # and one track with no observation at all
    cam, pt, z = s['obs_cam'][keep], s['obs_pt'][keep], s['obs_z'][keep]
    flags = default_flags(12, 40)
    sensor = O.Sensor.gaussian(1.)
    a = (s['K'], s['R0'], s['t0'], s['X0'], cam, pt, z)
    load_problem(be, *a, *flags, sensor)
    # lambda = 0: a one-observation track has an exactly rank-2 HPP -> the cutoff applies
    HCC, HPP, W, bC, bP = O.normal_blocks(sensor, *a, 12, 40)
    HPPi = O.invert_point_blocks(HPP, 1e-5)
    S0, b0 = O.schur_complement(HCC, HPPi, W, bC, bP, cam, pt, flags[0])
    be.linearize(0)
    be.schur(0, 0., 1e-5)
    Hi = be.get_point_inverses()
    close(Hi, HPPi, 1e-9)
    assert np.all(Hi[8] == 0)
    for k in (3, 17, 29):
        assert np.linalg.matrix_rank(Hi[k], tol=1e-9 * np.abs(Hi[k]).max()) == 2
        assert np.linalg.matrix_rank(HPP[k], tol=1e-9 * np.abs(HPP[k]).max()) == 2
    S, b = be.get_reduced()
    close(S, S0, 1e-10)
    close(b, b0, 1e-10)
    # lambda > 0 regularises them; the whole step still matches
    mu, su, parts = O.compute_update(sensor, *a, *flags, damping=0.5, return_parts=True)
    dC, dP = hip_update(be, .5)
    close(be.get_point_inverses(), parts['HPP_inv'], 1e-9)
    close(-dC, mu, SOLVE)
    close(-dP, su, SOLVE)
    from pysfm_amd.backend import SingularPointBlock
    be.linearize(0)
    with pytest.raises(SingularPointBlock):
        be.schur(0, .5, None)


def test_plain_inverse_mode_matches_numpy_inv(be):
    g = load_golden('scene_4x10_cauchy')
    load_problem(be, *scene(g), g['l2_cam_opt_pos'], g['l2_pt_opt'], sensor_of(g))
    be.linearize(0)
    be.schur(0, 2., None)
    HCC, HPP, W, bC, bP = O.normal_blocks(sensor_of(g), *scene(g), 4, 10)
    close(be.get_point_inverses(), np.linalg.inv(O.damp_blocks(HPP, 2.)), 1e-11)


def test_empty_and_degenerate_inputs(be):
    K = np.eye(3)
    R = np.stack([np.eye(3)] * 3)
    t = np.zeros((3, 3))
    X = np.ones((4, 3))
    flags = default_flags(3, 4)
    empty_i, empty_z = np.zeros(0, np.int32), np.zeros((0, 2))
    be.set_problem(3, 4, empty_i, empty_i, empty_z, K, *flags)
    be.set_sensor(0, np.eye(2).reshape(4))
    be.set_params(0, R, t, X)
    assert be.cost(0) == 0.
    be.linearize(0)
    be.schur(0, 1., 1e-5)
    S, b = be.get_reduced()
    assert not S.any() and not b.any()
    # an empty shard of a sharded adjuster (no tracks at all): its half of a trial runs and contributes zeros
    be.set_problem(3, 0, empty_i, empty_i, empty_z, K, flags[0], np.zeros(0, np.uint8))
    be.set_sensor(0, np.eye(2).reshape(4))
    be.set_params(0, R, t, np.zeros((0, 3)))
    be.lm_trial_begin(10., 1e-5)
    S, b = be.get_reduced()
    assert not S.any() and not b.any() and be.cost(0) == 0.
    with pytest.raises(ValueError):
        be.set_option('schur', 'no-such-kernel')
    with pytest.raises(ValueError):
        be.set_option('no-such-option', 1)
    be.set_problem(3, 4, [0, 1], [1, 0], np.zeros((2, 2)), K, *flags)          # any observation order is accepted
    with pytest.raises(ValueError):
        be.set_problem(3, 4, [1, 1], [2, 2], np.zeros((2, 2)), K, *flags)      # a (camera, track) pair twice
    with pytest.raises(ValueError):
        be.set_problem(3, 4, [0, 5], [0, 1], np.zeros((2, 2)), K, *flags)      # camera out of range


# ------------------------------------------------------------------ scenes without a band: conjugate gradients over the blocks the tracks define
@pytest.mark.parametrize('damping,masked', [(10., False), (1., False), (10., True), (.1, False)])
def test_unordered_photo_collection_solved_by_conjugate_gradients_vs_oracle(be, damping, masked):
    """An unordered photo collection (every camera shares tracks with cameras drawn at random from ALL the others: no order of the
    cameras makes the reduced system a narrow band) - the reference's dense S does not care (bundle_adjuster.py:259-312).  The
    library's solver for such scenes: conjugate gradients with a block-Jacobi preconditioner over the blocks of S the tracks define
    (csrc/ba_pcg.h; chosen by itself from 1500 cameras on, by option here).  S, b against the oracle as for any scene; dC and dP
    against the oracle's LAPACK solve to 1e-8; masked parameters exactly zero; the dense Cholesky of the same device-resident system
    agrees to 1e-9."""
    from pysfm_amd import synthetic_data as sd
    nc, nt = 300, 6000
    s = sd.generate_collection_scene(nc, nt, partners=8, track_len=3)
    flags = default_flags(nc, nt)
    sensor = O.Sensor.cauchy(.05)
    a = (s['K'], s['R0'], s['t0'], s['X0'], s['obs_cam'], s['obs_pt'], s['obs_z'])
    load_problem(be, *a, *flags, sensor)
    assert be.half_bandwidth > 150                                   # (no band worth the name)
    n = be.nco * 6
    mask = None
    if masked:
        mask = (np.arange(n) % 9 != 4).astype(np.uint8)
        mask[6 * 17:6 * 19] = 0
    be.set_option('solver', 'pcg')
    be.linearize(0)
    be.schur(0, damping, 1e-5)
    be.solve_reduced(mask)
    assert be.last_solve_kind == 'pcg' and be.last_solve_path == 'pcg'
    info = be.pcg_info()
    assert 0 < info['iterations'] < 1000 and info['rel_residual'] <= 1e-12 and info['band_fill'] < .25, info
    x = be.get_solution().reshape(-1)
    dP = be.backsubstitute(0)
    mu, su, parts = O.compute_update(sensor, *a, *flags, damping=damping, cam_param_mask=None if mask is None else mask.astype(bool), return_parts=True)
    S, b = be.get_reduced()
    close(S, parts['S'], 1e-11)
    close(b, parts['b'], 1e-11)
    # the pattern: exactly the blocks the oracle's S has
    nz = np.abs(parts['S']).sum(axis=(2, 3)) > 0
    assert info['blocks'] == int(np.triu(nz).sum())
    close(-x.reshape(-1, 6), mu, 1e-8)
    close(-dP, su, 1e-8)
    assert mask is None or np.all(x[mask == 0] == 0)
    be.set_option('solver', 'dense')
    be.solve_reduced(mask)
    assert be.last_solve_kind == 'dense_cholesky'
    close(x, be.get_solution().reshape(-1), 1e-9)


def test_unordered_photo_collection_lm_walk_vs_oracle():
    """optimize() of an unordered collection with every reduced system solved by conjugate gradients: the walk of the oracle
    (LAPACK's LU of the dense S, bundle_adjuster.py:117-162) - same decisions, costs to 1e-6 - down to damping 1e-4, where the
    block-Jacobi preconditioned iteration needs a few hundred steps."""
    from pysfm_amd import Bundle, BundleAdjuster, sensor_model, synthetic_data as sd
    nc, nt = 300, 6000
    s = sd.generate_collection_scene(nc, nt, partners=8, track_len=3)
    a = (s['K'], s['R0'], s['t0'], s['X0'], s['obs_cam'], s['obs_pt'], s['obs_z'])
    trace = []
    ref = O.lm_optimize(O.Sensor.gaussian(1.), *a, *default_flags(nc, nt), max_steps=7, trace=trace)
    b = Bundle.FromObservations(*a, sensor_model=sensor_model.GaussianModel(1.))
    ba = BundleAdjuster(verbose=False)
    ba.backend.set_option('solver', 'pcg')               # (before the problem is set: the packed store, no band at all)
    ba.set_bundle(b)
    assert ba.backend.problem_info()['packed_store'] == 1 and ba.backend.S_doubles == 36 * ba.backend.pcg_info()['blocks']
    ba.optimize(max_steps=7)
    assert ba.backend.last_solve_kind == 'pcg'
    assert [(d, o == 'accepted') for d, o, _ in ba.trial_log] == [(t['damping'], t['next'] < t['cur']) for t in trace]
    close(ba.costs, ref['costs'], 1e-6)
    close(ba.bundle.reconstruction, ref['X'], 1e-6)
    ba.backend.close()


def test_conjugate_gradients_report_what_they_cannot_solve(be):
    """A reduced system that is not positive definite (negative damping) and one they cannot finish within their budget: *info > 0 -
    the caller raises the damping, as for a LinAlgError of the reference (bundle_adjuster.py:302-305) - never a wrong answer."""
    from pysfm_amd import synthetic_data as sd
    from pysfm_amd.backend import ReducedSystemSingular
    nc, nt = 200, 3000
    s = sd.generate_collection_scene(nc, nt, partners=8, track_len=3)
    load_problem(be, s['K'], s['R0'], s['t0'], s['X0'], s['obs_cam'], s['obs_pt'], s['obs_z'], *default_flags(nc, nt), O.Sensor.gaussian(1.))
    be.set_option('solver', 'pcg')
    be.linearize(0)
    be.schur(0, -.9, 1e-5)
    with pytest.raises(ReducedSystemSingular):
        be.solve_reduced(None)
    be.schur(0, 1., 1e-5)
    be.set_option('pcg_max_iter', 3)
    with pytest.raises(ReducedSystemSingular):
        be.solve_reduced(None)
    assert be.last_solve_kind == 'pcg'
    be.set_option('pcg_max_iter', 0)
    be.solve_reduced(None)
    assert be.pcg_info()['rel_residual'] <= 1e-12


# ------------------------------------------------------------------ full-size properties
@pytest.fixture(scope='module')
def config3():
    return banded(1000, 100000)


# ------------------------------------------------------------------ the solve where the walk is sensitive: a golden [S | b], the same bits every run
EPS = 2. ** -52


@pytest.fixture(scope='module')
def golden_reduced():
    """tests/golden/config3_reduced_damping1e-3.npz (oracle/gen_golden_reduced.py): the ORACLE's reduced system of config 3 after five
    LM steps at damping 1e-3 (condition number ~1e13), LAPACK's LU and Cholesky solutions of exactly these numbers."""
    g = load_golden('config3_reduced_damping1e-3')
    band, rhs = g['band'], g['b'].reshape(-1)
    nco, hb = band.shape[0], band.shape[1] - 1
    A = np.zeros((nco, nco, 6, 6))
    for d in range(hb + 1):
        i = np.arange(nco - d)
        A[i, i + d] = band[i, d]
        A[i + d, i] = band[i, d].transpose(0, 2, 1)
    A = A.transpose(0, 2, 1, 3).reshape(6 * nco, 6 * nco)
    return dict(g, A=A, rhs=rhs, nco=nco, hb=hb)


def solve_golden_reduced(be, config3, g, refine, mask=None):
    """The golden [S | b] copied into the device's band (no device arithmetic before the solve), solved by ba_solve_reduced."""
    import torch
    s = config3
    load_problem(be, s['K'], s['R0'], s['t0'], s['X0'], s['obs_cam'], s['obs_pt'], s['obs_z'], *default_flags(1000, 100000), O.Sensor.gaussian(1.))
    assert be.nco == g['nco'] and be.half_bandwidth == g['hb'] and not be.problem_info()['cameras_permuted']
    be.set_option('refine', refine)
    be.linearize(0)
    be.schur(0, float(g['damping']), 1e-5)               # (state and the damping the option looks at; the golden system replaces what it formed)
    be.synchronize()                                         # (the handle works on a stream of its own: its kernels first, then the copies)
    S_t, b_t = be.reduced_tensors()
    S_t.copy_(torch.from_numpy(np.ascontiguousarray(g['band']).reshape(-1)))
    b_t.copy_(torch.from_numpy(g['rhs']))
    torch.cuda.synchronize()
    before = be.problem_info()['solves_refined']
    be.solve_reduced(mask)
    assert be.last_solve_kind == 'bcr'
    assert be.problem_info()['solves_refined'] - before == (0 if refine == '0' else 1)
    return be.get_solution().reshape(-1)


def backward_error(A, rhs, x):
    """||A x - b|| / || |A| |x| + |b| ||: the normwise backward error in the measure of Oettli and Prager (what LAPACK's own refinement
    drivers bound) - ||A||_2 ||x|| in its place is 1e3 times larger on these systems and lets everything through."""
    return np.linalg.norm(A @ x - rhs) / np.linalg.norm(np.abs(A) @ np.abs(x) + np.abs(rhs))


def test_golden_reduced_system_backward_error(be, config3, golden_reduced):
    """bundle_adjuster.py:302-305 solves the reduced system with LAPACK's gesv.  On the golden system the device's solution - the block
    cyclic reduction followed by one step of iterative refinement through its kept factors (csrc/ba_bcr_refine.h; on by default below
    damping 1e-2) - has a backward error ||S x - b|| / || |S| |x| + |b| || of at most 2 eps AND leaves a residual ||S x - b|| / ||b|| no
    larger than 1.5 times the larger of LAPACK's two (LU, Cholesky).  The input is the same bits every run; the solution is not (the
    cyclic reduction adds its Schur complements with atomics): scripts/golden_solve_spread.py, profiles/r06_golden_solve_spread.txt -
    over 40 runs the refined solution's backward error is at most 0.46 eps (LAPACK: 0.58), its residual at most 3.8e-16 (LAPACK:
    4.8e-16 and 2.8e-16); unrefined: 1.48 eps, 1.2e-15."""
    g = golden_reduced
    A, rhs = g['A'], g['rhs']
    res = lambda x: np.linalg.norm(A @ x - rhs) / np.linalg.norm(rhs)
    lapack = max(res(g['x_lu']), res(g['x_chol']))
    x = solve_golden_reduced(be, config3, g, 'auto')
    assert np.all(np.isfinite(x))
    assert backward_error(A, rhs, x) <= 2 * EPS, (backward_error(A, rhs, x) / EPS, backward_error(A, rhs, g['x_lu']) / EPS)
    assert res(x) <= 1.5 * lapack, (res(x), res(g['x_lu']), res(g['x_chol']))
    # the cyclic reduction alone (refine = 0) is within a few units of round-off as well - and the step does not make it worse
    x0 = solve_golden_reduced(be, config3, g, '0')
    assert backward_error(A, rhs, x0) <= 4 * EPS and res(x0) <= 4 * lapack, (res(x0), lapack)
    assert res(x) <= max(1.05 * res(x0), lapack)


def test_golden_reduced_system_solution_vs_lapack(be, config3, golden_reduced):
    """The device's solution of the golden system agrees with LAPACK's Cholesky solution at least as closely as LAPACK's own LU does
    (times two): what is left is the system's sensitivity (condition number ~1e13), not the solver's.  Separate from the backward-error
    test so that neither can hide the other."""
    g = golden_reduced
    x = solve_golden_reduced(be, config3, g, 'auto')
    d_lu = np.linalg.norm(g['x_lu'] - g['x_chol'])
    assert np.linalg.norm(x - g['x_chol']) <= 2. * d_lu + 1e-14 * np.linalg.norm(g['x_chol']), (np.linalg.norm(x - g['x_chol']), d_lu)
    assert np.linalg.norm(x - g['x_lu']) <= 3. * d_lu + 1e-14 * np.linalg.norm(g['x_chol'])


def test_golden_reduced_system_refinement_with_masked_parameters(be, config3, golden_reduced):
    """param_mask deletes rows and columns (bundle_adjuster.py:292-300): the refinement's residual is that of the REDUCED system -
    masked rows stay exactly zero, the rest solves the deleted system as LAPACK does."""
    g = golden_reduced
    n = 6 * g['nco']
    mask = (np.arange(n) % 7 != 2).astype(np.uint8)
    mask[6 * 400:6 * 403] = 0                                  # three whole cameras
    keep = np.nonzero(mask)[0]
    x = solve_golden_reduced(be, config3, g, '1', mask)
    assert np.all(x[mask == 0] == 0.)
    A, rhs = g['A'][np.ix_(keep, keep)], g['rhs'][keep]
    ref = np.linalg.solve(A, rhs)
    # (not "1.5 x LAPACK's residual": on this deleted system LAPACK's happens to be 0.19 eps in the backward-error measure, the device's
    #  0.1 ... 0.6 eps from run to run, and a step of refinement - the device's or one done on the host in long double - does not lower
    #  it: rounding x + dx costs as much; profiles/r06_golden_solve_spread.txt)
    assert backward_error(A, rhs, x[keep]) <= 2 * EPS, (backward_error(A, rhs, x[keep]) / EPS, backward_error(A, rhs, ref) / EPS)
    x0 = solve_golden_reduced(be, config3, g, '0', mask)
    assert backward_error(A, rhs, x0[keep]) <= 4 * EPS
    close(x[keep], x0[keep], 1e-4)                             # (two answers of an ill-conditioned system: equal to its sensitivity)


def test_config3_properties_1000x100k(be, config3):
    """BASELINE config 3 size (1000 cams / 100k pts / 1M obs).  The oracle is too slow for
    the full Schur here, so check size-independent properties plus the cheap oracle parts."""
    s = config3
    nc, nt = 1000, 100000
    flags = default_flags(nc, nt)
    sensor = O.Sensor.gaussian(1.)
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
    # (1) symmetry and band structure (each track spans 10 consecutive cameras)
    close(S, S.transpose(1, 0, 3, 2), 1e-13)
    i, j = np.nonzero(np.abs(S).sum(axis=(2, 3)))
    assert np.max(np.abs(i - j)) == 9
    # (2) b against the oracle (O(N) to compute)
    HPPi = O.invert_point_blocks(O.damp_blocks(HPP, 10.), 1e-5)
    T = W @ HPPi[s['obs_pt']]
    b0 = np.zeros((nc - 1, 6))
    pos = flags[0][s['obs_cam']]
    kk = pos >= 0
    b0[np.arange(nc - 1)] = bC[1:]
    np.subtract.at(b0, pos[kk], np.einsum('nij,nj->ni', T[kk], bP[s['obs_pt'][kk]]))
    close(b, b0, 1e-10)
    # (3) linearity over points: S(first half) + S(second half) - diag(HCC) terms == S(all)
    half = nt // 2
    parts = []
    for lo, hi in ((0, half), (half, nt)):
        m = (s['obs_pt'] >= lo) & (s['obs_pt'] < hi)
        be.set_problem(nc, hi - lo, s['obs_cam'][m], s['obs_pt'][m] - lo, s['obs_z'][m], s['K'], flags[0],
                       np.ones(hi - lo, np.uint8))
        be.set_sensor(0, np.eye(2).reshape(4))
        be.set_params(0, s['R0'], s['t0'], s['X0'][lo:hi])
        be.linearize(0)
        be.schur(0, 10., 1e-5)
        parts.append(be.get_reduced())
    close(parts[0][0] + parts[1][0], S, 1e-11)
    close(parts[0][1] + parts[1][1], b, 1e-11)
    # (4) a few rows of S against the oracle restricted to the tracks that touch them
    rows = [0, 499, 998]
    touching = np.zeros(nt, bool)
    for r in rows:
        touching[s['obs_pt'][pos == r]] = True
    m = touching[s['obs_pt']]
    S0, _ = O.schur_complement(O.damp_blocks(HCC, 10.), HPPi, W[m], bC, bP, s['obs_cam'][m], s['obs_pt'][m], flags[0])
    for r in rows:
        close(S[r], S0[r], 1e-10)


@pytest.mark.parametrize('init_mode', ['pose', 'params'])
def test_config3_full_lm_converges(config3, init_mode):
    """init_mode 'params' is round 1's start (Camera.perturb on the raw parameters: cameras thrown up to 2 units off,
    initial cost 7.6e6): the hard case for the LM schedule."""
    from pysfm_amd import Bundle, BundleAdjuster
    from pysfm_amd.synthetic_data import reprojection_rmse
    s = config3 if init_mode == 'pose' else banded(1000, 100000, init_mode='params')
    b0 = Bundle.FromObservations(s['K'], s['R0'], s['t0'], s['X0'], s['obs_cam'], s['obs_pt'], s['obs_z'])
    ba = BundleAdjuster(b0, verbose=False)
    ba.optimize(max_steps=25)
    assert all(c1 < c0 for c0, c1 in zip(ba.costs, ba.costs[1:]))       # accepted steps only ever lower the cost
    out = ba.bundle
    be = ba.backend
    e = be.eval_observations(0, e=True, r=False, Jc=False, Jp=False)['e']
    rmse = reprojection_rmse(e)
    assert rmse < 1.1 * .02 * np.sqrt(2)                                 # down to the measurement noise (sigma .02 per axis)
    assert ba.costs[-1] < (.2 if init_mode == 'params' else .7) * ba.costs[0]      # ('pose' starts within 1.2 x of the noise floor)
    # near-idempotence: restarting from the result never raises the cost and gains < 1 %
    ba2 = BundleAdjuster(out, verbose=False)
    ba2.optimize(max_steps=3)
    assert ba2.costs[0] == pytest.approx(ba.costs[-1], rel=1e-9)
    assert ba2.costs[-1] <= ba2.costs[0] and ba2.costs[0] - ba2.costs[-1] <= 1e-2 * ba.costs[-1]


def _device_lu(be, mask):
    """The same device-resident system through LU with partial pivoting down the band (option solver = lu, ba_band_lu.h):
    an independent factorisation to compare the Cholesky solvers with."""
    be.set_option('solver', 'lu')
    be.solve_reduced(mask)
    assert be.last_solve_kind == 'band_lu'
    x = be.get_solution().reshape(-1)
    be.set_option('solver', 'auto')
    return x


def test_band_solver_vs_dense_lu(be):
    """k_band_solve (block Cholesky on the band) against the dense LU path on the same
    device-resident system, with and without masked camera parameters."""
    s = banded(80, 4000, track_len=7)
    flags = default_flags(80, 4000)
    sensor = O.Sensor.cauchy(.05)
    a = (s['K'], s['R0'], s['t0'], s['X0'], s['obs_cam'], s['obs_pt'], s['obs_z'])
    load_problem(be, *a, *flags, sensor)
    assert be.half_bandwidth == 6 and be.S_doubles == 79 * 7 * 36
    be.linearize(0)
    be.schur(0, 1., 1e-5)
    n = 79 * 6
    for mask in (None, (np.arange(n) % 5 != 0).astype(np.uint8), np.r_[np.zeros(12, np.uint8), np.ones(n - 12, np.uint8)]):
        be.solve_reduced(mask)
        assert be.last_solve_path == 'band'
        x = be.get_solution().reshape(-1)
        xd = _device_lu(be, mask)
        close(x, xd, 1e-9)
        close(xd, _dense_reference(be, mask), 1e-9)
        if mask is not None:
            assert np.all(x[mask == 0] == 0)
        mu, su = O.compute_update(sensor, *a, *flags, damping=1., cam_param_mask=None if mask is None else mask.astype(bool))
        close(-x.reshape(-1, 6), mu, SOLVE)
        close(-be.backsubstitute(0), su, SOLVE)


@pytest.mark.parametrize('nc,L', [(80, 7), (37, 3), (200, 10), (23, 2), (64, 5), (1500, 10), (700, 11), (400, 12), (300, 13), (500, 16), (400, 22), (97, 19), (300, 23), (260, 24),
                                  (120, 4), (150, 6), (160, 8), (180, 9),      # (with these every half-bandwidth 1..11 of the node kernels: 16 x 16 tiles with and without the 4 x 4 x 4 edge)
                                  (1000, 13), (1000, 14), (333, 14), (29, 13), (3100, 14), (53, 15)])      # (nodes of 12 and 13 cameras: three matrices in LDS, ba_bcr.h; 3100: levels wider than the chip)
def test_cyclic_reduction_vs_sequential_band_solver(be, nc, L):
    """The multi-CU block-cyclic-reduction solve against the single-workgroup band Cholesky and
    the dense LU on the same device-resident system (odd sizes: padded last super-block,
    non-power-of-two level counts), with and without masked parameters."""
    npt = (50 if nc < 500 else 12) * nc
    s = banded(nc, npt, track_len=L)
    flags = default_flags(nc, npt)
    load_problem(be, s['K'], s['R0'], s['t0'], s['X0'], s['obs_cam'], s['obs_pt'], s['obs_z'], *flags, O.Sensor.gaussian(1.))
    assert be.half_bandwidth == L - 1
    be.linearize(0)
    be.schur(0, 10., 1e-5)
    n = (nc - 1) * 6
    for mask in (None, (np.arange(n) % 7 != 3).astype(np.uint8)):
        sol = {}
        for solver in ('bcr', 'bcr1', 'band'):            # hb <= 11: a node over three CUs (bcr) / on one (bcr1); 12, 13: over three CUs (bcr) / ba_bcr_wide.h (bcr1); .. 23: ba_bcr_wide.h
            be.set_option('solver', solver)
            be.solve_reduced(mask)
            if L > 22 and solver == 'band':                # no single-workgroup band solver this wide: the dense Cholesky stands in
                assert be.last_solve_kind == 'dense_cholesky'
            else:
                assert be.last_solve_path == 'band'
                few = (nc - 1 + L - 2) // (L - 1) < 4        # (the wide solver wants four nodes to reduce over)
                assert be.last_solve_kind == ('band' if solver == 'band' or (few and L > 14) or (few and L > 12 and solver == 'bcr1') else
                                              'bcr' if L <= 12 or (L <= 14 and solver == 'bcr') else 'bcr_wide'), (solver, be.last_solve_kind)
            sol[solver] = be.get_solution().reshape(-1)
        xd = _device_lu(be, mask)
        close(sol['bcr'], xd, 1e-9)
        close(sol['band'], xd, 1e-9)
        close(sol['bcr'], sol['band'], 1e-10)
        close(sol['bcr1'], sol['band'], 1e-10)
        be.set_option('solver', 'bcr')                    # the back-substitution level by level instead of in one launch
        be.set_option('fused_backsolve', 0)
        be.solve_reduced(mask)
        be.set_option('fused_backsolve', 1)
        close(be.get_solution().reshape(-1), sol['bcr'], 1e-13)
        if mask is not None:
            assert np.all(sol['bcr'][mask == 0] == 0)
        # one step of iterative refinement through the kept factors (csrc/ba_bcr_refine.h; the damping of this test does not switch it
        # on by itself): the same solution, a residual that is no larger, masked rows still exactly zero - nodes over three compute
        # units and on one, every node size, level counts that are not powers of two, levels wider than the chip
        if L <= 14:
            for solver in ('bcr', 'bcr1') if L <= 12 else ('bcr',):
                be.set_option('solver', solver)
                be.set_option('refine', '1')
                before = be.problem_info()['solves_refined']
                be.solve_reduced(mask)
                be.set_option('refine', 'auto')
                assert be.last_solve_kind == 'bcr' and be.problem_info()['solves_refined'] == before + 1
                xr = be.get_solution().reshape(-1)
                close(xr, sol[solver], 1e-10)
                assert mask is None or np.all(xr[mask == 0] == 0)
                if nc <= 1000:
                    S, rhs = be.get_reduced()
                    A, rhs = S.transpose(0, 2, 1, 3).reshape(n, n), rhs.reshape(n)
                    keep = np.arange(n) if mask is None else np.nonzero(mask)[0]
                    A, rhs = A[np.ix_(keep, keep)], rhs[keep]
                    # (in the measure with a margin - backward_error above: at the level of round-off a step of refinement need not lower
                    #  ||S x - b||, rounding x + dx costs what it gains: profiles/r06_golden_solve_spread.txt - and loosely against the
                    #  unrefined residual)
                    res = lambda v: np.linalg.norm(A @ v[keep] - rhs) / np.linalg.norm(rhs)
                    assert backward_error(A, rhs, xr[keep]) <= 2 * EPS, (solver, backward_error(A, rhs, xr[keep]) / EPS)
                    assert res(xr) <= 2. * res(sol[solver]) + 4 * EPS, (solver, res(xr), res(sol[solver]))


@pytest.mark.parametrize('nc,L,sensor', [(40, 10, O.Sensor.cauchy(.05)), (30, 4, O.Sensor.gaussian(1.)),
                                         (26, 13, O.Sensor.huber(.06)), (60, 7, O.Sensor.gaussian(1.)),
                                         (40, 11, O.Sensor.gaussian(1.)), (50, 16, O.Sensor.cauchy(.05)),
                                         (44, 17, O.Sensor.gaussian(1.)), (48, 20, O.Sensor.gaussian(1.)),
                                         (100, 22, O.Sensor.huber(.06)), (110, 24, O.Sensor.gaussian(1.)),
                                         (150, 25, O.Sensor.gaussian(1.)), (160, 32, O.Sensor.cauchy(.05)),
                                         (200, 40, O.Sensor.huber(.06))])    # (sparse enough not to be 'dense visibility')
def test_group_reduction_kernel_equals_pair_kernel(be, nc, L, sensor):
    """k_schur_groups (register accumulation over runs of points with identical camera lists; 1 and 2 pair rounds),
    k_schur_groups_mfma2 (the same groups on the fp64 matrix cores, track length <= 10) and k_schur_groups_mfma3 (any
    track length up to 40: one to ten launches over the tile columns of the window, with and without the LDS
    accumulation window) against k_schur_pairs and the oracle, also with groups broken up by dropped observations
    (ragged track lengths) and frozen cameras in the middle of the sequence."""
    s = banded(nc, 40 * nc, track_len=L, outlier_frac=.05)
    keep = np.ones(len(s['obs_cam']), bool)
    keep[::97] = False                                   # ragged tracks: many size-1 groups
    cam, pt, z = s['obs_cam'][keep], s['obs_pt'][keep], s['obs_z'][keep]
    cam_opt_pos = np.arange(nc, dtype=np.int32) - 1
    cam_opt_pos[nc // 2] = -1                            # a frozen camera inside the window
    cam_opt_pos[nc // 2 + 1:] -= 1
    pt_opt = np.ones(40 * nc, np.uint8)
    a = (s['K'], s['R0'], s['t0'], s['X0'], cam, pt, z)
    out = {}
    for kern in ('pairs', 'groups', 'mfma2', 'mfma', 'mfma-nowindow'):    # 'groups' / 'mfma2' fall back to pairs beyond L = 15 / 10
        be.set_option('lds_window', 0 if kern == 'mfma-nowindow' else 1)
        be.set_option('schur', kern.split('-')[0])
        load_problem(be, *a, cam_opt_pos, pt_opt, sensor)
        info = be.problem_info()
        if kern.startswith('mfma') and kern != 'mfma2':
            assert info['schur_kernel'] == 4 and info['schur_mfma'] == 1
            assert kern != 'mfma-nowindow' or info['lds_window_rows'] == 0
        be.linearize(0)
        be.schur(0, 3., 1e-5)
        out[kern] = be.get_reduced()
    for kern in ('groups', 'mfma2', 'mfma', 'mfma-nowindow'):
        close(out[kern][0], out['pairs'][0], 1e-12)
        close(out[kern][1], out['pairs'][1], 1e-12)
    mu, su, parts = O.compute_update(sensor, *a, cam_opt_pos, pt_opt, damping=3., return_parts=True)
    close(out['mfma'][0], parts['S'], TIGHT)
    close(out['mfma'][1], parts['b'], TIGHT)
    # the whole trial through the same kernels (camera blocks and right-hand side inside the reduction)
    be.set_option('schur', 'auto')
    be.set_option('lds_window', 1)
    load_problem(be, *a, cam_opt_pos, pt_opt, sensor)
    assert be.problem_info()['schur_kernel'] in (3, 4)       # the fixed-shape kernel for unbroken runs up to L = 10, else the general one
    info, cost = be.lm_trial(3., 1e-5, None)
    assert info == 0
    St, bt = be.get_reduced()
    close(St, parts['S'], TIGHT)
    close(bt, parts['b'], TIGHT)
    R2, t2, X2 = O.apply_update(s['R0'], s['t0'], s['X0'], mu, su, cam_opt_pos, pt_opt)
    close(cost, O.cost(sensor, s['K'], R2, t2, X2, cam, pt, z, cam_opt_pos, pt_opt), 1e-8)


def test_ragged_track_lengths_through_the_matrix_core_reduction(be):
    """Track lengths 2 ... 24 in ONE scene (a third of the observations dropped at random), tracks in random order: runs
    of every length share the launches of k_schur_groups_mfma3 (windows of fewer tiles than the launch covers, batches of
    different sizes sharing the staging buffers)."""
    nc, nt, L = 70, 6000, 24
    s = banded(nc, nt, track_len=L)
    rs = np.random.RandomState(5)
    keep = rs.rand(len(s['obs_cam'])) > .33
    keep[::L] = True                                     # every track keeps at least two observations
    keep[1::L] = True
    o = rs.permutation(int(keep.sum()))
    cam, pt, z = s['obs_cam'][keep][o], s['obs_pt'][keep][o], s['obs_z'][keep][o]
    flags = default_flags(nc, nt)
    a = (s['K'], s['R0'], s['t0'], s['X0'], cam, pt, z)
    sensor = O.Sensor.cauchy(.05)
    out = {}
    for kern in ('pairs', 'mfma'):
        be.set_option('schur', kern)
        load_problem(be, *a, *flags, sensor)
        be.linearize(0)
        be.schur(0, 1., 1e-5)
        out[kern] = be.get_reduced()
    assert be.problem_info()['schur_kernel'] in (3, 4)       # the fixed-shape kernel for unbroken runs up to L = 10, else the general one
    close(out['mfma'][0], out['pairs'][0], 1e-12)
    close(out['mfma'][1], out['pairs'][1], 1e-12)
    mu, su, parts = O.compute_update(sensor, *a, *flags, damping=1., return_parts=True)
    close(out['mfma'][0], parts['S'], TIGHT)
    close(out['mfma'][1], parts['b'], TIGHT)
    be.set_option('schur', 'auto')
    load_problem(be, *a, *flags, sensor)
    info, cost = be.lm_trial(1., 1e-5, None)
    assert info == 0
    Rg, tg, Xg = be.get_params(1)
    R2, t2, X2 = O.apply_update(s['R0'], s['t0'], s['X0'], mu, su, *flags)
    close(Xg, X2, 1e-8)
    close(tg, t2, 1e-8)


@pytest.mark.parametrize('rcond', [None, 1e-13, 1e-5])
def test_factorised_point_inverses_with_ill_conditioned_blocks(be, rcond):
    """The producer / consumer MFMA reduction works from HPPinv = L D L^T (ba_math.h sym3_ldl).  Points far
    from the cameras make the 3 x 3 blocks ill-conditioned along the viewing direction (eigenvalue ratios up
    to ~1e9); without damping, with the plain inverse (rcond None), a tiny and the default pinv cut-off the
    reduction must still agree with the pair kernel, which multiplies by HPPinv itself - the tolerance of
    the factorisation's "cut direction" test follows rcond."""
    nc, nt = 40, 1600
    s = banded(nc, nt, track_len=10)
    X0 = s['X0'].copy()
    X0[::3, 2] *= 400.                                     # every third point far away
    cam_opt_pos, pt_opt = default_flags(nc, nt)
    a = (s['K'], s['R0'], s['t0'], X0, s['obs_cam'], s['obs_pt'], s['obs_z'])
    load_problem(be, *a, cam_opt_pos, pt_opt, O.Sensor.gaussian(1.))
    out = {}
    for kern in ('pairs', 'mfma2', 'mfma'):
        be.set_option('schur', kern)
        be.linearize(0)
        be.schur(0, 1e-9, rcond)
        out[kern] = be.get_reduced()
    blk = be.get_blocks()
    w = np.linalg.eigvalsh(blk['HPP'])
    assert (w[:, -1] / np.maximum(w[:, 0], 1e-300)).max() > 1e6       # the scene really is ill-conditioned
    for kern in ('mfma2', 'mfma'):
        close(out[kern][0], out['pairs'][0], 1e-10)
        close(out[kern][1], out['pairs'][1], 1e-10)


def test_wide_band_takes_dense_path(be):
    g = load_golden('scene_oleg_40x100')
    load_problem(be, *scene(g), g['l10_cam_opt_pos'], g['l10_pt_opt'], sensor_of(g))
    assert be.half_bandwidth == 38
    dC, dP = hip_update(be, 10.)
    assert be.last_solve_path == 'dense_cholesky'
    close(-dC, g['update_l10_motion'], 1e-7)


def test_dense_visibility_reduction_equals_pair_kernels(be):
    """ba_set_dense_visibility: the reduction as one symmetric matrix product over all points
    (k_dense_stage / k_dense_syrk / k_dense_apply / k_dense_rhs, factorised point inverses) against the
    per-pair kernels on the same linearisation, and against the oracle."""
    g = load_golden('scene_oleg_40x100')
    a = scene(g)
    load_problem(be, *a, g['l10_cam_opt_pos'], g['l10_pt_opt'], sensor_of(g))
    assert be._dense                                         # the host picked the dense form for this scene
    be.linearize(0)
    be.schur(0, 10., 1e-5)
    Sd, bd = be.get_reduced()
    be._check(be._lib.ba_set_dense_visibility(be._h, 0))
    be.schur(0, 10., 1e-5)
    Sp, bp = be.get_reduced()
    be._check(be._lib.ba_set_dense_visibility(be._h, 1))
    close(Sd, Sp, 1e-11)
    close(bd, bp, 1e-11)
    mu, su, parts = O.compute_update(sensor_of(g), *a, g['l10_cam_opt_pos'], g['l10_pt_opt'], damping=10., return_parts=True)
    close(Sd, parts['S'], TIGHT)
    close(bd, parts['b'], TIGHT)


def _dense_reference(be, mask=None):
    S, b = be.get_reduced()
    n = be.nco * 6
    A, r = S.transpose(0, 2, 1, 3).reshape(n, n), b.reshape(n)
    keep = np.arange(n) if mask is None else np.nonzero(mask)[0]
    x = np.zeros(n)
    x[keep] = np.linalg.solve(A[np.ix_(keep, keep)], r[keep])
    return x


@pytest.mark.parametrize('nc,L', [(160, 30), (23, 23), (58, 40), (9, 9)])
def test_dense_cholesky_on_the_device_matches_lapack(be, nc, L):
    """Band too wide for the cyclic reduction: ba_solve_reduced factors the whole matrix on the device
    (ba_dense.h: block columns of 48; 6 (nc - 1) is not a multiple of 48 in any of these, one ends in a
    block of 6, one has fewer panel rows than a workgroup takes) - against LAPACK on the same system,
    with and without deleted camera parameters."""
    s = banded(nc, 20 * nc, track_len=L)
    flags = default_flags(nc, 20 * nc)
    load_problem(be, s['K'], s['R0'], s['t0'], s['X0'], s['obs_cam'], s['obs_pt'], s['obs_z'], *flags, O.Sensor.gaussian(1.))
    be.set_option('solver', 'dense')                  # (narrower bands and more nodes: the cyclic reductions)
    be.linearize(0)
    be.schur(0, 5., 1e-5)
    rng = np.random.RandomState(nc)
    mask = (rng.rand(be.nco * 6) > .1).astype(np.uint8)
    for m in (None, mask):
        be.solve_reduced(m)
        assert be.last_solve_path == 'dense_cholesky'
        x = be.get_solution().reshape(-1)
        ref = _dense_reference(be, m)
        assert np.abs(x - ref).max() <= 1e-9 * max(1., np.abs(ref).max())


@pytest.mark.parametrize('nc,L', [(160, 30), (400, 26), (333, 41), (700, 33), (217, 25), (600, 81), (900, 175), (1000, 200)])
def test_big_node_cyclic_reduction_matches_lapack(be, nc, L):
    """Half-bandwidths beyond 23 (tracks of 25 and more cameras) with at least four nodes of hb cameras: ba_bcr_big.h - every
    level a batched partial dense Cholesky of one 3B x 3B matrix per eliminated node (B = 150 .. 480 and, nodes wider than one
    round of the back-substitution's 1024 threads, B = 1044 and 1200; node counts that
    are and are not powers of two, a padded last node, hb odd and even).  Against LAPACK on the same system and against the
    dense blocked Cholesky, with and without deleted camera parameters; the status word of a system that is not positive
    definite."""
    s = banded(nc, 12 * nc, track_len=L)
    flags = default_flags(nc, 12 * nc)
    load_problem(be, s['K'], s['R0'], s['t0'], s['X0'], s['obs_cam'], s['obs_pt'], s['obs_z'], *flags, O.Sensor.gaussian(1.))
    assert be.half_bandwidth == L - 1
    be.linearize(0)
    be.schur(0, 5., 1e-5)
    rng = np.random.RandomState(nc)
    mask = (rng.rand(be.nco * 6) > .1).astype(np.uint8)
    for m in (None, mask):
        be.set_option('solver', 'auto')
        be.solve_reduced(m)
        assert be.last_solve_kind == 'bcr_big' and be.last_solve_path == 'band'
        x = be.get_solution().reshape(-1)
        ref = _dense_reference(be, m)
        assert np.abs(x - ref).max() <= 1e-9 * max(1., np.abs(ref).max())
        be.set_option('solver', 'dense')
        be.solve_reduced(m)
        assert be.last_solve_kind == 'dense_cholesky'
        close(be.get_solution().reshape(-1), x, 1e-10)
        if m is not None:
            assert np.all(x[m == 0] == 0)
    be.set_option('solver', 'bcr')
    if nc <= 217:
        be.schur(0, -3., 1e-5)                        # negative damping: not positive definite - reported, solved again by LU down the band
        be.solve_reduced(None)
        assert be.last_solve_kind == 'band_lu' and be.last_solve_path == 'lu'
        close(be.get_solution().reshape(-1), _dense_reference(be), 1e-7)


def test_dense_cholesky_equals_cyclic_reduction(be):
    """The same banded system through the cyclic reduction and (option solver=dense) the dense factorisation."""
    s = banded(120, 3000, track_len=8)
    flags = default_flags(120, 3000)
    load_problem(be, s['K'], s['R0'], s['t0'], s['X0'], s['obs_cam'], s['obs_pt'], s['obs_z'], *flags, O.Sensor.gaussian(1.))
    be.linearize(0)
    be.schur(0, 2., 1e-5)
    be.solve_reduced(None)
    assert be.last_solve_kind == 'bcr'
    x0 = be.get_solution()
    be.set_option('solver', 'dense')
    be.solve_reduced(None)
    assert be.last_solve_kind == 'dense_cholesky' and be.last_solve_path == 'dense_cholesky'
    close(be.get_solution(), x0, 1e-10)


def test_lu_down_the_band_by_option(be):
    """Option solver=lu keeps the device Cholesky out: LU with partial pivoting down the band (ba_band_lu.h, the path
    systems that are not positive definite take beyond nodes of 11 cameras) must agree with LAPACK."""
    nc, L = 160, 30
    s = banded(nc, 20 * nc, track_len=L)
    flags = default_flags(nc, 20 * nc)
    load_problem(be, s['K'], s['R0'], s['t0'], s['X0'], s['obs_cam'], s['obs_pt'], s['obs_z'], *flags, O.Sensor.gaussian(1.))
    be.set_option('solver', 'lu')
    dC, dP = hip_update(be, 5.)
    assert be.last_solve_kind == 'band_lu' and be.last_solve_path == 'lu'
    close(dC.reshape(-1), _dense_reference(be), 1e-9)
    mask = (np.random.RandomState(3).rand(be.nco * 6) > .1).astype(np.uint8)
    be.solve_reduced(mask)
    close(be.get_solution().reshape(-1), _dense_reference(be, mask), 1e-9)


def test_dense_cholesky_reports_non_positive_pivot(be):
    g = load_golden('scene_oleg_40x100')
    load_problem(be, *scene(g), g['l10_cam_opt_pos'], g['l10_pt_opt'], sensor_of(g))
    be.linearize(0)
    be.schur(0, -3., 1e-5)
    be.solve_reduced(None)
    assert be.last_solve_kind == 'band_lu' and be.last_solve_path == 'lu'                  # LU took over
    x = be.get_solution().reshape(-1)
    S, b = be.get_reduced()
    n = be.nco * 6
    close(S.transpose(0, 2, 1, 3).reshape(n, n) @ x, b.reshape(-1), 1e-8)


def test_band_solver_reports_non_positive_pivot(be):
    """An indefinite reduced system must not be 'solved' by Cholesky: the single-workgroup band solver (forced: systems this
    small are one node of the cyclic reduction by default) reports info > 0 and the dense LU takes over; the default path
    reports it too and solves again with an LU node on the device."""
    g = load_golden('scene_4x10_cauchy')
    load_problem(be, *scene(g), g['l2_cam_opt_pos'], g['l2_pt_opt'], sensor_of(g))
    be.linearize(0)
    be.schur(0, -3., 1e-5)                    # (1 + lambda) < 0 flips the sign of every diagonal
    be.solve_reduced(None)
    assert be.last_solve_kind == 'bcr_lu' and be.last_solve_path == 'lu'
    x_lu = be.get_solution().reshape(-1)
    be.set_option('solver', 'band')
    be.solve_reduced(None)
    assert be.last_solve_kind == 'band_lu' and be.last_solve_path == 'lu'
    close(x_lu, be.get_solution().reshape(-1), 1e-9)
    S, b = be.get_reduced()
    x = be.get_solution().reshape(-1)
    A = S.transpose(0, 2, 1, 3).reshape(18, 18)
    close(A @ x, b.reshape(-1), 1e-9)


def test_a_system_that_is_not_positive_definite_without_the_device_lu(be):
    """Option device_lu = 0 (round 2's behaviour): a failed Cholesky is answered like the reference's LinAlgError
    (NormalEquationsIllconditioned -> the LM loop raises the damping, bundle_adjuster.py:134-140).  With the device LU (the
    default) the same system is SOLVED, as the reference's LU solves it."""
    from pysfm_amd.backend import ReducedSystemSingular
    nc = 380
    s = banded(nc, 3 * nc, track_len=6)
    flags = default_flags(nc, 3 * nc)
    load_problem(be, s['K'], s['R0'], s['t0'], s['X0'], s['obs_cam'], s['obs_pt'], s['obs_z'], *flags, O.Sensor.gaussian(1.))
    be.linearize(0)
    be.schur(0, -3., 1e-5)                    # (1 + lambda) < 0: indefinite
    be.set_option('device_lu', 0)
    with pytest.raises(ReducedSystemSingular):
        be.solve_reduced(None)
    info, _ = be.lm_trial(-3., 1e-5, None)
    assert info > 0
    be.set_option('device_lu', 1)
    be.schur(0, -3., 1e-5)
    be.solve_reduced(None)
    assert be.last_solve_kind == 'bcr_lu' and be.last_solve_path == 'lu'
    info, _ = be.lm_trial(-3., 1e-5, None)    # (the one-batch trial only reports: the caller solves again, stepwise)
    assert info > 0
    be.schur(0, 10., 1e-5)                    # and the same scene, damped properly, solves on the device
    be.solve_reduced(None)
    assert be.last_solve_path == 'band'


def test_lm_trial_entry_equals_stepwise_calls(be):
    """ba_lm_trial (one batch, one synchronisation) against the same step made of the
    individual entry points, including the accept / swap bookkeeping."""
    s = banded(50, 2500, track_len=8, outlier_frac=.05)
    flags = default_flags(50, 2500)
    sensor = O.Sensor.cauchy(.05)
    a = (s['K'], s['R0'], s['t0'], s['X0'], s['obs_cam'], s['obs_pt'], s['obs_z'])
    load_problem(be, *a, *flags, sensor)
    mask = np.ones(49 * 6, np.uint8)
    mask[[3, 40]] = 0
    for lam, m in ((10., None), (.1, mask)):
        be.set_params(0, s['R0'], s['t0'], s['X0'])
        info, c_fused = be.lm_trial(lam, 1e-5, m)
        assert info == 0
        Rf, tf, Xf = be.get_params(1)
        dC_f = be.get_solution()
        be.set_params(0, s['R0'], s['t0'], s['X0'])
        dC, dP = hip_update(be, lam, keep=None if m is None else np.nonzero(m)[0])
        be.apply_update(0, 1)
        c_step = be.cost(1)
        Rs, ts, Xs = be.get_params(1)
        close(dC_f, dC, 1e-12)
        close(Rf, Rs, 1e-13)
        close(tf, ts, 1e-13)
        close(Xf, Xs, 1e-13)
        assert abs(c_fused - c_step) <= 1e-12 * c_step
        mu, su = O.compute_update(sensor, *a, *flags, damping=lam, cam_param_mask=None if m is None else m.astype(bool))
        R2, t2, X2 = O.apply_update(s['R0'], s['t0'], s['X0'], mu, su, *flags)
        close(c_fused, O.cost(sensor, s['K'], R2, t2, X2, *a[4:], *flags), 1e-8)
    be.swap_params()                                        # accept: the trial set becomes current
    close(be.get_params(0)[2], Xs, 0.)


@pytest.mark.parametrize('L,rcond', [(10, 1e-5), (7, 1e-5), (3, 1e-5), (10, None)])
def test_trial_lineariser_inverts_the_point_blocks_itself(be, L, rcond):
    """ba_lm_trial with a matrix-core reduction (round 6, option fuse_invert): k_linearize_groups_trial damps, inverts and factorises
    the point blocks of a workgroup's groups once they are summed and clears [S | b] - no k_point_invert_schur_init launch.  The
    inverses are the SAME BITS as that launch's (the same device function on the same sums), the trial is the one the two launches
    make, a trial after a rejected one (new damping, linearisation kept) still takes the launch, and plain-inverse mode (rcond None,
    bundle_adjuster.py:252-256 without the pseudo-inverse) counts its singular blocks the same way."""
    nc, nt = 60, 3000
    s = banded(nc, nt, track_len=L, outlier_frac=.03)
    cam_opt_pos = np.arange(nc, dtype=np.int32) - 1
    pt_opt = np.ones(nt, np.uint8)
    pt_opt[::11] = 0
    sensor = O.Sensor.cauchy(.05)
    out = {}
    for fuse in (1, 0):
        be.set_option('fuse_invert', fuse)
        load_problem(be, s['K'], s['R0'], s['t0'], s['X0'], s['obs_cam'], s['obs_pt'], s['obs_z'], cam_opt_pos, pt_opt, sensor)
        assert be.problem_info()['schur_mfma'] == 1            # (the fused path belongs to the matrix-core reductions)
        res = []
        for lam in (10., 1., 1.):                              # (no swap in between: the second and third trial reuse the linearisation)
            info, cost = be.lm_trial(lam, rcond, None)
            assert info == 0
            res.append((cost, be.get_point_inverses().copy(), be.get_solution().copy(), be.get_params(1)[2].copy()))
        out[fuse] = res
    be.set_option('fuse_invert', 1)
    for (c1, i1, d1, x1), (c0, i0, d0, x0) in zip(out[1], out[0]):
        assert np.array_equal(i1, i0)
        close(d1, d0, 1e-11)
        close(x1, x0, 1e-12)
        assert abs(c1 - c0) <= 1e-12 * c0
    HPP = O.normal_blocks(sensor, s['K'], s['R0'], s['t0'], s['X0'], s['obs_cam'], s['obs_pt'], s['obs_z'], nc, nt)[1]
    close(out[1][1][1], O.invert_point_blocks(O.damp_blocks(HPP, 1.), rcond), 1e-9)


@pytest.mark.parametrize('L', [2, 3, 7, 10, 13, 15])
def test_group_packed_point_kernels_equal_lanes_per_point_kernels(be, L):
    """k_linearize_groups / k_backsub_groups (lane = (point slot, observation), trial cost fused into the
    back-substitution) against k_linearize / k_backsub / k_cost (option point_kernels=v1) for track lengths that
    do and do not divide 64, with a second frozen camera and points that are not optimised; both against
    the oracle."""
    nc, nt = 40, 1200
    s = banded(nc, nt, track_len=L, outlier_frac=.03)
    cam_opt_pos = np.arange(nc, dtype=np.int32) - 1
    cam_opt_pos[17] = -1                                   # a frozen camera in the middle of the sequence
    cam_opt_pos[18:] -= 1
    pt_opt = np.ones(nt, np.uint8)
    pt_opt[::7] = 0
    sensor = O.Sensor.cauchy(.05)
    a = (s['K'], s['R0'], s['t0'], s['X0'], s['obs_cam'], s['obs_pt'], s['obs_z'])
    out = {}
    for tag in ('groups', 'v1'):
        if tag == 'v1':
            be.set_option('point_kernels', 'v1')
        load_problem(be, *a, cam_opt_pos, pt_opt, sensor)
        be.linearize(0)
        blk = be.get_blocks()
        info, cost = be.lm_trial(3., 1e-5, None)
        assert info == 0
        out[tag] = (blk['HPP'], blk['bP'], be.get_params(1), cost, be.cost(1))
    g, v = out['groups'], out['v1']
    close(g[0], v[0], 1e-13)
    close(g[1], v[1], 1e-12)
    for x, y in zip(g[2], v[2]):
        close(x, y, 1e-11)
    assert abs(g[3] - v[3]) <= 1e-10 * v[3] and abs(g[3] - g[4]) <= 1e-12 * g[4]     # fused cost = k_cost of the trial set
    mu, su, parts = O.compute_update(sensor, *a, cam_opt_pos, pt_opt, damping=3., return_parts=True)
    close(g[0], parts['HPP'], TIGHT)
    close(g[1], parts['bP'], TIGHT)
    R2, t2, X2 = O.apply_update(s['R0'], s['t0'], s['X0'], mu, su, cam_opt_pos, pt_opt)
    close(g[3], O.cost(sensor, s['K'], R2, t2, X2, *a[4:], cam_opt_pos, pt_opt), 1e-8)


def _two_rank_worker(rank, world, port, out_dir):
    import sys
    for p in (ROOT, os.path.join(ROOT, 'tests')):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0')
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from pysfm_amd import Bundle, BundleAdjuster, sensor_model
    from pysfm_amd import synthetic_data as sd
    from pysfm_amd.distributed import ShardComm, shard_tracks
    s = sd.generate_banded_scene(60, 3000, track_len=8, outlier_frac=.02)
    b = Bundle.FromObservations(s['K'], s['R0'], s['t0'], s['X0'], s['obs_cam'], s['obs_pt'], s['obs_z'],
                                sensor_model=sensor_model.CauchyModel(.05))
    comm = ShardComm()
    ba = BundleAdjuster(device=0, comm=comm, verbose=False)          # both ranks on GPU 0
    ba.set_bundle(b, track_ids=shard_tracks(b, rank, world))
    ba.optimize(max_steps=6)
    X = comm.gather_points(ba)
    R, t, _ = ba.backend.get_params(0)
    np.savez(os.path.join(out_dir, 'rank%d.npz' % rank), costs=np.array(ba.costs), X=X, R=R, t=t, trials=ba.lm_trials,
             nbytes=comm.bytes_reduced)
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_on_one_gpu_walk_the_single_gpu_trajectory(tmp_path):
    """Two processes, each with its half of the tracks on the real HIP backend (same GPU, gloo group
    staging the collectives through the host): ba_lm_trial_begin -> all-reduce of [S | b] ->
    ba_lm_trial_end, band layout agreed over the ranks, trial costs summed.  Must reproduce the
    unsharded run."""
    import socket
    import torch.multiprocessing as mp
    from pysfm_amd import Bundle, BundleAdjuster, sensor_model
    from pysfm_amd import synthetic_data as sd
    sk = socket.socket(); sk.bind(('127.0.0.1', 0)); port = sk.getsockname()[1]; sk.close()
    mp.spawn(_two_rank_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    s = sd.generate_banded_scene(60, 3000, track_len=8, outlier_frac=.02)
    b = Bundle.FromObservations(s['K'], s['R0'], s['t0'], s['X0'], s['obs_cam'], s['obs_pt'], s['obs_z'],
                                sensor_model=sensor_model.CauchyModel(.05))
    ba = BundleAdjuster(verbose=False)
    ba.set_bundle(b)
    ba.optimize(max_steps=6)
    R1, t1, X1 = ba.backend.get_params(0)
    r0, r1 = (np.load(os.path.join(str(tmp_path), 'rank%d.npz' % r)) for r in (0, 1))
    for r in (r0, r1):
        assert int(r['trials']) == ba.lm_trials and int(r['nbytes']) > 0
        close(r['costs'], np.array(ba.costs), 1e-9)
        close(r['t'], t1, 1e-8)
        close(r['X'], X1, 1e-8)


@pytest.mark.parametrize('collectives', ['library', 'torch'])
def test_sharded_trial_path_matches_single_gpu_path(collectives):
    """The multi-GPU trial with a one-rank RCCL group must walk the same LM trajectory as the single-GPU
    ba_lm_trial.  'library': the collectives are issued by the library itself on its own stream
    (ba_comm_init; ba_lm_trial is then the sharded trial); 'torch' (ShardComm(collectives='torch')): ba_lm_trial_begin ->
    torch.distributed all-reduce of [S | b] -> ba_lm_trial_end -> all-reduce of the trial record."""
    import torch
    import torch.distributed as dist
    from pysfm_amd import Bundle, BundleAdjuster, sensor_model
    from pysfm_amd.distributed import ShardComm
    s = banded(60, 3000, track_len=8, outlier_frac=.02)
    model = sensor_model.CauchyModel(.05)

    def run(comm):
        b = Bundle.FromObservations(s['K'], s['R0'], s['t0'], s['X0'], s['obs_cam'], s['obs_pt'], s['obs_z'], sensor_model=model)
        ba = BundleAdjuster(comm=comm, verbose=False)
        ba.set_bundle(b)
        ba.optimize(max_steps=6)
        return ba

    single = run(None)
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29533')
    created = not dist.is_initialized()
    if created:
        torch.cuda.set_device(0)
        dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    try:
        comm = ShardComm(collectives=collectives)
        sharded = run(comm)
        if collectives == 'torch':
            assert comm.bytes_reduced > 0 and not sharded.backend.direct_comm      # the collective really ran
        else:
            assert sharded.backend.direct_comm and comm.direct is sharded.backend
            got = sharded.backend.comm_allreduce_sum([1.5, -2.])
            assert got[0] == 1.5 and got[1] == -2.
    finally:
        if created:
            dist.destroy_process_group()
    assert len(sharded.costs) == len(single.costs)
    close(np.array(sharded.costs), np.array(single.costs), 1e-9)
    assert sharded.lm_trials == single.lm_trials
    Rs, ts, Xs = sharded.backend.get_params(0)
    R1, t1, X1 = single.backend.get_params(0)
    close(Xs, X1, 1e-9)
    close(ts, t1, 1e-9)


def test_timing_counters(be):
    g = load_golden('scene_4x10_cauchy')
    load_problem(be, *scene(g), g['l2_cam_opt_pos'], g['l2_pt_opt'], sensor_of(g))
    be.enable_timing(True)
    be.timings(reset=True)
    for _ in range(3):
        hip_update(be, 2.)
        be.cost(0)
    tm = be.timings(reset=True)
    be.enable_timing(False)
    assert tm['schur_pairs']['launches'] == 3 and tm['linearize']['launches'] == 3 and tm['cost']['launches'] == 3
    assert tm['schur_pairs']['ms'] > 0


# ------------------------------------------------------------------ rows either side of the path
@pytest.mark.parametrize('name', ['scene_planar_lm', 'scene_oleg_10x50', 'scene_oleg_40x100'])
def test_triangulation_vs_reference(be, name):
    """k_triangulate against the reference's Bundle.triangulate_all (the fixture's X)."""
    g = load_golden(name)
    nc, nt = len(g['R']), len(g['X'])
    load_problem(be, g['K'], g['R'], g['t'], np.zeros((nt, 3)), g['obs_cam'], g['obs_pt'], g['obs_z'],
                 *default_flags(nc, nt), sensor_of(g))
    X = be.triangulate(0)
    close(X, g['X'], 1e-7)
    close(be.get_params(0)[2], X, 0.)


def test_triangulation_degenerate_tracks(be):
    """one observation (rank 2) and no observation: minimum-norm answer like numpy.linalg.lstsq"""
    s = banded(12, 30, track_len=6)
    keep = np.ones(len(s['obs_cam']), bool)
    keep[np.nonzero(s['obs_pt'] == 4)[0][1:]] = False
    keep[s['obs_pt'] == 9] = False
    cam, pt, z = s['obs_cam'][keep], s['obs_pt'][keep], s['obs_z'][keep]
    load_problem(be, s['K'], s['R'], s['t'], np.zeros((30, 3)), cam, pt, z, *default_flags(12, 30), O.Sensor.gaussian(1.))
    X = be.triangulate(0)
    X0 = O.triangulate_all(s['K'], s['R'], s['t'], cam, pt, z, 30)
    close(X, X0, 1e-7)
    assert np.all(X[9] == 0)
    ok = (pt != 4) & (pt != 9)               # well-posed tracks reproject to within the measurement noise
    e = O.reproj_error(s['K'], s['R'], s['t'], X, cam[ok], pt[ok], z[ok])
    assert np.sqrt(np.mean(np.sum(e * e, axis=1))) < 3 * .02


@pytest.mark.parametrize('nc,L', [(64, 4), (300, 10), (1000, 10), (257, 12), (230, 13), (120, 16), (200, 23), (150, 32), (400, 80), (30, 30)])
def test_not_positive_definite_systems_are_solved_like_the_reference_lu(be, nc, L):
    """The reference solves its reduced system by LU (numpy.linalg.solve, bundle_adjuster.py:302-305): a symmetric matrix that is
    NOT positive definite is still solved.  A NEGATIVE damping makes such a system on purpose (diag(H) scaled by 0.4: indefinite,
    far from singular): the device Cholesky must report it - whichever solver the band width selects: half-bandwidths 3 .. 11
    (cyclic reduction), 15 and 22 (three kernels per level), 31 and 79 (nodes in device memory), 29 of 29 cameras (dense) -, the
    LU must solve it (k_bcr_eliminate_lu: partial pivoting inside a node, up to 11 cameras per node; ba_band_lu.h: partial
    pivoting down the band, gesv's own pivot choices, beyond), and the solution must be LAPACK's - for the whole system and
    with parameters masked."""
    nt = (30 if L <= 12 else 12) * nc
    s = banded(nc, nt, track_len=L)
    flags = default_flags(nc, nt)
    load_problem(be, s['K'], s['R0'], s['t0'], s['X0'], s['obs_cam'], s['obs_pt'], s['obs_z'], *flags, O.Sensor.gaussian(1.))
    be.linearize(0)
    be.schur(0, -.6, 1e-5)
    be.synchronize()
    St, bt = be.reduced_tensors()
    hb = be.half_bandwidth
    band = St.cpu().numpy().reshape(nc - 1, hb + 1, 6, 6)
    b = bt.cpu().numpy().reshape(-1)
    n = 6 * (nc - 1)
    A = np.zeros((n, n))
    for d in range(hb + 1):
        for i in range(nc - 1 - d):
            A[6 * i:6 * i + 6, 6 * (i + d):6 * (i + d) + 6] = band[i, d]
            if d:
                A[6 * (i + d):6 * (i + d) + 6, 6 * i:6 * i + 6] = band[i, d].T
    if n <= 2000:
        w = np.linalg.eigvalsh(A)
        assert w[0] < 0 < w[-1] and np.min(np.abs(w)) > 1e-9 * np.max(np.abs(w))      # indefinite, not singular
    else:
        with pytest.raises(np.linalg.LinAlgError):
            np.linalg.cholesky(A)
    for mask in (None, (np.arange(n) % 11 != 3).astype(np.uint8)):
        be.set_option('device_lu', 0)
        with pytest.raises(Exception):                                                # (without the device LU: reported, not solved)
            be.solve_reduced(mask)
        be.set_option('device_lu', 1)
        be.solve_reduced(mask)
        assert be.last_solve_kind == ('bcr_lu' if hb <= 11 else 'band_lu') and be.last_solve_path == 'lu'
        x = be.get_solution().reshape(-1)
        keep = np.arange(n) if mask is None else np.nonzero(mask)[0]
        ref = np.linalg.solve(A[np.ix_(keep, keep)], b[keep])
        cond = np.linalg.cond(A[np.ix_(keep, keep)]) if n <= 2000 else 1e6
        close(x[keep], ref, max(1e-9, 1e-13 * cond))
        res = A[np.ix_(keep, keep)] @ x[keep] - b[keep]
        assert np.max(np.abs(res)) <= 1e-9 * (np.max(np.abs(A)) * np.max(np.abs(x)) + np.max(np.abs(b)))
        assert mask is None or np.all(x[mask == 0] == 0)
        # the solution goes on into the back-substitution like any other
        dP = be.backsubstitute(0)
        assert np.all(np.isfinite(dP))


@pytest.mark.parametrize('nc,nt,L,reps', [(1000, 20000, 10, 200), (97, 3000, 7, 300), (523, 9000, 11, 150), (12, 400, 3, 300), (1000, 12000, 13, 150), (611, 8000, 14, 150)])
def test_one_launch_solve_repeats_itself_and_equals_the_per_level_launches(be, nc, nt, L, reps):
    """k_bcr_eliminate_fused hands data from workgroup to workgroup INSIDE one launch (words in memory, relaxed agent-scope
    accesses, no fences): a stale read would show as a solve that differs from the others.  The same reduced system solved
    `reps` times through the one launch (every third time from poisoned LDS and workspace, re-reduced) must agree with the
    solve made of one launch per level to 1e-13 of its largest entry (only the order of two fp64 atomics may differ), and no
    solve may time out.  (scripts/bcr_fused_stress.py runs thousands, also with two processes sharing the GPU.)"""
    s = banded(nc, nt, track_len=L)
    flags = default_flags(nc, nt)
    load_problem(be, s['K'], s['R0'], s['t0'], s['X0'], s['obs_cam'], s['obs_pt'], s['obs_z'], *flags, O.Sensor.gaussian(1.))
    be.linearize(0)
    be.schur(0, 10., 1e-5)
    be.set_option('fused_eliminate', 0)
    be.solve_reduced(None)
    assert be.last_solve_kind == 'bcr' and be.last_solve_path == 'band'
    ref = be.get_solution()
    be.set_option('fused_eliminate', 1)
    for r in range(reps):
        if r % 3 == 2:
            be.debug_poison()
            be.linearize(0)
            be.schur(0, 10., 1e-5)
        be.solve_reduced(None)
        assert be.last_solve_kind == 'bcr' and be.last_solve_path == 'band'
        x = be.get_solution()
        assert np.all(np.isfinite(x)) and np.max(np.abs(x - ref)) <= 1e-13 * np.max(np.abs(ref)), r
    # ... and the refinement step behind it (round 6): k_bcr_refine hands vectors from workgroup to workgroup the same way - residual
    # items, forward items up the tree, backward items down, slots marked by the solve's k_bcr_assemble.  Every third time from
    # poisoned workspace; the corrected solutions agree with each other and with the plain solve.
    be.set_option('refine', '1')
    first = None
    for r in range(reps // 2):
        if r % 3 == 2:
            be.debug_poison()
            be.linearize(0)
            be.schur(0, 10., 1e-5)
        be.solve_reduced(None)
        assert be.last_solve_kind == 'bcr'
        x = be.get_solution()
        assert np.all(np.isfinite(x)) and np.max(np.abs(x - ref)) <= 1e-12 * np.max(np.abs(ref)), r
        first = x if first is None else first
        assert np.max(np.abs(x - first)) <= 1e-13 * np.max(np.abs(ref)), r
    assert be.problem_info()['solves_refined'] >= reps // 2


@pytest.mark.parametrize('nc,nt,L,reps', [(600, 8000, 16, 120), (500, 6000, 22, 90), (2200, 20000, 15, 60)])
def test_one_launch_back_substitution_of_the_wide_cyclic_reduction_repeats_itself(be, nc, nt, L, reps):
    """k_bcrw_backsolve_fused (nodes of 14 .. 21 cameras): every node's workgroup holds its share of P, Q, G^-1 in registers and polls
    the solution entries of its neighbours one level up - the hand-over protocol of k_bcr_backsolve_fused.  The same system solved
    `reps` times (every third time from poisoned workspace, re-reduced) must agree with the back-substitution of one launch per
    level to 1e-13 of its largest entry, and no solve may time out."""
    s = banded(nc, nt, track_len=L)
    flags = default_flags(nc, nt)
    load_problem(be, s['K'], s['R0'], s['t0'], s['X0'], s['obs_cam'], s['obs_pt'], s['obs_z'], *flags, O.Sensor.gaussian(1.))
    be.linearize(0)
    be.schur(0, 10., 1e-5)
    be.set_option('fused_backsolve', 0)
    be.solve_reduced(None)
    assert be.last_solve_kind == 'bcr_wide'
    ref = be.get_solution()
    be.set_option('fused_backsolve', 1)
    for r in range(reps):
        if r % 3 == 2:
            be.debug_poison()
            be.linearize(0)
            be.schur(0, 10., 1e-5)
        be.solve_reduced(None)
        assert be.last_solve_kind == 'bcr_wide'
        x = be.get_solution()
        assert np.all(np.isfinite(x)) and np.max(np.abs(x - ref)) <= 1e-13 * np.max(np.abs(ref)), r


def test_triangulation_of_low_parallax_tracks_equals_lstsq(be):
    """triangulate.py:17 hands the 2L x 3 system to numpy.linalg.lstsq.  Tracks seen under very little parallax (cameras a
    few 1e-4 apart looking at points 5 - 50 units away: condition numbers 1e4 ... 1e6) must come out like lstsq's: the QR in
    k_triangulate keeps cond * eps, the 3 x 3 normal equations of round 2 lost cond^2 * eps (1e-4 relative at 1e6)."""
    rs = np.random.RandomState(5)
    nc, nt, L = 12, 400, 6
    K = np.array([[1.2, 0., .1], [0., 1.1, -.05], [0., 0., 1.]])
    w = rs.randn(nc, 3) * 1e-3
    R = O.so3_exp(w)
    centers = np.cumsum(rs.rand(nc, 3) * 3e-4, axis=0)                     # a baseline of ~1e-3 over the whole sequence
    t = -np.einsum('nij,nj->ni', R, centers)
    X = np.column_stack((rs.rand(nt) * 2 - 1, rs.rand(nt) * 2 - 1, 5 + 45 * rs.rand(nt)))
    first = rs.randint(0, nc - L + 1, nt)
    cam = (first[:, None] + np.arange(L)[None, :]).reshape(-1).astype(np.int32)
    pt = np.repeat(np.arange(nt, dtype=np.int32), L)
    p = np.einsum('nij,nj->ni', R[cam], X[pt]) + t[cam]
    p = p @ K.T
    z = p[:, :2] / p[:, 2:3] + rs.randn(len(cam), 2) * 1e-9
    be.set_problem(nc, nt, cam, pt, z, K, np.arange(nc, dtype=np.int32), np.ones(nt, np.uint8))
    be.set_sensor(0, np.eye(2).reshape(4))
    be.set_params(0, R, t, np.zeros((nt, 3)))
    got = be.triangulate(0)
    worst_cond = 0.
    for k in range(nt):
        A = np.empty((2 * L, 3)); b = np.empty(2 * L)
        for q in range(L):
            n = k * L + q
            for r in range(2):
                A[2 * q + r] = (K[r] - z[n, r] * K[2]) @ R[cam[n]]
                b[2 * q + r] = (z[n, r] * K[2] - K[r]) @ t[cam[n]]
        ref = np.linalg.lstsq(A, b, rcond=None)[0]
        cond = np.linalg.cond(A)
        worst_cond = max(worst_cond, cond)
        assert np.max(np.abs(got[k] - ref)) <= 1e-12 * cond * np.max(np.abs(ref)) + 1e-12, (k, cond, got[k], ref)
    assert worst_cond > 1e5                                              # (the normal equations would be off by 1e-6 and more here)


def _with_long_tracks(s, nc, nt, every, Llong, seed=3, holes=0.):
    """The banded scene with every `every`-th point seen by Llong consecutive cameras instead of 10 (a feature that survives
    for a long stretch of the video; `holes`: the fraction of those frames in which it was not detected): measurements from
    the true parameters + the scene's noise."""
    rs = np.random.RandomState(seed)
    cam, pt, z = [], [], []
    L = len(s['obs_cam']) // nt
    for k in range(nt):
        if k % every == every // 2:
            c0 = int(np.clip(s['obs_cam'][k * L] - Llong // 2, 0, nc - Llong))
            cs = np.arange(c0, c0 + Llong)
            if holes > 0:
                keep = rs.rand(Llong) >= holes
                keep[[0, -1]] = True
                cs = cs[keep]
            p = np.einsum('nij,j->ni', s['R'][cs], s['X'][k]) + s['t'][cs]
            zz = p[:, :2] / p[:, 2:3] + rs.randn(len(cs), 2) * .02
            cam.append(cs); pt.append(np.full(len(cs), k)); z.append(zz)
        else:
            sl = slice(k * L, (k + 1) * L)
            cam.append(s['obs_cam'][sl]); pt.append(s['obs_pt'][sl]); z.append(s['obs_z'][sl])
    return np.concatenate(cam).astype(np.int32), np.concatenate(pt).astype(np.int32), np.concatenate(z)


@pytest.mark.parametrize('nc,nt,every,Llong,holes,sensor,L', [
    (300, 6000, 40, 60, 0., O.Sensor.gaussian(1.), 10),
    (300, 6000, 40, 140, 0., O.Sensor.cauchy(.05), 10),
    (300, 6000, 7, 75, .4, O.Sensor.gaussian(1.), 10),            # many of them, with holes
    (150, 1200, 1, 50, .2, O.Sensor.cauchy(.05), 10),             # nothing but long tracks
    (83, 900, 1, 83, .5, O.Sensor.gaussian(1.), 10),              # ... every one of them over all cameras (a short last segment)
    (300, 3000, 20, 90, .1, O.Sensor.cauchy(.05), 30),            # beside tracks of 30 cameras: wide window groups (k_schur_wide_mfma) + pairs of segments
    (260, 4000, 30, 64, 0., O.Sensor.gaussian(1.), 18),           # beside tracks of 18 cameras: three launches of k_schur_groups_mfma3 + pairs of segments
])
def test_long_tracks_stay_on_the_matrix_cores(be, nc, nt, every, Llong, holes, sensor, L):
    """Tracks that span more than the widest window of the matrix-core reduction (40 cameras): their cameras are cut along
    segments of 32 positions - inside a segment they are members of a window group (a point list), between two segments
    k_schur_rect_mfma forms the rectangular products.  S, b, the solve and the whole trial against the oracle, and against
    the pair kernel alone."""
    s = banded(nc, nt, track_len=L)
    cam, pt, z = _with_long_tracks(s, nc, nt, every, Llong, holes=holes)
    flags = default_flags(nc, nt)
    a = (s['K'], s['R0'], s['t0'], s['X0'], cam, pt, z)
    out = {}
    for kern in ('pairs', 'auto'):
        be.set_option('schur', kern)
        load_problem(be, *a, *flags, sensor)
        be._check(be._lib.ba_set_dense_visibility(be._h, 0))       # (the last two scenes are dense enough for the SYRK reduction: not the subject here)
        info = be.problem_info()
        assert info['schur_kernel'] == (0 if kern == 'pairs' else 4) and Llong - 2 <= info['half_bandwidth'] <= Llong - 1      # (camera 0 is fixed)
        be.linearize(0)
        be.schur(0, 3., 1e-5)
        out[kern] = be.get_reduced()
    close(out['auto'][0], out['pairs'][0], 1e-12)
    close(out['auto'][1], out['pairs'][1], 1e-12)
    mu, su, parts = O.compute_update(sensor, *a, *flags, damping=3., return_parts=True)
    close(out['auto'][0], parts['S'], TIGHT)
    close(out['auto'][1], parts['b'], TIGHT)
    info, cost = be.lm_trial(3., 1e-5, None)
    assert info == 0
    St, bt = be.get_reduced()
    close(St, parts['S'], TIGHT)
    close(bt, parts['b'], TIGHT)
    R2, t2, X2 = O.apply_update(s['R0'], s['t0'], s['X0'], mu, su, *flags)
    close(be.get_params(1)[2], X2, 1e-8)
    close(cost, O.cost(sensor, s['K'], R2, t2, X2, cam, pt, z, *flags), 1e-8)


def test_distributed_solve_entry_points_fail_loudly_when_misused(be):
    """include/pysfm_ba.h ba_dist_*: a stage without an enabled plan, an exchange buffer that is too small, a plan for a rank
    count that is not a power of two - status codes and messages, never a silent fall-back."""
    import ctypes as C
    from pysfm_amd import _capi as capi
    s = banded(300, 6000)
    flags = default_flags(300, 6000)
    load_problem(be, s['K'], s['R0'], s['t0'], s['X0'], s['obs_cam'], s['obs_pt'], s['obs_z'], *flags, O.Sensor.gaussian(1.))
    lib, h = be._lib, be._h
    n = C.c_int64()
    assert lib.ba_dist_stage(h, 1, None, C.byref(n)) == capi.BA_ERR_STATE and b'ba_dist_enable' in lib.ba_last_error(h)
    assert be.dist_plan(299, 9, 3) is None and be.dist_plan(299, 9, 2) is not None
    info = be.dist_enable(0, 3)                      # no plan for three ranks: stays off, no error
    assert info['on'] == 0 and not be.dist_on
    info = be.dist_enable(1, 2)
    assert info['on'] == 1 and info['cams_per_node'] >= 9 and info['node_lo'] == info['nodes_per_rank']
    assert lib.ba_dist_bind_exchange(h, C.c_void_p(be._dist_t.data_ptr()), 8) == capi.BA_ERR_INVALID_ARG
    assert lib.ba_dist_stage(h, 1, None, C.byref(n)) == capi.BA_ERR_STATE            # (no reduction yet: ba_lm_trial_begin first)
    be.lm_trial_begin(10., 1e-5)
    assert lib.ba_dist_stage(h, 7, None, C.byref(n)) == capi.BA_ERR_INVALID_ARG
    assert be.dist_stage(1).numel() == info['exchange1_doubles']
    be.dist_enable(0, 1)                             # off again: the shared backend goes back to the plain trial
    assert not be.dist_on
    assert be.lm_trial(10., 1e-5, None)[0] == 0


def test_window_slam_vs_reference():
    from pysfm_amd import Bundle, window_slam
    g = load_golden('scene_window_slam')
    b = Bundle.FromObservations(*scene(g))
    out, hist = window_slam.run(b, 4, verbose=False)
    assert len(hist) == 7
    assert [len(h) - 1 <= n for h, n in zip(hist, g['ws_num_steps'])]
    close([h[0] for h in hist], g['ws_first_cost'], LM)
    close([h[-1] for h in hist], g['ws_last_cost'], LM)
    close(out.Rs(), g['ws_R'], LM)
    close(out.ts(), g['ws_t'], LM, 1e-6)
    close(out.reconstruction, g['ws_X'], LM)


def test_batch_driver_end_to_end(tmp_path):
    from conftest import GOLDEN
    from pysfm_amd import batch_ba, bundle_io
    import os
    ba = batch_ba.main([os.path.join(GOLDEN, 'oleg_tracks_head5.txt'), os.path.join(GOLDEN, 'oleg_poses.txt'),
                        str(tmp_path), '--cameras', '40', '--tracks', '5', '--max-steps', '6'])
    assert len(ba.camera_ids) == 40 and len(ba.track_ids) == 5
    assert all(c1 < c0 for c0, c1 in zip(ba.costs, ba.costs[1:])) and len(ba.costs) >= 2
    lines = open(tmp_path / 'adjusted_poses.txt').read().strip().split('\n')
    assert len(lines) == 40 and len(lines[0].split()) == 12
    # against the oracle on the same files
    b = bundle_io.load(os.path.join(GOLDEN, 'oleg_tracks_head5.txt'), os.path.join(GOLDEN, 'oleg_poses.txt'))
    cam, trk, z = b.observation_table()
    X0 = O.triangulate_all(b.K, b.Rs(), b.ts(), cam, trk, z, 5)
    m = cam < 40
    mask = np.ones(39 * 6, bool)
    mask[3] = False
    ref = O.lm_optimize(O.Sensor.gaussian(1.), b.K, b.Rs()[:40], b.ts()[:40], X0, cam[m], trk[m], z[m],
                        np.arange(40, dtype=np.int32) - 1, np.ones(5, bool), cam_param_mask=mask, max_steps=6)
    close(ba.costs, ref['costs'], LM)


# ------------------------------------------------------------------ a foreign Bundle class
def test_a_reference_style_bundle_of_plain_objects_walks_the_golden_trajectory():
    """BundleAdjuster takes any object with the reference Bundle's attributes (bundle.py:128-146: K, cameras[i].R / .t,
    tracks[j].measurements, reconstruction, sensor_model, clone_params, check_consistency) - here plain classes that share
    nothing with pysfm_amd.bundle, observations gathered by _select_observations_generic - and reproduces the reference's
    own LM trajectory on bundle_unittest.create_test_bundle (36 observations, Cauchy)."""
    from pysfm_amd import BundleAdjuster
    g = load_golden('scene_4x10_cauchy')

    class Cam(object):
        def __init__(self, R, t):
            self.R, self.t = np.array(R, float), np.array(t, float)

    class Trk(object):
        def __init__(self):
            self.measurements = {}

    class Cauchy(object):                                         # (only the name and .sigma travel to the device: sensor_model.device_params_of)
        def __init__(self, sigma):
            self.sigma = sigma

    class ForeignBundle(object):
        def __init__(self):
            self.K, self.cameras, self.tracks, self.reconstruction, self.sensor_model = None, [], [], None, None

        def check_consistency(self):
            assert np.shape(self.K) == (3, 3) and np.shape(self.reconstruction) == (len(self.tracks), 3)

        def clone_params(self):
            c = ForeignBundle()
            c.K, c.tracks, c.sensor_model = self.K.copy(), self.tracks, self.sensor_model
            c.cameras = [Cam(k.R, k.t) for k in self.cameras]
            c.reconstruction = self.reconstruction.copy()
            return c

    Cauchy.__name__ = 'CauchyModel'
    fb = ForeignBundle()
    fb.K = np.array(g['K'], float)
    fb.cameras = [Cam(R, t) for R, t in zip(g['R'], g['t'])]
    fb.tracks = [Trk() for _ in range(len(g['X']))]
    order = np.random.RandomState(0).permutation(len(g['obs_cam']))          # (dict insertion order must not matter)
    for n in order:
        fb.tracks[int(g['obs_pt'][n])].measurements[int(g['obs_cam'][n])] = np.array(g['obs_z'][n], float)
    fb.reconstruction = np.array(g['X'], float)
    fb.sensor_model = Cauchy(float(g['sensor_sigma']))
    ba = BundleAdjuster(fb, verbose=False)
    ba.optimize(max_steps=10)
    assert ba.num_steps == int(g['lm_num_steps']) and ba.converged == bool(g['lm_converged'])
    close(np.array(ba.costs), g['lm_costs'], 1e-7)
    out = ba.bundle
    assert isinstance(out, ForeignBundle) and out is not fb
    assert np.array_equal(fb.cameras[2].R, g['R'][2])                        # the caller's bundle is never mutated (bundle_adjuster.py:151)
    close(np.array([c.R for c in out.cameras]), g['lm_R'], 1e-6)
    close(np.array([c.t for c in out.cameras]), g['lm_t'], 1e-6, 1e-9)
    close(out.reconstruction, g['lm_X'], 1e-6)
    ba.backend.close()


# ------------------------------------------------------------------ the sensor-model plug-in point (sensor_model.py:19-32)
def test_a_caller_defined_robustifier_runs_on_the_device(be):
    """A Geman-McClure model defined HERE, as a plain object with the reference's four methods, handed to the adjuster as
    bundle.sensor_model: the kernels evaluate it through the sampled table (BA_SENSOR_TABLE).  (1) the reference's own
    self-check, sensor_model.validate, on the device form; (2) residual and Jacobian on the device against the Python
    model over eight decades of |e|; (3) an LM run on the reference's 5-camera / 50-point scene against the oracle driven
    by the same Python model, called per observation as the reference calls it (bundle.py:251-252, 269-273): same
    decisions, costs to 1e-6, parameters to 1e-6."""
    from conftest import GemanMcClure
    from pysfm_amd import Bundle, BundleAdjuster, sensor_model
    from pysfm_amd import synthetic_data as sd
    m = GemanMcClure(.05)
    sensor_model.validate(sensor_model.TabulatedModel(m))
    be.set_sensor(*sensor_model.device_params_of(m))
    rs = np.random.RandomState(1)
    e = rs.randn(4000, 2) * 10 ** rs.uniform(-5, 3, (4000, 1))
    e[:3] = [[0., 0.], [1e-300, 0.], [0., -3e-14]]
    r, J = be.eval_sensor(e)
    r0 = np.array([m.residual_from_error(x) for x in e])
    J0 = np.array([m.Jresidual_from_error(x) for x in e])
    assert np.max(np.abs(r - r0) / np.maximum(np.abs(r0).max(axis=1, keepdims=True), 1e-300)) <= 1e-11
    assert np.max(np.abs(J - J0) / np.abs(J0).max(axis=(1, 2), keepdims=True)) <= 2e-9
    K, Rs, ts, pts, msm = sd.generate_sequence(5, 50)
    b = Bundle.FromArrays(K, Rs, ts, pts, msm)
    b.sensor_model = m
    ba = BundleAdjuster(b, verbose=False)
    ba.optimize(max_steps=8)
    cam, pt, z = b.select_observations(range(5), range(50))
    flags = (np.arange(5, dtype=np.int32) - 1, np.ones(50, bool))
    trace = []
    ref = O.lm_optimize(O.Sensor.from_model(m), K, Rs, ts, pts, cam, pt, z, *flags, max_steps=8, trace=trace)
    got = [(d, o == 'accepted') for d, o, c in ba.trial_log]
    want = [(tr['damping'], tr['next'] < tr['cur']) for tr in trace]
    assert len(got) == len(want) and all(g[1] == w[1] and abs(g[0] - w[0]) <= 1e-12 * w[0] for g, w in zip(got, want)), (got, want)
    assert ba.num_steps == ref['num_steps'] and ba.converged == ref['converged']
    close(np.array(ba.costs), np.array(ref['costs']), 1e-6)
    out = ba.bundle
    close(out.ts(), ref['t'], 1e-6, 1e-9)
    close(out.reconstruction, ref['X'], 1e-6)
    assert ba.costs[-1] < ba.costs[0]
    ba.backend.close()


def test_a_caller_defined_robustifier_on_an_unordered_collection(be):
    """The table sensor model through the sparse path's kernels (k_schur_blocks<TABLE>, conjugate gradients): a Geman-McClure model
    defined in the tests on a 150-camera collection - S, b, dC against the oracle driven by the same Python object."""
    from conftest import GemanMcClure
    from pysfm_amd import sensor_model
    from pysfm_amd import synthetic_data as sd
    m = GemanMcClure(.05)
    nc, nt = 150, 1500
    s = sd.generate_collection_scene(nc, nt, partners=6, track_len=3)
    flags = default_flags(nc, nt)
    a = (s['K'], s['R0'], s['t0'], s['X0'], s['obs_cam'], s['obs_pt'], s['obs_z'])
    be.set_problem(nc, nt, s['obs_cam'], s['obs_pt'], s['obs_z'], s['K'], *flags)
    be.set_sensor(*sensor_model.device_params_of(m))
    be.set_params(0, s['R0'], s['t0'], s['X0'])
    be.debug_poison()
    be.set_option('solver', 'pcg')
    mu, su, parts = O.compute_update(O.Sensor.from_model(m), *a, *flags, damping=1., return_parts=True)
    info, cost = be.lm_trial(1., 1e-5, None)
    assert info == 0 and be.last_solve_kind == 'pcg'
    S, b = be.get_reduced()
    close(S, parts['S'], 2e-9)                               # (the table interpolates the model's Jacobian to 3e-10)
    close(b, parts['b'], 2e-9)
    close(-be.get_solution(), mu, 1e-7)


@pytest.mark.parametrize('name,tag,damping', [('scene_5x50_gauss', 'l10_', 10.), ('scene_4x10_cauchy', 'l2_', 2.), ('scene_oleg_10x50', 'l10_', 10.)])
@pytest.mark.parametrize('option', [('refine', '1'), ('solver', 'pcg')])
def test_small_reference_scenes_through_the_round_6_solver_paths(be, name, tag, damping, option):
    """The reference's own small scenes (one node of the cyclic reduction: the refinement's tree is its root alone; a handful of block
    rows for the conjugate gradients) through the refinement step and through conjugate gradients, by option: dC and dP equal to the
    REFERENCE's own (the golden vectors) to 1e-8."""
    g = load_golden(name)
    load_problem(be, *scene(g), g[tag + 'cam_opt_pos'], g[tag + 'pt_opt'].astype(np.uint8), sensor_of(g))
    be.set_option(*option)
    be.linearize(0)
    be.schur(0, damping, 1e-5)
    before = be.problem_info()['solves_refined']
    be.solve_reduced(None)
    if option[0] == 'refine':
        assert be.last_solve_kind == 'bcr' and be.problem_info()['solves_refined'] == before + 1
    else:
        assert be.last_solve_kind == 'pcg' and be.pcg_info()['rel_residual'] <= 1e-12
    close(be.get_solution(), g[tag + 'dC'], 1e-8)
    dP = be.backsubstitute(0)
    close(dP[g[tag + 'pt_opt'].astype(bool)], g[tag + 'dP'], 1e-8)
