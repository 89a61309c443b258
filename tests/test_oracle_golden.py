"""Pin the CPU oracle (oracle/ba_oracle.py) to golden vectors captured from the
reference itself by oracle/gen_golden.py.  CPU only."""
import os

import numpy as np
import pytest

from conftest import load_golden, GOLDEN
from oracle import ba_oracle as O

RTOL = 1e-9          # restatement reproduces every golden to <= 1e-9


def sensor_of(g, prefix='sensor_'):
    return O.Sensor(int(g[prefix + 'kind']), L=g[prefix + 'L'], sigma=float(g[prefix + 'sigma']))


def scene_args(g):
    return (g['K'], g['R'], g['t'], g['X'], g['obs_cam'], g['obs_pt'], g['obs_z'])


def close(a, b, rtol=RTOL, atol=0.0):
    a, b = np.asarray(a, float), np.asarray(b, float)
    assert a.shape == b.shape, (a.shape, b.shape)
    scale = max(np.max(np.abs(b)), 1e-300) if b.size else 1.0
    err = np.max(np.abs(a - b)) if b.size else 0.0
    assert err <= rtol * scale + atol, 'max abs err %.3e vs scale %.3e' % (err, scale)


# ---------------------------------------------------------------- functions
def test_so3_exp_matches_reference():
    g = load_golden('functions')
    close(O.so3_exp(g['so3_m']), g['so3_R'], 1e-14)


@pytest.mark.parametrize('tag', ['gauss_iso', 'gauss_diag', 'gauss_full', 'cauchy', 'cauchy2'])
def test_sensor_models_match_reference(tag):
    g = load_golden('functions')
    s = O.Sensor(int(g[tag + '_kind']), L=g[tag + '_L'], sigma=float(g[tag + '_sigma']))
    e = g['sens_e']
    close(O.sensor_residual(s, e), g[tag + '_r'], 1e-13)
    close(O.sensor_jacobian(s, e), g[tag + '_J'], 1e-12)
    close(O.sensor_cost(s, e), g[tag + '_cost'], 1e-13)


def test_huber_meets_reference_validate_criteria():
    """Huber is not in the reference (parity unpinned): apply the reference's own
    sensor_model.validate checks (sensor_model.py:76-99) + agreement with the
    pinned models where they coincide."""
    s = O.Sensor.huber(0.7)
    e = np.array([[1., 2.], [.1, .2], [0., 0.], [-3., .5], [.7, 0.], [.69, .0]])
    r = O.sensor_residual(s, e)
    close(np.sum(r * r, axis=1), O.sensor_cost(s, e), 1e-13, 1e-15)
    assert O.sensor_cost(s, np.zeros((1, 2)))[0] == 0.0
    J = O.sensor_jacobian(s, e)
    h = 1e-7
    for n in range(len(e)):
        if abs(np.linalg.norm(e[n]) - s.k) < 1e-3:
            continue
        for c in range(2):
            d = np.zeros(2)
            d[c] = h
            fd = (O.sensor_residual(s, e[n:n + 1] + d) - O.sensor_residual(s, e[n:n + 1] - d))[0] / (2 * h)
            assert np.max(np.abs(fd - J[n][:, c])) < 1e-5
    # inside the threshold Huber == Gaussian(1); continuity at the threshold
    inside = np.array([[.1, .2], [-.3, .4]])
    close(O.sensor_residual(s, inside), O.sensor_residual(O.Sensor.gaussian(1.), inside), 1e-15)
    edge = np.array([[.7 * (1 + 1e-12), 0.]])
    close(O.sensor_residual(s, edge), edge, 1e-9)


# ---------------------------------------------------------------- 4x10 scene
def test_per_observation_blocks_4x10():
    g = load_golden('scene_4x10_cauchy')
    s = sensor_of(g)
    close(O.reproj_error(*scene_args(g)), g['e'])
    r, Jc, Jp = O.jacobians(s, *scene_args(g))
    close(r, g['r'])
    close(Jc, g['Jc'])
    close(Jp, g['Jp'])
    assert len(g['obs_cam']) == 36
    close(O.complete_cost(s, *scene_args(g)), g['complete_cost'])
    # known answers quoted in SURVEY.md section 8(c)
    assert abs(g['complete_cost'] - 64.231038642231454) < 1e-9
    assert abs(g['l0_cost'] - 48.497433885068325) < 1e-9


@pytest.mark.parametrize('lam,tag', [(0., 'l0_'), (2., 'l2_')])
def test_normal_blocks_and_schur_4x10(lam, tag):
    g = load_golden('scene_4x10_cauchy')
    s = sensor_of(g)
    nc, nt = len(g['R']), len(g['X'])
    HCC, HPP, W, bC, bP = O.normal_blocks(s, *scene_args(g), nc, nt)
    for k, v in (('HCC', HCC), ('HPP', HPP), ('W', W), ('bC', bC), ('bP', bP)):
        close(v, g[tag + k])
    HPPi = O.invert_point_blocks(O.damp_blocks(HPP, lam), 1e-5)
    close(HPPi, g[tag + 'HPP_inv'])
    S, b = O.schur_complement(O.damp_blocks(HCC, lam), HPPi, W, bC, bP,
                              g['obs_cam'], g['obs_pt'], g[tag + 'cam_opt_pos'])
    close(S, g[tag + 'S'])
    close(b, g[tag + 'b'])
    if lam > 0:
        # (at lambda = 0 the reduced system is singular along the scale gauge: cond(S) ~ 1e17,
        #  so the reference's own dC there is round-off, not a pin; S and b above are the pin)
        dC = O.solve_reduced(S, b, np.ones(S.shape[0] * 6, bool))
        close(dC, g[tag + 'dC'])
        dP = O.backsubstitute(dC, HPPi, W, bP, g['obs_cam'], g['obs_pt'], g[tag + 'cam_opt_pos'], nt)
        close(dP, g[tag + 'dP'])
    else:
        assert np.linalg.cond(O.flatten_reduced(S, b)[0]) > 1e12
    close(O.cost(s, *scene_args(g), g[tag + 'cam_opt_pos'], g[tag + 'pt_opt']), g[tag + 'cost'])


def test_block_schur_equals_dense_schur_4x10():
    """The reference's own tests (bundle_adjuster_unittest.py:16-67) restated."""
    g = load_golden('scene_4x10_cauchy')
    s = sensor_of(g)
    r, J = O.dense_jacobian(s, *scene_args(g))
    J = J[:, 6:]
    Sd, bd = O.dense_schur_complement(J.T @ J, J.T @ r, 18)
    close(Sd, g['dense_S_l0'], 1e-9)
    close(bd, g['dense_b_l0'], 1e-9)
    Sflat, bflat = O.flatten_reduced(g['l0_S'], g['l0_b'])
    assert np.sum((Sflat - Sd) ** 2) <= 1e-7          # numpy_test.py:112-118 tolerance
    assert np.sum((bflat - bd) ** 2) <= 1e-7
    mu, su = O.compute_update(s, *scene_args(g), g['l2_cam_opt_pos'], g['l2_pt_opt'], damping=2.)
    close(mu, g['update_l2_motion'])
    close(su, g['update_l2_structure'])
    delta = np.concatenate((mu.reshape(-1), su.reshape(-1)))
    assert np.sum((delta - g['dense_delta_l2']) ** 2) <= 1e-7


def test_loop_variant_equals_vectorised():
    g = load_golden('scene_4x10_cauchy')
    s = sensor_of(g)
    a = O.normal_blocks(s, *scene_args(g), 4, 10)
    b = O.normal_blocks_loop(s, *scene_args(g), 4, 10)
    for x, y in zip(a, b):
        close(x, y, 1e-13)


# ---------------------------------------------------------------- subset scene
def test_subset_schur_matches_reference():
    g = load_golden('scene_subset')
    s = sensor_of(g)
    mu, su, parts = O.compute_update(s, *scene_args(g), g['l2_cam_opt_pos'], g['l2_pt_opt'],
                                     damping=2., return_parts=True)
    assert parts['S'].shape == (1, 1, 6, 6)
    close(parts['S'], g['l2_S'])
    close(parts['b'], g['l2_b'])
    close(parts['HCC'], g['l2_HCC'])
    close(parts['HPP'], g['l2_HPP'])
    close(mu, g['update_l2_motion'])
    close(su, g['update_l2_structure'])
    assert su.shape == (1, 3)
    close(O.cost(s, *scene_args(g), g['l2_cam_opt_pos'], g['l2_pt_opt']), g['l2_cost'])


# ---------------------------------------------------------------- LM trajectories
@pytest.mark.parametrize('name,steps', [('scene_4x10_cauchy', 10), ('scene_5x50_gauss', 5),
                                        ('scene_5x50_cauchy_masked', 8), ('scene_planar_lm', 50)])
def test_lm_trajectory_matches_reference(name, steps):
    g = load_golden(name)
    s = sensor_of(g)
    nc, nt = len(g['R']), len(g['X'])
    cam_opt_pos = np.arange(nc, dtype=np.int32) - 1          # first camera frozen
    pt_opt = np.ones(nt, bool)
    trace = []
    out = O.lm_optimize(s, *scene_args(g), cam_opt_pos, pt_opt, max_steps=steps, trace=trace)
    assert out['num_steps'] == int(g['lm_num_steps'])
    assert out['converged'] == bool(g['lm_converged'])
    close(out['costs'], g['lm_costs'], 1e-7)
    assert len(trace) == len(g['lm_trials'])
    close([tr['damping'] for tr in trace], g['lm_trials'][:, 0], 1e-12)
    close([tr['next'] for tr in trace], g['lm_trials'][:, 1], 1e-7)
    close(out['R'], g['lm_R'], 1e-6)
    close(out['t'], g['lm_t'], 1e-6, 1e-9)
    close(out['X'], g['lm_X'], 1e-6)


def test_loop_closure_scene_with_renumbered_cameras_matches_reference():
    """oracle/gen_golden_layout.py: 60 cameras renumbered at random, tracks of 5 + four loop-closure points - the reference's dense
    reduced system does not know about camera order (bundle_adjuster.py:259-312); the oracle on the same arrays must reproduce its
    S, b, dC, dP, update and LM walk."""
    g = load_golden('scene_loop_closure_60x424')
    s = sensor_of(g)
    nc, nt = len(g['R']), len(g['X'])
    HCC, HPP, W, bC, bP = O.normal_blocks(s, *scene_args(g), nc, nt)
    for k, v in (('HCC', HCC), ('HPP', HPP), ('bC', bC), ('bP', bP)):
        close(v, g['l2_' + k])
    HPPi = O.invert_point_blocks(O.damp_blocks(HPP, 2.), 1e-5)
    S, b = O.schur_complement(O.damp_blocks(HCC, 2.), HPPi, W, bC, bP, g['obs_cam'], g['obs_pt'], g['l2_cam_opt_pos'])
    close(S, g['l2_S'])
    close(b, g['l2_b'])
    dC = O.solve_reduced(S, b, np.ones(S.shape[0] * 6, bool))
    close(dC, g['l2_dC'])
    close(O.backsubstitute(dC, HPPi, W, bP, g['obs_cam'], g['obs_pt'], g['l2_cam_opt_pos'], nt), g['l2_dP'])
    close(O.cost(s, *scene_args(g), g['l2_cam_opt_pos'], g['l2_pt_opt']), g['l2_cost'])
    mu, su = O.compute_update(s, *scene_args(g), g['l2_cam_opt_pos'], g['l2_pt_opt'], damping=2.)
    close(mu, g['update_l2_motion'])
    close(su, g['update_l2_structure'])
    trace = []
    out = O.lm_optimize(s, *scene_args(g), g['l2_cam_opt_pos'], g['l2_pt_opt'], max_steps=4, trace=trace)
    assert out['num_steps'] == int(g['lm_num_steps']) and out['converged'] == bool(g['lm_converged'])
    close(out['costs'], g['lm_costs'], 1e-7)
    close([tr['next'] for tr in trace], g['lm_trials'][:, 1], 1e-7)
    # the scene is what its name says: the cameras of a track are far apart in the caller's numbering, and four tracks tie
    # cameras that are 30 apart along the sequence
    spread = np.zeros(nt, int)
    for k in range(nt):
        c = g['obs_cam'][g['obs_pt'] == k]
        spread[k] = c.max() - c.min()
    assert np.median(spread) > 20


def test_known_answers_5x50():
    g = load_golden('scene_5x50_gauss')
    want = [0.1515770559897651, 0.14507108508956248, 0.12678104023107856,
            0.11966878251986428, 0.11537244502638065, 0.11100988334711709]
    close(g['lm_costs'], want, 1e-10)
    assert abs(g['complete_cost'] - 0.1841512398522971) < 1e-12
    assert len(g['obs_cam']) == 250


# ---------------------------------------------------------------- pixel-unit scenes
@pytest.mark.parametrize('name', ['scene_oleg_10x50', 'scene_oleg_40x100'])
def test_oleg_subsets(name):
    g = load_golden(name)
    s = sensor_of(g)
    nc, nt = len(g['R']), len(g['X'])
    mu, su, parts = O.compute_update(s, *scene_args(g), g['l10_cam_opt_pos'], g['l10_pt_opt'],
                                     damping=10., return_parts=True)
    close(parts['b'], g['l10_b'], 1e-9)
    close(parts['HPP_inv'], g['l10_HPP_inv'], 1e-8)
    if 'l10_S' in g:
        close(parts['S'], g['l10_S'], 1e-9)
    else:
        assert abs(np.linalg.norm(parts['S']) / g['l10_S_fro'] - 1) < 1e-10
    close(mu, g['update_l10_motion'], 1e-7)
    close(su, g['update_l10_structure'], 1e-7)
    close(O.complete_cost(s, *scene_args(g)), g['complete_cost'])


@pytest.mark.skipif(not os.path.exists(os.path.join(GOLDEN, 'scene_oleg_100x1000.npz')),
                    reason='big spot-check fixture not generated')
def test_oleg_full_spot_check():
    g = load_golden('scene_oleg_100x1000')
    s = sensor_of(g)
    z = g['obs_z'].astype(float)
    a = (g['K'], g['R'], g['t'], g['X'], g['obs_cam'], g['obs_pt'], z)
    nc, nt = len(g['R']), len(g['X'])
    assert len(z) == 100000
    cam_opt_pos = np.arange(nc, dtype=np.int32) - 1
    mu, su, parts = O.compute_update(s, *a, cam_opt_pos, np.ones(nt, bool), damping=10., return_parts=True)
    close(parts['b'], g['l10_b'], 1e-9)
    assert abs(np.linalg.norm(parts['S']) / g['l10_S_fro'] - 1) < 1e-10
    close(-mu, g['l10_dC'], 1e-6)
    close(-su[:20], g['l10_dP_head'], 1e-6)
    close(O.cost(s, *a, cam_opt_pos, np.ones(nt, bool)), g['l10_cost'])
