"""Host logic of pysfm_amd (set_bundle bookkeeping, masks, the LM schedule, the Bundle
data model) on CPU.  The arithmetic comes from the OracleBackend test double, the
expected values from the reference's golden vectors - so these tests pin the Python
side of the drop-in boundary, not the kernels."""
import numpy as np
import pytest

from conftest import load_golden
from oracle_backend import OracleBackend
import pysfm_amd
from pysfm_amd import Bundle, BundleAdjuster, Camera, Track, sensor_model
from pysfm_amd import backend as backend_mod
from pysfm_amd.bundle_adjuster import select, NormalEquationsIllconditioned


@pytest.fixture(autouse=True)
def oracle_default_backend(monkeypatch):
    """Bundle.predict()/residual()/... use the process-wide default backend."""
    monkeypatch.setitem(backend_mod._default, 0, OracleBackend())


def model_of(g):
    if int(g['sensor_kind']) == 0:
        m = sensor_model.GaussianModel(1.)
        m.L = g['sensor_L']
        m.covinv = m.L @ m.L.T
        m.cov = np.linalg.inv(m.covinv)
        return m
    return sensor_model.CauchyModel(float(g['sensor_sigma']))


def bundle_of(g, prefix=''):
    return Bundle.FromObservations(g['K'], g[prefix + 'R'], g[prefix + 't'], g[prefix + 'X'],
                                   g[prefix + 'obs_cam'], g[prefix + 'obs_pt'], g[prefix + 'obs_z'],
                                   sensor_model=model_of(g))


def close(a, b, rtol=1e-9, atol=0.):
    a, b = np.asarray(a, float), np.asarray(b, float)
    assert a.shape == b.shape, (a.shape, b.shape)
    scale = np.max(np.abs(b)) if b.size else 1.
    assert np.max(np.abs(a - b)) <= rtol * scale + atol if b.size else True


# ------------------------------------------------------------------ select()
def test_select_bool_and_id_masks():
    s, idx = select([3, 1, 7], [False, True, True])
    assert list(s) == [1, 7] and idx == [1, 2]
    s, idx = select([3, 1, 7], np.array([7, 3]))
    assert list(s) == [7, 3] and idx == [2, 0]
    with pytest.raises(AssertionError):
        select([3, 1, 7], np.array([5]))
    with pytest.raises(AssertionError):
        select([3, 1, 7], [True, False])


# ------------------------------------------------------------------ LM trajectories
@pytest.mark.parametrize('name,steps', [('scene_4x10_cauchy', 10), ('scene_5x50_gauss', 5),
                                        ('scene_5x50_cauchy_masked', 8), ('scene_planar_lm', 50)])
def test_optimize_reproduces_reference_trajectory(name, steps):
    g = load_golden(name)
    b0 = bundle_of(g)
    R_before = b0.Rs().copy()
    ba = BundleAdjuster(b0, backend=OracleBackend(), verbose=False)
    assert ba.optim_camera_ids == list(range(1, len(g['R'])))
    ba.optimize(max_steps=steps)
    assert ba.num_steps == int(g['lm_num_steps'])
    assert ba.converged == bool(g['lm_converged'])
    close(ba.costs, g['lm_costs'], 1e-7)
    out = ba.bundle
    assert out is not b0                                   # bundle_adjuster.py:151: a new bundle
    assert np.array_equal(b0.Rs(), R_before)               # the caller's bundle is never mutated
    close(out.Rs(), g['lm_R'], 1e-6)
    close(out.ts(), g['lm_t'], 1e-6, 1e-9)
    close(out.reconstruction, g['lm_X'], 1e-6)


def test_step_is_one_outer_iteration():
    g = load_golden('scene_5x50_gauss')
    ba = BundleAdjuster(bundle_of(g), backend=OracleBackend(), verbose=False)
    conv = ba.step()
    assert conv is False and ba.num_steps == 1
    close(ba.costs, g['lm_costs'][:2], 1e-7)
    ba.step()
    close(ba.costs, g['lm_costs'][:3], 1e-7)


def test_every_trial_relinearises():
    """No cached outputs: each LM trial runs linearize -> schur -> solve -> backsub -> cost."""
    g = load_golden('scene_planar_lm')
    be = OracleBackend()
    ba = BundleAdjuster(bundle_of(g), backend=be, verbose=False)
    ba.optimize(max_steps=50)
    ntrials = len(g['lm_trials'])
    assert be.calls.count('linearize') == ntrials
    assert be.calls.count('schur') == ntrials
    assert be.calls.count('backsub') == ntrials


# ------------------------------------------------------------------ subsets and masks
def test_subset_schur_matches_reference():
    """bundle_adjuster_unittest.py:70-119."""
    g = load_golden('scene_subset')
    full = Bundle.FromObservations(g['K'], g['full_R'], g['full_t'], g['full_X'], g['full_obs_cam'],
                                   g['full_obs_pt'], g['full_obs_z'], sensor_model=model_of(g))
    ba = BundleAdjuster(backend=OracleBackend(), verbose=False)
    ba.set_bundle(full, [3, 1], [0, 1, 2], [False, True], [False, True, False])
    assert ba.optim_camera_ids == [1] and ba.optim_camera_indices == [1]
    assert ba.optim_track_ids == [1] and ba.optim_track_indices == [1]
    ba.prepare_schur_complement()
    ba.apply_damping(2.)
    A, b = ba.compute_schur_complement()
    assert A.shape == (1, 1, 6, 6) and b.shape == (1, 6)
    close(A, g['l2_S'])
    close(b, g['l2_b'])
    close(ba.HCCs, g['l2_HCC'] * (np.ones((6, 6)) + 2 * np.eye(6)))     # damped in place, like the reference
    close(ba.HPPs, g['l2_HPP'] * (np.ones((3, 3)) + 2 * np.eye(3)))
    close(ba.compute_cost(full), g['l2_cost'])
    mu, su = ba.compute_update(2.)
    close(mu, g['update_l2_motion'])
    close(su, g['update_l2_structure'])
    assert mu.shape == (1, 6) and su.shape == (1, 3)


def test_integer_id_masks():
    g = load_golden('scene_subset')
    full = Bundle.FromObservations(g['K'], g['full_R'], g['full_t'], g['full_X'], g['full_obs_cam'],
                                   g['full_obs_pt'], g['full_obs_z'], sensor_model=model_of(g))
    ba = BundleAdjuster(backend=OracleBackend(), verbose=False)
    ba.set_bundle(full, g['int_camera_ids'].tolist(), g['int_track_ids'].tolist(),
                  g['int_cam_mask'], g['int_track_mask'])
    assert ba.optim_camera_ids == [1, 3] and ba.optim_camera_indices == [3, 2]
    assert ba.optim_track_ids == [8, 5, 4] and ba.optim_track_indices == [3, 0, 1]
    close(ba.compute_cost(full), g['int_cost'])
    mu, su = ba.compute_update(.5)
    close(mu, g['int_update_motion'])
    close(su, g['int_update_structure'])


def test_compute_update_and_manual_pipeline_4x10():
    """bundle_adjuster_unittest.py:16-67 call shapes."""
    g = load_golden('scene_4x10_cauchy')
    bnd = bundle_of(g)
    ba = BundleAdjuster(bnd, backend=OracleBackend(), verbose=False)
    ba.prepare_schur_complement()
    ba.apply_damping(0.)
    A, b = ba.compute_schur_complement()
    close(A, g['l0_S'])
    close(b, g['l0_b'])
    nc = len(ba.optim_camera_ids)
    Aflat = A.transpose((0, 2, 1, 3)).reshape((6 * nc, 6 * nc))
    assert np.sum((Aflat - g['dense_S_l0']) ** 2) <= 1e-7
    close(ba.HCPs[g['obs_cam'], g['obs_pt']], g['l0_W'])
    assert ba.HCPs.shape == (4, 10, 6, 3)
    close(ba.HPP_invs, g['l0_HPP_inv'])
    mu, su = ba.compute_update(2.)
    close(mu, g['update_l2_motion'])
    close(su, g['update_l2_structure'])
    # host-array solve + backsubstitute, the way the reference chains them
    ba.prepare_schur_complement()
    ba.apply_damping(2.)
    S, b = ba.compute_schur_complement()
    dC = ba.solve_motion_normal_eqns(S, b, np.ones(nc * 6, bool))
    close(dC, g['l2_dC'])
    close(ba.backsubstitute(dC), g['l2_dP'])


def test_param_mask_semantics():
    g = load_golden('scene_4x10_cauchy')
    ba = BundleAdjuster(bundle_of(g), backend=OracleBackend(), verbose=False)
    nparams = 6 * 3 + 3 * 10
    mask = np.ones(nparams, bool)
    mask[[0, 7]] = False
    mu, su = ba.compute_update(2., mask)
    assert mu[0, 0] == 0 and mu[1, 1] == 0 and np.all(mu.reshape(-1)[[1, 2, 3]] != 0)
    with pytest.raises(AssertionError):
        ba.compute_update(2., np.ones(nparams + 1, bool))
    bad = mask.copy()
    bad[-1] = False
    with pytest.raises(AssertionError):
        ba.compute_update(2., bad)            # 'Eliminating point parameters not implemented'
    with pytest.raises(AssertionError):
        ba.compute_update(2., np.ones(nparams, int))


def test_singular_reduced_system_raises_reference_exception():
    g = load_golden('scene_4x10_cauchy')
    ba = BundleAdjuster(bundle_of(g), backend=OracleBackend(), verbose=False)
    S = np.zeros((3, 3, 6, 6))
    with pytest.raises(NormalEquationsIllconditioned):
        ba.solve_motion_normal_eqns(S, np.zeros((3, 6)), np.ones(18, bool))


def test_update_motion_and_structure_on_clone():
    g = load_golden('scene_4x10_cauchy')
    bnd = bundle_of(g)
    ba = BundleAdjuster(bnd, backend=OracleBackend(), verbose=False)
    mu, su = ba.compute_update(2.)
    bnext = ba.bundle.clone_params()
    ba.update_motion(mu, bnext)
    ba.update_structure(su, bnext)
    from oracle import ba_oracle as O
    R2, t2, X2 = O.apply_update(g['R'], g['t'], g['X'], mu, su, g['l2_cam_opt_pos'], g['l2_pt_opt'])
    close(bnext.Rs(), R2, 1e-14)
    close(bnext.ts(), t2, 1e-14)
    close(bnext.reconstruction, X2, 1e-14)
    assert np.array_equal(bnd.Rs(), g['R'])
    with pytest.raises(AssertionError):
        ba.update_motion(mu[:2], bnext)


def test_set_bundle_assertions():
    g = load_golden('scene_4x10_cauchy')
    bnd = bundle_of(g)
    ba = BundleAdjuster(backend=OracleBackend(), verbose=False)
    with pytest.raises(AssertionError):
        ba.set_bundle(bnd, camera_ids=[0])                 # 'Cannot optimize just one camera'
    with pytest.raises(AssertionError):
        ba.set_bundle(bnd, camera_ids=[0, 9])
    with pytest.raises(AssertionError):
        ba.set_bundle(bnd, track_ids=[-1, 2])
    empty = Bundle(2, 3)
    with pytest.raises(AssertionError):
        ba.set_bundle(empty)                               # 'reconstruction must be initialized'


# ------------------------------------------------------------------ Bundle data model
def test_from_arrays_equals_from_observations():
    g = load_golden('scene_4x10_cauchy')
    nc, nt = 4, 10
    msm = np.zeros((nc, nt, 2))
    mask = np.zeros((nc, nt), bool)
    msm[g['obs_cam'], g['obs_pt']] = g['obs_z']
    mask[g['obs_cam'], g['obs_pt']] = True
    a = Bundle.FromArrays(g['K'], g['R'], g['t'], g['X'], msm, mask)
    b = bundle_of(g)
    for x, y in zip(a.observation_table(), b.observation_table()):
        assert np.array_equal(x, y)
    assert len(b.tracks) == 10 and b.tracks[3].has_measurement(int(g['obs_cam'][g['obs_pt'] == 3][0]))
    assert sorted(a.tracks[2].camera_ids()) == sorted(b.tracks[2].camera_ids())
    assert np.array_equal(a.measurement(0, 0), b.measurement(0, 0))
    ci, ti, z = a.select_observations([3, 1], [0, 1, 2])
    gs = load_golden('scene_subset')
    assert np.array_equal(ci, gs['obs_cam']) and np.array_equal(ti, gs['obs_pt']) and np.array_equal(z, gs['obs_z'])


def test_bundle_per_observation_api():
    g = load_golden('scene_4x10_cauchy')
    b = bundle_of(g)
    n = 7
    i, j = int(g['obs_cam'][n]), int(g['obs_pt'][n])
    close(b.reproj_error(i, j), g['e'][n])
    close(b.residual(i, j), g['r'][n])
    Jc, Jp = b.Jresidual(i, j)
    close(Jc, g['Jc'][n])
    close(Jp, g['Jp'][n])
    close(b.predict(i, j), g['e'][n] + g['obs_z'][n])
    close(b.complete_cost(), g['complete_cost'])
    assert b.residuals().shape == (72,)
    J, rows, cols = b.Jresiduals_extended()
    assert J.shape == (72, 4 * 6 + 10 * 3) and rows.shape == (72, 2) and cols.shape == (54, 2)
    assert b.Jresiduals_partial([3, 1], [0, 1, 2]).shape[1] == 2 * 6 + 3 * 3
    assert b.num_params() == 54


def test_clone_params_and_perturb():
    g = load_golden('scene_4x10_cauchy')
    b = bundle_of(g)
    c = b.clone_params()
    assert c.tracks is b.tracks and c.sensor_model is b.sensor_model
    c.perturb(np.full(c.num_params(), .01))
    assert np.array_equal(b.Rs(), g['R']) and not np.allclose(c.Rs(), g['R'])
    close(c.reconstruction, g['X'] + .01, 1e-15)
    cam = Camera(np.eye(3), np.zeros(3))
    cam.perturb(np.array([0, 0, 0, 1., 2., 3.]))
    assert np.array_equal(cam.R, np.eye(3)) and np.array_equal(cam.t, [1, 2, 3])
    assert cam.projection_matrix().shape == (3, 4)


def test_track_and_incremental_building():
    b = Bundle()
    b.add_camera()
    b.add_camera(Camera(np.eye(3), np.ones(3)))
    t = b.add_track(Track([0, 1], [np.zeros(2), np.ones(2)]))
    assert t.has_measurement(1) and not t.has_measurement(2)
    t.add_measurement(0, np.array([2., 3.]))
    assert np.array_equal(b.measurement(0, 0), [2., 3.])
    assert b.reconstruction.shape == (1, 3)
    with pytest.raises(Exception):
        b.add_track(Track([5], [np.zeros(2)]))
    assert list(b.measurement_ids()) == [(0, 0), (1, 0)]


def test_sensor_model_run_tests_entry_point(capsys):
    sensor_model.run_tests()
    assert 'Huber' in capsys.readouterr().out


def test_sensor_models_protocol():
    for m in (sensor_model.GaussianModel([2., 3.]), sensor_model.CauchyModel(2.), sensor_model.HuberModel(.7)):
        assert sensor_model.validate(m)
        assert type(m.clone()) is type(m)
    kind, params = sensor_model.device_params_of(sensor_model.GaussianModel(.1))
    assert kind == 0 and np.allclose(params.reshape(2, 2), np.eye(2) / np.sqrt(.1))
    with pytest.raises(TypeError):
        sensor_model.device_params_of(object())


def test_synthetic_generator_matches_reference_scene():
    from pysfm_amd import synthetic_data as sd
    g = load_golden('scene_5x50_gauss')
    K, Rs, ts, pts, msm = sd.generate_sequence(5, 50)
    close(Rs, g['R'], 1e-15)
    close(ts, g['t'], 1e-15)
    close(pts, g['X'], 1e-15)
    close(msm.transpose(1, 0, 2).reshape(-1, 2), g['obs_z'], 1e-14)
    b = Bundle.FromArrays(K, Rs, ts, pts, msm)
    ba = BundleAdjuster(b, backend=OracleBackend(), verbose=False)
    ba.optimize(max_steps=5)
    close(ba.costs, g['lm_costs'], 1e-7)


def test_banded_scene_shape():
    from pysfm_amd import synthetic_data as sd
    s = sd.generate_banded_scene(30, 200, track_len=10, outlier_frac=.1)
    assert len(s['obs_cam']) == 2000 and np.all(np.diff(s['obs_pt']) >= 0)
    assert s['outliers'].sum() == 200
    span = s['obs_cam'].reshape(200, 10)
    assert np.all(span[:, -1] - span[:, 0] == 9)
    p = np.einsum('nij,nj->ni', s['R'][s['obs_cam']], s['X'][s['obs_pt']]) + s['t'][s['obs_cam']]
    assert np.all(p[:, 2] > 3)
    assert np.array_equal(s['R0'][0], s['R'][0])


# ------------------------------------------------------------------ round-2 regressions (ADVICE)
def test_compute_cost_evaluates_the_bundle_it_is_given_after_an_accepted_step():
    """The reference's compute_cost evaluates its argument (bundle_adjuster.py:165-171).  After optimize() the adjuster's
    device copy has moved on; the caller's original bundle must still cost what it cost."""
    g = load_golden('scene_5x50_gauss')
    b0 = bundle_of(g)
    ba = BundleAdjuster(b0, backend=OracleBackend(), verbose=False)
    c0 = ba.compute_cost(b0)
    ba.optimize(max_steps=3)
    assert ba.costs[-1] < c0
    assert ba.compute_cost(b0) == pytest.approx(c0, rel=1e-12)          # not the optimised cost
    assert ba.compute_cost(ba.bundle) == pytest.approx(ba.costs[-1], rel=1e-12)
    # in-place edits of the adjuster's current bundle are seen too
    cur = ba.bundle
    cur.reconstruction[3] += .05
    assert ba.compute_cost(cur) > ba.costs[-1]


def test_compute_cost_of_the_unchanged_current_bundle_keeps_the_linearisation():
    """round-3 ADVICE: prepare_schur_complement() -> compute_cost(ba.bundle) -> compute_schur_complement() is a sequence the
    reference allows (its compute_cost touches no block array, bundle_adjuster.py:165-171).  An unchanged bundle must neither
    trip 'call prepare_schur_complement() first' nor be uploaded again; an edited one is uploaded and invalidates the blocks."""
    g = load_golden('scene_4x10_cauchy')
    ba = BundleAdjuster(bundle_of(g), backend=OracleBackend(), verbose=False)
    uploads = []
    orig = ba.backend.set_params
    ba.backend.set_params = lambda which, *a: (uploads.append(which), orig(which, *a))[1]
    ba.prepare_schur_complement()
    ba.apply_damping(2.)
    c = ba.compute_cost(ba.bundle)
    assert uploads == [] and c == pytest.approx(float(g['l2_cost']), rel=1e-12)
    S, b = ba.compute_schur_complement()                              # still linearised, still damped
    close(S, g['l2_S'])
    close(b, g['l2_b'])
    cur = ba.bundle
    cur.reconstruction[2] += .01                                       # edited in place: seen, uploaded, blocks gone
    assert ba.compute_cost(cur) != pytest.approx(c, rel=1e-9) and uploads == [0]
    with pytest.raises(AssertionError):
        ba.compute_schur_complement()


def test_solver_timeout_is_not_reported_as_ill_conditioned():
    """round-3 ADVICE: status BA_SOLVE_TIMED_OUT of the one-launch cyclic reduction is a solver fault: warned about on its
    own and the trial repeated stepwise - never folded into 'not positive definite' (a damping increase would hide it)."""
    from pysfm_amd._capi import SOLVE_TIMED_OUT
    g = load_golden('scene_5x50_gauss')
    be = OracleBackend()
    ba = BundleAdjuster(bundle_of(g), backend=be, verbose=False)
    ref = BundleAdjuster(bundle_of(g), backend=OracleBackend(), verbose=False)
    be.lm_trial = lambda damping, rcond, mask=None: (SOLVE_TIMED_OUT, float('nan'))
    with pytest.warns(RuntimeWarning, match='timed out'):
        ba.optimize(max_steps=2)
    ref.optimize(max_steps=2)
    assert ba.solver_timeouts >= 2 and getattr(ba, 'cholesky_rejections', 0) == 0
    close(ba.costs, ref.costs)


def test_block_properties_follow_the_latest_linearisation():
    """HCCs / HPPs / bCs / bPs / HCPs always reflect the last prepare_schur_complement / compute_update, as the
    reference's arrays do - also after a step moved the linearisation point (no stale cache, W on demand)."""
    g = load_golden('scene_5x50_gauss')
    ba = BundleAdjuster(bundle_of(g), backend=OracleBackend(), verbose=False)
    ba.prepare_schur_complement()
    H0, P0 = ba.HCCs.copy(), ba.HPPs.copy()
    ba.optimize(max_steps=2)
    ba.compute_update(1.)
    H1 = ba.HCCs
    assert np.abs(H1 / 2. - H0)[1:].max() > 1e-6 * np.abs(H0).max()      # a different point, and damped diagonals (x 2)
    W = ba.HCPs                                                         # needs W: linearised again on demand
    assert W.shape == (5, 50, 6, 3) and np.abs(W).max() > 0
    ba.prepare_schur_complement()
    close(ba.HPPs, ba._blocks()['HPP'])
    assert np.abs(ba.HPPs - P0).max() > 0


def test_shard_tracks_any_order_and_empty_shards():
    """Shards are consecutive stretches of the tracks ORDERED BY FIRST CAMERA, observation-balanced, each track in
    exactly one shard - whatever order the tracks come in; more ranks than tracks leaves some ranks empty, and
    set_bundle takes an empty track list."""
    from pysfm_amd import synthetic_data as sd
    from pysfm_amd.distributed import shard_tracks, shard_bounds
    s = sd.generate_banded_scene(40, 600, track_len=6)
    rs = np.random.RandomState(3)
    new_id = rs.permutation(600)
    X0 = np.empty_like(s['X0'])
    X0[new_id] = s['X0']
    o = rs.permutation(len(s['obs_cam']))
    b = Bundle.FromObservations(s['K'], s['R0'], s['t0'], X0, s['obs_cam'][o], new_id[s['obs_pt'][o]], s['obs_z'][o])
    cam, trk, _ = b.observation_table()
    shards = [shard_tracks(b, r, 4) for r in range(4)]
    assert sorted(sum(shards, [])) == list(range(600))
    spans = []
    for ids in shards:
        m = np.isin(trk, ids)
        assert abs(m.sum() - len(trk) / 4.) <= 6 * 2                    # balanced to within a couple of tracks
        spans.append((cam[m].min(), cam[m].max()))
    for (lo0, hi0), (lo1, hi1) in zip(spans, spans[1:]):
        assert lo0 <= lo1 and hi0 <= hi1 and hi0 - lo0 <= 40 // 4 + 8     # one stretch of the camera sequence each
    assert shard_bounds([3, 3], 4)[-1] == 2
    tiny = Bundle.FromObservations(s['K'], s['R0'], s['t0'], s['X0'][:2], s['obs_cam'][:12], s['obs_pt'][:12], s['obs_z'][:12])
    parts = [shard_tracks(tiny, r, 4) for r in range(4)]
    assert sorted(sum(parts, [])) == [0, 1] and any(len(p) == 0 for p in parts)
    ba = BundleAdjuster(backend=OracleBackend(), verbose=False)
    empty = [p for p in parts if not p][0]
    ba.set_bundle(tiny, camera_ids=list(range(12)), track_ids=empty)     # must not raise
    assert ba.track_ids == [] and ba.optim_track_ids == []


# ------------------------------------------------------------------ round 3: the cut of the distributed reduced solve (no GPU needed)
def _dist_plan(nco, hb, world):
    """ba_dist_plan through ctypes: a pure function of its arguments (HipBackend.dist_plan without a device)."""
    import ctypes as C
    from pysfm_amd import _capi
    lib = _capi.load()
    cb, N, P = C.c_int32(), C.c_int32(), C.c_int32()
    if lib.ba_dist_plan(int(nco), int(hb), int(world), C.byref(cb), C.byref(N), C.byref(P)) != _capi.BA_OK:
        return None
    return cb.value, N.value, P.value


def test_distributed_solve_plan_and_track_cut():
    """csrc/ba_dist.h: the elimination tree of the cyclic reduction cut along the ranks.  The plan for BASELINE config 5
    (9999 optimised cameras, half-bandwidth 9, 8 ranks) is 1000 nodes of 10 cameras, 128 per rank; it exists only for 2^g
    ranks and systems with enough nodes; shard_tracks(plan=...) cuts the tracks where the plan says - consecutive ranges
    that cover every track, each rank's tracks STARTING inside its own interval of camera positions."""
    from pysfm_amd import synthetic_data as sd
    from pysfm_amd.distributed import shard_tracks, tree_cut
    assert _dist_plan(9999, 9, 8) == (10, 1000, 128)
    assert _dist_plan(9999, 9, 2) == (10, 1000, 512)
    assert _dist_plan(9999, 9, 3) is None and _dist_plan(9999, 9, 1) is None       # 2^g ranks only
    assert _dist_plan(50, 9, 8) is None                                             # too few nodes to cut
    assert _dist_plan(9999, 12, 8) is None                                          # beyond the narrow cyclic reduction
    cb, N, P = _dist_plan(2399, 9, 4)
    assert cb >= 9 and N == -(-2399 // cb) and 3 * P < N <= 4 * P and P & (P - 1) == 0
    nc, nt, world = 2400, 6000, 4
    s = sd.generate_banded_scene(nc, nt)
    b = Bundle.FromObservations(s['K'], s['R0'], s['t0'], s['X0'], s['obs_cam'], s['obs_pt'], s['obs_z'])
    cuts = tree_cut(b, world, _dist_plan)
    assert cuts[0] == 0 and cuts[-1] == nc and cuts[1:-1] == [r * P * cb + 1 for r in range(1, world)]
    first = np.full(nt, nc, np.int64)
    np.minimum.at(first, s['obs_pt'], s['obs_cam'])
    seen = []
    for r in range(world):
        ids = shard_tracks(b, r, world, plan=_dist_plan)
        seen += ids
        pos = np.maximum(first[ids], 1) - 1                     # first OPTIMISED position (camera 0 is frozen)
        assert len(ids) > 0 and pos.min() >= r * P * cb and (pos.max() < (r + 1) * P * cb or r == world - 1)
    assert sorted(seen) == list(range(nt))
    # without a plan (or where none exists) the cut is the observation-balanced one
    sizes = [len(shard_tracks(b, r, 3)) for r in range(3)]
    assert sum(sizes) == nt and max(sizes) - min(sizes) <= 2
    assert [len(shard_tracks(b, r, 3, plan=_dist_plan)) for r in range(3)] == sizes


def test_projection_jacobian_helpers_and_plane_rotations():
    """The module-level helpers the reference's own tests use (bundle.py:22-50, geometry.py:28-41; bundle_unittest.py:146-162
    checks exactly this): Jproject_R / _t / _x against central differences of project2 / project, Jproject_cam / _all
    consistent with them; rotation_xy / _xz / _yz with the reference's sign conventions."""
    from pysfm_amd import bundle as B, geometry as G
    K0 = np.array([[1.2, 0., .1], [0., .9, -.05], [0., 0., 1.]])
    R0 = G.rotation_xy(1.1)
    t0 = np.array([3., 4., 5.5])
    x0 = np.array([-1., 2., 6.])

    def fd(f, at):
        J = np.empty((2, 3))
        for k in range(3):
            d = np.zeros(3)
            d[k] = 1e-6
            J[:, k] = (f(at + d) - f(at - d)) / 2e-6
        return J
    close(B.Jproject_R(K0, R0, t0, x0), fd(lambda m: B.project2(K0, R0, m, t0, x0), np.zeros(3)), 1e-7)
    close(B.Jproject_t(K0, R0, t0, x0), fd(lambda t: B.project(K0, R0, t, x0), t0), 1e-7)
    close(B.Jproject_x(K0, R0, t0, x0), fd(lambda x: B.project(K0, R0, t0, x), x0), 1e-7)
    Jc, Jx = B.Jproject_all(K0, R0, t0, x0)
    close(Jc, np.hstack((B.Jproject_R(K0, R0, t0, x0), B.Jproject_t(K0, R0, t0, x0))), 1e-14)
    close(Jc, B.Jproject_cam(K0, R0, t0, x0), 1e-14)
    close(Jx, B.Jproject_x(K0, R0, t0, x0), 1e-14)
    th = .37
    c, s_ = np.cos(th), np.sin(th)
    close(G.rotation_xy(th), [[c, -s_, 0], [s_, c, 0], [0, 0, 1]], 1e-15)
    close(G.rotation_xz(th), [[c, 0, -s_], [0, 1, 0], [s_, 0, c]], 1e-15)
    close(G.rotation_yz(th), [[1, 0, 0], [0, c, -s_], [0, s_, c]], 1e-15)
    for R in (G.rotation_xy(th), G.rotation_xz(th), G.rotation_yz(th)):
        close(R @ R.T, np.eye(3), 1e-15)
        assert abs(np.linalg.det(R) - 1.) < 1e-15


# ------------------------------------------------------------------ round-3 ADVICE
def test_few_of_many_tracks_after_an_edit_of_the_track_objects():
    """The few-of-many-tracks path of select_observations must see an edit of a Track-object bundle that keeps every count
    it could compare (a measurement moved from one track to another): no offsets cached beside mutable tracks."""
    b = Bundle()
    for i in range(3):
        b.add_camera(Camera(np.eye(3), np.array([0., 0., float(i)])))
    for j in range(12):
        b.add_track(Track([0, 2], np.array([[.1 * j, 0.], [.2 * j, 0.]])))
    b.reconstruction = np.ones((12, 3))
    cam0, trk0, _ = b.select_observations([0, 1, 2], [1, 5])
    assert cam0.tolist() == [0, 2, 0, 2] and trk0.tolist() == [0, 0, 1, 1]
    z = b.tracks[5].measurements.pop(0)                  # move track 5's measurement in camera 0 to track 1 as camera 1's
    b.tracks[1].measurements[1] = z
    cam1, trk1, z1 = b.select_observations([0, 1, 2], [1, 5])
    assert cam1.tolist() == [0, 1, 2, 2] and trk1.tolist() == [0, 0, 0, 1]
    assert np.array_equal(z1[1], z)
    # an array-native bundle (immutable table) still caches its offsets
    t = Bundle.FromObservations(np.eye(3), [np.eye(3)] * 3, np.zeros((3, 3)), np.ones((12, 3)),
                                [0, 2] * 12, np.repeat(np.arange(12), 2), np.zeros((24, 2)))
    c_w, t_w, _ = t.select_observations([0, 1, 2], [1, 5])                 # a window of consecutive cameras: binary searches in the table's key
    assert c_w.tolist() == [0, 2, 0, 2] and t_w.tolist() == [0, 0, 1, 1]
    assert getattr(t, '_table_key', None) is not None and getattr(b, '_table_key', None) is None
    c_s, t_s, _ = t.select_observations([2, 0], [1, 5])                    # any other camera list: the tracks' rows
    assert c_s.tolist() == [0, 1, 0, 1] and t_s.tolist() == [0, 0, 1, 1]      # (positions in the camera list, the list's order)
    assert getattr(t, '_track_offsets', None) is not None and getattr(b, '_track_offsets', None) is None
    assert t.clone_params().__dict__.get('_table_key') is t._table_key


def test_a_timed_out_solve_survives_the_sum_over_the_ranks():
    """The shards' trial records are SUMMED over the ranks; a time-out on one rank (or on all of them) must come out as
    SOLVE_TIMED_OUT, a pivot index as a pivot index (csrc/ba_types.h trial_status_word / trial_status_of_sum)."""
    from pysfm_amd._capi import SOLVE_TIMED_OUT
    from pysfm_amd.distributed import TRIAL_TIMED_OUT_WORD, trial_status_of_sum
    word = lambda st: TRIAL_TIMED_OUT_WORD if st == SOLVE_TIMED_OUT else float(st)
    for world in (2, 4, 8):
        for own_parts in (False, True):
            assert trial_status_of_sum(0., world, own_parts) == 0
            assert trial_status_of_sum(word(SOLVE_TIMED_OUT) + (world - 1) * word(0), world, own_parts) == SOLVE_TIMED_OUT
            assert trial_status_of_sum(world * word(SOLVE_TIMED_OUT), world, own_parts) == SOLVE_TIMED_OUT
            assert trial_status_of_sum(word(SOLVE_TIMED_OUT) + (world - 1) * word(2 ** 31 - 1), world, own_parts) == SOLVE_TIMED_OUT
        assert trial_status_of_sum(world * word(37), world, False) == 37              # every rank solved the same system
        assert trial_status_of_sum(word(37), world, True) == 37                       # one rank's part failed
        assert trial_status_of_sum(world * word(2 ** 31 - 1), world, False) != SOLVE_TIMED_OUT


def test_algebra_and_optimize_helper_modules():
    """The reference's public helper modules (algebra.py:5-56, optimize.py:7-15) keep their names for ported callers."""
    from pysfm_amd import algebra, optimize
    assert np.allclose(algebra.pr([2., 4., 2.]), [1., 2.])
    assert np.allclose(algebra.pr(np.array([[2., 4., 2.], [3., 3., 3.]])), [[1., 2.], [1., 1.]])
    assert np.allclose(algebra.unpr([1., 2.]), [1., 2., 1.])
    assert algebra.unpr(np.zeros((4, 2))).shape == (4, 3)
    H = np.array([[2., 0., 1.], [0., 2., 1.], [0., 0., 2.]])
    assert np.allclose(algebra.prdot(H, [1., 2.]), [1.5, 2.5])
    assert np.allclose(algebra.prdot(H, np.array([[1., 2.], [0., 0.]])), [[1.5, 2.5], [.5, .5]])
    with pytest.raises(Exception):
        algebra.pr(np.zeros((2, 2, 2)))
    assert np.allclose(algebra.dots(H, H, np.eye(3)), H @ H) and algebra.ssq(np.array([1., 2., 3.])) == 14.
    assert np.allclose(algebra.skew([1., 2., 3.]) @ [4., 5., 6.], np.cross([1., 2., 3.], [4., 5., 6.]))
    A = np.ones((3, 3))
    Bd = optimize.apply_lm_damping(A, 2.)
    assert np.allclose(np.diag(Bd), 3.) and np.allclose(A, 1.) and Bd[0, 1] == 1.
    optimize.apply_lm_damping_inplace(A, 1.)
    assert np.allclose(np.diag(A), 2.) and optimize.skew is algebra.skew


def test_a_caller_defined_sensor_model_is_sampled_into_a_table():
    """The reference's plug-in point (sensor_model.py:19-32, bundle.py:269-273): any object with the four methods.  An
    isotropic one gets a device form - BA_SENSOR_TABLE, h(rho) and dh/dlog2(rho) on a grid the kernels interpolate; here
    the interpolation formula of csrc/ba_math.h mirrored in NumPy against the model itself; an anisotropic one is refused."""
    from conftest import GemanMcClure
    from pysfm_amd._capi import SENSOR_TABLE
    m = GemanMcClure(.05)
    kind, p = sensor_model.device_params_of(m)
    assert kind == SENSOR_TABLE and sensor_model.device_params_of(m)[1] is p          # sampled once, cached on the object
    u0, inv_du, n = p[0], p[1], int(p[2])
    tab = p[3:].reshape(n, 2)
    rs = np.random.RandomState(0)
    for _ in range(300):
        e = rs.randn(2) * 10 ** rs.uniform(-4, 2)
        rho2 = e.dot(e)
        t = (0.5 * np.log2(rho2) - u0) * inv_du
        i = min(int(t), n - 2)
        f, du = t - i, 1. / inv_du
        h0, m0, h1, m1 = tab[i, 0], tab[i, 1] * du, tab[i + 1, 0], tab[i + 1, 1] * du
        c2, c3 = 3 * (h1 - h0) - 2 * m0 - m1, 2 * (h0 - h1) + m0 + m1
        h = h0 + f * (m0 + f * (c2 + f * c3))
        q = (m0 + f * (2 * c2 + 3 * f * c3)) * inv_du / np.log(2.) / rho2
        close(h * e, m.residual_from_error(e), 1e-11)
        close(h * np.eye(2) + q * np.outer(e, e), m.Jresidual_from_error(e), 1e-8)

    class Anisotropic(GemanMcClure):
        def residual_from_error(self, e):
            return GemanMcClure.residual_from_error(self, e) * np.array([1., 2.])

        def Jresidual_from_error(self, e):
            return GemanMcClure.Jresidual_from_error(self, e) * np.array([[1.], [2.]])
    with pytest.raises(TypeError, match='not isotropic'):
        sensor_model.device_params_of(Anisotropic(.05))
    with pytest.raises(TypeError, match='no device form'):
        sensor_model.device_params_of(object())


# ------------------------------------------------------------------ the loop taken on the device, replayed from its log
class _ResidentDouble(OracleBackend):
    """OracleBackend + the two entry points of the resident loop (backend.lm_resident_fits / lm_resident), with the schedule of
    csrc/ba_resident.h restated in Python: runs trials until done, until its log is full (`capacity`), or until it meets a
    trial it refuses (`refuse`: indices of trials, counted over the whole run, that end the launch with exit reason 2)."""

    def __init__(self, capacity=1000, refuse=(), lose_launch=None):
        OracleBackend.__init__(self)
        self.capacity, self.refuse, self.seen, self.launches = capacity, set(refuse), 0, 0
        self.lose_launch = lose_launch        # the launch (counted from 1) whose workgroups "lose each other": exit reason 4, nothing done

    def lm_resident_fits(self):
        return True

    def lm_resident(self, max_steps, steps_taken, in_step, converged, damping, improvement_threshold, rcond, cur_cost, cam_param_mask=None):
        from types import SimpleNamespace
        self.launches += 1
        log = SimpleNamespace(ntrials=0, nsteps=steps_taken, converged=int(bool(converged)), in_step=int(bool(in_step)), exit_reason=0,
                              exit_info=0, accepted=0, have_cost0=0, damping=damping, cost0=0., cur_cost=-1. if cur_cost is None else cur_cost,
                              trial_damping=[], trial_cost=[], trial_accepted=[])
        if self.launches == self.lose_launch:
            log.exit_reason, log.nsteps, log.in_step, log.converged, log.damping = 4, 0, 0, 0, 0.      # (the C side zeroes the header)
            return log
        while True:
            if not log.in_step:
                if log.converged or log.nsteps >= max_steps:
                    break
                log.nsteps += 1
                log.in_step = 1
            if log.converged or not (log.damping < 1e+8):
                log.in_step = 0
                continue
            if log.ntrials >= self.capacity:
                log.exit_reason = 1
                break
            if not log.have_cost0:
                log.cost0, log.have_cost0 = self.cost(0), 1
                if log.cur_cost < 0:
                    log.cur_cost = log.cost0
            if self.seen in self.refuse:
                self.refuse.discard(self.seen)
                log.exit_reason, log.exit_info = 2, 7
                break
            self.seen += 1
            self.linearize(0); self.schur(0, log.damping, rcond); self.solve_reduced(cam_param_mask)
            self.backsubstitute(0, fetch=False); self.apply_update(0, 1)
            cost = self.cost(1)
            acc = cost < log.cur_cost
            log.trial_damping.append(log.damping); log.trial_cost.append(cost); log.trial_accepted.append(int(acc))
            log.ntrials += 1
            if acc:
                self.swap_params()
                log.damping *= .1
                log.converged = int(abs(log.cur_cost - cost) < improvement_threshold)
                log.cur_cost, log.accepted, log.in_step = cost, 1, 0
            else:
                log.damping *= 10.
                log.converged = int(log.damping > 1e+8)
        return log


@pytest.mark.parametrize('name,steps', [('scene_4x10_cauchy', 10), ('scene_5x50_gauss', 5), ('scene_planar_lm', 50)])
@pytest.mark.parametrize('capacity,refuse', [(1000, ()), (3, ()), (1000, (0, 4, 5)), (2, (1, 6))])
def test_the_log_of_the_resident_loop_replays_into_the_reference_walk(name, steps, capacity, refuse):
    g = load_golden(name)
    plain = BundleAdjuster(bundle_of(g), backend=OracleBackend(), verbose=False)
    plain.optimize(max_steps=steps)
    be = _ResidentDouble(capacity, refuse)
    ba = BundleAdjuster(bundle_of(g), backend=be, verbose=False)
    ba.optimize(max_steps=steps)
    assert be.launches >= (2 if refuse or capacity < 5 else 1)
    assert (ba.num_steps, ba.converged, ba.lm_trials) == (plain.num_steps, plain.converged, plain.lm_trials)
    assert ba.num_steps == int(g['lm_num_steps']) and ba.converged == bool(g['lm_converged'])
    assert [(d, o) for d, o, _ in ba.trial_log] == [(d, o) for d, o, _ in plain.trial_log]
    close(ba.costs, plain.costs, 1e-12)
    close(ba.costs, g['lm_costs'], 1e-6)
    assert ba._damping == plain._damping
    close(ba.bundle.reconstruction, plain.bundle.reconstruction, 1e-12)
    # step by step, the same walk
    be2 = _ResidentDouble(capacity, refuse)
    st = BundleAdjuster(bundle_of(g), backend=be2, verbose=False)
    st.num_steps, st.converged, st.costs, st.trial_log, st.lm_trials, st._damping, st._cur_cost = 0, False, [], [], 0, 10., None
    while not st.converged and st.num_steps < steps:
        st.step()
    assert st.trial_log == ba.trial_log and st.costs == ba.costs and st.num_steps == ba.num_steps
    # a parameter mask, a switched-off loop: the Python loop
    off = BundleAdjuster(bundle_of(g), backend=_ResidentDouble(), verbose=False)
    off.resident = False
    off.optimize(max_steps=2)
    assert off.backend.launches == 0
    # a mask over camera parameters travels with the launch: the same walk as the Python loop with that mask
    nparams = len(ba.optim_camera_ids) * 6 + len(ba.optim_track_ids) * 3
    m = np.ones(nparams, bool)
    m[[2, 7]] = False
    pm = BundleAdjuster(bundle_of(g), backend=OracleBackend(), verbose=False)
    pm.optimize(param_mask=m, max_steps=3)
    masked = BundleAdjuster(bundle_of(g), backend=_ResidentDouble(), verbose=False)
    masked.optimize(param_mask=m, max_steps=3)
    assert masked.backend.launches >= 1
    assert [(d, o) for d, o, _ in masked.trial_log] == [(d, o) for d, o, _ in pm.trial_log]
    close(masked.costs, pm.costs, 1e-12)


@pytest.mark.parametrize('capacity,lose', [(1000, 1), (3, 2), (2, 3)])
def test_a_resident_launch_that_times_out_hands_over_to_the_python_loop(capacity, lose):
    """exit reason 4 (the workgroups lost each other) leaves the device's current set and the schedule untouched;
    optimize() must neither raise nor lose the state - it finishes on the general loop and walks the reference's walk."""
    g = load_golden('scene_5x50_gauss')
    plain = BundleAdjuster(bundle_of(g), backend=OracleBackend(), verbose=False)
    plain.optimize(max_steps=5)
    be = _ResidentDouble(capacity, lose_launch=lose)
    ba = BundleAdjuster(bundle_of(g), backend=be, verbose=False)
    with pytest.warns(RuntimeWarning, match='resident loop timed out'):
        ba.optimize(max_steps=5)
    assert be.launches == lose and ba.resident is False and ba.resident_timeouts == 1
    assert (ba.num_steps, ba.converged, ba.lm_trials) == (plain.num_steps, plain.converged, plain.lm_trials)
    assert [(d, o) for d, o, _ in ba.trial_log] == [(d, o) for d, o, _ in plain.trial_log]
    close(ba.costs, plain.costs, 1e-12)
    close(ba.bundle.reconstruction, plain.bundle.reconstruction, 1e-12)


def test_select_observations_after_add_camera_on_an_array_native_bundle():
    """the cached (track, camera) key of the consecutive-camera fast path carries its own multiplier."""
    rs = np.random.RandomState(11)
    nc, nt = 6, 9
    cam, trk = np.nonzero(rs.rand(nc, nt) < .7)
    b = Bundle.FromObservations(np.eye(3), np.tile(np.eye(3), (nc, 1, 1)), np.zeros((nc, 3)), np.ones((nt, 3)), cam, trk, rs.randn(len(cam), 2))
    want = b.select_observations(range(1, 4), [7, 2, 3])
    b.add_camera()
    got = b.select_observations(range(1, 4), [7, 2, 3])                     # first call after the camera count changed
    for a, w in zip(got, want):
        assert np.array_equal(a, w)
    b2 = Bundle.FromObservations(np.eye(3), np.tile(np.eye(3), (nc, 1, 1)), np.zeros((nc, 3)), np.ones((nt, 3)), cam, trk, rs.randn(len(cam), 2))
    first = b2.select_observations(range(1, 4), [7, 2, 3])                  # key cached with nc = 6 ...
    b2.add_camera()
    again = b2.select_observations(range(1, 4), [7, 2, 3])                  # ... and used with nc = 7
    for a, w in zip(again, first):
        assert np.array_equal(a, w)
    ci, ti, _ = again
    assert ci.min() >= 0 and ci.max() <= 2 and ti.max() <= 2


def test_a_foreign_sensor_model_is_sampled_again_when_its_parameters_change():
    """the table cached on a caller's model carries a fingerprint of the model it was sampled from."""
    class Geman(object):
        def __init__(self, s): self.s = s
        def cost_from_error(self, e): r = self.residual_from_error(e); return float(np.dot(r, r))
        def residual_from_error(self, e):
            e = np.asarray(e, float); return e / np.sqrt(self.s * self.s + e.dot(e))
        def Jresidual_from_error(self, e):
            e = np.asarray(e, float); d = self.s * self.s + e.dot(e)
            return np.eye(2) / np.sqrt(d) - np.outer(e, e) / d ** 1.5
    m = Geman(.5)
    k1, p1 = sensor_model.device_params_of(m)
    k1b, p1b = sensor_model.device_params_of(m)
    assert p1b is p1                                                         # unchanged model: the cached table
    m.s = 2.
    k2, p2 = sensor_model.device_params_of(m)
    assert not np.array_equal(p1, p2)
    _, fresh = sensor_model.tabulate(Geman(2.))
    assert np.array_equal(p2, fresh)


def test_cameras_over_stacked_arrays_concatenate_and_copy():
    """list operations that read the raw storage must not hand out the lazy placeholders."""
    R = np.tile(np.eye(3), (4, 1, 1)); t = np.arange(12.).reshape(4, 3)
    b = Bundle.FromObservations(np.eye(3), R, t, np.ones((2, 3)), [0, 1, 2, 3], [0, 0, 1, 1], np.zeros((4, 2)))
    extra = Camera(np.eye(3), np.ones(3))
    both = b.cameras + [extra]
    assert type(both) is list and len(both) == 5 and all(c is not None for c in both) and np.array_equal(both[2].t, t[2])
    assert all(c is not None for c in [extra] + b.cameras) and all(c is not None for c in b.cameras * 2)
    assert all(c is not None for c in b.cameras.copy()) and len(2 * b.cameras) == 8


def test_cameras_over_stacked_arrays_behave_like_a_list():
    """Bundle.cameras of an array-native bundle makes Camera objects on first access (bundle._Cameras): a list all the same."""
    import copy
    import pickle
    rs = np.random.RandomState(3)
    R = np.array([np.linalg.qr(rs.randn(3, 3))[0] for _ in range(6)])
    t = rs.randn(6, 3)
    b = Bundle.FromObservations(np.eye(3), R, t, np.ones((4, 3)), [0, 1, 2, 3, 4, 5, 0, 1], [0, 0, 1, 1, 2, 2, 3, 3], np.zeros((8, 2)))
    cams = b.cameras
    assert isinstance(cams, list) and len(cams) == 6
    assert np.array_equal(cams[2].R, R[2]) and cams[-1].idx == 5 and cams[2] is cams[2]
    cams[2].perturb(np.array([.1, 0, 0, 1., 2., 3.]))                          # an object that was handed out is the truth
    cams[4].t[:] = 7.                                                         # ... also when edited in place
    assert np.allclose(b.ts()[2], t[2] + [1., 2., 3.]) and np.all(b.ts()[4] == 7.) and np.array_equal(b.ts()[0], t[0])
    c = b.clone_params()
    assert np.array_equal(c.Rs(), b.Rs()) and np.array_equal(c.ts(), b.ts())
    c.cameras[4].t[:] = 0.
    c.cameras[0].R = np.eye(3)
    assert np.all(b.ts()[4] == 7.) and np.array_equal(b.Rs()[0], R[0])          # a clone is deep
    assert [k.idx for k in cams[1:4]] == [1, 2, 3] and [k.idx for k in reversed(cams)] == [5, 4, 3, 2, 1, 0]
    d = copy.deepcopy(b)
    p = pickle.loads(pickle.dumps(b.cameras))
    assert type(p) is list and np.array_equal(p[2].t, b.ts()[2]) and np.array_equal(d.ts(), b.ts())
    # entries that move: everything becomes an object first
    e = b.clone_params()
    last = e.cameras.pop()
    assert last.idx == 5 and len(e.cameras) == 5 and np.array_equal(e.Rs(), b.Rs()[:5])
    e.cameras.insert(0, last)
    assert np.array_equal(e.ts()[0], b.ts()[5]) and np.array_equal(e.ts()[1:], b.ts()[:5])
    del e.cameras[0]
    e.add_camera(Camera(np.eye(3), np.array([1., 1., 1.])))
    assert len(e.cameras) == 6 and e.cameras[5].idx == 5 and np.array_equal(e.ts()[5], [1., 1., 1.])
    assert e.cameras[3] in e.cameras and e.cameras.index(e.cameras[3]) == 3
    e.cameras[1:3] = [Camera(np.eye(3), np.zeros(3)), Camera(np.eye(3), np.ones(3))]
    assert np.array_equal(e.ts()[2], np.ones(3)) and np.array_equal(e.ts()[3], b.ts()[3])


def test_set_poses_of_stacked_cameras():
    rs = np.random.RandomState(4)
    R = np.array([np.linalg.qr(rs.randn(3, 3))[0] for _ in range(5)])
    t = rs.randn(5, 3)
    b = Bundle.FromObservations(np.eye(3), R, t, np.ones((2, 3)), [0, 1, 2, 3], [0, 0, 1, 1], np.zeros((4, 2)))
    held = b.cameras[3]                                  # an object that has been handed out
    newR, newt = np.array([np.eye(3), 2 * np.eye(3)]), np.array([[1., 2., 3.], [4., 5., 6.]])
    b.cameras.set_poses(np.array([1, 3]), newR, newt)
    assert np.array_equal(b.cameras[1].t, [1., 2., 3.]) and np.array_equal(held.t, [4., 5., 6.]) and np.array_equal(held.R, 2 * np.eye(3))
    assert np.array_equal(b.ts()[[0, 2, 4]], t[[0, 2, 4]]) and np.array_equal(b.Rs()[1], np.eye(3))
    newt[0, 0] = 99.                                     # the bundle keeps copies
    assert b.cameras[1].t[0] == 1. and b.ts()[1, 0] == 1.
    b.add_camera(Camera(np.eye(3), np.zeros(3)))
    b.cameras.set_poses(np.array([5, 0]), newR, newt)
    assert np.array_equal(b.ts()[5], newt[0]) and np.array_equal(b.ts()[0], newt[1])


# ------------------------------------------------------------------ the internal camera order (csrc/ba_order.hip; no GPU needed)
def _order_cameras(nco, lists):
    import ctypes as C
    from pysfm_amd import _capi as capi
    lib = capi.load()
    off = np.zeros(len(lists) + 1, np.int32)
    off[1:] = np.cumsum([len(l) for l in lists])
    pos = np.concatenate([np.asarray(l, np.int32) for l in lists]).astype(np.int32) if lists else np.zeros(0, np.int32)
    new = np.empty(nco, np.int32)
    hb = C.c_int32()
    rc = lib.ba_order_cameras(nco, len(lists), capi.iptr(off), capi.iptr(pos), capi.iptr(new), C.cast(C.byref(hb), capi._ip))
    assert rc == capi.BA_OK
    assert sorted(new.tolist()) == list(range(nco))                  # a permutation
    spread = max([int(new[l].max() - new[l].min()) for l in map(np.asarray, lists) if len(l)] + [0])
    assert spread == hb.value
    return new, hb.value


@pytest.mark.parametrize('nco,L,drop,seed', [(60, 10, 0., 1), (400, 10, 0., 2), (400, 10, .3, 3), (300, 16, .1, 4), (150, 3, 0., 5), (1000, 10, .2, 6)])
def test_camera_order_recovers_the_band_of_a_shuffled_sequence(nco, L, drop, seed):
    """Tracks of L consecutive cameras (some observations missing), camera positions shuffled: the caller's order spreads a
    track over most of the sequence, Cuthill-McKee on the co-visibility hypergraph finds an order as narrow as the sequence's own."""
    rs = np.random.RandomState(seed)
    perm = rs.permutation(nco)
    lists, true_hb = [], 0
    for k in range(6 * nco):
        c0 = rs.randint(0, nco - L + 1)
        cams = np.arange(c0, c0 + L)
        cams = cams[rs.rand(L) >= drop]
        if len(cams) < 2:
            continue
        true_hb = max(true_hb, int(cams.max() - cams.min()))
        lists.append(perm[cams])
    caller_hb = max(int(l.max() - l.min()) for l in lists)
    new, hb = _order_cameras(nco, lists)
    assert caller_hb > nco // 2
    assert hb <= true_hb + (0 if drop == 0. else 2), (hb, true_hb)


def test_camera_order_components_and_unseen_cameras():
    """Two sequences that share no track and a few cameras nobody sees: every component gets a stretch of its own, unseen cameras
    go to the end in their own order."""
    rs = np.random.RandomState(9)
    nco = 90
    perm = rs.permutation(nco)
    lists = [perm[np.arange(c, c + 5)] for c in range(0, 36)] + [perm[np.arange(c, c + 5)] for c in range(45, 76)]      # cameras 0..39 and 45..79; 40..44, 80..89 unseen
    new, hb = _order_cameras(nco, lists)
    assert hb == 4
    a, b = np.sort(new[perm[np.arange(0, 40)]]), np.sort(new[perm[np.arange(45, 80)]])
    assert a[-1] - a[0] == 39 and b[-1] - b[0] == 34                 # contiguous stretches
    unseen = perm[np.r_[40:45, 80:90]]
    assert np.all(new[unseen] >= 75) and np.all(np.diff(new[np.sort(unseen)]) > 0)
    # degenerate inputs
    new, hb = _order_cameras(5, [])
    assert new.tolist() == [0, 1, 2, 3, 4] and hb == 0
    new, hb = _order_cameras(1, [[0]])
    assert new.tolist() == [0] and hb == 0


def _plan_layout(nco, lists, points=None, allow_border=True):
    import ctypes as C
    from pysfm_amd import _capi as capi
    lib = capi.load()
    off = np.zeros(len(lists) + 1, np.int32)
    off[1:] = np.cumsum([len(l) for l in lists])
    pos = np.concatenate([np.asarray(l, np.int32) for l in lists]).astype(np.int32)
    pts = None if points is None else np.asarray(points, np.int32)
    new = np.empty(nco, np.int32)
    n1, hb = C.c_int32(), C.c_int32()
    rc = lib.ba_plan_camera_layout(nco, len(lists), capi.iptr(off), capi.iptr(pos), capi.iptr(pts), int(allow_border), capi.iptr(new),
                                   C.cast(C.byref(n1), capi._ip), C.cast(C.byref(hb), capi._ip))
    assert rc == capi.BA_OK and sorted(new.tolist()) == list(range(nco))
    # the band the layout leaves: the spread of the band cameras of every list
    band = 0
    for l in lists:
        p = new[np.asarray(l)]
        p = p[p < n1.value]
        if len(p):
            band = max(band, int(p.max() - p.min()))
    assert band <= hb.value
    return new, n1.value, hb.value


def _sequence_lists(nco, L, per_list, rs):
    lists, pts = [], []
    for c0 in range(0, nco - L + 1):
        lists.append(np.arange(c0, c0 + L))
        pts.append(per_list)
    return lists, pts


@pytest.mark.parametrize('shuffle', [False, True])
def test_camera_layout_puts_loop_closures_into_a_border(shuffle):
    """A sequence of 400 cameras (tracks of 10, a hundred points a camera list) plus 8 single-point tracks that tie camera i to camera
    i + 200: the far ends become the border, the band of the others is the sequence's own - also when the cameras come in no
    particular order (the weak ties are left out of the ordering, or the breadth-first walk folds the loop into the band)."""
    rs = np.random.RandomState(3)
    nco, L = 400, 10
    lists, pts = _sequence_lists(nco, L, 100, rs)
    ties = [(int(i), int(i) + 200) for i in rs.choice(150, 8, replace=False) + 5]
    for a, b in ties:
        lists.append(np.array([a, b])); pts.append(1)
    perm = rs.permutation(nco) if shuffle else np.arange(nco)
    lists = [perm[l] for l in lists]
    new, n1, hb = _plan_layout(nco, lists, pts)
    assert hb == L - 1 and nco - n1 == 8, (hb, n1)
    # one camera of every tie is in the border
    for a, b in ties:
        assert (new[perm[a]] >= n1) != (new[perm[b]] >= n1)
    # without the border the band is what the ties make it
    new0, n10, hb0 = _plan_layout(nco, lists, pts, allow_border=False)
    assert n10 == nco and hb0 > 20


def test_camera_layout_leaves_a_sequence_alone_and_cuts_a_ring_once():
    rs = np.random.RandomState(4)
    nco, L = 300, 8
    lists, pts = _sequence_lists(nco, L, 50, rs)
    new, n1, hb = _plan_layout(nco, lists, pts)
    assert n1 == nco and hb == L - 1 and np.array_equal(new, np.arange(nco))      # the caller's order wins: nothing moves
    # a ring: 40 tracks see the first four and the last four cameras
    ring = lists + [np.r_[0:4, nco - 4:nco]] * 1
    new, n1, hb = _plan_layout(nco, ring, pts + [40])
    assert hb <= 11 and nco - n1 == 4, (hb, n1)
    side = new[np.r_[0:4]] >= n1
    assert side.all() or (new[np.r_[nco - 4:nco]] >= n1).all()


def test_camera_layout_leaves_an_unordered_collection_as_it_comes():
    """A photo collection (every camera shares tracks with cameras drawn at random from all the others) has no band under any order:
    Cuthill-McKee comes out a few per cent narrower than the caller's numbering, far beyond every band solver - not worth setting
    the problem up a second time (the solver of such scenes, conjugate gradients over the blocks the tracks define, does not care
    about the order).  The planner keeps the caller's order; a renumbered SEQUENCE is still found."""
    from pysfm_amd import synthetic_data as sd
    nco = 600
    s = sd.generate_collection_scene(nco + 1, 12000, partners=8, track_len=3)
    cams = s['obs_cam'].reshape(-1, 3)
    lists = [np.sort(c[c > 0]) - 1 for c in cams]                 # optimised positions (camera 0 is the gauge camera)
    lists = [l for l in lists if len(l) >= 2]
    new, n1, hb = _plan_layout(nco, lists, [1] * len(lists))
    assert n1 == nco and np.array_equal(new, np.arange(nco)) and hb > nco // 2, (n1, hb)
    rs = np.random.RandomState(1)
    seq, pts = _sequence_lists(nco, 10, 60, rs)
    perm = rs.permutation(nco)
    new, n1, hb = _plan_layout(nco, [perm[l] for l in seq], pts)
    assert hb == 9 and n1 == nco and not np.array_equal(new, np.arange(nco))


# ------------------------------------------------------------------ the bench line the driver parses
def test_bench_headline_of_a_full_record_is_short_and_keeps_the_contract():
    """Round 5's last stdout line was the whole 31 KB record and the driver could not parse it.  `bench.headline` reduces a full
    record (here: the committed one of that very run) to the line the driver reads: at most 4 KB, the contract's keys, `roofline`
    and `cpu_baseline` with their numbers; a record with everything the flags can add stays under the limit too."""
    import json
    import os
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import bench
    full = json.loads(open(os.path.join(ROOT, 'profiles', 'r05f_bench_default.json')).read().strip().splitlines()[-1])
    assert len(json.dumps(full)) > 20000
    h = bench.headline(full, os.path.join(ROOT, 'bench_detail.json'))
    line = json.dumps(h)
    assert len(line) <= bench.HEADLINE_MAX_BYTES == 4096, len(line)
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data'):
        assert h[k] == full[k], k
    assert h['config']['workload'] == full['config']['workload'] and 'model' not in h['config']
    assert h['config']['cameras'] == 1000 and h['config']['points'] == 100000 and h['config']['observations'] == 1000000
    r = h['roofline']
    for k in ('bound', 'limited_by', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'avg_launch_ms', 'algorithmic_bytes_per_launch', 'flops', 'hbm_frac'):
        assert r[k] == full['roofline'][k], k
    assert r['kernel'] == 'k_bcr_eliminate_fused' and 'per_kernel' not in r
    c = h['cpu_baseline']
    assert c['value'] == full['cpu_baseline']['value'] and c['kind'] == 'port' and c['cores'] >= 1 and c['unit'] == 'obs/s' and len(c['sample']) <= 320
    assert h['final_reproj_rmse'] == full['final_reproj_rmse'] and h['final_reproj_rmse_oracle'] == full['final_reproj_rmse_oracle']
    assert h['detail'] == 'bench_detail.json'
    for k in ('other_configs', 'small_problems', 'problem_info', 'roofline_linearise_schur_pass'):
        assert k not in h
    # the worst case: long option lists, a long workload string, many timers
    fat = dict(full)
    fat['config'] = dict(full['config'], workload=full['config']['workload'] * 6, library_options=['solver=bcr1'] * 40)
    fat['kernel_ms_per_step'] = {'kernel_number_%d' % i: .001 * (i + 1) for i in range(80)}
    assert len(json.dumps(bench.headline(fat, None))) <= 4096
    assert bench.headline({'error': 'a rank gave up'}) == {'error': 'a rank gave up'}


def test_a_timed_out_solve_falls_back_to_the_lu_solver_with_or_without_a_border():
    """Round-5 ADVICE: the time-out fallback of HipBackend.solve_reduced sets option solver = lu and solves again - which
    ba_solve_reduced used to refuse for a problem with border cameras (BA_ERR_STATE: a time-out of the one-launch cyclic reduction
    on a bordered scene aborted the optimisation).  Since round 6 the C library solves band + border by LU too (border_solve_lu,
    tests/test_gpu_configs.py), so the fallback is ONE path: warn, solver = lu, solve, the caller's option back - and no host
    solve anywhere.  No GPU: the C library is a double that injects the time-out status."""
    import ctypes as C
    from pysfm_amd import backend as B
    from pysfm_amd._capi import SOLVE_TIMED_OUT
    calls = []

    class Lib(object):
        def __init__(self):
            self.solver = b'auto'

        def ba_solve_reduced(self, h, mask, info):
            calls.append(('solve', self.solver))
            info._obj.value = SOLVE_TIMED_OUT if self.solver != b'lu' else 0
            return 0

        def ba_set_option(self, h, name, value):
            calls.append(('option', name, value))
            if name == b'solver':
                self.solver = value
            return 0

        def ba_last_solve_kind(self, h):
            return 5

    for border in (0, 3):
        del calls[:]
        be = B.HipBackend.__new__(B.HipBackend)
        be._lib, be._h, be.nco, be.device_lu, be._options = Lib(), None, 4, True, {}
        be.problem_info = lambda: {'border_cameras': border}
        be._check = lambda rc: None
        with pytest.warns(RuntimeWarning, match='timed out'):
            be.solve_reduced(None)
        assert calls == [('solve', b'auto'), ('option', b'solver', b'lu'), ('solve', b'lu'), ('option', b'solver', b'auto')]
        assert be.last_solve_kind == 'bcr_lu' and be.last_solve_path == 'lu'
    assert not hasattr(B.HipBackend, '_host_lu_of_bordered_system')
