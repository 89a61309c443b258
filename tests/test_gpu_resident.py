"""The resident Levenberg-Marquardt loop (csrc/ba_resident.h: optimize() / step() of a small problem as ONE launch of a few
workgroups) against the oracle, the reference's golden trajectories, and the Python loop over ba_lm_trial.  Needs a real
MI355X: run with ``-m gpu``.

Tolerances: the reduced system of a trial 1e-11 of its largest entry, its solution 1e-8, cost trajectories and final
parameters 1e-6 (the north-star tolerance) against the oracle / the goldens; against the Python loop over the general
kernels (same arithmetic, other summation orders) 1e-9, with IDENTICAL accept / reject decisions.
"""
import numpy as np
import pytest

from conftest import load_golden
from oracle import ba_oracle as O

pytestmark = pytest.mark.gpu

LM = 1e-6


def close(a, b, rtol, atol=0.):
    a, b = np.asarray(a, float), np.asarray(b, float)
    assert a.shape == b.shape, (a.shape, b.shape)
    if not b.size:
        return
    scale = np.max(np.abs(b))
    err = np.max(np.abs(a - b))
    assert err <= rtol * scale + atol, 'max abs err %.3e vs scale %.3e (rtol %.1e)' % (err, scale, rtol)


def model_of(sensor):
    from pysfm_amd import sensor_model
    if sensor.kind == O.GAUSS:
        m = sensor_model.GaussianModel(1.)
        m.L = np.asarray(sensor.L, float).reshape(2, 2)
        return m
    if sensor.kind == O.CAUCHY:
        return sensor_model.CauchyModel(float(sensor.sigma))
    return sensor_model.HuberModel(float(sensor.k))


def small_scene(nc, nt, L, seed, sensor, outliers=0., perturbation=.03, ragged=False, shuffle=False):
    from pysfm_amd import Bundle, synthetic_data as sd
    s = sd.generate_banded_scene(nc, nt, track_len=L, seed=seed, msm_noise=.01, init_perturbation=perturbation, outlier_frac=outliers)
    cam, pt, z = s['obs_cam'], s['obs_pt'], s['obs_z']
    rs = np.random.RandomState(seed + 1)
    if ragged:                                  # tracks of different lengths: drop a third of the observations, keep two per track
        keep = rs.rand(len(cam)) > .33
        first = np.concatenate(([True], pt[1:] != pt[:-1]))
        keep |= first | np.concatenate(([False], first[:-1]))
        cam, pt, z = cam[keep], pt[keep], z[keep]
    if shuffle:
        o = rs.permutation(len(cam))
        cam, pt, z = cam[o], pt[o], z[o]
    arrays = (s['K'], s['R0'], s['t0'], s['X0'], cam, pt, z)
    return Bundle.FromObservations(*arrays, sensor_model=model_of(sensor)), arrays


def run(bundle, resident, **kw):
    from pysfm_amd import BundleAdjuster
    opt = {k: kw.pop(k) for k in ('max_steps', 'init_damping') if k in kw}
    ba = BundleAdjuster(verbose=False)
    ba.resident = resident
    ba.set_bundle(bundle, **kw)
    assert ba._resident_applies(None) == resident
    ba.optimize(**opt)
    return ba


def same_walk(a, r, rtol=1e-9):
    """Adjuster r (resident) took the walk of adjuster a (Python loop over ba_lm_trial)."""
    assert [(d, o) for d, o, _ in a.trial_log] == [(d, o) for d, o, _ in r.trial_log]
    assert (a.num_steps, a.converged, a.lm_trials) == (r.num_steps, r.converged, r.lm_trials)
    close(r.costs, a.costs, rtol)
    # (a rejected trial is an overshoot at a damping too small for the step: its cost carries the conditioning of that solve)
    for (_, o, ca), (_, _, cr) in zip(a.trial_log, r.trial_log):
        close([cr], [ca], rtol if o == 'accepted' else 1e-2)
    assert r._damping == a._damping
    ba, br = a.bundle, r.bundle
    close(br.Rs(), ba.Rs(), 1e-8)
    close(br.ts(), ba.ts(), 1e-8, 1e-10)
    close(br.reconstruction, ba.reconstruction, 1e-8)


SENSORS = [O.Sensor.gaussian(1.), O.Sensor.cauchy(.05), O.Sensor.huber(.06), O.Sensor.gaussian(np.array([[2., .3], [.3, 1.5]]))]


@pytest.mark.parametrize('nc,nt,L,kw', [(5, 50, 5, {}), (10, 100, 10, {}), (10, 37, 4, dict(ragged=True)), (8, 200, 6, dict(shuffle=True)),
                                        (4, 300, 4, {}), (11, 512, 11, dict(ragged=True, shuffle=True)), (9, 1024, 7, {}), (14, 120, 12, {}), (17, 256, 16, dict(ragged=True)),
                                        (12, 90, 5, dict(shuffle=True))])
@pytest.mark.parametrize('si', range(len(SENSORS)))
def test_resident_loop_takes_the_walk_of_the_python_loop(nc, nt, L, kw, si):
    sensor = SENSORS[si]
    b, _ = small_scene(nc, nt, L, 100 * nc + nt + si, sensor, outliers=0. if sensor.kind == O.GAUSS else .05, **kw)
    same_walk(run(b, False), run(b, True))


def test_resident_loop_vs_oracle_trajectory():
    for sensor, seed in ((O.Sensor.gaussian(1.), 3), (O.Sensor.cauchy(.05), 4), (O.Sensor.huber(.06), 5)):
        b, arrays = small_scene(6, 60, 6, seed, sensor, outliers=0. if sensor.kind == O.GAUSS else .05)
        flags = (np.arange(6, dtype=np.int32) - 1, np.ones(60, np.uint8))
        ref = O.lm_optimize(sensor, *arrays, *flags, max_steps=12)
        r = run(b, True, max_steps=12)
        assert r.num_steps == ref['num_steps'] and r.converged == ref['converged']
        close(r.costs, ref['costs'], LM)
        out = r.bundle
        close(out.Rs(), ref['R'], LM)
        close(out.ts(), ref['t'], LM, 1e-9)
        close(out.reconstruction, ref['X'], LM)


@pytest.mark.parametrize('name,steps', [('scene_4x10_cauchy', 10), ('scene_5x50_gauss', 5), ('scene_planar_lm', 50)])
def test_resident_loop_walks_the_golden_trajectories(name, steps):
    from pysfm_amd import Bundle, sensor_model
    g = load_golden(name)
    if int(g['sensor_kind']) == 0:
        m = sensor_model.GaussianModel(1.)
        m.L = g['sensor_L']
    else:
        m = sensor_model.CauchyModel(float(g['sensor_sigma']))
    b0 = Bundle.FromObservations(g['K'], g['R'], g['t'], g['X'], g['obs_cam'], g['obs_pt'], g['obs_z'], sensor_model=m)
    for resident in (True, False):              # (both loops are pinned to the reference, whichever optimize() would pick)
        ba = run(b0, resident, max_steps=steps)
        assert ba.num_steps == int(g['lm_num_steps']) and ba.converged == bool(g['lm_converged'])
        close(ba.costs, g['lm_costs'], LM)
        out = ba.bundle
        close(out.Rs(), g['lm_R'], LM)
        close(out.ts(), g['lm_t'], LM, 1e-9)
        close(out.reconstruction, g['lm_X'], LM)


def test_first_trial_reduced_system_and_solution_vs_oracle():
    from pysfm_amd import BundleAdjuster
    for sensor, kw in ((O.Sensor.gaussian(1.), {}), (O.Sensor.cauchy(.05), dict(ragged=True, shuffle=True))):
        b, arrays = small_scene(9, 150, 7, 11, sensor, outliers=0. if sensor.kind == O.GAUSS else .05, **kw)
        ba = BundleAdjuster(verbose=False)
        ba.backend.set_option('solve_trace', '1')
        try:
            ba.set_bundle(b)
            log = ba.backend.lm_resident(1, 0, False, False, 7.5, 1e-4, ba.SCHUR_COMPLIMENT_PINV_THRESHOLD, None)
            S, rhs, dC = ba.backend.lm_resident_debug()
        finally:
            ba.backend.set_option('solve_trace', '0')
        flags = (np.arange(9, dtype=np.int32) - 1, np.ones(150, np.uint8))
        mu, su, parts = O.compute_update(sensor, *arrays, *flags, damping=7.5, return_parts=True)
        So, bo = O.flatten_reduced(parts['S'], parts['b'])
        close(S, So, 1e-11)
        close(rhs, bo, 1e-11)
        close(-dC, mu.reshape(-1), 1e-8)
        assert log.ntrials == 1 and log.have_cost0
        close([log.cost0], [O.cost(sensor, *arrays, *flags)], 1e-12)
        R2, t2, X2 = O.apply_update(arrays[1], arrays[2], arrays[3], mu, su, *flags)
        close([log.trial_cost[0]], [O.cost(sensor, arrays[0], R2, t2, X2, *arrays[4:], *flags)], 1e-8)


def test_masks_and_frozen_cameras_and_tracks():
    sensor = O.Sensor.cauchy(.05)
    b, _ = small_scene(12, 120, 8, 21, sensor, outliers=.05)
    # a window of 9 of the 12 cameras, two of them frozen, every third track frozen (bundle_adjuster.py:54-114)
    kw = dict(camera_ids=list(range(2, 11)), track_ids=list(range(10, 110)), camera_mask=[3, 4, 6, 7, 8, 9, 10],
              track_mask=[j for j in range(10, 110) if j % 3])
    same_walk(run(b, False, **kw), run(b, True, **kw))
    # tracks of 16 observations (the longest a lane group takes), six of the sixteen cameras frozen
    b, _ = small_scene(16, 70, 16, 22, sensor, outliers=.05)
    kw = dict(camera_mask=list(range(6, 16)))
    same_walk(run(b, False, **kw), run(b, True, **kw))


def test_a_mask_over_camera_parameters():
    """solve_motion_normal_eqns deletes masked camera parameters from the system (bundle_adjuster.py:290-299): their update is zero."""
    from pysfm_amd import BundleAdjuster
    b, _ = small_scene(7, 80, 6, 35, O.Sensor.cauchy(.05), outliers=.05)
    m = np.ones(6 * 6 + 3 * 80, bool)
    m[[0, 1, 2, 9, 17, 30, 35]] = False               # camera 1's rotation, single parameters elsewhere
    out = []
    for resident in (False, True):
        ba = BundleAdjuster(verbose=False)
        ba.resident = resident
        ba.set_bundle(b)
        ba.optimize(param_mask=m, max_steps=8)
        out.append(ba)
    same_walk(out[0], out[1])
    assert np.array_equal(out[1].bundle.cameras[1].R, b.cameras[1].R)          # a rotation that was not allowed to move


def test_step_by_step_equals_optimize():
    from pysfm_amd import BundleAdjuster
    b, _ = small_scene(7, 80, 6, 33, O.Sensor.gaussian(1.))
    whole = run(b, True, max_steps=9)
    ba = BundleAdjuster(verbose=False)
    ba.set_bundle(b)
    ba.num_steps, ba.converged, ba.costs, ba.trial_log, ba.lm_trials = 0, False, [], [], 0
    ba._damping = 10.
    while not ba.converged and ba.num_steps < 9:
        ba.step()
    assert ba.trial_log == whole.trial_log and ba.costs == whole.costs and ba.num_steps == whole.num_steps
    assert np.array_equal(ba.bundle.reconstruction, whole.bundle.reconstruction)


def test_a_trial_the_resident_loop_cannot_take_goes_through_the_general_path():
    from pysfm_amd import BundleAdjuster
    # next to no damping from the start: the reduced system of a monocular reconstruction is singular along its gauge up to
    # that damping - Cholesky stops at a pivot <= 0 or passes on round-off, the reference's gesv solves it either way.
    # (Exactly zero is the reference's own trap: an ill-conditioned trial multiplies the damping by ten, bundle_adjuster.py:141.)
    b, _ = small_scene(6, 40, 6, 8, O.Sensor.gaussian(1.))
    a, r = run(b, False, init_damping=1e-13), run(b, True, init_damping=1e-13)
    assert [(d, o) for d, o, _ in a.trial_log] == [(d, o) for d, o, _ in r.trial_log]
    close(r.costs[-1:], a.costs[-1:], 1e-6)
    assert r.backend.lm_resident_fits()
    # plain-inverse mode with a point that two cameras on one line of sight cannot fix: the reference raises LinAlgError
    # from numpy.linalg.inv (bundle_adjuster.py:254) - so do both loops
    from pysfm_amd import Bundle, sensor_model
    K = np.eye(3)
    R = np.array([np.eye(3)] * 3)
    t = np.array([[0., 0, 0], [0, 0, 1.], [.5, 0, 0]])
    X = np.array([[0., 0, 5.], [1., .5, 6.], [-.7, .2, 4.], [.3, -.4, 5.5]])
    cam = np.array([0, 1, 0, 1, 2, 0, 1, 2, 0, 1, 2], np.int32)          # point 0 lies on the line through cameras 0 and 1
    pt = np.array([0, 0, 1, 1, 1, 2, 2, 2, 3, 3, 3], np.int32)
    z = np.array([(R[c] @ X[p] + t[c])[:2] / (R[c] @ X[p] + t[c])[2] for c, p in zip(cam, pt)])
    bundle = Bundle.FromObservations(K, R, t, X, cam, pt, z, sensor_model=sensor_model.GaussianModel(1.))
    for resident in (False, True):
        ba = BundleAdjuster(verbose=False)
        ba.resident = resident
        ba.SCHUR_COMPLIMENT_PINV_THRESHOLD = None
        ba.set_bundle(bundle)
        with pytest.raises(np.linalg.LinAlgError):
            ba.optimize(max_steps=2)


def test_what_is_not_a_resident_problem_takes_the_python_loop():
    from pysfm_amd import BundleAdjuster, sensor_model
    from conftest import GemanMcClure
    b, _ = small_scene(18, 60, 12, 2, O.Sensor.gaussian(1.))
    ba = BundleAdjuster(b, verbose=False)                                  # 17 optimised cameras: one more than the loop takes
    assert not ba._resident_applies(None)
    ba.set_bundle(b, camera_ids=list(range(8)))
    assert ba._resident_applies(None)
    b.sensor_model = GemanMcClure(.3)                                      # a caller-defined model travels as a table: its own instance of the loop
    kw = dict(camera_ids=list(range(8)))
    same_walk(run(b, False, max_steps=6, **kw), run(b, True, max_steps=6, **kw), rtol=1e-8)
    b.sensor_model = sensor_model.GaussianModel(1.)
    ba.backend.set_option('resident', '0')
    try:
        ba.set_bundle(b, camera_ids=list(range(8)))
        assert not ba._resident_applies(None)
    finally:
        ba.backend.set_option('resident', '1')


@pytest.mark.parametrize('nc,nt,L', [(10, 40, 5), (10, 100, 10), (14, 120, 12), (17, 256, 16), (9, 1024, 4)])
def test_both_forms_of_the_exchange_end_on_the_same_bits(nc, nt, L):
    """Exchange 1 of the resident loop (csrc/ba_resident.h): every workgroup adds up every record, or - from five workgroups
    on - each adds up a slice of all records and everybody fetches the sums.  The same numbers in the same order either way."""
    from pysfm_amd import BundleAdjuster
    b, _ = small_scene(nc, nt, L, 7, O.Sensor.cauchy(.05), outliers=.05, ragged=True)
    out = []
    for scatter_min in ('1000', '2'):
        ba = BundleAdjuster(verbose=False)
        ba.backend.set_option('resident_scatter_min', scatter_min)
        ba.set_bundle(b)
        assert ba._resident_applies(None)
        ba.optimize(max_steps=6)
        out.append((ba.trial_log, ba.costs, ba.bundle.Rs(), ba.bundle.ts(), np.asarray(ba.bundle.reconstruction)))
    assert out[0][0] == out[1][0] and out[0][1] == out[1][1]
    for x, y in zip(out[0][2:], out[1][2:]):
        assert np.array_equal(x, y)


def test_host_side_and_device_side_set_up_agree():
    """ba_set_problem orders small problems on the host (csrc/ba_problem.hip host_front_end): the same internal order,
    the same work lists, the same numbers as the device pipeline."""
    from pysfm_amd import BundleAdjuster
    for kw in (dict(), dict(ragged=True), dict(shuffle=True), dict(ragged=True, shuffle=True)):
        b, _ = small_scene(12, 300, 9, 17, O.Sensor.cauchy(.05), outliers=.05, **kw)
        out = []
        for host in ('1', '0'):
            ba = BundleAdjuster(verbose=False)
            ba.resident = False
            ba.backend.set_option('host_setup', host)
            try:
                ba.set_bundle(b)
                info = ba.backend.problem_info()
                ba.optimize(max_steps=4)
            finally:
                ba.backend.set_option('host_setup', '1')
            out.append((info, ba.costs, ba.trial_log, ba.bundle.reconstruction.copy()))
        assert out[0][0] == out[1][0]                       # the same groups, windows, band, permutation flags
        # (the general kernels add into [S | b] with atomics: equal to round-off, not bit for bit, even run to run)
        assert [(d, o) for d, o, _ in out[0][2]] == [(d, o) for d, o, _ in out[1][2]]
        close(out[0][1], out[1][1], 1e-10)
        close(out[0][3], out[1][3], 1e-9)
    # and the same refusals
    from pysfm_amd.backend import HipBackend
    be = HipBackend(0)
    try:
        K, pos, opt = np.eye(3), np.array([-1, 0, 1], np.int32), np.ones(2, np.uint8)
        for cam, pt, what in (([0, 1, 3], [0, 0, 1], 'obs_cam[2]=3'), ([0, 1, 2], [0, 0, 2], 'obs_pt[2]=2'), ([0, 1, 1, 2], [0, 0, 0, 1], 'track 0 has two')):
            for host in ('1', '0'):
                be.set_option('host_setup', host)
                with pytest.raises(ValueError, match=what.replace('[', r'\[').replace(']', r'\]')):
                    be.set_problem(3, 2, np.array(cam, np.int32), np.array(pt, np.int32), np.zeros((len(cam), 2)), K, pos, opt)
    finally:
        be.close()


def test_window_slam_same_result_with_either_loop():
    from pysfm_amd import Bundle, BundleAdjuster, sensor_model, window_slam
    g = load_golden('scene_oleg_100x1000')
    b = Bundle.FromObservations(g['K'], g['R'].reshape(-1, 3, 3), g['t'], g['X'], g['obs_cam'], g['obs_pt'], g['obs_z'],
                                sensor_model=sensor_model.GaussianModel(1.))
    res = []
    try:
        for resident in (True, False):
            BundleAdjuster.resident = resident
            out, hist = window_slam.run(b, 10, num_tracks=100, max_steps=6, verbose=False)
            res.append((out, hist))
    finally:
        BundleAdjuster.resident = True
    (o1, h1), (o2, h2) = res
    assert [len(h) for h in h1] == [len(h) for h in h2]
    close(np.concatenate(h1), np.concatenate(h2), 1e-8)
    close(o1.Rs(), o2.Rs(), 1e-7)
    close(o1.ts(), o2.ts(), 1e-7, 1e-9)
    close(o1.reconstruction, o2.reconstruction, 1e-7)


def test_window_slam_with_the_next_window_set_up_during_this_one():
    """window_slam.run(overlap=True): two adjusters taking turns, window i + 1's problem set up while window i's loop runs on
    the device (optimize_begin / optimize_end), its values when window i is back.  The same bits as one adjuster doing
    everything in turn - with the loop on the device, and with the Python loop (where optimize_end does all the work)."""
    from pysfm_amd import Bundle, BundleAdjuster, sensor_model, window_slam
    g = load_golden('scene_oleg_100x1000')
    b = Bundle.FromObservations(g['K'], g['R'].reshape(-1, 3, 3), g['t'], g['X'], g['obs_cam'], g['obs_pt'], g['obs_z'],
                                sensor_model=sensor_model.GaussianModel(1.))
    try:
        for resident in (True, False):
            BundleAdjuster.resident = resident
            seen = []
            o1, h1 = window_slam.run(b, 10, num_tracks=100, max_steps=5, verbose=False, overlap=False)
            o2, h2 = window_slam.run(b, 10, num_tracks=100, max_steps=5, verbose=False, overlap=True,
                                     on_window=lambda i, ba: seen.append((i, ba.num_steps, list(ba.camera_ids))))
            if resident:                              # (every sum of the device loop has a fixed order; the general kernels add with atomics)
                assert h1 == h2
                assert np.array_equal(o1.Rs(), o2.Rs()) and np.array_equal(o1.ts(), o2.ts()) and np.array_equal(o1.reconstruction, o2.reconstruction)
            assert [len(h) for h in h1] == [len(h) for h in h2]
            close(np.concatenate(h1), np.concatenate(h2), 1e-8)
            close(o1.Rs(), o2.Rs(), 1e-7)
            close(o1.ts(), o2.ts(), 1e-7, 1e-9)
            close(o1.reconstruction, o2.reconstruction, 1e-7)
            assert [i for i, _, _ in seen] == list(range(91)) and all(c == list(range(i, i + 10)) for i, _, c in seen)
    finally:
        BundleAdjuster.resident = True


def test_optimize_in_two_halves():
    """optimize_begin / optimize_end = optimize, for a problem the device loop takes (a launch and a wait) and for one it does
    not (optimize_end does everything); with a parameter mask; a trial that has to go through the general path in between."""
    from pysfm_amd import BundleAdjuster
    mask = np.ones(6 * 7 + 3 * 60, bool)
    mask[[2, 9, 17]] = False
    for nc, nt, L, kw in ((10, 100, 10, {}), (20, 200, 10, {}), (8, 60, 6, dict(param_mask=mask))):
        b, _ = small_scene(nc, nt, L, 3, O.Sensor.gaussian(1.))
        a = BundleAdjuster(b, verbose=False)
        a.optimize(max_steps=6, **kw)
        r = BundleAdjuster(b, verbose=False)
        r.optimize_begin(max_steps=6, **kw)
        r.optimize_end()
        assert (a.num_steps, a.converged, a.lm_trials) == (r.num_steps, r.converged, r.lm_trials)
        assert [(d, o) for d, o, _ in a.trial_log] == [(d, o) for d, o, _ in r.trial_log]
        if a._resident_applies(kw.get('param_mask')):      # (every sum of the device loop has a fixed order; the general kernels add with atomics)
            assert a.trial_log == r.trial_log and a.costs == r.costs
            assert np.array_equal(a.bundle.Rs(), r.bundle.Rs()) and np.array_equal(a.bundle.reconstruction, r.bundle.reconstruction)
        close(r.costs, a.costs, 1e-9)
        close(r.bundle.Rs(), a.bundle.Rs(), 1e-8)
        close(r.bundle.reconstruction, a.bundle.reconstruction, 1e-8)
    # a trial the device loop cannot take (next to no damping: the reduced system is singular along its gauge) goes through the
    # general path between two launches, the first of which optimize_begin made
    b6, _ = small_scene(6, 40, 6, 8, O.Sensor.gaussian(1.))
    a, r = BundleAdjuster(b6, verbose=False), BundleAdjuster(b6, verbose=False)
    a.optimize(init_damping=1e-13)
    r.optimize_begin(init_damping=1e-13)
    r.optimize_end()
    assert [(d, o) for d, o, _ in a.trial_log] == [(d, o) for d, o, _ in r.trial_log] and (a.num_steps, a.converged) == (r.num_steps, r.converged)
    close(r.costs[-1:], a.costs[-1:], 1e-6)
    # a launch that was never collected is refused loudly, not silently overwritten
    r = BundleAdjuster(b, verbose=False)
    r.optimize_begin(max_steps=2)
    with pytest.raises(Exception):
        r.backend.lm_resident_begin(2, 0, False, False, 10., 1e-4, r.SCHUR_COMPLIMENT_PINV_THRESHOLD, None)
    r.optimize_end()


def test_degenerate_shapes():
    """One workgroup, one point; tracks nobody in the window sees; a camera that sees nothing; two cameras."""
    from pysfm_amd import Bundle
    sensor = O.Sensor.gaussian(1.)
    # tracks 0 .. 11 of a 6-camera scene, cameras 3 .. 5 only: the first tracks have no observation in the window
    b, _ = small_scene(6, 40, 3, 41, sensor)
    kw = dict(camera_ids=[3, 4, 5], track_ids=list(range(0, 30)))
    a, r = run(b, False, **kw), run(b, True, **kw)
    same_walk(a, r)
    # two cameras, one of them free; a single track
    for nc, nt, L in ((2, 7, 2), (3, 1, 3), (2, 1, 2)):
        b, _ = small_scene(nc, nt, L, 50 + nt, sensor)
        same_walk(run(b, False, max_steps=6), run(b, True, max_steps=6), rtol=1e-7)
    # a camera in the window that observes none of the selected tracks (its block of the reduced system is zero: the
    # reference's solve raises, the damping grows - both loops alike)
    K, R, t, X, cam, pt, z = small_scene(5, 30, 3, 43, sensor)[1]
    keep = cam != 2
    b = Bundle.FromObservations(K, R, t, X, cam[keep], pt[keep], z[keep], sensor_model=model_of(sensor))
    a, r = run(b, False, max_steps=4), run(b, True, max_steps=4)
    assert [(d, o) for d, o, _ in a.trial_log] == [(d, o) for d, o, _ in r.trial_log]
    close(r.costs, a.costs, 1e-8)


@pytest.mark.parametrize('fault', [0, 3, 6])
def test_a_launch_whose_workgroups_disagree_changes_nothing(fault):
    """Round-5 ADVICE: workgroups of the resident loop that lose each other need not agree on how they ended - one passes its last
    wait and ends DONE (it used to write its slice of the current set back) while another's spin budget runs out on the same
    epoch (it wrote nothing): a partially updated set the host believed untouched.  Now every workgroup writes to a staging copy
    and reports (reason, trials); the host commits the copy only when all reports are alike.  Option resident_fault makes ONE
    workgroup report a time-out after a perfectly normal run: the launch must then count as not having happened - the current
    set on the device bit for bit the one it was given - and the Python loop takes over and ends where it ends without the fault."""
    import warnings
    from pysfm_amd import BundleAdjuster
    b, _ = small_scene(10, 100, 10, 77, O.Sensor.cauchy(.05), outliers=.05)
    clean = run(b, False, max_steps=6)
    ba = BundleAdjuster(verbose=False)
    ba.set_bundle(b)
    be = ba.backend
    assert ba._resident_applies(None) and (be.nt + 15) // 16 == 7
    be.set_option('resident_fault', fault)
    before = be.get_params(0)
    log = be.lm_resident(6, 0, False, False, 10., 1e-4, 1e-5, -1., None)
    from pysfm_amd._capi import RESIDENT_TIMED_OUT
    assert log.exit_reason == RESIDENT_TIMED_OUT and log.ntrials == 0 and not log.accepted
    after = be.get_params(0)
    for x, y in zip(before, after):
        assert np.array_equal(x, y)
    # ... and through the adjuster: a warning, the loop over ba_lm_trial from the same start, the same end
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        ba.optimize(max_steps=6)
    assert any('resident loop timed out' in str(x.message) for x in w) and ba.resident_timeouts == 1 and ba.resident is False
    same_walk(clean, ba)
    be.set_option('resident_fault', -1)
    # without the fault the same handle's next launch commits
    ba2 = BundleAdjuster(verbose=False)
    ba2.set_bundle(b)
    ba2.optimize(max_steps=6)
    assert getattr(ba2, 'resident_timeouts', 0) == 0
    same_walk(clean, ba2)
    ba.backend.close(); ba2.backend.close(); clean.backend.close()
