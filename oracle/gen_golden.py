#!/usr/bin/env python3
"""Generate golden vectors from the reference (alexflint/pysfm) itself.

TEST INFRASTRUCTURE ONLY - runs in the build container, never on the GPU box.

The reference is Python-2 source.  This script copies ``/root/reference/*.py``
into a temporary directory OUTSIDE the repository, translates it there with
``lib2to3`` (plus the one ``i/6 -> i//6`` fix the reference's own test needs),
imports it, drives the reference's own scene builders and ``BundleAdjuster``
and stores inputs + outputs as ``tests/golden/*.npz``.  Only data (inputs and
expected outputs) is written to the repository - no reference source, bytecode
or translation of it.

    python oracle/gen_golden.py [--ref /root/reference] [--out tests/golden] [--big]

``--big`` additionally produces the 100-camera x 1000-track spot check
(~1 minute in the reference).
"""
import argparse
import contextlib
import io
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np


# the reference modules on (or next to) the hot path; two unrelated modules of
# the reference (fundamental.py, fmat_uncertainty.py) do not even parse for lib2to3
REF_MODULES = ['algebra', 'lie', 'numpy_test', 'finite_differences', 'sensor_model', 'triangulate',
               'bundle', 'optimize', 'schur', 'bundle_adjuster', 'sequence', 'synthetic_data',
               'bundle_io', 'bundle_unittest', 'test_bundle', 'draw_bundle', 'geometry',
               'window_slam', 'draw_bundle_pca', 'pca']


def import_reference(ref_dir):
    tmp = tempfile.mkdtemp(prefix='pysfm_py3_')
    for f in REF_MODULES:
        shutil.copy(os.path.join(ref_dir, f + '.py'), tmp)
    subprocess.run([sys.executable, '-m', 'lib2to3', '-w', '-n', tmp],
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
    sys.path.insert(0, tmp)
    os.environ.setdefault('MPLBACKEND', 'Agg')
    return tmp


@contextlib.contextmanager
def quiet():
    with contextlib.redirect_stdout(io.StringIO()):
        yield


def bundle_arrays(bundle, camera_ids=None, track_ids=None):
    """Reference Bundle -> SoA in the order the reference's loops visit the
    observations (tracks outer, cameras inner; bundle_adjuster.py:222-226)."""
    camera_ids = list(range(len(bundle.cameras))) if camera_ids is None else list(camera_ids)
    track_ids = list(range(len(bundle.tracks))) if track_ids is None else list(track_ids)
    cam, pt, z = [], [], []
    for jpos, j in enumerate(track_ids):
        tr = bundle.tracks[j]
        for ipos, i in enumerate(camera_ids):
            if tr.has_measurement(i):
                cam.append(ipos)
                pt.append(jpos)
                z.append(np.asarray(tr.get_measurement(i), float))
    return dict(
        K=np.asarray(bundle.K, float),
        R=np.array([bundle.cameras[i].R for i in camera_ids], float),
        t=np.array([bundle.cameras[i].t for i in camera_ids], float),
        X=np.array([bundle.reconstruction[j] for j in track_ids], float),
        obs_cam=np.array(cam, np.int32), obs_pt=np.array(pt, np.int32),
        obs_z=np.array(z, float).reshape(-1, 2),
    )


def sensor_arrays(sm):
    name = type(sm).__name__
    if name == 'GaussianModel':
        return dict(sensor_kind=0, sensor_L=np.asarray(sm.L, float), sensor_sigma=1.0)
    if name == 'CauchyModel':
        return dict(sensor_kind=1, sensor_L=np.eye(2), sensor_sigma=float(sm.sigma))
    raise ValueError(name)


def per_observation(bundle, arrs, camera_ids=None, track_ids=None):
    camera_ids = list(range(len(bundle.cameras))) if camera_ids is None else list(camera_ids)
    track_ids = list(range(len(bundle.tracks))) if track_ids is None else list(track_ids)
    e, r, Jc, Jp = [], [], [], []
    for n in range(len(arrs['obs_cam'])):
        i = camera_ids[arrs['obs_cam'][n]]
        j = track_ids[arrs['obs_pt'][n]]
        e.append(bundle.reproj_error(i, j))
        r.append(bundle.residual(i, j))
        a, b = bundle.Jresidual(i, j)
        Jc.append(a)
        Jp.append(b)
    return dict(e=np.array(e), r=np.array(r), Jc=np.array(Jc), Jp=np.array(Jp))


def adjuster_blocks(ref, bundle, damping, out, tag, camera_ids=None, track_ids=None,
                    camera_mask=None, track_mask=None):
    """prepare -> damp -> schur -> solve -> backsub through the reference."""
    with quiet():
        ba = ref['bundle_adjuster'].BundleAdjuster()
        ba.set_bundle(bundle, camera_ids, track_ids, camera_mask, track_mask)
        ba.prepare_schur_complement()
    arrs = bundle_arrays(bundle, ba.camera_ids, ba.track_ids)
    out[tag + 'HCC'] = ba.HCCs.copy()
    out[tag + 'HPP'] = ba.HPPs.copy()
    out[tag + 'bC'] = ba.bCs.copy()
    out[tag + 'bP'] = ba.bPs.copy()
    out[tag + 'W'] = np.array([ba.HCPs[arrs['obs_cam'][n], arrs['obs_pt'][n]]
                               for n in range(len(arrs['obs_cam']))]).reshape(-1, 6, 3)
    ba.apply_damping(damping)
    S, b = ba.compute_schur_complement()
    out[tag + 'S'] = S
    out[tag + 'b'] = b
    out[tag + 'HPP_inv'] = ba.HPP_invs.copy()
    nco = len(ba.optim_camera_ids)
    try:
        dC = ba.solve_motion_normal_eqns(S, b, np.ones(nco * 6, bool))
        out[tag + 'dC'] = dC
        out[tag + 'dP'] = ba.backsubstitute(dC)
    except ref['bundle_adjuster'].NormalEquationsIllconditioned:
        pass
    cam_opt_pos = -np.ones(len(ba.camera_ids), np.int32)
    for pos, idx in enumerate(ba.optim_camera_indices):
        cam_opt_pos[idx] = pos
    pt_opt = np.zeros(len(ba.track_ids), bool)
    pt_opt[list(ba.optim_track_indices)] = True
    out[tag + 'cam_opt_pos'] = cam_opt_pos
    out[tag + 'pt_opt'] = pt_opt
    out[tag + 'cost'] = ba.compute_cost(bundle)
    return ba, arrs


def traced_optimize(ref, bundle, **kw):
    """Run the reference's optimize() and record every LM trial."""
    BA = ref['bundle_adjuster'].BundleAdjuster
    with quiet():
        ba = BA(bundle)
    trials = []
    orig_update, orig_cost = ba.compute_update, ba.compute_cost
    state = {}

    def upd(damping, param_mask=None):
        mu, su = orig_update(damping, param_mask)
        state['damping'] = damping
        state['mu'] = np.array(mu)
        state['su'] = np.array(su)
        state['await'] = True
        return mu, su

    def cst(b):
        c = orig_cost(b)
        if state.get('await'):
            trials.append((state['damping'], c, np.linalg.norm(state['mu']),
                           np.linalg.norm(state['su'])))
            state['await'] = False
        return c

    ba.compute_update, ba.compute_cost = upd, cst
    with quiet():
        ba.optimize(**kw)
    return ba, np.array(trials, float).reshape(-1, 4)


def save(out_dir, name, d):
    path = os.path.join(out_dir, name + '.npz')
    np.savez_compressed(path, **d)
    print('wrote %s (%d arrays, %.1f KB)' % (path, len(d), os.path.getsize(path) / 1024.))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--ref', default='/root/reference')
    ap.add_argument('--out', default=os.path.join(os.path.dirname(__file__), '..', 'tests', 'golden'))
    ap.add_argument('--big', action='store_true')
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)

    tmp = import_reference(args.ref)
    try:
        import importlib
        names = ['bundle', 'bundle_adjuster', 'sensor_model', 'lie', 'schur', 'optimize',
                 'synthetic_data', 'bundle_io', 'triangulate', 'bundle_unittest', 'test_bundle',
                 'algebra', 'window_slam', 'geometry']
        with quiet():
            ref = {n: importlib.import_module(n) for n in names}
        generate(ref, args)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def generate(ref, args):
    B, SM = ref['bundle'], ref['sensor_model']

    # ---- (0) small function-level vectors: SO3.exp, sensor models -------
    rng = np.random.RandomState(7)
    ms = np.concatenate((rng.randn(20, 3), rng.randn(5, 3) * 1e-9, [[0, 0, 0], [1., 3., -1.]]))
    es = np.concatenate((rng.randn(30, 2), rng.randn(5, 2) * 1e-7, rng.randn(5, 2) * 1e-3,
                         [[0., 0.], [1., 2.]]))
    d = dict(so3_m=ms, so3_R=np.array([ref['lie'].SO3.exp(m) for m in ms]), sens_e=es)
    for tag, sm in (('gauss_iso', SM.GaussianModel(.1)), ('gauss_diag', SM.GaussianModel([2., 3.])),
                    ('gauss_full', SM.GaussianModel(np.array([[2., .3], [.3, 1.]]))),
                    ('cauchy', SM.CauchyModel(.05)), ('cauchy2', SM.CauchyModel(2.))):
        sa = sensor_arrays(sm)
        d[tag + '_kind'] = sa['sensor_kind']
        d[tag + '_L'] = sa['sensor_L']
        d[tag + '_sigma'] = sa['sensor_sigma']
        d[tag + '_r'] = np.array([sm.residual_from_error(e) for e in es])
        d[tag + '_J'] = np.array([sm.Jresidual_from_error(e) for e in es])
        d[tag + '_cost'] = np.array([sm.cost_from_error(e) for e in es])
    save(args.out, 'functions', d)

    # ---- (1) bundle_unittest.create_test_bundle: 4 cams x 10 pts Cauchy --
    with quiet():
        b = ref['bundle_unittest'].create_test_bundle()
    d = bundle_arrays(b)
    d.update(sensor_arrays(b.sensor_model))
    d.update(per_observation(b, d))
    d['complete_cost'] = b.complete_cost()
    for lam, tag in ((0., 'l0_'), (2., 'l2_')):
        adjuster_blocks(ref, b, lam, d, tag)
    # dense oracle of the reference's own test (bundle_adjuster_unittest.py:16-44)
    r = b.residuals()
    J = b.Jresiduals()[:, 6:]
    JTJ, JTr = J.T @ J, J.T @ r
    d['dense_S_l0'], d['dense_b_l0'] = ref['schur'].get_schur_complement(JTJ, JTr, 6 * 3)
    JTJ2 = JTJ.copy()
    ref['optimize'].apply_lm_damping_inplace(JTJ2, 2.)
    d['dense_delta_l2'] = -np.linalg.solve(JTJ2, JTr)
    with quiet():
        ba = ref['bundle_adjuster'].BundleAdjuster(b)
        mu, su = ba.compute_update(2.)
    d['update_l2_motion'], d['update_l2_structure'] = np.array(mu), np.array(su)
    # one accepted LM trajectory on this scene
    ba, trials = traced_optimize(ref, b, max_steps=10)
    d['lm_costs'] = np.array(ba.costs)
    d['lm_trials'] = trials
    d['lm_num_steps'] = ba.num_steps
    d['lm_converged'] = ba.converged
    fin = bundle_arrays(ba.bundle)
    d['lm_R'], d['lm_t'], d['lm_X'] = fin['R'], fin['t'], fin['X']
    save(args.out, 'scene_4x10_cauchy', d)

    # ---- (2) test_subset_schur configuration ------------------------------
    d = {}
    camera_ids, track_ids = [3, 1], [0, 1, 2]
    cam_mask, track_mask = [False, True], [False, True, False]
    ba, arrs = adjuster_blocks(ref, b, 2., d, 'l2_', camera_ids, track_ids, cam_mask, track_mask)
    d.update(arrs)
    d.update(sensor_arrays(b.sensor_model))
    d['camera_ids'], d['track_ids'] = np.array(camera_ids), np.array(track_ids)
    d['cam_mask'], d['track_mask'] = np.array(cam_mask), np.array(track_mask)
    full = bundle_arrays(b)
    d['full_R'], d['full_t'], d['full_X'] = full['R'], full['t'], full['X']
    d['full_obs_cam'], d['full_obs_pt'], d['full_obs_z'] = full['obs_cam'], full['obs_pt'], full['obs_z']
    with quiet():
        ba2 = ref['bundle_adjuster'].BundleAdjuster()
        ba2.set_bundle(b, camera_ids, track_ids, cam_mask, track_mask)
        mu, su = ba2.compute_update(2.)
    d['update_l2_motion'], d['update_l2_structure'] = np.array(mu), np.array(su)
    # integer-id masks (select() int path, bundle_adjuster.py:17-20)
    with quiet():
        ba3 = ref['bundle_adjuster'].BundleAdjuster()
        ba3.set_bundle(b, [0, 2, 3, 1], [5, 4, 7, 8, 1], np.array([1, 3]), np.array([8, 5, 4]))
        mu, su = ba3.compute_update(.5)
        d['int_cost'] = ba3.compute_cost(b)
    d['int_camera_ids'], d['int_track_ids'] = np.array([0, 2, 3, 1]), np.array([5, 4, 7, 8, 1])
    d['int_cam_mask'], d['int_track_mask'] = np.array([1, 3]), np.array([8, 5, 4])
    d['int_update_motion'], d['int_update_structure'] = np.array(mu), np.array(su)
    save(args.out, 'scene_subset', d)

    # ---- (3)/(4) synthetic_data.generate_sequence(5, 50) -------------------
    def seq_bundle(nframes, npts, mask=None, sensor=None):
        with quiet():
            seq = ref['synthetic_data'].generate_sequence(nframes, npts)
        msm = np.array([[np.asarray(tr.measurements[i]) for tr in seq.tracks] for i in range(nframes)])
        bb = B.Bundle.FromArrays(seq.K, seq.initial_Rs, seq.initial_ts, seq.initial_xs, msm, mask)
        if sensor is not None:
            bb.sensor_model = sensor
        return bb

    b5 = seq_bundle(5, 50)
    d = bundle_arrays(b5)
    d.update(sensor_arrays(b5.sensor_model))
    d['complete_cost'] = b5.complete_cost()
    adjuster_blocks(ref, b5, 10., d, 'l10_')
    ba, trials = traced_optimize(ref, b5, max_steps=5)
    d['lm_costs'], d['lm_trials'] = np.array(ba.costs), trials
    d['lm_num_steps'], d['lm_converged'] = ba.num_steps, ba.converged
    fin = bundle_arrays(ba.bundle)
    d['lm_R'], d['lm_t'], d['lm_X'] = fin['R'], fin['t'], fin['X']
    save(args.out, 'scene_5x50_gauss', d)

    np.random.seed(4309)                      # bundle_unittest.py:52-57 pattern, 20 % missing
    mask = np.ones((5, 50), bool)
    for i in range(5):
        mask[i, np.random.permutation(50)[:10]] = False
    b5c = seq_bundle(5, 50, mask, SM.CauchyModel(.05))
    d = bundle_arrays(b5c)
    d.update(sensor_arrays(b5c.sensor_model))
    d['complete_cost'] = b5c.complete_cost()
    adjuster_blocks(ref, b5c, 10., d, 'l10_')
    ba, trials = traced_optimize(ref, b5c, max_steps=8)
    d['lm_costs'], d['lm_trials'] = np.array(ba.costs), trials
    d['lm_num_steps'], d['lm_converged'] = ba.num_steps, ba.converged
    fin = bundle_arrays(ba.bundle)
    d['lm_R'], d['lm_t'], d['lm_X'] = fin['R'], fin['t'], fin['X']
    save(args.out, 'scene_5x50_cauchy_masked', d)

    # ---- (5) test_bundle.create_test_problem(noise=0) -> optimize(50) ------
    with quiet():
        b_true, b_init = ref['test_bundle'].create_test_problem(noise=0)
    d = bundle_arrays(b_init)
    d.update(sensor_arrays(b_init.sensor_model))
    d['complete_cost'] = b_init.complete_cost()
    ba, trials = traced_optimize(ref, b_init, max_steps=50)
    d['lm_costs'], d['lm_trials'] = np.array(ba.costs), trials
    d['lm_num_steps'], d['lm_converged'] = ba.num_steps, ba.converged
    fin = bundle_arrays(ba.bundle)
    d['lm_R'], d['lm_t'], d['lm_X'] = fin['R'], fin['t'], fin['X']
    tru = bundle_arrays(b_true)
    d['true_R'], d['true_t'], d['true_X'] = tru['R'], tru['t'], tru['X']
    # triangulation pin (triangulate.py:6-18): b_init.reconstruction IS the triangulation
    with quiet():
        _, b_pert = ref['test_bundle'].create_test_problem(noise=0)
    d['triangulated_X'] = np.array([b_pert.triangulate(tr) for tr in b_pert.tracks])
    save(args.out, 'scene_planar_lm', d)

    # ---- (6) data/oleg_synthetic subsets, pixel units, K f=1500 -----------
    data = os.path.join(args.ref, 'data', 'oleg_synthetic')
    with quiet():
        bo = ref['bundle_io'].load(os.path.join(data, 'tracks.txt'), os.path.join(data, 'poses.txt'))
    for ncam, ntr in ((10, 50), (40, 100)):
        cams, trs = list(range(ncam)), list(range(ntr))
        with quiet():
            sub = B.Bundle()
            sub.K = bo.K.copy()
            for i in cams:
                sub.add_camera(B.Camera(bo.cameras[i].R.copy(), bo.cameras[i].t.copy()))
            for j in trs:
                tr = bo.tracks[j]
                ids = [i for i in cams if tr.has_measurement(i)]
                sub.add_track(B.Track(ids, [np.asarray(tr.get_measurement(i), float) for i in ids]))
            sub.triangulate_all()
        d = bundle_arrays(sub)
        d.update(sensor_arrays(sub.sensor_model))
        d['complete_cost'] = sub.complete_cost()
        adjuster_blocks(ref, sub, 10., d, 'l10_')
        with quiet():
            ba = ref['bundle_adjuster'].BundleAdjuster(sub)
            mu, su = ba.compute_update(10.)
        d['update_l10_motion'], d['update_l10_structure'] = np.array(mu), np.array(su)
        if ncam == 40:      # keep the fixture small: drop the bulky per-block arrays
            for k in ('l10_W', 'l10_S'):
                d[k + '_fro'] = np.linalg.norm(d.pop(k))
        save(args.out, 'scene_oleg_%dx%d' % (ncam, ntr), d)
    # first lines of the on-disk format, as data, for the loader tests
    d = dict(K=bo.K, R0=bo.cameras[0].R, t0=bo.cameras[0].t, R99=bo.cameras[99].R, t99=bo.cameras[99].t,
             ncameras=len(bo.cameras), ntracks=len(bo.tracks),
             track0_cams=np.array(sorted(bo.tracks[0].measurements.keys())),
             track0_z=np.array([bo.tracks[0].measurements[i] for i in sorted(bo.tracks[0].measurements.keys())], float),
             nobs=sum(len(t.measurements) for t in bo.tracks))
    save(args.out, 'oleg_io', d)

    # ---- (6b) window_slam.run on the first 10 cameras / 100 tracks, window of 4 --------
    with quiet():
        ws = B.Bundle()
        ws.K = bo.K.copy()
        for i in range(10):
            ws.add_camera(B.Camera(bo.cameras[i].R.copy(), bo.cameras[i].t.copy()))
        for j in range(100):
            tr = bo.tracks[j]
            ids = [i for i in range(10) if tr.has_measurement(i)]
            ws.add_track(B.Track(ids, [np.asarray(tr.get_measurement(i), float) for i in ids]))
        ws.triangulate_all()
    d = bundle_arrays(ws)
    d.update(sensor_arrays(ws.sensor_model))
    # the reference's run() keeps no history: wrap BundleAdjuster.optimize to record it
    BA = ref['bundle_adjuster'].BundleAdjuster
    hist = []
    orig_opt = BA.optimize

    def rec_opt(self, *a, **kw):
        orig_opt(self, *a, **kw)
        hist.append((self.num_steps, self.converged, list(self.costs)))
    BA.optimize = rec_opt
    try:
        import copy
        state = {}
        orig_run_ba = ref['window_slam'].BundleAdjuster
        with quiet():
            # run() returns nothing: capture the last adjuster's bundle through the recorder
            last = {}
            orig_set = BA.set_bundle

            def rec_set(self, *a, **kw):
                last['ba'] = self
                return orig_set(self, *a, **kw)
            BA.set_bundle = rec_set
            ref['window_slam'].run(copy.deepcopy(ws), 4)
            BA.set_bundle = orig_set
    finally:
        BA.optimize = orig_opt
    fin = bundle_arrays(last['ba'].bundle)
    d['ws_R'], d['ws_t'], d['ws_X'] = fin['R'], fin['t'], fin['X']
    d['ws_num_windows'] = len(hist)
    d['ws_num_steps'] = np.array([h[0] for h in hist])
    d['ws_converged'] = np.array([h[1] for h in hist])
    d['ws_first_cost'] = np.array([h[2][0] for h in hist])
    d['ws_last_cost'] = np.array([h[2][-1] for h in hist])
    save(args.out, 'scene_window_slam', d)

    # ---- (7) config-2-sized spot check: 100 cams x 1000 tracks ------------
    if args.big:
        with quiet():
            bo.triangulate_all()
        d = bundle_arrays(bo)
        d.update(sensor_arrays(bo.sensor_model))
        tmp = {}
        adjuster_blocks(ref, bo, 10., tmp, 'l10_')
        d['l10_b'], d['l10_dC'] = tmp['l10_b'], tmp['l10_dC']
        d['l10_S_fro'] = np.linalg.norm(tmp['l10_S'])
        d['l10_dP_norm'] = np.linalg.norm(tmp['l10_dP'])
        d['l10_dP_head'] = tmp['l10_dP'][:20]
        d['l10_cost'] = tmp['l10_cost']
        d['obs_z'] = d['obs_z'].astype(np.int32)     # integer pixels on disk
        save(args.out, 'scene_oleg_100x1000', d)


if __name__ == '__main__':
    main()
