"""BundleAdjuster - pysfm's Levenberg-Marquardt bundle adjuster with its arithmetic
on an MI355X (reference: bundle_adjuster.py:33-343).

Same constructor, methods, return shapes, attributes and exceptions as the
reference, plus ``step()`` (one outer LM iteration).  What stays in Python is what
the reference keeps in Python: id / mask bookkeeping (``set_bundle``) and the
damping / accept / reject schedule (``optimize``).  Every numeric step is a HIP
kernel behind the C ABI of include/pysfm_ba.h:

    set_bundle                -> ba_set_problem             validation, internal order, work lists: on the device
    optimize / step           -> ba_lm_trial                one LM trial = one batch of launches, one synchronisation
    compute_cost              -> ba_cost                    (k_cost; inside a trial: fused into k_backsub_groups)
    prepare_schur_complement  -> ba_linearize               (k_linearize_groups | k_linearize, k_camera_blocks)
    apply_damping + compute_schur_complement -> ba_schur    (k_point_invert_schur_init, k_schur_groups_mfma2;
                                                             other scenes: k_schur_groups, k_schur_pairs, k_dense_*)
    solve_motion_normal_eqns  -> ba_solve_reduced           (block cyclic reduction k_bcr_* / k_bcrw_*; k_band_solve for tiny
                                                             systems; dense blocked Cholesky k_dense_* for wide bands;
                                                             LU with partial pivoting, k_bcr_eliminate_lu / k_lu_*, when
                                                             the system is not positive definite)
    backsubstitute            -> ba_backsubstitute          (k_backsub_groups | k_backsub)
    update_motion / update_structure -> ba_apply_update     (k_apply_update; inside a trial: fused into the back-substitution)

During ``optimize`` the accepted and the trial parameter sets both live on the GPU;
``self.bundle`` is materialised on the host only when somebody reads it.
"""

import numpy as np

from ._capi import PARAMS_CUR, PARAMS_TRIAL, SOLVE_TIMED_OUT as SOLVER_TIMED_OUT
from .backend import ReducedSystemSingular
from .sensor_model import device_params_of


############################################################################
def _select_arrays(L, mask):
    """select() on arrays: (subset, positions of the subset in L), both as int64 arrays."""
    L = np.asarray(L)
    mask = np.asarray(mask)
    if mask.dtype.kind == 'b':
        assert len(mask) == len(L)
        idx = np.nonzero(mask)[0]
        return L[idx], idx
    assert mask.dtype.kind == L.dtype.kind
    order = np.argsort(L, kind='stable')
    where = np.searchsorted(L, mask, sorter=order) if len(L) else np.zeros(len(mask), np.int64)
    ok = len(L) > 0 and np.all(where < len(L))
    idx = order[np.minimum(where, max(len(L) - 1, 0))] if len(L) else where
    assert ok and np.array_equal(L[idx], mask), 'Mask contained some items not in L'
    return mask, idx


def select(L, mask):
    """Subset of L by boolean mask or by explicit ids (bundle_adjuster.py:11-24)."""
    subset, idx = _select_arrays(L, mask)
    return subset, idx.tolist()


def _id_list(name):
    """The reference keeps its id / index lists as Python lists (callers read them: bundle_adjuster_unittest.py:38); at a
    million tracks building them costs more than the whole device-side set-up, so they live as arrays and become lists
    when somebody asks."""
    arr, cache = '_%s_arr' % name, '_%s_list' % name

    def get(self):
        lst = self.__dict__.get(cache)
        if lst is None:
            a = self.__dict__.get(arr)
            if a is None:
                return None
            lst = self.__dict__[cache] = a.tolist()
        return lst

    def put(self, value):
        self.__dict__[arr] = np.asarray(value, np.int64).reshape(-1)
        self.__dict__[cache] = None
    return property(get, put)


############################################################################
class NormalEquationsIllconditioned(Exception):
    '''Thrown when the bundle adjuster fails to solve a set of normal
    equations (bundle_adjuster.py:27-30)'''
    pass


############################################################################
class BundleAdjuster(object):
    # The threshold applied to singular values when inverting the lower-diagonal
    # (HPP) blocks for the schur compliment.  None = full inverse
    # (bundle_adjuster.py:37).
    SCHUR_COMPLIMENT_PINV_THRESHOLD = 1e-5

    # ids / positions of the selected and of the optimised cameras and tracks (bundle_adjuster.py:63-101): lists, as in the reference
    camera_ids = _id_list('camera_ids')
    track_ids = _id_list('track_ids')
    optim_camera_ids = _id_list('optim_camera_ids')
    optim_camera_indices = _id_list('optim_camera_indices')
    optim_track_ids = _id_list('optim_track_ids')
    optim_track_indices = _id_list('optim_track_indices')

    @property
    def camera_id_set(self):
        return set(self.camera_ids)

    def __init__(self, bundle=None, backend=None, device=0, comm=None, verbose=True):
        '''bundle: a Bundle (ours, or any object with the reference Bundle's attributes).
        backend: compute backend; default = a new HipBackend on `device`.
        comm: optional pysfm_amd.distributed.ShardComm - this process then holds one
        shard of the tracks and the reduced camera system is all-reduced over RCCL.
        A reduced system the device Cholesky finds not positive definite is solved again on the device by LU with
        partial pivoting, like the reference's (numpy.linalg.solve, bundle_adjuster.py:303), at any size.'''
        self.num_steps = 0
        self.converged = False
        self.costs = []
        self.verbose = verbose
        self._backend = backend
        self._device = device
        self._comm = comm
        self._host_bundle = None
        self._host_stale = False
        self._damp_factor = 1.
        self._have_blocks = False
        self._damping = 10.
        self.lm_trials = 0
        self.trial_log = []       # one (damping, outcome, cost of the trial set) per LM trial; outcome 'accepted' / 'rejected' / 'ill-conditioned'
        if bundle is not None:
            self.set_bundle(bundle)

    def _say(self, msg):
        if self.verbose:
            print(msg)

    @property
    def backend(self):
        if self._backend is None:
            from .backend import HipBackend      # raises if libpysfm_ba.so / the GPU is missing
            self._backend = HipBackend(self._device)
        if self._comm is not None and not getattr(self, '_comm_checked', False):
            # the shards' collectives move into the library when they can (RCCL on the handle's own stream);
            # every rank takes the same decision (ShardComm.enable_direct agrees on it)
            self._comm_checked = True
            enable = getattr(self._comm, 'enable_direct', None)
            if enable is not None and hasattr(self._backend, 'comm_attach'):
                enable(self._backend)
        return self._backend

    # ------------------------------------------------------------------ bundle <-> device
    @property
    def bundle(self):
        '''The current (accepted) bundle.  After optimize() this is a NEW bundle; the
        one passed to set_bundle is never mutated (bundle_adjuster.py:151).'''
        if self._host_stale:
            R, t, X = self.backend.get_params(PARAMS_CUR)
            b = self._host_bundle.clone_params()
            set_poses = getattr(b.cameras, 'set_poses', None)
            if set_poses is not None:                # (rows of the stacked arrays: no Camera object is made for a pose nobody looks at)
                set_poses(self.camera_ids, R, t)
            else:
                for pos, idx in enumerate(self.camera_ids):
                    b.cameras[idx].R = R[pos].copy()
                    b.cameras[idx].t = t[pos].copy()
            if self._all_tracks:
                b.reconstruction = X
            else:
                b.reconstruction[self._track_ids_arr] = X
            self._host_bundle = b
            self._host_stale = False
        return self._host_bundle

    @bundle.setter
    def bundle(self, b):
        self._host_bundle = b
        self._host_stale = False
        if self.camera_ids is not None:
            self._upload(b, PARAMS_CUR)

    def _params_of(self, bundle):
        poses = getattr(bundle.cameras, 'poses', None)
        if poses is not None:
            R, t = poses(self.camera_ids)
        else:
            R = np.array([bundle.cameras[i].R for i in self.camera_ids], float).reshape(-1, 3, 3)
            t = np.array([bundle.cameras[i].t for i in self.camera_ids], float).reshape(-1, 3)
        X = np.asarray(bundle.reconstruction, float)
        X = (X if self._all_tracks else X[self._track_ids_arr]).reshape(-1, 3)
        return R, t, X

    def _upload(self, bundle, which):
        params = self._params_of(bundle)
        self.backend.set_params(which, *params)
        if which == PARAMS_CUR:
            self._cur_cost = None                    # cached cost of the current set
            # what the device's current set holds (compute_cost compares) - a copy where `params` aliases the caller's arrays
            self._uploaded_cur = params[:2] + (params[2].copy() if self._all_tracks else params[2],)

    # ------------------------------------------------------------------ set_bundle
    def set_bundle(self, bundle, camera_ids=None, track_ids=None, camera_mask=None, track_mask=None, upload=True):
        '''Configure the bundle this adjuster operates on (bundle_adjuster.py:54-114).
        camera_ids / track_ids select the cameras / tracks whose measurements are used;
        camera_mask / track_mask (bool array over the selection, or a list of ids)
        select what is updated.  Default: all cameras but the first, all tracks.
        upload=False: the problem only - which cameras, tracks and observations; the parameter values follow with
        `adjuster.bundle = b` (a bundle of the same structure), before anything is computed.'''
        bundle.check_consistency()
        self._host_bundle = bundle
        self._host_stale = False

        ncam_all, ntrk_all = len(bundle.cameras), len(bundle.tracks)
        if camera_ids is None:
            cam_ids = np.arange(ncam_all, dtype=np.int64)
        else:
            cam_ids = np.asarray(list(camera_ids) if not isinstance(camera_ids, np.ndarray) else camera_ids)
            if len(cam_ids):
                assert cam_ids.dtype.kind in 'iu'
                assert cam_ids.min() >= 0
                assert cam_ids.max() < ncam_all
            cam_ids = cam_ids.astype(np.int64).reshape(-1)
        if track_ids is None:
            trk_ids = np.arange(ntrk_all, dtype=np.int64)
        else:
            trk_ids = np.asarray(list(track_ids) if not isinstance(track_ids, np.ndarray) else track_ids)
            if len(trk_ids):              # (a shard of a sharded adjuster may be empty)
                assert trk_ids.dtype.kind in 'iu'
                assert trk_ids.min() >= 0
                assert trk_ids.max() < ntrk_all
            trk_ids = trk_ids.astype(np.int64).reshape(-1)
        self.camera_ids, self.track_ids = cam_ids, trk_ids
        self._all_tracks = track_ids is None or (len(trk_ids) == ntrk_all and np.array_equal(trk_ids, np.arange(ntrk_all)))

        if camera_mask is None:
            assert len(cam_ids) > 1, 'Cannot optimize just one camera'
            self.optim_camera_ids, self.optim_camera_indices = cam_ids[1:], np.arange(1, len(cam_ids))
        else:
            self.optim_camera_ids, self.optim_camera_indices = _select_arrays(cam_ids, camera_mask)
        if track_mask is None:
            self.optim_track_ids, self.optim_track_indices = trk_ids, np.arange(len(trk_ids))
        else:
            self.optim_track_ids, self.optim_track_indices = _select_arrays(trk_ids, track_mask)
        optim_camera_indices, optim_track_indices = self._optim_camera_indices_arr, self._optim_track_indices_arr

        nc, nt = len(cam_ids), len(trk_ids)
        # device problem: observation SoA ordered by track position, flags, positions
        if hasattr(bundle, 'select_observations'):
            obs_cam, obs_pt, obs_z = bundle.select_observations(cam_ids, trk_ids)
        else:
            obs_cam, obs_pt, obs_z = _select_observations_generic(bundle, self.camera_ids, self.track_ids)
        cam_opt_pos = -np.ones(nc, np.int32)
        cam_opt_pos[optim_camera_indices] = np.arange(len(optim_camera_indices))
        if track_mask is None:
            pt_opt = np.ones(nt, np.uint8)
        else:
            pt_opt = np.zeros(nt, np.uint8)
            pt_opt[optim_track_indices] = 1
        self._cam_opt_pos, self._pt_opt = cam_opt_pos, pt_opt
        self._nobs = len(obs_cam)

        be = self.backend
        if self._comm is not None and hasattr(be, 'set_min_half_bandwidth'):
            # the ranks add their [S | b] buffers element by element: one band layout for all, i.e. ONE order of the optimised
            # cameras (planned by every rank from the whole bundle: the same input, the same order) and the
            # widest spread of optimised-camera positions inside a track over ALL shards
            layout = self._shared_camera_layout(bundle, cam_ids, cam_opt_pos) if hasattr(be, 'set_camera_layout') else None
            if hasattr(be, 'set_camera_layout'):
                be.set_camera_layout(layout)
            self._shared_layout = layout
            if hasattr(be, 'set_pattern_lists'):
                if getattr(self, '_sparse_options_set', False):      # (an earlier bundle of this adjuster took the sparse path: the choice is made afresh)
                    be.set_option('solver', 'auto'); be.set_option('schur', 'auto')
                    self._sparse_options_set = False
                be.set_pattern_lists(*(getattr(self, '_shared_lists', None) or (None, None)))
            pos = cam_opt_pos[np.asarray(obs_cam, int)]
            if layout is not None:
                pos = np.where(pos >= 0, layout[np.maximum(pos, 0)], -1)
            ok = pos >= 0
            lo = np.full(nt, np.iinfo(np.int32).max, np.int64)
            hi = np.full(nt, -1, np.int64)
            np.minimum.at(lo, np.asarray(obs_pt, int)[ok], pos[ok])
            np.maximum.at(hi, np.asarray(obs_pt, int)[ok], pos[ok])
            spread = np.where(hi >= 0, hi - lo, 0)
            be.set_min_half_bandwidth(int(self._comm.allreduce_max(int(spread.max()) if nt else 0)))
        be.set_problem(nc, nt, obs_cam, obs_pt, obs_z, np.asarray(bundle.K, float), cam_opt_pos, pt_opt)
        if self._comm is not None and hasattr(be, 'set_pattern_lists') and getattr(self, '_shared_lists', None) is not None:
            # A sharded scene without a band (an unordered photo collection): the sparse path - conjugate gradients over the blocks the
            # tracks of the WHOLE scene define, [S | b] stored and summed over the ranks as that list - when all ranks see a scene for it
            # (the same lists, the same agreed band width: the same answer) or the caller says so (`sparse` = True | False | None).
            want = self.sparse
            if want is None:
                want = be.nco >= self.SPARSE_MIN_CAMERAS and be.half_bandwidth > 23 and be.pcg_info()['band_fill'] <= .1
            if self._comm._agree(bool(want)) and not be.problem_info().get('packed_store', 0):
                be.set_option('solver', 'pcg'); be.set_option('schur', 'pairs')
                self._sparse_options_set = True
                be.set_problem(nc, nt, obs_cam, obs_pt, obs_z, np.asarray(bundle.K, float), cam_opt_pos, pt_opt)
        self._configure_distributed_solve(be, cam_opt_pos, obs_cam, obs_pt, nt)
        be.set_sensor(*device_params_of(bundle.sensor_model))
        if upload:
            self._upload(bundle, PARAMS_CUR)
        self._have_blocks = False
        self._blocks_cache = None
        self._have_W = False
        self._damp_factor = 1.
        self._say('Configured a bundle adjuster for %d cameras, %d tracks' % (nc, nt))

    def _shared_camera_layout(self, bundle, cam_ids, cam_opt_pos):
        """The order of the optimised cameras every rank of a sharded adjuster uses (csrc/ba_order.hip through
        ba_plan_camera_layout, no border: the all-reduce payload is the band), planned from ALL tracks of the bundle so that
        every rank arrives at the same one; None when the caller's order is as narrow as an order can be.  The ranks check
        that they agree (a checksum over the group) and keep the caller's order otherwise."""
        be = self.backend
        cam, trk, _ = bundle.observation_table()
        cpos = -np.ones(max(len(bundle.cameras), 1), np.int64)
        cpos[cam_ids] = np.arange(len(cam_ids))
        sel = cpos[np.asarray(cam, np.int64)]
        pos = np.where(sel >= 0, cam_opt_pos[np.maximum(sel, 0)], -1).astype(np.int64)
        ok = pos >= 0
        pos, trk = pos[ok], np.asarray(trk, np.int64)[ok]            # (the table is sorted by (track, camera): so is what is left)
        nco = int((cam_opt_pos >= 0).sum())
        layout = None
        self._shared_lists = None
        if len(pos) and nco >= 3:
            start = np.flatnonzero(np.r_[True, trk[1:] != trk[:-1]])
            cnt = np.diff(np.r_[start, len(trk)])
            lo, hi = np.minimum.reduceat(pos, start), np.maximum.reduceat(pos, start)
            if int((hi - lo).max()) > max(1, int(cnt.max()) - 1):
                # distinct camera lists, approximately (a 64-bit mix of length, ends, sum and sum of squares: two lists that collide
                # are taken for one, which can only cost band width, never correctness)
                p64 = pos.astype(np.uint64)
                key = (cnt.astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)) ^ (lo.astype(np.uint64) * np.uint64(0xC2B2AE3D27D4EB4F)) \
                    ^ (hi.astype(np.uint64) * np.uint64(0x165667B19E3779F9)) ^ (np.add.reduceat(p64, start) * np.uint64(0xD6E8FEB86659FD93)) \
                    ^ (np.add.reduceat(p64 * p64, start) * np.uint64(0xFF51AFD7ED558CCD))
                _, rep = np.unique(key, return_index=True)
                rep = rep[cnt[rep] >= 2]
                off = np.r_[0, np.cumsum(cnt[rep])]
                idx = np.repeat(start[rep] - off[:-1], cnt[rep]) + np.arange(off[-1])
                # (the distinct lists of ALL tracks: also what a sharded scene on the sparse path defines its blocks from - set_bundle)
                self._shared_lists = (off.astype(np.int32), pos[idx].astype(np.int32))
                new, n1, hb = be.plan_camera_layout(nco, off, pos[idx], None, allow_border=False)
                if not np.array_equal(new, np.arange(nco)):
                    layout = new.astype(np.int64)
        chk = 0. if layout is None else float(np.dot(layout % 1000003, np.arange(1, len(layout) + 1) % 1000003) % 1000003) + 1.
        same = self._comm.allreduce_max(chk) == chk and self._comm.allreduce_max(-chk) == -chk
        return layout if self._comm._agree(same) else None

    # bytes of [S | b] above which the sharded adjuster spreads the reduced SOLVE over its ranks instead of summing the whole
    # band and solving it on every rank (csrc/ba_dist.h): below, one all-reduce of a few MB and a 0.1 ms solve are the faster way
    DISTRIBUTED_SOLVE_MIN_BYTES = 4 << 20
    distributed_solve = 'auto'           # 'auto' | True | False
    sparse = None                        # sharded scenes: None = the library's rule (1500 optimised cameras, a tenth of the band's blocks), True | False
    SPARSE_MIN_CAMERAS = 1500

    def _configure_distributed_solve(self, be, cam_opt_pos, obs_cam, obs_pt, nt):
        """Sharded adjuster: switch the trial to the solve that is spread over the ranks when it applies - the band is large,
        the library has a cut for (cameras, band width, ranks), and EVERY rank's tracks start inside its own interval
        (distributed.shard_tracks(plan=...) cuts them that way).  All ranks decide together."""
        self._dist = False
        if self._comm is None or not hasattr(be, 'dist_enable'):
            return
        comm = self._comm
        want = self.distributed_solve
        ok = bool(want) and comm.world_size > 1
        ok = ok and not be.problem_info().get('packed_store', 0)      # (the sparse path has no band to cut: every rank solves)
        if want == 'auto':
            ok = ok and 8 * be.S_doubles >= self.DISTRIBUTED_SOLVE_MIN_BYTES
        cut = be.dist_plan(be.nco, be.half_bandwidth, comm.world_size) if ok else None
        ok = ok and cut is not None
        if ok:
            cb, N, P = cut
            pos = cam_opt_pos[np.asarray(obs_cam, int)]
            layout = getattr(self, '_shared_layout', None)
            if layout is not None:                        # (the tree is cut along the library's camera positions)
                pos = np.where(pos >= 0, layout[np.maximum(pos, 0)], -1)
            first = np.full(nt, np.iinfo(np.int64).max, np.int64)
            np.minimum.at(first, np.asarray(obs_pt, int)[pos >= 0], pos[pos >= 0])
            first = first[first < np.iinfo(np.int64).max]
            lo, hi = comm.rank * P * cb, (comm.rank + 1) * P * cb
            ok = len(first) == 0 or (first.min() >= lo and (first.max() < hi or comm.rank == comm.world_size - 1))
        ok = comm._agree(ok)
        info = be.dist_enable(comm.rank, comm.world_size if ok else 1)
        self._dist = bool(info['on'])
        if ok and not self._dist:
            raise RuntimeError('ba_dist_enable refused a plan ba_dist_plan had accepted')

    # ------------------------------------------------------------------ LM loop
    def optimize(self, param_mask=None, max_steps=25, init_damping=10., improvement_threshold=1e-4):
        '''Optimize the current bundle to convergence (bundle_adjuster.py:117-162).'''
        self._damping = init_damping
        self.num_steps = 0
        self.lm_trials = 0
        self.trial_log = []
        self.converged = False
        if self._resident_applies(param_mask):
            self._cur_cost = None
            self.costs = []
            self._resident_steps(max_steps, improvement_threshold, param_mask)
        else:
            self._cur_cost = self._cost(PARAMS_CUR)
            self.costs = [self._cur_cost]
            while not self.converged and self.num_steps < max_steps:
                self.step(param_mask, improvement_threshold)
        if self.converged:
            self._say('Converged after %d steps' % self.num_steps)
        else:
            self._say('Failed to converge after %d steps' % self.num_steps)

    def optimize_begin(self, param_mask=None, max_steps=25, init_damping=10., improvement_threshold=1e-4):
        """optimize() in two halves, for a caller with host work that does not depend on the result (window_slam.run prepares
        the next window on a second adjuster): where the loop runs on the device as one launch (_resident_applies) this starts
        it and returns; optimize_end() waits and replays the log.  Anywhere else optimize_end() runs the whole of optimize().
        Nothing else may be asked of this adjuster in between."""
        self._begun = (param_mask, max_steps, init_damping, improvement_threshold, False)
        be = self.backend
        if not (self._resident_applies(param_mask) and hasattr(be, 'lm_resident_begin')):
            return
        self._damping = init_damping
        self.num_steps = 0
        self.lm_trials = 0
        self.trial_log = []
        self.converged = False
        self._cur_cost = None
        self.costs = []
        be.lm_resident_begin(max_steps, 0, False, False, init_damping, improvement_threshold, self.SCHUR_COMPLIMENT_PINV_THRESHOLD,
                             None, self._resident_cam_mask(param_mask))
        self._begun = (param_mask, max_steps, init_damping, improvement_threshold, True)

    def optimize_end(self):
        """The second half of optimize_begin()."""
        param_mask, max_steps, init_damping, improvement_threshold, launched = self._begun
        self._begun = None
        if not launched:
            return self.optimize(param_mask, max_steps, init_damping, improvement_threshold)
        self._resident_steps(max_steps, improvement_threshold, param_mask, launched=True)
        self._say('Converged after %d steps' % self.num_steps if self.converged else 'Failed to converge after %d steps' % self.num_steps)

    def step(self, param_mask=None, improvement_threshold=1e-4):
        '''One outer iteration of optimize(): retry with growing damping until a trial
        lowers the cost (bundle_adjuster.py:128-157; cf. optimize.py:110-135).
        Returns self.converged.'''
        if self._resident_applies(param_mask):
            if self.converged:                       # (the loop below would not run either)
                self.num_steps += 1
                return True
            return self._resident_steps(self.num_steps + 1, improvement_threshold, param_mask)
        # the cost of the current set is the cost of the trial that was accepted last (same kernel, same
        # data, deterministic summation order): no need to evaluate it again at the top of every step
        cur_cost = getattr(self, '_cur_cost', None)
        if cur_cost is None:
            cur_cost = self._cur_cost = self._cost(PARAMS_CUR)
        if not self.costs:
            self.costs = [cur_cost]
        self.num_steps += 1
        self._say('Step %d: cost=%f, damping=%f' % (self.num_steps, cur_cost, self._damping))
        while not self.converged and self._damping < 1e+8:
            accepted, next_cost = self.trial(self._damping, param_mask, cur_cost)
            self.trial_log.append((self._damping, 'ill-conditioned' if accepted is None else 'accepted' if accepted else 'rejected', next_cost))
            if accepted is None:                       # ill-conditioned: raise damping
                self._damping *= 10.
                self.converged = self._damping > 1e+8
                continue
            if accepted:
                self._damping *= .1
                self.costs.append(next_cost)
                self.converged = abs(cur_cost - next_cost) < improvement_threshold
                break
            else:
                self._damping *= 10.
                self.converged = self._damping > 1e+8
        return self.converged

    # ------------------------------------------------------------------ the loop on the device (small problems)
    resident = True          # False: always the Python loop over ba_lm_trial

    def _resident_applies(self, param_mask):
        """A problem that fits one compute unit (a sliding window, the reference's own test scenes) runs the whole
        loop of optimize() / step() as ONE resident workgroup (csrc/ba_resident.h): the same schedule, taken on the
        device, replayed here from its log."""
        if not self.resident or self._comm is not None:
            return False
        be = self.backend
        return hasattr(be, 'lm_resident_fits') and be.lm_resident_fits()

    def _resident_cam_mask(self, param_mask):
        """Camera parameters deleted from the solve, or None (the common case costs no array work)."""
        if param_mask is None:
            return None
        cam_mask = self._cam_param_mask(param_mask)
        return None if np.all(cam_mask) else cam_mask

    def _resident_steps(self, max_steps, improvement_threshold, param_mask=None, launched=False):
        """Outer iterations until self.num_steps == max_steps or convergence (the loop of optimize(), bundle_adjuster.py:128-157),
        on the device (launched: the first launch has been made, optimize_begin).  A trial the resident loop cannot take (a reduced system that is not positive definite: the reference
        solves it by LU; a singular point block in plain-inverse mode: the reference raises) goes through trial() and the
        loop resumes on the device."""
        from ._capi import RESIDENT_DONE, RESIDENT_LOG_FULL, RESIDENT_TIMED_OUT
        be = self.backend
        in_step = False
        cam_mask = self._resident_cam_mask(param_mask)
        if not hasattr(self, 'costs') or self.costs is None:
            self.costs = []
        while True:
            steps_before = self.num_steps
            if launched:                             # (optimize_begin has made this call's launch)
                log, launched = be.lm_resident_end(), False
            else:
                log = be.lm_resident(max_steps, self.num_steps, in_step, self.converged, self._damping, improvement_threshold,
                                     self.SCHUR_COMPLIMENT_PINV_THRESHOLD, self._cur_cost, cam_mask)
            if log.exit_reason == RESIDENT_TIMED_OUT:
                # the launch's workgroups lost each other (compute units taken away underneath it, a partitioned device):
                # nothing happened on the device, the schedule stands where it stood - the Python loop takes over for good
                import warnings
                warnings.warn('pysfm_amd: the resident loop timed out; continuing with the loop over ba_lm_trial', RuntimeWarning)
                self.resident = False
                self.resident_timeouts = getattr(self, 'resident_timeouts', 0) + 1
                if in_step:
                    self.num_steps -= 1                 # (step() counts the step we are in the middle of again)
                while not self.converged and self.num_steps < max_steps:
                    self.step(param_mask, improvement_threshold)
                return self.converged
            if self._cur_cost is None and log.have_cost0:
                self._cur_cost = log.cost0
            if not self.costs and self._cur_cost is not None:
                self.costs = [self._cur_cost]
            step_no = steps_before
            nlog = log.ntrials                      # (one slice per array: a ctypes index per entry costs more than the loop body)
            for damping, cost, acc in zip(log.trial_damping[:nlog], log.trial_cost[:nlog], log.trial_accepted[:nlog]):
                if not in_step:
                    step_no += 1
                    in_step = True
                    self._say('Step %d: cost=%f, damping=%f' % (step_no, self._cur_cost, damping))
                self.trial_log.append((damping, 'accepted' if acc else 'rejected', cost))
                if acc:
                    self.costs.append(cost)
                    self._cur_cost = cost
                    in_step = False
            self.lm_trials += log.ntrials
            self.num_steps, self.converged, self._damping, in_step = log.nsteps, bool(log.converged), log.damping, bool(log.in_step)
            if log.accepted:
                self._host_stale = True
            self._have_blocks = False
            self._blocks_cache = None
            self._have_W = False
            if log.exit_reason == RESIDENT_DONE:
                break
            if log.exit_reason == RESIDENT_LOG_FULL:
                continue
            # one trial through the general path, the schedule of step() around it
            if self._cur_cost is None:
                self._cur_cost = self._cost(PARAMS_CUR)
                self.costs = self.costs or [self._cur_cost]
            cur_cost = self._cur_cost
            accepted, next_cost = self.trial(self._damping, param_mask, cur_cost)
            self.trial_log.append((self._damping, 'ill-conditioned' if accepted is None else 'accepted' if accepted else 'rejected', next_cost))
            if accepted:
                self._damping *= .1
                self.costs.append(next_cost)
                self.converged = abs(cur_cost - next_cost) < improvement_threshold
                in_step = False
            else:
                self._damping *= 10.
                self.converged = self._damping > 1e+8
                in_step = True
        if self._cur_cost is None:                    # no trial ran (no step to take, or the damping already beyond its range)
            self._cur_cost = self._cost(PARAMS_CUR)
        if not self.costs:
            self.costs = [self._cur_cost]
        return self.converged

    def trial(self, damping, param_mask, cur_cost):
        '''One LM trial, entirely device-resident: linearise, damp, Schur, solve,
        back-substitute, apply to the trial set, evaluate.  Accepts (swaps the
        parameter sets) iff the cost went down.  Returns (accepted | None, next_cost).'''
        self.lm_trials += 1
        next_cost = None
        be = self.backend
        self._blocks_cache = None
        self._have_W = False
        if (self._comm is None or getattr(be, 'direct_comm', False)) and hasattr(be, 'lm_trial'):
            # single GPU - or shards whose collectives the library issues itself (ba_comm_init): the whole trial is
            # one batch of launches with one synchronisation; the cost that comes back is the sum over the shards
            cam_param_mask = None                              # (the common case costs no array work)
            if param_mask is not None:
                cam_param_mask = self._cam_param_mask(param_mask)
                if np.all(cam_param_mask):
                    cam_param_mask = None
            self._damp_factor = 1. + damping
            info, cost = be.lm_trial(damping, self.SCHUR_COMPLIMENT_PINV_THRESHOLD, cam_param_mask)
            self._have_blocks = True
            if info == 0:
                next_cost = cost
            elif info == SOLVER_TIMED_OUT:
                self._note_solver_timeout(damping)                 # a solver bug, not a property of the system: said loudly,
            # (info > 0: not positive definite - the stepwise path below solves again, by LU with partial pivoting on the device)
        elif self._comm is not None and hasattr(be, 'lm_trial_begin'):
            # sharded: the same batch in two halves around the all-reduce of [S | b]; the ranks' trial
            # costs are summed on the device, one synchronisation per trial
            cam_param_mask = None
            if param_mask is not None:
                cam_param_mask = self._cam_param_mask(param_mask)
                if np.all(cam_param_mask):
                    cam_param_mask = None
            self._damp_factor = 1. + damping
            be.lm_trial_begin(damping, self.SCHUR_COMPLIMENT_PINV_THRESHOLD)
            if getattr(self, '_dist', False):
                # the solve spread over the ranks: three small sums instead of the whole band (csrc/ba_dist.h)
                for stage in (1, 2, 3):
                    self._comm.allreduce_exchange(be, be.dist_stage(stage, cam_param_mask))
                be.dist_stage(4)
                be.lm_trial_finish()
                pre = 0
            else:
                self._comm.allreduce_reduced(be)
                pre = be.lm_trial_end(cam_param_mask)
            self._have_blocks = True
            if pre == 0:
                from ._capi import TRIAL_PARTIALS
                cost, nsing, info = self._comm.allreduce_trial_result(be, TRIAL_PARTIALS)
                if nsing > 0 and self.SCHUR_COMPLIMENT_PINV_THRESHOLD is None:
                    raise np.linalg.LinAlgError('singular 3x3 point block(s) in plain-inverse mode')
                if info == 0:
                    next_cost = cost
                elif info == SOLVER_TIMED_OUT:
                    self._note_solver_timeout(damping)
        if next_cost is None:
            try:
                self._compute_update_device(damping, param_mask, fetch=False)
            except NormalEquationsIllconditioned:
                return None, None
            be.apply_update(PARAMS_CUR, PARAMS_TRIAL)            # bnext = clone; update_motion/structure
            next_cost = self._cost(PARAMS_TRIAL)
        if next_cost < cur_cost:
            self.backend.swap_params()                          # self.bundle = bnext
            self._host_stale = True
            self._have_blocks = False
            self._blocks_cache = None
            self._cur_cost = next_cost
            return True, next_cost
        return False, next_cost

    def _note_solver_timeout(self, damping):
        """A workgroup of the one-launch cyclic reduction gave up waiting for its neighbours (status word
        BA_SOLVE_TIMED_OUT of include/pysfm_ba.h): that is a fault of the solver or of the GPU, never a property of
        the matrix - it must not hide behind a damping increase.  Counted and always reported; the trial is then
        repeated through the stepwise entry points."""
        import warnings
        self.solver_timeouts = getattr(self, 'solver_timeouts', 0) + 1
        warnings.warn('pysfm_amd: the device solve of the reduced system timed out at damping %g (status 0x%x); '
                      'repeating the trial stepwise' % (damping, SOLVER_TIMED_OUT), RuntimeWarning)

    # ------------------------------------------------------------------ cost
    def _cost(self, which):
        c = self.backend.cost(which)
        if self._comm is not None:
            c = self._comm.allreduce_scalar(c)
        return c

    def compute_cost(self, bundle):
        '''Sum of squared residuals over optim_track_ids x optim_camera_ids
        (bundle_adjuster.py:165-171).'''
        # The reference evaluates the bundle it is given.  The device copy of the current set stands in for it only
        # when it IS that bundle: the caller's object, not stale (no step accepted since it was uploaded).  It is
        # uploaded again even then - the caller may have edited it in place (cameras[i].perturb, transform).
        if bundle is self._host_bundle and not self._host_stale:
            # ... but only if it differs from what was uploaded last: prepare_schur_complement() -> compute_cost(ba.bundle)
            # -> compute_schur_complement() is a sequence the reference allows, and an unchanged bundle must not
            # throw the device's linearisation away
            up = getattr(self, '_uploaded_cur', None)
            params = self._params_of(bundle)
            if up is None or any(a.shape != b.shape or not np.array_equal(a, b) for a, b in zip(params, up)):
                self._upload(bundle, PARAMS_CUR)
                self._have_blocks = False
                self._blocks_cache = None
            return self._cost(PARAMS_CUR)
        self._upload(bundle, PARAMS_TRIAL)
        return self._cost(PARAMS_TRIAL)

    # ------------------------------------------------------------------ update
    def _cam_param_mask(self, param_mask):
        nc = len(self.optim_camera_ids)
        nt = len(self.optim_track_ids)
        nparams = 6 * nc + 3 * nt
        if param_mask is None:
            return np.ones(nc * 6, bool)
        param_mask = np.asarray(param_mask)
        assert param_mask.dtype.kind == 'b'
        assert np.shape(param_mask) == (nparams,), \
            'param_mask had shape %s but there are %d parameters' % (str(np.shape(param_mask)), nparams)
        assert np.all(param_mask[nc * 6:]), 'Eliminating point parameters not implemented'
        return param_mask[:nc * 6]

    def _compute_update_device(self, damping, param_mask, fetch):
        cam_param_mask = self._cam_param_mask(param_mask)
        be = self.backend
        be.linearize(PARAMS_CUR)
        self._have_blocks = True
        self._blocks_cache = None                     # the block properties must reflect THIS linearisation
        self._have_W = False
        self._damp_factor = 1. + damping
        self._schur_device()
        try:
            be.solve_reduced(None if np.all(cam_param_mask) else cam_param_mask)
        except ReducedSystemSingular:
            raise NormalEquationsIllconditioned
        if getattr(be, 'last_solve_path', None) == 'lu':
            self.lu_solves = getattr(self, 'lu_solves', 0) + 1               # (diagnostics: how often the LU semantics were needed)
        dC = be.get_solution() if fetch else None
        dP = be.backsubstitute(PARAMS_CUR, None, fetch=fetch)
        return dC, dP

    def _schur_device(self):
        be = self.backend
        be.schur(PARAMS_CUR, self._damp_factor - 1., self.SCHUR_COMPLIMENT_PINV_THRESHOLD)
        if self._comm is not None:
            self._comm.allreduce_reduced(be)

    def compute_update(self, damping, param_mask=None):
        '''Solve the normal equations using the Schur complement.  Returns
        (update-for-cameras [nco,6], update-for-points [nto,3]) - bundle_adjuster.py:176-208.'''
        dC, dP = self._compute_update_device(damping, param_mask, fetch=True)
        return -dC, -dP[self._optim_track_indices_arr]

    def prepare_schur_complement(self):
        '''Hessian blocks HCC, HPP, HCP and gradients bC, bP (bundle_adjuster.py:211-234).'''
        self.backend.linearize(PARAMS_CUR, store_W=True)
        self._have_blocks = True
        self._have_W = True
        self._damp_factor = 1.
        self._blocks_cache = None

    def _blocks(self, W=False):
        assert self._have_blocks, 'call prepare_schur_complement() first'
        if W and not getattr(self, '_have_W', False):
            # the last linearisation (compute_update / a trial) did not keep the per-observation blocks:
            # linearise the same point again, this time with W
            self.backend.linearize(PARAMS_CUR, store_W=True)
            self._have_W = True
            self._blocks_cache = None
        c = getattr(self, '_blocks_cache', None)
        if c is None or (W and c['W'] is None):
            d = self.backend.get_blocks(W=W)
            if self._comm is not None:          # camera blocks are sums over all shards
                d['HCC'] = self._comm.allreduce_array(d['HCC'])
                d['bC'] = self._comm.allreduce_array(d['bC'])
            self._blocks_cache = d
        return self._blocks_cache

    @property
    def HCCs(self):
        f = np.ones((6, 6)) + (self._damp_factor - 1.) * np.eye(6)
        return self._blocks()['HCC'] * f

    @property
    def HPPs(self):
        f = np.ones((3, 3)) + (self._damp_factor - 1.) * np.eye(3)
        return self._blocks()['HPP'] * f

    @property
    def bCs(self):
        return self._blocks()['bC']

    @property
    def bPs(self):
        return self._blocks()['bP']

    @property
    def HCPs(self):
        '''Dense (nc, nt, 6, 3) array like the reference's (bundle_adjuster.py:107); built
        on demand from the per-observation blocks - only sensible for small scenes.'''
        nc, nt = len(self.camera_ids), len(self.track_ids)
        if nc * nt * 18 * 8 > 2 ** 31:
            raise MemoryError('dense HCPs would need %.1f GB; use backend.get_blocks(W=True)' % (nc * nt * 144 / 1e9))
        out = np.zeros((nc, nt, 6, 3))
        cam, pt, _ = (self._host_bundle.select_observations(self.camera_ids, self.track_ids)
                      if hasattr(self._host_bundle, 'select_observations')
                      else _select_observations_generic(self._host_bundle, self.camera_ids, self.track_ids))
        out[cam, pt] = self._blocks(W=True)['W']
        return out

    @property
    def HPP_invs(self):
        return self.backend.get_point_inverses()

    def apply_damping(self, damping):
        '''diag *= (1 + damping) on every HCC / HPP block (bundle_adjuster.py:238-242).
        Calls compound, as in the reference.'''
        assert self._have_blocks, 'call prepare_schur_complement() first'
        self._damp_factor *= (1. + damping)

    def compute_schur_complement(self):
        '''Reduced camera system (S [nco,nco,6,6], b [nco,6]) - bundle_adjuster.py:247-278.'''
        assert self._have_blocks, 'call prepare_schur_complement() first'
        self._schur_device()
        return self.backend.get_reduced()

    def solve_motion_normal_eqns(self, S, b, param_mask):
        '''Solve the (masked) reduced system given as host arrays (bundle_adjuster.py:281-312).
        This host step is the reference's own numpy.linalg.solve call; compute_update()
        uses the device-resident system instead.'''
        nc = len(self.optim_camera_ids)
        assert np.shape(S) == (nc, nc, 6, 6)
        assert np.shape(b) == (nc, 6)
        assert np.shape(param_mask) == (nc * 6,), 'shape was ' + str(np.shape(param_mask))
        param_mask = np.asarray(param_mask, bool)
        AC = np.asarray(S).transpose((0, 2, 1, 3)).reshape((nc * 6, nc * 6))
        bC = np.asarray(b).reshape(-1)
        AC_reduced = AC[param_mask].T[param_mask].T
        bC_reduced = bC[param_mask]
        try:
            dC_reduced = np.linalg.solve(AC_reduced, bC_reduced)
        except np.linalg.LinAlgError:
            raise NormalEquationsIllconditioned
        dC = np.zeros(nc * 6)
        dC[param_mask] = dC_reduced
        return dC.reshape(nc, 6)

    def backsubstitute(self, dC):
        '''Point updates from the camera update (bundle_adjuster.py:316-331).'''
        dP = self.backend.backsubstitute(PARAMS_CUR, np.asarray(dC, float), fetch=True)
        return dP[self._optim_track_indices_arr]

    # ------------------------------------------------------------------ parameter update
    def _apply_on_device(self, motion, structure, bundle):
        be = self.backend
        self._upload(bundle, PARAMS_TRIAL)
        be.apply_update(PARAMS_TRIAL, PARAMS_TRIAL, motion, structure)
        R, t, X = be.get_params(PARAMS_TRIAL)
        return R, t, X

    def update_motion(self, delta, bundle):
        '''cam.R <- R exp(delta[:3]), cam.t += delta[3:] for the optimised cameras
        (bundle_adjuster.py:334-337), evaluated by k_apply_update.'''
        assert np.shape(delta) == (len(self.optim_camera_ids), 6)
        R, t, _ = self._apply_on_device(delta, np.zeros((len(self.track_ids), 3)), bundle)
        for pos, idx in zip(self.optim_camera_indices, self.optim_camera_ids):
            bundle.cameras[idx].R = R[pos].copy()
            bundle.cameras[idx].t = t[pos].copy()

    def update_structure(self, delta, bundle):
        '''reconstruction[idx] += delta for the optimised tracks (bundle_adjuster.py:340-343).'''
        assert np.shape(delta) == (len(self.optim_track_ids), 3)
        full = np.zeros((len(self.track_ids), 3))
        full[self._optim_track_indices_arr] = delta
        _, _, X = self._apply_on_device(np.zeros((len(self.optim_camera_ids), 6)), full, bundle)
        for pos, idx in zip(self.optim_track_indices, self.optim_track_ids):
            bundle.reconstruction[idx] = X[pos]


def _select_observations_generic(bundle, camera_ids, track_ids):
    """select_observations for foreign Bundle objects (e.g. the reference's own class)."""
    cam_pos = {c: p for p, c in enumerate(camera_ids)}
    cam, pt, z = [], [], []
    for jpos, j in enumerate(track_ids):
        tr = bundle.tracks[j]
        items = [(cam_pos[i], m) for i, m in tr.measurements.items() if i in cam_pos]
        items.sort(key=lambda it: it[0])
        for ipos, m in items:
            cam.append(ipos)
            pt.append(jpos)
            z.append(np.asarray(m, float))
    return np.array(cam, np.int32), np.array(pt, np.int32), np.array(z, float).reshape(-1, 2)
