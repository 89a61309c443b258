"""HipBackend - NumPy-facing wrapper of one ``ba_handle`` (one GPU, one stream).

This is the only compute backend of the package.  All arithmetic of the
bundle-adjustment inner loop runs in the HIP kernels of libpysfm_ba.so; this
class only marshals arrays, owns the torch tensors that alias the reduced camera
system (for the RCCL all-reduce and the dense reduced solve) and turns C status
codes into Python exceptions.
"""
import ctypes as C

import numpy as np

from . import _capi as capi
from ._capi import PARAMS_CUR, PARAMS_TRIAL  # noqa: F401  (re-exported)

# the dense-visibility REDUCTION (ba_set_dense_visibility) is worth it once the band is this wide; it does not decide the
# solver: half-bandwidths up to 13 go through the one-launch cyclic reduction, up to 23 (kBcrwMaxHB of ba_bcr_wide.h) through the wide one, the dense
# blocked Cholesky takes over beyond
DENSE_MIN_HALF_BANDWIDTH = 21
DENSE_MIN_FILL = 0.25             # observed fraction of the (camera, track) pairs
DENSE_MAX_BYTES = 2 << 30         # of the staged operand


class SingularPointBlock(np.linalg.LinAlgError):
    """plain-inverse mode met a singular HPP block (numpy.linalg.inv would raise)."""


class ReducedSystemSingular(Exception):
    """The LU of the reduced camera system met an exactly zero pivot (where numpy.linalg.solve raises LinAlgError)."""


class HipBackend(object):
    device_lu = True                    # ba_solve_reduced solves a not-positive-definite system again by LU with partial pivoting (option device_lu)
    poison_after_set_problem = False    # tests/conftest.py sets it: every problem starts from NaNs in all LDS / workspace (ba_debug_poison)

    def __init__(self, device=0):
        self._lib = capi.load()
        self._h = C.c_void_p()
        rc = self._lib.ba_create(int(device), C.byref(self._h))
        if rc != capi.BA_OK:
            msg = self._lib.ba_last_error(None).decode()
            self._h = None
            raise capi.HipDeviceError('ba_create(device=%d) failed (%d): %s' % (device, rc, msg))
        self.device = int(device)
        self.nc = self.nt = self.nco = self.nobs = 0
        self._torch = None
        self._S_t = self._b_t = self._Sb_t = None
        self._dense = self._dense_keep = None
        self._attach_torch()

    # ---------------------------------------------------------------- plumbing
    def _attach_torch(self):
        """Run on torch's current stream so torch ops (all-reduce, solve) and our
        kernels are ordered without extra synchronisation."""
        try:
            import torch
        except ImportError:       # the C library works stand-alone; only the large solve needs torch
            return
        if not torch.cuda.is_available():
            return
        self._torch = torch
        torch.cuda.set_device(self.device)
        # The kernels run on a torch stream of our own: torch's default stream is the NULL stream, which the
        # C ABI reads as "make your own" - and a private HIP stream is not ordered against torch ops.  Every
        # torch op that touches the library's device buffers (the RCCL all-reduce of [S | b], the sum of the
        # trial-cost partials) runs under stream_ctx(), i.e. on this same stream.
        self._tstream = torch.cuda.Stream(device=self.device)
        self._check(self._lib.ba_set_stream(self._h, C.c_void_p(self._tstream.cuda_stream)))

    def stream_ctx(self):
        """Context manager: torch ops inside are enqueued on the stream the kernels run on."""
        return self._torch.cuda.stream(self._tstream)

    def _check(self, rc):
        if rc == capi.BA_OK:
            return
        msg = self._lib.ba_last_error(self._h).decode()
        if rc == capi.BA_ERR_SINGULAR:
            raise SingularPointBlock(msg)
        if rc == capi.BA_ERR_INVALID_ARG:
            raise ValueError(msg)
        raise capi.HipDeviceError('libpysfm_ba error %d: %s' % (rc, msg))

    def close(self):
        if getattr(self, '_h', None):
            self._lib.ba_destroy(self._h)
            self._h = None
        self._S_t = self._b_t = self._trial_t = None
        self._dense = self._dense_keep = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self):
        self._check(self._lib.ba_synchronize(self._h))

    def debug_poison(self):
        """Test aid (ba_debug_poison): NaNs into every LDS and workspace buffer, cached intermediates forgotten."""
        self._check(self._lib.ba_debug_poison(self._h))

    def set_option(self, name, value):
        """Test / measurement switch of the library (include/pysfm_ba.h ba_set_option); the defaults are the
        product path.  Options that shape the work lists take effect at the next set_problem."""
        if isinstance(value, bool):
            value = '1' if value else '0'
        self._check(self._lib.ba_set_option(self._h, str(name).encode(), str(value).encode()))
        self._options = dict(getattr(self, '_options', {}), **{str(name): str(value)})
        if name == 'device_lu':
            self.device_lu = str(value) not in ('0', 'False')

    # ---------------------------------------------------------------- problem
    def set_min_half_bandwidth(self, min_hb):
        """Lower bound for the band width of the next set_problem (ranks of the sharded adjuster
        must store [S | b] in one common layout)."""
        self._check(self._lib.ba_set_min_half_bandwidth(self._h, int(min_hb)))

    def set_camera_layout(self, new_pos):
        """Impose the layout of the optimised cameras for the following set_problem calls (ba_set_camera_layout); None: the
        library chooses again.  The ranks of a sharded adjuster share one layout."""
        if new_pos is None:
            self._check(self._lib.ba_set_camera_layout(self._h, None, 0))
            return
        new_pos = capi.i32(new_pos)
        self._check(self._lib.ba_set_camera_layout(self._h, capi.iptr(new_pos), len(new_pos)))

    def camera_layout(self):
        """(new_pos[nco], band_cameras) of the problem as ba_set_problem laid it out (ba_get_camera_layout): the caller's optimised
        position p sits at internal position new_pos[p]; the first band_cameras internal positions form the band, the border
        cameras follow.  The device views of the reduced system (reduced_tensors) are in that internal order."""
        new_pos = np.zeros(max(1, self.nco), np.int32)
        band = C.c_int32(0)
        self._check(self._lib.ba_get_camera_layout(self._h, capi.iptr(new_pos), C.byref(band)))
        return new_pos[:self.nco], band.value

    def set_pattern_lists(self, list_off=None, list_pos=None):
        """ba_set_pattern_lists: the camera lists of ALL tracks of a sharded scene (optimised positions, the caller's order) for the
        following set_problem calls, so that every rank's list of blocks is the same; None: forget them."""
        if list_off is None or len(list_off) < 2:
            self._check(self._lib.ba_set_pattern_lists(self._h, 0, None, None))
            return
        off, pos = capi.i32(list_off), capi.i32(list_pos)
        self._check(self._lib.ba_set_pattern_lists(self._h, len(off) - 1, capi.iptr(off), capi.iptr(pos)))

    def pcg_info(self):
        """ba_pcg_info: the blocks of S (upper triangle) the tracks define, iterations and ||r|| / ||b|| of the last solve by
        conjugate gradients (csrc/ba_pcg.h), the fraction of the band those blocks fill."""
        blocks, it, rel, fill = C.c_int64(0), C.c_int32(0), C.c_double(0.), C.c_double(0.)
        self._check(self._lib.ba_pcg_info(self._h, C.byref(blocks), C.byref(it), C.byref(rel), C.byref(fill)))
        return dict(blocks=blocks.value, iterations=it.value, rel_residual=rel.value, band_fill=fill.value)

    def plan_camera_layout(self, nco, list_off, list_pos, list_points=None, allow_border=False):
        """ba_plan_camera_layout (a pure function: no GPU work): (new_pos, band_cameras, half_bandwidth)."""
        list_off, list_pos = capi.i32(list_off), capi.i32(list_pos)
        pts = None if list_points is None else capi.i32(list_points)
        new = np.empty(int(nco), np.int32)
        n1, hb = C.c_int32(), C.c_int32()
        self._check(self._lib.ba_plan_camera_layout(int(nco), len(list_off) - 1, capi.iptr(list_off), capi.iptr(list_pos), capi.iptr(pts), int(bool(allow_border)),
                                                    capi.iptr(new), C.cast(C.byref(n1), capi._ip), C.cast(C.byref(hb), capi._ip)))
        return new, n1.value, hb.value

    def set_problem(self, nc, nt, obs_cam, obs_pt, obs_z, K, cam_opt_pos, pt_opt):
        obs_cam, obs_pt = capi.i32(obs_cam), capi.i32(obs_pt)
        obs_z = capi.f64(obs_z, (-1, 2))
        K = capi.f64(K, (3, 3))
        cam_opt_pos = capi.i32(cam_opt_pos)
        pt_opt = np.ascontiguousarray(pt_opt, dtype=np.uint8)
        assert len(obs_cam) == len(obs_pt) == len(obs_z)
        assert len(cam_opt_pos) == nc and len(pt_opt) == nt
        self._check(self._lib.ba_set_problem(self._h, nc, nt, len(obs_cam), capi.iptr(obs_cam),
                                             capi.iptr(obs_pt), capi.dptr(obs_z), capi.dptr(K),
                                             capi.iptr(cam_opt_pos), capi.bptr(pt_opt)))
        self.nc, self.nt, self.nobs = int(nc), int(nt), len(obs_cam)
        nco, hb, nS = C.c_int32(), C.c_int32(), C.c_int64()
        self._check(self._lib.ba_reduced_layout(self._h, C.byref(nco), C.byref(hb), C.byref(nS)))
        self.nco, self.half_bandwidth, self.S_doubles = nco.value, hb.value, nS.value
        self._S_t = self._b_t = None
        if self._torch is not None:
            self._bind_reduced()
        self._bind_dense()
        if self.poison_after_set_problem:
            self.debug_poison()

    def problem_info(self):
        """What ba_set_problem made of the scene (include/pysfm_ba.h BA_INFO_*), as a dict."""
        out = (C.c_int64 * len(capi.INFO_KEYS))()
        self._check(self._lib.ba_problem_info(self._h, out, len(capi.INFO_KEYS)))
        return dict(zip(capi.INFO_KEYS, [int(v) for v in out]))

    def _bind_reduced(self):
        torch = self._torch
        dev = torch.device('cuda', self.device)
        nS, nb = max(1, self.S_doubles), max(1, self.nco * 6)
        # (the sliding-window caller sets problem after problem of one shape: the same tensor serves them all)
        if getattr(self, '_Sb_shape', None) != (nS, nb):
            self._Sb_t = torch.empty(nS + nb, dtype=torch.float64, device=dev)   # [S | b]: one collective
            self._Sb_shape = (nS, nb)
        self._S_t, self._b_t = self._Sb_t[:nS], self._Sb_t[nS:]
        self._check(self._lib.ba_bind_reduced_buffers(self._h, C.c_void_p(self._S_t.data_ptr()),
                                                      C.c_void_p(self._b_t.data_ptr())))

    def set_sensor(self, kind, params):
        p = capi.f64(params).reshape(-1)
        self._check(self._lib.ba_set_sensor(self._h, int(kind), capi.dptr(p), len(p)))

    def set_params(self, which, R, t, X):
        R, t, X = capi.f64(R, (-1, 9)), capi.f64(t, (-1, 3)), capi.f64(X, (-1, 3))
        assert len(R) == len(t) == self.nc and len(X) == self.nt
        self._check(self._lib.ba_set_params(self._h, which, capi.dptr(R), capi.dptr(t), capi.dptr(X)))

    def get_params(self, which):
        R = np.empty((self.nc, 3, 3))
        t = np.empty((self.nc, 3))
        X = np.empty((self.nt, 3))
        self._check(self._lib.ba_get_params(self._h, which, capi.dptr(R), capi.dptr(t), capi.dptr(X)))
        return R, t, X

    def swap_params(self):
        self._check(self._lib.ba_swap_params(self._h))

    # ---------------------------------------------------------------- evaluation
    def cost(self, which):
        out = C.c_double()
        self._check(self._lib.ba_cost(self._h, which, C.byref(out)))
        return out.value

    def eval_observations(self, which, e=True, r=True, Jc=True, Jp=True):
        N = self.nobs
        oe = np.empty((N, 2)) if e else None
        orr = np.empty((N, 2)) if r else None
        oJc = np.empty((N, 2, 6)) if Jc else None
        oJp = np.empty((N, 2, 3)) if Jp else None
        self._check(self._lib.ba_eval_observations(self._h, which, capi.dptr(oe), capi.dptr(orr),
                                                   capi.dptr(oJc), capi.dptr(oJp)))
        return dict(e=oe, r=orr, Jc=oJc, Jp=oJp)

    def eval_sensor(self, e):
        e = capi.f64(e, (-1, 2))
        r = np.empty_like(e)
        J = np.empty((len(e), 2, 2))
        self._check(self._lib.ba_eval_sensor(self._h, len(e), capi.dptr(e), capi.dptr(r), capi.dptr(J)))
        return r, J

    # ---------------------------------------------------------------- normal equations
    def linearize(self, which, store_W=False):
        self._check(self._lib.ba_linearize(self._h, which, int(bool(store_W))))

    def get_blocks(self, W=False):
        HCC = np.empty((self.nc, 6, 6))
        bC = np.empty((self.nc, 6))
        HPP = np.empty((self.nt, 3, 3))
        bP = np.empty((self.nt, 3))
        Wa = np.empty((self.nobs, 6, 3)) if W else None
        self._check(self._lib.ba_get_blocks(self._h, capi.dptr(HCC), capi.dptr(bC), capi.dptr(HPP),
                                            capi.dptr(bP), capi.dptr(Wa)))
        return dict(HCC=HCC, bC=bC, HPP=HPP, bP=bP, W=Wa)

    def schur(self, which, damping, rcond):
        """rcond None -> plain inverse (SCHUR_COMPLIMENT_PINV_THRESHOLD = None)."""
        self._check(self._lib.ba_schur(self._h, which, float(damping), -1.0 if rcond is None else float(rcond)))

    def _bind_dense(self):
        """Dense-visibility reduction (include/pysfm_ba.h ba_set_dense_visibility): when the band is too wide
        for the cyclic reduction anyway and most (camera, track) pairs are observed, ba_schur forms the
        reduction as one symmetric matrix product on the matrix cores."""
        self._dense = False
        if self.nco == 0 or self.nt == 0:
            return
        fill = self.nobs / float(self.nco * self.nt)
        words = 3 * self.nt * 6 * self.nco
        self._dense = bool(self.half_bandwidth > DENSE_MIN_HALF_BANDWIDTH and fill >= DENSE_MIN_FILL
                           and 8 * words <= DENSE_MAX_BYTES)
        self._check(self._lib.ba_set_dense_visibility(self._h, int(self._dense)))

    def reduced_tensors(self):
        """torch views (S_band[nco*(hb+1)*36], b[nco*6]) of the device-resident reduced
        system in block-band layout (include/pysfm_ba.h ba_reduced_layout)."""
        if self._S_t is None:
            raise capi.HipDeviceError('reduced_tensors needs torch with a visible GPU')
        return self._S_t, self._b_t

    def reduced_payload(self):
        """[S_blocks | b] as ONE contiguous torch tensor: the all-reduce payload."""
        self.reduced_tensors()
        return self._Sb_t

    def get_reduced(self):
        S = np.empty((self.nco, self.nco, 6, 6))
        b = np.empty((self.nco, 6))
        self._check(self._lib.ba_get_reduced(self._h, capi.dptr(S), capi.dptr(b)))
        return S, b

    def get_point_inverses(self):
        out = np.empty((self.nt, 3, 3))
        self._check(self._lib.ba_get_point_inverses(self._h, capi.dptr(out)))
        return out

    def solve_reduced(self, cam_param_mask=None):
        """Solve the reduced camera system with the masked camera parameters deleted
        (solve_motion_normal_eqns, bundle_adjuster.py:281-312), entirely on the device (ba_solve_reduced): Cholesky by
        band shape (last_solve_kind 'bcr', 'bcr_wide', 'bcr_big', 'band', 'dense_cholesky'), and - when the system is not
        positive definite - LU with partial pivoting, the reference's own factorisation ('bcr_lu' for nodes of up to 11
        cameras, 'band_lu' otherwise; last_solve_path 'lu'; band + border systems too: csrc/ba_border.hip border_solve_lu).  The solution stays on the device for backsubstitute();
        get_solution() fetches it.  Raises ReducedSystemSingular where the reference's solve raises (an exactly zero pivot)."""
        n = self.nco * 6
        mask = None if cam_param_mask is None else np.ascontiguousarray(cam_param_mask, dtype=np.uint8)
        assert mask is None or mask.shape == (n,)
        info = C.c_int32(0)
        self._check(self._lib.ba_solve_reduced(self._h, capi.bptr(mask), C.byref(info)))
        if info.value == capi.SOLVE_TIMED_OUT:
            import warnings
            warnings.warn('pysfm_amd: the device solve of the reduced system timed out (status 0x%x): a solver fault, '
                          'not a property of the matrix; solving through LU' % info.value, RuntimeWarning)
            before = getattr(self, '_options', {}).get('solver', 'auto')      # (a caller's / test's own choice comes back afterwards)
            self._check(self._lib.ba_set_option(self._h, b'solver', b'lu'))
            try:
                self._check(self._lib.ba_solve_reduced(self._h, capi.bptr(mask), C.byref(info)))
            finally:
                self._check(self._lib.ba_set_option(self._h, b'solver', before.encode()))
        self._note_solve(info.value)
        if info.value != 0:
            raise ReducedSystemSingular

    def get_solution(self):
        """dC[nco,6] of the last solve_reduced()."""
        dC = np.empty((self.nco, 6))
        self._check(self._lib.ba_get_solution(self._h, capi.dptr(dC)))
        return dC

    def backsubstitute(self, which, dC=None, fetch=True):
        """dC None: use the solution of solve_reduced()."""
        if dC is not None:
            dC = capi.f64(dC, (-1, 6))
            assert len(dC) == self.nco
        dP = np.empty((self.nt, 3)) if fetch else None
        self._check(self._lib.ba_backsubstitute(self._h, which, capi.dptr(dC), capi.dptr(dP)))
        return dP

    def apply_update(self, src, dst, motion=None, structure=None):
        if motion is not None:
            motion, structure = capi.f64(motion, (-1, 6)), capi.f64(structure, (-1, 3))
            assert len(motion) == self.nco and len(structure) == self.nt
        self._check(self._lib.ba_apply_update(self._h, src, dst, capi.dptr(motion), capi.dptr(structure)))

    def lm_trial(self, damping, rcond, cam_param_mask=None):
        """One LM trial as a single batch of launches with one synchronisation
        (ba_lm_trial).  Returns (info, next_cost); info != 0 means the device solve did
        not apply (band too wide / not SPD) and the caller must take the stepwise path."""
        mask = None if cam_param_mask is None else np.ascontiguousarray(cam_param_mask, dtype=np.uint8)
        cost, info = C.c_double(), C.c_int32()
        self._check(self._lib.ba_lm_trial(self._h, float(damping), -1.0 if rcond is None else float(rcond),
                                          capi.bptr(mask), C.byref(cost), C.byref(info)))
        self._note_solve(info.value)
        return info.value, cost.value

    def lm_resident_fits(self):
        """ba_lm_resident_fits: does the problem (and its sensor model, and the options) fit the one-workgroup loop?"""
        return bool(self._lib.ba_lm_resident_fits(self._h))

    def lm_resident(self, max_steps, steps_taken, in_step, converged, damping, improvement_threshold, rcond, cur_cost, cam_param_mask=None):
        """ba_lm_resident: the loop of optimize() / step() on the device, from the given state of the schedule.
        Returns the log (capi.ResidentLog); the current parameter set has moved iff log.accepted."""
        if getattr(self, '_res_log', None) is None:
            self._res_log = capi.ResidentLog()
        log = self._res_log
        mask = None if cam_param_mask is None else np.ascontiguousarray(cam_param_mask, dtype=np.uint8)
        assert mask is None or mask.shape == (self.nco * 6,)
        self._check(self._lib.ba_lm_resident(self._h, int(max_steps), int(steps_taken), int(bool(in_step)), int(bool(converged)),
                                             float(damping), float(improvement_threshold), -1.0 if rcond is None else float(rcond),
                                             -1.0 if cur_cost is None else float(cur_cost), capi.bptr(mask), C.byref(log)))
        return log

    def lm_resident_begin(self, max_steps, steps_taken, in_step, converged, damping, improvement_threshold, rcond, cur_cost, cam_param_mask=None):
        """ba_lm_resident_begin: lm_resident's launch and nothing that waits for it; lm_resident_end() collects the log.  Nothing
        else may be asked of this backend in between."""
        mask = None if cam_param_mask is None else np.ascontiguousarray(cam_param_mask, dtype=np.uint8)
        assert mask is None or mask.shape == (self.nco * 6,)
        self._check(self._lib.ba_lm_resident_begin(self._h, int(max_steps), int(steps_taken), int(bool(in_step)), int(bool(converged)),
                                                   float(damping), float(improvement_threshold), -1.0 if rcond is None else float(rcond),
                                                   -1.0 if cur_cost is None else float(cur_cost), capi.bptr(mask)))

    def lm_resident_end(self):
        """ba_lm_resident_end: waits for the launch of lm_resident_begin; the log as lm_resident returns it."""
        if getattr(self, '_res_log', None) is None:
            self._res_log = capi.ResidentLog()
        self._check(self._lib.ba_lm_resident_end(self._h, C.byref(self._res_log)))
        return self._res_log

    def lm_resident_debug(self):
        """ba_lm_resident_debug (option solve_trace): [S | b] and dC of the first trial of the last lm_resident."""
        n = 6 * self.nco
        S, b, dC = np.empty((n, n)), np.empty(n), np.empty(n)
        self._check(self._lib.ba_lm_resident_debug(self._h, capi.dptr(S), capi.dptr(b), capi.dptr(dC)))
        return S, b, dC

    # the same trial in two halves around the all-reduce of the sharded adjuster
    def lm_trial_begin(self, damping, rcond):
        """ba_lm_trial_begin: linearise + Schur reduction of this rank's shard, nothing read back."""
        self._check(self._lib.ba_lm_trial_begin(self._h, float(damping), -1.0 if rcond is None else float(rcond)))

    def lm_trial_end(self, cam_param_mask=None):
        """ba_lm_trial_end: solve, back-substitute, update the trial set, trial cost - nothing read
        back; the rank's cost partials + status words are in trial_result().  Returns the
        pre-check (non-zero: the band solver does not apply, take the stepwise path)."""
        if getattr(self, '_trial_t', None) is None:
            torch = self._torch
            self._trial_t = torch.zeros(capi.TRIAL_PARTIALS + 2, dtype=torch.float64, device=torch.device('cuda', self.device))
            self._check(self._lib.ba_bind_trial_result(self._h, C.c_void_p(self._trial_t.data_ptr())))
        mask = None if cam_param_mask is None else np.ascontiguousarray(cam_param_mask, dtype=np.uint8)
        pre = C.c_int32()
        self._check(self._lib.ba_lm_trial_end(self._h, capi.bptr(mask), C.byref(pre)))
        self._note_solve(pre.value)
        return pre.value

    # ---- the reduced solve spread over the ranks (include/pysfm_ba.h ba_dist_*; pysfm_amd/csrc/ba_dist.h)
    dist_on = False

    def dist_plan(self, nco, half_bandwidth, nranks):
        """(cameras per node, nodes, nodes per rank) of the cut the library would make, or None when the distributed
        solve does not apply (ba_dist_plan: a pure function of its arguments, the same on every rank)."""
        cb, N, P = C.c_int32(), C.c_int32(), C.c_int32()
        if self._lib.ba_dist_plan(int(nco), int(half_bandwidth), int(nranks), C.byref(cb), C.byref(N), C.byref(P)) != capi.BA_OK:
            return None
        return cb.value, N.value, P.value

    def dist_enable(self, rank, nranks):
        """Switch the trial to the distributed solve (nranks <= 1: off).  Returns ba_dist_info as a dict."""
        self._check(self._lib.ba_dist_enable(self._h, int(rank), int(nranks)))
        out = (C.c_int64 * len(capi.DIST_INFO_KEYS))()
        self._check(self._lib.ba_dist_info(self._h, out, len(capi.DIST_INFO_KEYS)))
        info = dict(zip(capi.DIST_INFO_KEYS, [int(v) for v in out]))
        self.dist_on = bool(info['on'])
        self._dist_info = info
        self._dist_t = None
        if self.dist_on and self._torch is not None:
            # the exchange buffer as a torch tensor (the caller's collectives sum it in place between the stages)
            torch = self._torch
            n = max(info['exchange1_doubles'], info['exchange2_doubles'], info['exchange3_doubles'])
            self._dist_t = torch.zeros(n, dtype=torch.float64, device=torch.device('cuda', self.device))
            self._check(self._lib.ba_dist_bind_exchange(self._h, C.c_void_p(self._dist_t.data_ptr()), n))
        return info

    def dist_stage(self, stage, cam_param_mask=None):
        """ba_dist_stage: returns the slice of the exchange tensor to be summed over the ranks before the next stage
        (None after stage 4)."""
        mask = None if cam_param_mask is None else np.ascontiguousarray(cam_param_mask, dtype=np.uint8)
        n = C.c_int64()
        self._check(self._lib.ba_dist_stage(self._h, int(stage), capi.bptr(mask), C.byref(n)))
        return self._dist_t[:n.value] if n.value else None

    def lm_trial_finish(self):
        """ba_lm_trial_finish: back-substitution, trial set and trial cost once the solution is on the device; the
        rank's record is in trial_result()."""
        if getattr(self, '_trial_t', None) is None:
            torch = self._torch
            self._trial_t = torch.zeros(capi.TRIAL_PARTIALS + 2, dtype=torch.float64, device=torch.device('cuda', self.device))
            self._check(self._lib.ba_bind_trial_result(self._h, C.c_void_p(self._trial_t.data_ptr())))
        self._check(self._lib.ba_lm_trial_finish(self._h))
        self._note_solve(0)

    def _note_solve(self, info):
        """last_solve_kind: the device solver whose state ba_solve_reduced left ('bcr', 'bcr_wide', 'band', 'dense_cholesky',
        'bcr_big'; 'bcr_lu' / 'band_lu' = LU with partial pivoting after the Cholesky solver found the system not positive
        definite, or by option); last_solve_path: 'lu' for those two, 'dense_cholesky' or 'band' for the Cholesky solvers."""
        self.last_solve_kind = capi.SOLVE_KINDS[self._lib.ba_last_solve_kind(self._h)]
        self.last_solve_path = 'lu' if self.last_solve_kind in ('bcr_lu', 'band_lu') else \
            'dense_cholesky' if self.last_solve_kind == 'dense_cholesky' else 'pcg' if self.last_solve_kind == 'pcg' else 'band'

    def trial_result(self):
        """Device tensor [TRIAL_PARTIALS cost partials | singular point blocks | solver status]."""
        return self._trial_t

    def triangulate(self, which, rcond=None, fetch=True):
        """Linear least-squares re-initialisation of every point from the cameras of
        parameter set `which` (Bundle.triangulate_all, bundle.py:320-321)."""
        X = np.empty((self.nt, 3)) if fetch else None
        self._check(self._lib.ba_triangulate(self._h, which, -1.0 if rcond is None else float(rcond), capi.dptr(X)))
        return X

    # ---------------------------------------------------------------- instrumentation
    # ---- the shards' collectives inside the library (include/pysfm_ba.h ba_comm_*)
    direct_comm = False

    def comm_load(self, librccl_path):
        """Resolve RCCL from the given shared object (the one torch has loaded).  False if that fails."""
        return self._lib.ba_comm_load(librccl_path.encode() if librccl_path else None) == 0

    def comm_unique_id(self):
        buf = C.create_string_buffer(128)
        if self._lib.ba_comm_unique_id(buf) != 0:
            raise capi.HipDeviceError('ba_comm_unique_id failed')
        return buf.raw

    def comm_attach(self, id128, rank, nranks):
        assert len(id128) == 128
        self._check(self._lib.ba_comm_init(self._h, C.c_char_p(id128), int(rank), int(nranks)))
        self.direct_comm = True

    def comm_detach(self):
        self._check(self._lib.ba_comm_destroy(self._h))
        self.direct_comm = False

    def comm_allreduce_reduced(self):
        self._check(self._lib.ba_comm_allreduce_reduced(self._h))

    def comm_allreduce_sum(self, values):
        v = np.ascontiguousarray(values, dtype=np.float64).copy()
        self._check(self._lib.ba_comm_allreduce_sum(self._h, capi.dptr(v), v.size))
        return v

    def measure_copy_bandwidth(self, nbytes=1 << 30, repeats=10):
        """GB/s (read + write) of a streaming copy on this device: the achievable HBM rate."""
        out = C.c_double()
        self._check(self._lib.ba_measure_copy_bandwidth(self._h, int(nbytes), int(repeats), C.byref(out)))
        return out.value

    def enable_timing(self, on=True, only=None, stride=1):
        """HIP-event timing of the kernels; `only` = iterable of kernel names to bracket
        (each event pair costs a few microseconds of stream time)."""
        mask = (1 << 64) - 1
        if only is not None:
            mask = 0
            for name in only:
                mask |= 1 << capi.KERNEL_IDS.index(name)
        self._check(self._lib.ba_set_timing_mask(self._h, C.c_uint64(mask)))
        self._check(self._lib.ba_set_timing_stride(self._h, int(stride)))
        self._check(self._lib.ba_enable_timing(self._h, int(bool(on))))

    def timings(self, reset=False):
        ms = np.zeros(capi.K_COUNT)
        n = np.zeros(capi.K_COUNT, dtype=np.int64)
        self._check(self._lib.ba_get_timings(self._h, capi.dptr(ms), n.ctypes.data_as(C.POINTER(C.c_int64)),
                                             int(bool(reset))))
        return {name: dict(ms=float(ms[i]), launches=int(n[i])) for i, name in enumerate(capi.KERNEL_IDS)}


_default = {}


def default_backend(device=0):
    """Process-wide backend used by the Bundle / sensor-model convenience methods."""
    if device not in _default:
        _default[device] = HipBackend(device)
    return _default[device]
