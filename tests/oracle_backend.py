"""OracleBackend - a TEST DOUBLE with HipBackend's interface, computed by the CPU
oracle (oracle/ba_oracle.py).

It exists so that the host logic of pysfm_amd (id / mask bookkeeping, the LM
schedule, sharding + the all-reduce path) can be exercised by the CPU-only test
suite.  It lives under tests/ and is never importable from the package: the
product has exactly one backend, the HIP one.
"""
import numpy as np
import torch

from oracle import ba_oracle as O
from pysfm_amd.backend import ReducedSystemSingular, SingularPointBlock


class OracleBackend(object):
    def __init__(self):
        self.p = [None, None]
        self.cur = 0
        self.sensor = O.Sensor(O.GAUSS)
        self.nc = self.nt = self.nco = self.nobs = 0
        self.calls = []

    # -- plumbing
    def close(self):
        pass

    def synchronize(self):
        pass

    def enable_timing(self, on=True):
        pass

    def timings(self, reset=False):
        return {}

    def _phys(self, which):
        return self.cur if which == 0 else 1 - self.cur

    # -- problem
    def set_problem(self, nc, nt, obs_cam, obs_pt, obs_z, K, cam_opt_pos, pt_opt):
        obs_pt = np.asarray(obs_pt, np.int32)
        if len(obs_pt) > 1 and np.any(np.diff(obs_pt) < 0):
            raise ValueError('observations must be ordered by track position')
        self.nc, self.nt, self.nobs = nc, nt, len(obs_cam)
        self.obs = (np.asarray(obs_cam, np.int32), obs_pt, np.asarray(obs_z, float).reshape(-1, 2))
        self.K = np.asarray(K, float).reshape(3, 3)
        self.cam_opt_pos = np.asarray(cam_opt_pos, np.int32)
        self.pt_opt = np.asarray(pt_opt).astype(bool)
        self.nco = int(np.sum(self.cam_opt_pos >= 0))
        self.p = [None, None]
        self.cur = 0
        self.Sb = torch.zeros(max(1, self.nco * self.nco * 36) + max(1, self.nco * 6), dtype=torch.float64)

    def set_sensor(self, kind, params):
        params = np.asarray(params, float).reshape(-1)
        if kind == O.GAUSS:
            self.sensor = O.Sensor(O.GAUSS, L=params.reshape(2, 2))
        elif kind == O.CAUCHY:
            self.sensor = O.Sensor(O.CAUCHY, sigma=params[0])
        else:
            self.sensor = O.Sensor(O.HUBER, k=params[0])

    def set_params(self, which, R, t, X):
        self.p[self._phys(which)] = (np.array(R, float).reshape(-1, 3, 3), np.array(t, float).reshape(-1, 3),
                                     np.array(X, float).reshape(-1, 3))

    def get_params(self, which):
        R, t, X = self.p[self._phys(which)]
        return R.copy(), t.copy(), X.copy()

    def swap_params(self):
        self.cur = 1 - self.cur

    def _args(self, which):
        R, t, X = self.p[self._phys(which)]
        return (self.sensor, self.K, R, t, X) + self.obs

    # -- evaluation
    def cost(self, which):
        self.calls.append('cost')
        return O.cost(*self._args(which), self.cam_opt_pos, self.pt_opt)

    def eval_observations(self, which, e=True, r=True, Jc=True, Jp=True):
        a = self._args(which)
        rr, jc, jp = O.jacobians(*a)
        return dict(e=O.reproj_error(*a[1:]) if e else None, r=rr if r else None,
                    Jc=jc if Jc else None, Jp=jp if Jp else None)

    def eval_sensor(self, e):
        e = np.asarray(e, float).reshape(-1, 2)
        return O.sensor_residual(self.sensor, e), O.sensor_jacobian(self.sensor, e)

    # -- normal equations
    def linearize(self, which, store_W=False):
        self.calls.append('linearize')
        self.blocks = O.normal_blocks(*self._args(which), self.nc, self.nt)

    def get_blocks(self, W=False):
        HCC, HPP, Wb, bC, bP = self.blocks
        return dict(HCC=HCC.copy(), bC=bC.copy(), HPP=HPP.copy(), bP=bP.copy(), W=Wb.copy() if W else None)

    def schur(self, which, damping, rcond):
        self.calls.append('schur')
        HCC, HPP, W, bC, bP = self.blocks
        try:
            self.HPP_inv = O.invert_point_blocks(O.damp_blocks(HPP, damping), rcond)
        except np.linalg.LinAlgError as e:
            raise SingularPointBlock(str(e))
        S, b = O.schur_complement(O.damp_blocks(HCC, damping), self.HPP_inv, W, bC, bP,
                                  self.obs[0], self.obs[1], self.cam_opt_pos)
        nS = self.nco * self.nco * 36
        self.Sb[:nS] = torch.from_numpy(S.reshape(-1))
        self.Sb[max(1, nS):max(1, nS) + self.nco * 6] = torch.from_numpy(b.reshape(-1))

    def reduced_payload(self):
        return self.Sb

    def reduced_tensors(self):
        nS = max(1, self.nco * self.nco * 36)
        return self.Sb[:nS], self.Sb[nS:]

    def get_reduced(self):
        S, b = self.reduced_tensors()
        return (S[:self.nco * self.nco * 36].numpy().reshape(self.nco, self.nco, 6, 6).copy(),
                b[:self.nco * 6].numpy().reshape(self.nco, 6).copy())

    def get_point_inverses(self):
        return self.HPP_inv.copy()

    def solve_reduced(self, cam_param_mask=None):
        self.calls.append('solve')
        S, b = self.get_reduced()
        A, rhs = O.flatten_reduced(S, b)
        n = self.nco * 6
        keep = np.arange(n) if cam_param_mask is None else np.nonzero(np.asarray(cam_param_mask))[0]
        try:
            x = np.linalg.solve(A[np.ix_(keep, keep)], rhs[keep])
        except np.linalg.LinAlgError:
            raise ReducedSystemSingular
        dC = np.zeros(n)
        dC[keep] = x
        self.dC = dC.reshape(-1, 6)

    def get_solution(self):
        return self.dC.copy()

    def backsubstitute(self, which, dC=None, fetch=True):
        self.calls.append('backsub')
        HCC, HPP, W, bC, bP = self.blocks
        if dC is not None:
            self.dC = np.asarray(dC, float).reshape(-1, 6)
        self.dP = O.backsubstitute(self.dC, self.HPP_inv, W, bP, self.obs[0], self.obs[1], self.cam_opt_pos, self.nt)
        return self.dP.copy() if fetch else None

    def triangulate(self, which, rcond=None, fetch=True):
        R, t, X = self.p[self._phys(which)]
        X2 = O.triangulate_all(self.K, R, t, self.obs[0], self.obs[1], self.obs[2], self.nt)
        self.p[self._phys(which)] = (R, t, X2)
        return X2.copy() if fetch else None

    def apply_update(self, src, dst, motion=None, structure=None):
        R, t, X = self.p[self._phys(src)]
        if motion is None:
            motion, structure = -self.dC, -self.dP
        structure = np.asarray(structure, float).reshape(-1, 3)[self.pt_opt]
        self.p[self._phys(dst)] = O.apply_update(R, t, X, np.asarray(motion, float).reshape(-1, 6), structure,
                                                 self.cam_opt_pos, self.pt_opt)
