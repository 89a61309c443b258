"""Pose helpers used by the sliding-window driver (geometry.py:5-24)."""
import numpy as np


def relative_pose(R0, t0, R1, t1):
    """(R01, t01) that takes the pose (R0, t0) to (R1, t1) (geometry.py:5-8)."""
    R_delta = np.dot(R1, R0.T)
    return R_delta, t1 - np.dot(R_delta, t0)


def propagate_pose_update(R0, t0, R0_updated, t0_updated, R1, t1):
    """Apply the update (R0,t0) -> (R0_updated,t0_updated) to (R1,t1) (geometry.py:13-17)."""
    R_delta = np.dot(R0_updated, R0.T)
    return np.dot(R_delta, R1), np.dot(R_delta, t1 - t0) + t0_updated


def propagate_pose_update_inplace(cam0, cam0_updated, cam1):
    """geometry.py:19-24."""
    R1, t1 = propagate_pose_update(cam0.R, cam0.t, cam0_updated.R, cam0_updated.t, cam1.R, cam1.t)
    cam1.R = R1
    cam1.t = t1


def _plane_rotation(i, j, th):
    """Rotation by `th` in the coordinate plane (i, j): entry (i, j) = -sin, (j, i) = +sin."""
    R = np.eye(3)
    R[i, i] = R[j, j] = np.cos(th)
    R[i, j] = -np.sin(th)
    R[j, i] = np.sin(th)
    return R


def rotation_xy(th):
    """geometry.py:28-31 (about z)."""
    return _plane_rotation(0, 1, th)


def rotation_xz(th):
    """geometry.py:33-36 (sign convention of the reference: entry (0, 2) = -sin)."""
    return _plane_rotation(0, 2, th)


def rotation_yz(th):
    """geometry.py:38-41 (about x)."""
    return _plane_rotation(1, 2, th)
