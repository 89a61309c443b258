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
