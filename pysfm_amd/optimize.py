"""Levenberg-Marquardt damping helper with the reference's name (optimize.py:7-9).
On the device the same rule is applied inside k_point_invert / k_schur_init."""
import numpy as np


def apply_lm_damping_inplace(A, damping):
    A = np.asarray(A)
    A[np.diag_indices(A.shape[0])] *= (1. + damping)


def apply_lm_damping(A, damping):
    B = np.array(A, dtype=float, copy=True)
    apply_lm_damping_inplace(B, damping)
    return B
