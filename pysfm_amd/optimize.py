"""Levenberg-Marquardt damping helpers with the reference's names (optimize.py:7-15).

``BundleAdjuster.apply_damping`` has the same effect on the device (the factor is folded into
``k_point_invert_schur_init`` and the reduction); these operate on host matrices for callers that damp their own.
The reference's generic dense LM classes (optimize.py:72-182) are outside the accelerated path."""
import numpy as np

from .algebra import *  # noqa: F401,F403  (the reference's optimize.py re-exports algebra)


def apply_lm_damping_inplace(A, damping):
    """diag(A) *= 1 + damping (optimize.py:7-9)."""
    A = np.asarray(A)
    A[np.diag_indices(A.shape[0])] *= (1. + damping)


def apply_lm_damping(A, damping):
    """Damped copy of A (optimize.py:11-15; the reference forgets to pass `damping` on and raises TypeError - this
    one does what its name says)."""
    B = np.array(A, copy=True)
    apply_lm_damping_inplace(B, damping)
    return B
