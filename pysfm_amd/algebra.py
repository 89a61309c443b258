"""Small projective / linear-algebra helpers with the names callers of pysfm import (algebra.py:5-56).

Data-model conveniences for code that builds or inspects scenes.  The adjuster never calls them: its projection,
products and cross-product matrices are inlined in the HIP kernels (csrc/ba_math.h)."""
from functools import reduce

import numpy as np

from .lie import skew  # noqa: F401  (algebra.py:51-56; one definition, in lie.py)


def pr(x):
    """Dehomogenise a vector, or every ROW of a matrix (algebra.py:5-12)."""
    x = np.asarray(x)
    if x.ndim not in (1, 2):
        raise Exception('Cannot pr() an array with %d dimensions' % x.ndim)
    return x[..., :-1] / x[..., -1:]


def unpr(x):
    """Append a homogeneous 1 to a vector, or to every row of a matrix (algebra.py:16-23)."""
    x = np.asarray(x)
    if x.ndim not in (1, 2):
        raise Exception('Cannot unpr() an array with %d dimensions' % x.ndim)
    return np.concatenate((x, np.ones(x.shape[:-1] + (1,))), axis=-1)


def prdot(H, X):
    """pr(H unpr(x)) for one point or for every row of X (algebra.py:28-39)."""
    H, X = np.asarray(H), np.asarray(X)
    assert H.ndim == 2, 'The shape of H was %s' % str(H.shape)
    assert X.ndim in (1, 2) and X.shape[-1] == H.shape[1] - 1, \
        'H.shape was %s, X.shape was %s' % (str(H.shape), str(X.shape))
    return pr(np.dot(unpr(X), H.T))


def dots(*m):
    """Product of any number of matrices, left to right (algebra.py:43-44)."""
    return reduce(np.dot, m)


def ssq(x):
    """Sum of squared elements of a vector (algebra.py:47-48)."""
    return np.dot(x, x)
