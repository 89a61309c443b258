"""Small host-side helpers with the names pysfm callers import (algebra.py:5-56).
They only shape data for the caller; the adjuster's arithmetic runs on the GPU."""
from functools import reduce

import numpy as np


def pr(x):
    """Project homogeneous vectors (rows of a matrix) - algebra.py:5-12."""
    x = np.asarray(x)
    if x.ndim == 1:
        return x[:-1] / x[-1]
    if x.ndim == 2:
        return x[:, :-1] / x[:, [-1]]
    raise Exception('Cannot pr() an array with %d dimensions' % x.ndim)


def unpr(x):
    """algebra.py:16-23."""
    x = np.asarray(x)
    if x.ndim == 1:
        return np.hstack((x, 1.))
    if x.ndim == 2:
        return np.hstack((x, np.ones((len(x), 1))))
    raise Exception('Cannot unpr() an array with %d dimensions' % x.ndim)


def prdot(H, X):
    """algebra.py:28-40."""
    H, X = np.asarray(H), np.asarray(X)
    assert H.ndim == 2, 'The shape of H was %s' % str(H.shape)
    if X.ndim == 1:
        assert len(X) == H.shape[1] - 1
        return pr(np.dot(H, unpr(X)))
    assert X.shape[1] == H.shape[1] - 1
    return pr(np.dot(unpr(X), H.T))


def dots(*m):
    """algebra.py:44-45."""
    return reduce(np.dot, m)


def ssq(x):
    """algebra.py:48-49."""
    return np.dot(x, x)


def skew(m):
    """algebra.py:51-56."""
    m = np.asarray(m)
    assert m.shape == (3,)
    return np.array([[0., -m[2], m[1]],
                     [m[2], 0., -m[0]],
                     [-m[1], m[0], 0.]])
