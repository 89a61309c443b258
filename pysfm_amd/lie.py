"""SO(3) helpers of the data model (lie.py:19-45).

``Camera.perturb`` uses these when a *caller* moves a single camera while building
a scene.  The adjuster itself never does: its parameter update is the
``k_apply_update`` HIP kernel (ba_math.h ``so3_exp``)."""
import numpy as np



def skew(m):
    """The cross-product matrix [m]x of a 3-vector (algebra.py:51-56): skew(m) @ v == cross(m, v)."""
    m = np.asarray(m, float)
    assert m.shape == (3,)
    return np.cross(m[None, :], -np.eye(3))


class SO3(object):
    @classmethod
    def exp(cls, m):
        """Rodrigues; identity when |m| < 1e-8 (lie.py:21-34)."""
        m = np.asarray(m, float)
        assert m.shape == (3,), 'shape was ' + str(m.shape)
        t = np.linalg.norm(m)
        if t < 1e-8:
            return np.eye(3)
        Kx = skew(m)
        return np.eye(3) + (np.sin(t) / t) * Kx + ((1. - np.cos(t)) / (t * t)) * np.dot(Kx, Kx)

    @classmethod
    def J_expm_x(cls, x):
        """d(exp(m) x)/dm at m = 0 (lie.py:38-40)."""
        return skew(-np.asarray(x, float))

    @classmethod
    def generator_field(cls, x):
        return skew(x)
