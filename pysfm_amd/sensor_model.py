"""Sensor models behind ``bundle.sensor_model`` (sensor_model.py:7-72).

Same four-method protocol as the reference (``cost_from_error``,
``residual_from_error``, ``Jresidual_from_error``, ``clone``).  Each model is a
*descriptor*: ``device_params()`` gives the (kind, params) the HIP kernels
evaluate (ba_math.h ``sensor_eval``).  ``residual_from_error`` /
``Jresidual_from_error`` run that same device function through
``ba_eval_sensor`` - there is no NumPy re-implementation to drift from it.
"""
import numpy as np

from . import _capi as capi


class _DeviceSensorModel(object):
    device = 0

    def device_params(self):
        raise NotImplementedError

    def _eval(self, x):
        from .backend import default_backend
        x = np.asarray(x, float)
        assert x.shape == (2,)
        be = default_backend(self.device)
        kind, params = self.device_params()
        be.set_sensor(kind, params)
        r, J = be.eval_sensor(x.reshape(1, 2))
        return r[0], J[0]

    def residual_from_error(self, x):
        return self._eval(x)[0]

    def Jresidual_from_error(self, x):
        return self._eval(x)[1]


class GaussianModel(_DeviceSensorModel):
    """cost = e^T cov^-1 e, residual = L e with L = chol(cov^-1) (sensor_model.py:7-32)."""

    def __init__(self, cov=1.):
        if np.isscalar(cov):
            self.cov = cov * np.eye(2)
        elif np.shape(cov) == (2,):
            self.cov = np.diag(cov)
        else:
            assert np.shape(cov) == (2, 2)
            self.cov = np.asarray(cov, float)
        self.covinv = np.linalg.inv(self.cov)
        self.L = np.linalg.cholesky(self.covinv)

    def device_params(self):
        return capi.SENSOR_GAUSS, np.asarray(self.L, float).reshape(4)

    def cost_from_error(self, x):
        x = np.asarray(x, float)
        assert x.shape == (2,)
        return float(np.dot(x, np.dot(self.covinv, x)))

    def clone(self):
        return GaussianModel(self.cov)


class CauchyModel(_DeviceSensorModel):
    """Vector residual with squared norm log(1 + |e|^2/sigma^2) (sensor_model.py:37-72)."""
    LinearWindowAboutZero = 1e-5

    def __init__(self, sigma):
        self.sigma = sigma
        self.sigmasqr = sigma * sigma

    def device_params(self):
        return capi.SENSOR_CAUCHY, np.array([self.sigma], float)

    def cost_from_error(self, x):
        x = np.asarray(x, float)
        return float(np.log(1. + np.dot(x, x) / self.sigmasqr))

    def clone(self):
        return CauchyModel(self.sigma)


class HuberModel(_DeviceSensorModel):
    """Huber robustifier under the same protocol (new: the reference has none).
    rho(s) = s^2 for s <= k, 2ks - k^2 beyond; residual = e sqrt(rho)/|e|."""

    def __init__(self, k):
        self.k = k

    def device_params(self):
        return capi.SENSOR_HUBER, np.array([self.k], float)

    def cost_from_error(self, x):
        x = np.asarray(x, float)
        s = np.sqrt(np.dot(x, x))
        return float(s * s if s <= self.k else 2. * self.k * s - self.k * self.k)

    def clone(self):
        return HuberModel(self.k)


def device_params_of(model):
    """(kind, params) for our models and for duck-typed reference models
    (classes named GaussianModel with .L / CauchyModel with .sigma)."""
    if hasattr(model, 'device_params'):
        return model.device_params()
    name = type(model).__name__
    if name == 'GaussianModel' and hasattr(model, 'L'):
        return capi.SENSOR_GAUSS, np.asarray(model.L, float).reshape(4)
    if name == 'CauchyModel' and hasattr(model, 'sigma'):
        return capi.SENSOR_CAUCHY, np.array([model.sigma], float)
    if name == 'HuberModel' and hasattr(model, 'k'):
        return capi.SENSOR_HUBER, np.array([model.k], float)
    raise TypeError('sensor model %r has no device form: use GaussianModel, CauchyModel or HuberModel' % (model,))


def validate(sensor_model):
    """The reference's self-check (sensor_model.py:76-99): cost == r.r, cost(0) == 0,
    analytic 2x2 Jacobian vs central differences."""
    error = np.asarray([1., 2.])
    cost = sensor_model.cost_from_error(error)
    residual = sensor_model.residual_from_error(error)
    assert np.abs(cost - np.dot(residual, residual)) < 1e-8, 'cost was not equal to residual.T * residual'
    cost_at_0 = sensor_model.cost_from_error([0, 0])
    assert np.isscalar(cost_at_0)
    assert abs(cost_at_0) < 1e-8, 'Cost at 0 must be 0'
    J = sensor_model.Jresidual_from_error(error)
    assert J.shape == (2, 2), 'shape was %s' % str(J.shape)
    h = 1e-6
    Jn = np.empty((2, 2))
    for c in range(2):
        d = np.zeros(2)
        d[c] = h
        Jn[:, c] = (sensor_model.residual_from_error(error + d) - sensor_model.residual_from_error(error - d)) / (2 * h)
    assert np.max(np.abs(J - Jn)) < 1e-5, 'Jacobian seems to be incorrect at ' + str(error)
    return True


def run_tests():
    """The reference's self-check entry point (sensor_model.py:104-110): every model of the package through validate()."""
    for name, model in (('Gaussian', GaussianModel([2., 3.])), ('Cauchy', CauchyModel(2.)), ('Huber', HuberModel(1.5))):
        print('Validating %s sensor model...' % name)
        validate(model)
