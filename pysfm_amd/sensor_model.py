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


# ---- any isotropic robustifier, sampled (the reference's plug-in point: an object with the four methods, sensor_model.py:19-32)
TABLE_LOG2_RHO_MIN, TABLE_LOG2_RHO_MAX, TABLE_NODES_PER_OCTAVE = -40., 40., 256


def tabulate(model):
    """(kind, params) of BA_SENSOR_TABLE for a model whose residual is isotropic, r = h(|e|) e.  Along e = (rho, 0) the
    model's own Jacobian is diag(h + h' rho, h): h and h' come from Jresidual_from_error, no differencing.  Sampled on a grid
    uniform in log2(rho), 256 nodes per octave from 2^-40 to 2^40, which the device interpolates (cubic Hermite: h to ~1e-13,
    its derivative to ~3e-10 where h is smooth).  Raises TypeError for a model that is not isotropic - its Jacobian is then
    not of the form h I + (h'/rho) e e^T, which is all the device evaluates."""
    n = int((TABLE_LOG2_RHO_MAX - TABLE_LOG2_RHO_MIN) * TABLE_NODES_PER_OCTAVE) + 1
    u = TABLE_LOG2_RHO_MIN + np.arange(n) / float(TABLE_NODES_PER_OCTAVE)
    rho = np.exp2(u)
    tab = np.empty((n, 2))
    for i in range(n):
        J = np.asarray(model.Jresidual_from_error(np.array([rho[i], 0.])), float)
        h = J[1, 1]
        tab[i, 0] = h
        tab[i, 1] = (J[0, 0] - h) * np.log(2.)              # dh/du = h'(rho) rho ln 2,  h' rho = J00 - J11
    if not np.all(np.isfinite(tab)):
        raise TypeError('sensor model %r: residual Jacobian is not finite on [2^%g, 2^%g]' % (model, TABLE_LOG2_RHO_MIN, TABLE_LOG2_RHO_MAX))
    # isotropy: at a few errors off the axes, r = h(|e|) e and J = h I + (h'/rho) e e^T with the sampled h, h'
    rs = np.random.RandomState(7)
    for scale in (1e-3, 3e-2, 1., 40.):
        e = rs.randn(2) * scale
        r_ = np.sqrt(e.dot(e))
        i = int(round((np.log2(r_) - TABLE_LOG2_RHO_MIN) * TABLE_NODES_PER_OCTAVE))
        en = e * (rho[i] / r_)                                # the same direction, on a node
        h, hp_rho = tab[i, 0], tab[i, 1] / np.log(2.)
        want_r = h * en
        want_J = h * np.eye(2) + (hp_rho / (rho[i] * rho[i])) * np.outer(en, en)
        got_r = np.asarray(model.residual_from_error(en), float)
        got_J = np.asarray(model.Jresidual_from_error(en), float)
        tol = 1e-9 * max(abs(h), 1e-300)
        if np.max(np.abs(got_r - want_r)) > tol * rho[i] or np.max(np.abs(got_J - want_J)) > 1e-7 * max(np.max(np.abs(want_J)), 1e-300):
            raise TypeError('sensor model %r is not isotropic (r = h(|e|) e): it has no device form' % (model,))
    return capi.SENSOR_TABLE, np.concatenate(([TABLE_LOG2_RHO_MIN, float(TABLE_NODES_PER_OCTAVE), float(n)], tab.reshape(-1)))


_FINGERPRINT_RADII = (2. ** -20, 1e-3, .05, .7, 1., 3.1, 40., 2. ** 20)


def _fingerprint(model):
    """The entries of the model's Jacobian along e = (rho, 0) at a few radii: what the table is built from."""
    out = []
    for rho in _FINGERPRINT_RADII:
        J = np.asarray(model.Jresidual_from_error(np.array([rho, 0.])), float)
        out += [float(J[0, 0]), float(J[1, 1])]
    return tuple(out)


def device_params_of(model):
    """(kind, params) for our models, for duck-typed reference models (classes named GaussianModel with .L / CauchyModel
    with .sigma), and - the reference's plug-in point, sensor_model.py:19-32 / bundle.py:269-273 - for ANY object with
    residual_from_error / Jresidual_from_error whose residual is isotropic: sampled once into a table (tabulate; cached on
    the object) that the kernels interpolate."""
    if hasattr(model, 'device_params'):
        return model.device_params()
    name = type(model).__name__
    if name == 'GaussianModel' and hasattr(model, 'L'):
        return capi.SENSOR_GAUSS, np.asarray(model.L, float).reshape(4)
    if name == 'CauchyModel' and hasattr(model, 'sigma'):
        return capi.SENSOR_CAUCHY, np.array([model.sigma], float)
    if name == 'HuberModel' and hasattr(model, 'k'):
        return capi.SENSOR_HUBER, np.array([model.k], float)
    if hasattr(model, 'residual_from_error') and hasattr(model, 'Jresidual_from_error'):
        # cached on the object WITH a fingerprint of the model as it is now - h and h' rho at a few radii - so that a model
        # whose parameters the caller changed afterwards (sigma = ...), or a clone() that carried the attribute along, is
        # sampled again instead of running the device on the old robustifier
        fp = _fingerprint(model)
        cached = getattr(model, '_pysfm_amd_table', None)
        if cached is None or cached[0] != fp:
            cached = (fp, tabulate(model))
            try:
                model._pysfm_amd_table = cached
            except AttributeError:
                pass
        return cached[1]
    raise TypeError('sensor model %r has no device form: it needs residual_from_error and Jresidual_from_error' % (model,))


class TabulatedModel(_DeviceSensorModel):
    """Device-evaluated view of a caller-defined isotropic model: residual_from_error / Jresidual_from_error go through
    the interpolated table on the GPU (ba_eval_sensor) - what the kernels see of the model - cost_from_error is the
    model's own.  `validate(TabulatedModel(m))` is the reference's self-check (sensor_model.py:76-99) of the device form."""

    def __init__(self, model):
        self.model = model
        self._params = device_params_of(model)

    def device_params(self):
        return self._params

    def cost_from_error(self, x):
        return self.model.cost_from_error(x)

    def clone(self):
        return TabulatedModel(self.model.clone() if hasattr(self.model, 'clone') else self.model)


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
