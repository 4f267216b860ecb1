"""MultiGaussianDist on the device: /root/reference/riemann/models/gaussian.py:21-66."""
import ctypes as C

import numpy as np

from .. import _lib
from ..sampling_errors import ParameterError
from .model import DeviceModel


class MultiGaussianDist(DeviceModel):
    """A multivariate Gaussian with known mean and covariance (gaussian.py:21-66)."""

    def __init__(self, mu, C_):
        mu = np.atleast_1d(np.asarray(mu, dtype=np.float64))
        Cm = np.atleast_2d(np.asarray(C_, dtype=np.float64))
        if Cm.shape[0] != Cm.shape[1]:
            raise ParameterError("C has non-square shape {}".format(Cm.shape))  # gaussian.py:35-36
        if Cm.shape[1] != mu.shape[0]:
            raise ParameterError("mu and C have incompatible shapes {}, {}"
                                 .format(mu.shape, Cm.shape))                   # gaussian.py:37-39
        self.mu = mu
        self.L = np.linalg.cholesky(Cm)                                         # gaussian.py:42
        self.logdetC = 2 * np.sum(np.log(np.diag(self.L)))
        self.Ndim = self.Npars = len(mu)
        h = C.c_void_p()
        _lib.call("rmh_model_gaussian_create", _lib.as_f64(mu), _lib.as_f64(self.L), self.Ndim,
                  C.byref(h))
        self._h = h

    def grad_log_likelihood(self, theta):
        """-C^-1 (theta - mu) (gaussian.py:54-58); log_prior is 0, so this is
        also the gradient of the log posterior."""
        return self.grad_log_posterior(theta)

    def draw(self, Ndraws=1):
        """Host-side draws with numpy's generator, as gaussian.py:60-66."""
        epsilon = np.random.normal(size=(Ndraws, self.mu.shape[0]))
        return np.dot(epsilon, self.L.T) + self.mu
