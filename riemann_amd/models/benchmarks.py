"""Benchmark target distributions: /root/reference/riemann/models/benchmarks.py:12-85.
Built on first use (constructing a device Model needs the GPU)."""
import ctypes as C

import numpy as np

from .. import _lib
from ..sampling_errors import ParameterError
from .gaussian import MultiGaussianDist
from .model import DeviceModel


class MixtureDist(DeviceModel):
    """A mixture of distributions (benchmarks.py:31-73) on the device.

    ``components``: list of ``(weight, dist)`` as in the reference.  The device kernel
    is built for MultiGaussianDist components of one dimension; densities follow the
    reference operation for operation (no log-sum-exp: far from every component the
    likelihood underflows to ``-inf``).  ``grad_log_posterior`` is an addition (the reference's
    class has no gradient): it lets MALA / HMC proposals run on mixture targets."""

    def __init__(self, components):
        weights, models = zip(*components)
        for m in models:
            if not isinstance(m, MultiGaussianDist):
                raise ParameterError("the device MixtureDist takes MultiGaussianDist components")
            if m.Ndim != models[0].Ndim:
                raise ParameterError("mixture components have different dimensions")
        self._model_list = models
        self._mixweights = np.array(weights, dtype=np.float64) / np.sum(weights)    # benchmarks.py:40
        self._cumulwts = np.cumsum(self._mixweights)
        self.Ndim = self.Npars = models[0].Ndim
        # log_prior of the reference (:54-62): the components' log_prior is 0.0
        logP0 = float(np.log(np.sum(self._mixweights * np.exp(np.zeros(len(models))))))
        mus = np.array([m.mu for m in models])
        Ls = np.array([m.L for m in models])
        h = C.c_void_p()
        _lib.call("rmh_model_mixture_create", _lib.as_f64(self._mixweights), _lib.as_f64(mus),
                  _lib.as_f64(Ls), len(models), self.Ndim, logP0, C.byref(h))
        self._h = h

    def draw(self, Ndraws=1):
        """Host-side draws with numpy's generator, as benchmarks.py:64-73."""
        idx = np.searchsorted(self._cumulwts, np.random.uniform(size=Ndraws))
        return np.array([self._model_list[i].draw(1) for i in idx])


_SPECS = {
    "benchmark_gauss1d": lambda: MultiGaussianDist(0.0, 1.0),
    "benchmark_gauss2d_iso": lambda: MultiGaussianDist(np.zeros(2), np.eye(2)),
    "benchmark_gauss2d_corr": lambda: MultiGaussianDist(np.zeros(2), 0.1 * np.eye(2) + 0.9 * np.ones((2, 2))),
    "benchmark_gauss100d_iso": lambda: MultiGaussianDist(np.zeros(100), np.eye(100)),
    "benchmark_gauss100d_corr": lambda: MultiGaussianDist(
        np.zeros(100), 0.1 * np.eye(100) + 0.9 * np.ones((100, 100))),
    # benchmarks.py:75-85
    "benchmark_gauss1d_mix2": lambda: MixtureDist(
        [[0.3, MultiGaussianDist(-5.0, 1.0)], [0.7, MultiGaussianDist(+5.0, 1.0)]]),
    "benchmark_gauss2d_mix2_cross": lambda: MixtureDist(
        [[0.3, MultiGaussianDist([0, 0], [[1.0, 0.9], [0.9, 1.0]])],
         [0.7, MultiGaussianDist([0, 0], [[1.0, -0.9], [-0.9, 1.0]])]]),
}
_cache = {}


def __getattr__(name):
    if name in _SPECS:
        if name not in _cache:
            _cache[name] = _SPECS[name]()
        return _cache[name]
    raise AttributeError(name)
