"""Benchmark target distributions: /root/reference/riemann/models/benchmarks.py:12-26.
Built on first use (constructing a device Model needs the GPU)."""
import numpy as np

from .gaussian import MultiGaussianDist

_SPECS = {
    "benchmark_gauss1d": lambda: (0.0, 1.0),
    "benchmark_gauss2d_iso": lambda: (np.zeros(2), np.eye(2)),
    "benchmark_gauss2d_corr": lambda: (np.zeros(2), 0.1 * np.eye(2) + 0.9 * np.ones((2, 2))),
    "benchmark_gauss100d_iso": lambda: (np.zeros(100), np.eye(100)),
    "benchmark_gauss100d_corr": lambda: (np.zeros(100),
                                         0.1 * np.eye(100) + 0.9 * np.ones((100, 100))),
}
_cache = {}


def __getattr__(name):
    if name in _SPECS:
        if name not in _cache:
            _cache[name] = MultiGaussianDist(*_SPECS[name]())
        return _cache[name]
    raise AttributeError(name)
