"""Synthetic inputs of the benchmark configurations (SURVEY.md section 8d,
BASELINE.md section 3).  Pure numpy, seeded; shared by bench.py, the tests and
tests/golden/make_golden.py so that every leg sees the same bytes."""
import numpy as np


def corr_gauss_cov(d=100):
    """C = 0.1 I + 0.9 11^T: /root/reference/riemann/models/benchmarks.py:25-26."""
    return 0.1 * np.eye(d) + 0.9 * np.ones((d, d))


def logreg_data(n, d, seed=2):
    """cfg3/cfg4 data: X ~ N(0,1)/sqrt(d), beta* ~ N(0,I), y ~ Bernoulli(sigmoid(X beta*))."""
    g = np.random.default_rng(seed)
    X = g.standard_normal((n, d)) / np.sqrt(d)
    beta = g.standard_normal(d)
    y = (g.random(n) < 1.0 / (1.0 + np.exp(-(X @ beta)))).astype(np.float64)
    return X, y


def changepoint_data(seed=3, k=5, n=100, xmin=1.0, xmax=3.0, hmin=1.0, hmax=3.0, sig=0.1):
    """cfg5 data as /root/reference/riemann/test_changepoint.py:137-150,167, seeded.

    Returns x, y, truth (packed cpx[k], cpv[k+1], sig) and the diagonal of the
    block random-walk covariance of test_changepoint.py:36-38."""
    g = np.random.default_rng(seed)
    cpx = np.sort(g.uniform(xmin, xmax, size=k))
    cpv = g.uniform(hmin, hmax, size=k + 1)
    truth = np.concatenate([cpx, cpv, [sig]])
    x = np.sort(xmin + (xmax - xmin) * g.uniform(size=n))
    y = cpv[np.searchsorted(cpx, x)] + sig * g.standard_normal(n)
    hscale = hmax - hmin
    cdiag = np.concatenate([np.full(k, 0.01 * (xmax - xmin) / (k + 1)),
                            np.full(k + 1, 0.01 * hscale ** 2 / n), [0.01 * hscale]])
    return x, y, truth, cdiag
