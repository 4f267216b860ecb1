import numpy as np

from riemann_amd import diagnostics as dg


def ar1(T, N, d, rho, seed=0):
    g = np.random.default_rng(seed)
    x = np.zeros((T, N, d))
    e = g.standard_normal((T, N, d)) * np.sqrt(1 - rho ** 2)
    for t in range(1, T):
        x[t] = rho * x[t - 1] + e[t]
    return x


def test_integrated_time_of_ar1():
    for rho in (0.0, 0.5, 0.9):
        tau = dg.integrated_time(ar1(20000, 8, 2, rho, seed=int(10 * rho)))
        np.testing.assert_allclose(tau, (1 + rho) / (1 - rho), rtol=0.12)


def test_ess_and_rhat():
    x = ar1(4000, 16, 3, 0.5)
    e = dg.ess(x)
    assert np.all(e > 0.25 * 4000 * 16) and np.all(e < 0.45 * 4000 * 16)
    assert np.all(np.abs(dg.split_rhat(x) - 1) < 0.02)
    y = x.copy()
    y[:, :8] += 3.0                      # two groups of chains that disagree
    assert np.all(dg.split_rhat(y) > 1.5)
    assert dg.accept_rate(np.array([[1, 0], [1, 1]], dtype=np.uint8)) == 0.75


def test_integrated_time_from_chain_averaged_acf():
    """The device-side reduction hands back chain-averaged autocorrelation functions truncated at K lags
    (Sampler.history_stats); tau from them equals the FFT estimator's when the Sokal window closes within
    K, and is flagged as a lower bound when it does not."""
    x = ar1(6000, 12, 2, 0.8, seed=3)
    T = x.shape[0]
    xc = x - x.mean(axis=0)
    a0 = np.sum(xc * xc, axis=0)
    K = 128
    acf = np.stack([np.mean(np.sum(xc[: T - k] * xc[k:], axis=0) / a0, axis=0) for k in range(K)])
    tau, closed = dg.integrated_time_from_acf(acf)
    assert closed.all()
    np.testing.assert_allclose(tau, dg.integrated_time(x), rtol=1e-9)
    tau8, closed8 = dg.integrated_time_from_acf(acf[:8])
    assert not closed8.any() and np.all(tau8 < tau)
