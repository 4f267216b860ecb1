"""Chain diagnostics: integrated autocorrelation time, ESS, split R-hat.

The reference only prints ``emcee.autocorr.integrated_time`` and the accept
rate (/root/reference/examples/test_randomwalk.py:42-46); emcee is neither in
the reference tree nor installed, so the estimator is restated here from its
published definition (FFT autocorrelation averaged over chains, Sokal's
automatic window with c = 5).  Parity: unpinned -- no reference test records a
tau or ESS value.  Host-side numpy on pooled samples; not part
of the hot path.
"""
import numpy as np


def _next_pow_two(n):
    i = 1
    while i < n:
        i <<= 1
    return i


def autocorr_1d(x):
    """Normalised autocorrelation function of a 1-D series via FFT."""
    x = np.asarray(x, dtype=np.float64)
    n = _next_pow_two(len(x))
    f = np.fft.rfft(x - np.mean(x), n=2 * n)
    acf = np.fft.irfft(f * np.conjugate(f))[: len(x)].real
    if acf[0] == 0:
        return np.ones_like(acf)
    return acf / acf[0]


def _auto_window(taus, c):
    m = np.arange(len(taus)) < c * taus
    return int(np.argmin(m)) if np.any(~m) else len(taus) - 1


def integrated_time(chain, c=5.0):
    """tau_int per parameter.  ``chain`` is (T, N, d) [or (T, d)]: T steps of N
    chains; the autocorrelation functions are averaged over chains first."""
    x = np.asarray(chain, dtype=np.float64)
    if x.ndim == 2:
        x = x[:, None, :]
    T, N, d = x.shape
    tau = np.empty(d)
    for j in range(d):
        f = np.zeros(T)
        for k in range(N):
            f += autocorr_1d(x[:, k, j])
        f /= N
        taus = 2.0 * np.cumsum(f) - 1.0
        tau[j] = taus[_auto_window(taus, c)]
    return tau


def integrated_time_from_acf(acf, c=5.0):
    """tau_int per parameter from the chain-averaged autocorrelation functions ``acf`` (K, d), lags
    0..K-1 (Sampler.history_stats: reduced on the device).  The estimate of a parameter whose
    Sokal window does not close within the K lags is a lower bound; ``closed`` says which did."""
    acf = np.asarray(acf, dtype=np.float64)
    K, d = acf.shape
    tau, closed = np.empty(d), np.empty(d, dtype=bool)
    for j in range(d):
        taus = 2.0 * np.cumsum(acf[:, j]) - 1.0
        m = np.arange(K) < c * taus
        closed[j] = bool(np.any(~m))
        tau[j] = taus[int(np.argmin(m)) if closed[j] else K - 1]
    return tau, closed


def ess(chain, c=5.0):
    """Effective sample size per parameter, pooled over chains: N*T / tau_int."""
    x = np.asarray(chain)
    T = x.shape[0]
    N = x.shape[1] if x.ndim == 3 else 1
    return N * T / np.maximum(integrated_time(chain, c), 1.0)


def split_rhat(chain):
    """Split R-hat per parameter for (T, N, d) samples."""
    x = np.asarray(chain, dtype=np.float64)
    if x.ndim == 2:
        x = x[:, None, :]
    T = x.shape[0] // 2
    halves = np.concatenate([x[:T], x[T:2 * T]], axis=1)       # (T, 2N, d)
    W = halves.var(axis=0, ddof=1).mean(axis=0)
    B = T * halves.mean(axis=0).var(axis=0, ddof=1)
    return np.sqrt(((T - 1) / T * W + B / T) / W)


def accept_rate(chain_accept):
    """Mean of the accept flags (the reference derives it from chain diffs,
    examples/test_randomwalk.py:45)."""
    return float(np.mean(chain_accept))
