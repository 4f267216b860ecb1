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


# ---- pooled diagnostics from per-rank reductions (riemann_amd.distributed.pool_history) ----------
# A sharded job must report the numbers a single process would: bit for bit, not "to rounding",
# or the same chains give a different R-hat on 4 GPUs than on 8.  Two devices make every sum over
# chains independent of how the chains are split over ranks:
#  * the autocorrelation sums arrive as integers (rmh_history_reduce_sums, RMH_ACF_SCALE): integer
#    addition is exact;
#  * sums of per-chain doubles (means, variances) are formed with math.fsum -- the correctly rounded
#    value of the exact sum -- from the ranks' partials, each rank contributing its exact partial sum
#    as three non-overlapping doubles (hi, lo, lo2: exact to 2^-150 of the partial).
ACF_SCALE = 1099511627776.0   # 2^40, include/riemann_hip.h RMH_ACF_SCALE


def exact_partial(x):
    """The exact sum of the 1-D array ``x`` as three doubles (hi, lo, lo2)."""
    import math
    v = np.asarray(x, dtype=np.float64).ravel().tolist()
    hi = math.fsum(v)
    lo = math.fsum(v + [-hi])
    lo2 = math.fsum(v + [-hi, -lo])
    return hi, lo, lo2


def _sum_over_ranks(rows, allreduce, rank, world):
    """Exact sums over all ranks of the rows of ``rows`` (q, n_local): each rank's exact partial
    (three doubles per row) travels in its own slot of a zero-padded vector -- x + 0 + ... + 0 is
    exact under any all-reduce order -- and every rank forms math.fsum over all partials."""
    import math
    rows = np.atleast_2d(np.asarray(rows, dtype=np.float64))
    q = rows.shape[0]
    buf = np.zeros((world, q, 3))
    for i in range(q):
        buf[rank, i] = exact_partial(rows[i])
    buf = np.asarray(allreduce(buf.ravel()), dtype=np.float64).reshape(world, q, 3)
    return np.array([math.fsum(buf[:, i, :].ravel().tolist()) for i in range(q)])


def pooled_history_stats(piece, allreduce=None, rank=0, world=1, c=5.0):
    """Diagnostics of a recorded run pooled over the chains of ALL ranks from per-rank reductions.

    ``piece`` (one rank; Sampler.history_pieces or its numpy twin): ``n`` kept states, ``mean`` /
    ``var`` (d, N_local) per-chain mean and biased variance, ``acf_sums`` (K, d) int64 integer
    autocorrelation sums, ``mean_h`` / ``var_h`` (2, d, N_local) the same for the two half chains of
    ``h`` states (absent when n < 4).  ``allreduce(vec)`` sums a float64 vector over the ranks (the
    library's rmh_comm_allreduce_host, a gloo all_reduce, HostGroup ...); None for one process.
    Returns pooled ``mean``, ``var`` (d,), ``acf`` (K, d), ``tau``, ``closed``, ``ess`` (d,),
    ``rhat`` (d,) and ``chains``: every entry the same on every rank and the same whatever the
    number of ranks the chains are sharded over."""
    if allreduce is None:
        if world != 1:
            raise ValueError("pooled_history_stats: an allreduce is needed for more than one rank")
        allreduce = lambda v: v                                           # noqa: E731
    mean, var = np.asarray(piece["mean"]), np.asarray(piece["var"])
    d, nloc = mean.shape
    n = int(piece["n"])
    cnt = np.zeros(world)
    cnt[rank] = nloc
    M = int(round(float(np.sum(allreduce(cnt)))))
    # autocorrelation: integer sums, split into two exactly representable halves for the all-reduce
    isum = np.asarray(piece["acf_sums"], dtype=np.int64)
    hi, lo = (isum >> 32).astype(np.float64), (isum & 0xFFFFFFFF).astype(np.float64)
    tot = np.asarray(allreduce(np.concatenate([hi.ravel(), lo.ravel()])), dtype=np.float64)
    K = isum.shape[0]
    thi, tlo = tot[:K * d].reshape(K, d), tot[K * d:].reshape(K, d)
    den = float(M) * ACF_SCALE
    acf = np.array([[float(int(thi[k, p]) * (1 << 32) + int(tlo[k, p])) / den for p in range(d)] for k in range(K)])
    tau, closed = integrated_time_from_acf(acf, c)
    out = {"chains": M, "acf": acf, "tau": tau, "closed": closed, "ess": M * n / np.maximum(tau, 1.0)}
    # pooled mean and variance per parameter: chains have equal length, so the pooled mean is the mean
    # of the chain means and the pooled variance the mean of var_c + (m_c - mu)^2
    mu = _sum_over_ranks(mean, allreduce, rank, world) / M
    out["mean"] = mu
    out["var"] = _sum_over_ranks(var + (mean - mu[:, None]) ** 2, allreduce, rank, world) / M
    # split R-hat (split_rhat above) over the 2M half chains
    if "mean_h" in piece:
        h = int(piece["h"])
        mh = np.concatenate([piece["mean_h"][0], piece["mean_h"][1]], axis=1)          # (d, 2 N_local)
        vh = np.concatenate([piece["var_h"][0], piece["var_h"][1]], axis=1)
        M2 = 2 * M
        W = _sum_over_ranks(vh * (h / (h - 1.0)), allreduce, rank, world) / M2
        mu_h = _sum_over_ranks(mh, allreduce, rank, world) / M2
        B = h * _sum_over_ranks((mh - mu_h[:, None]) ** 2, allreduce, rank, world) / (M2 - 1.0)
        out["rhat"] = np.sqrt(((h - 1.0) / h * W + B / h) / W)
    return out


def history_pieces_numpy(x, nlags=128):
    """The per-rank reductions of ``pooled_history_stats`` from a host array ``x`` (n, N, d) -- the
    numpy restatement of rmh_history_reduce_sums (tests, and users without a device history)."""
    x = np.asarray(x, dtype=np.float64)
    n, N, d = x.shape
    m = x.mean(axis=0)
    xc = x - m
    a0 = np.sum(xc * xc, axis=0)                                                      # (N, d)
    sums = np.empty((nlags, d), dtype=np.int64)
    for k in range(nlags):
        ak = np.sum(xc[: n - k] * xc[k:], axis=0) if k < n else np.zeros((N, d))
        rho = np.where(a0 == 0.0, 1.0, ak / np.where(a0 == 0.0, 1.0, a0))
        sums[k] = np.rint(rho * ACF_SCALE).astype(np.int64).sum(axis=0)
    piece = {"n": n, "mean": m.T.copy(), "var": x.var(axis=0).T.copy(), "acf_sums": sums}
    h = n // 2
    if h >= 2:
        piece["h"] = h
        piece["mean_h"] = np.stack([x[:h].mean(axis=0).T, x[h:2 * h].mean(axis=0).T])
        piece["var_h"] = np.stack([x[:h].var(axis=0).T, x[h:2 * h].var(axis=0).T])
    return piece
