"""The hot path the way a numpy practitioner would batch it on a CPU (test infrastructure: the
`cpu_baseline.batched_blas` leg of bench.py; never imported by the product).

oracle/rmh_oracle.c walks one chain at a time (one dot product per data row: the reference's
shape, which is what parity needs); oracle/py_twin.py is the reference's single chain.  Neither is
the CPU number a reader would recognise for MANY chains: with the chains as a matrix the
log-likelihoods become level-3 BLAS calls (theta @ X^T, triangular solves with many right-hand
sides) on all host cores.  Same arithmetic as py_twin, vectorised over the chain axis; normals and
uniforms from numpy's generator (a throughput baseline, not a parity run).

Reference lines: samplers/sampler.py:72-90, proposals/randomwalk.py:21-26,
proposals/hamiltonian.py:13-52,76-91, models/gaussian.py:46-58, models/changepoint.py:106-181.
"""
import time

import numpy as np


class BatchGaussian(object):
    def __init__(self, mu, C):
        from scipy.linalg import solve_triangular
        self._solve = solve_triangular
        self.mu = np.asarray(mu, dtype=np.float64)
        self.L = np.linalg.cholesky(np.atleast_2d(C))
        self.c = len(self.mu) * np.log(2 * np.pi) + 2 * np.sum(np.log(np.diag(self.L)))

    def log_posterior(self, th):                                  # (N, d) -> (N,)
        u = self._solve(self.L, (th - self.mu).T, lower=True)     # gaussian.py:51, N right-hand sides
        lp = -0.5 * (np.einsum("ij,ij->j", u, u) + self.c)
        return np.where(np.isfinite(lp), lp, -np.inf)


class BatchLogReg(object):
    def __init__(self, X, y, tau2):
        self.X, self.XT, self.y, self.tau2 = X, np.ascontiguousarray(X.T), y, tau2

    def log_posterior(self, th):
        eta = th @ self.XT                                        # (N, n): one dgemm
        lp = -np.einsum("ij,ij->i", th, th) / (2.0 * self.tau2) + eta @ self.y - np.logaddexp(0.0, eta).sum(axis=1)
        return np.where(np.isfinite(lp), lp, -np.inf)

    def grad(self, th):
        eta = th @ self.XT
        e = np.exp(-np.abs(eta))
        sg = np.where(eta >= 0, 1.0 / (1.0 + e), e / (1.0 + e))
        return (self.y - sg) @ self.X - th / self.tau2            # (N, n) @ (n, d): one dgemm


class BatchChangepoint(object):
    """models/changepoint.py:106-181 at fixed k, alpha = 1 as in BASELINE cfg 5 (general alpha kept)."""

    def __init__(self, x, y, k, xmin, xmax, lamb, alpha, beta):
        from scipy.special import gammaln
        self.x, self.y, self.k = np.asarray(x, float), np.asarray(y, float), k
        self.xmin, self.xmax, self.alpha, self.beta = xmin, xmax, alpha, beta
        kp = k + 1
        self.c = (kp * np.log(lamb) - gammaln(kp) - lamb + gammaln(2 * kp + 1) - kp * np.log(xmax - xmin)
                  + kp * (alpha * np.log(beta) - gammaln(alpha)))

    def log_posterior(self, th):
        k, n = self.k, len(self.x)
        cpx, cpv, sig = th[:, :k], th[:, k:2 * k + 1], th[:, 2 * k + 1]
        with np.errstate(all="ignore"):
            ds = np.diff(np.concatenate([np.full((len(th), 1), self.xmin), cpx, np.full((len(th), 1), self.xmax)],
                                        axis=1), axis=1)
            lprior = (self.c + np.sum((self.alpha - 1) * np.log(cpv) - self.beta * cpv, axis=1)
                      + np.sum(np.log(ds), axis=1) + np.log(1.0 / sig ** 2))
            seg = (self.x[None, :, None] > cpx[:, None, :]).sum(axis=2)          # searchsorted per chain
            ypred = np.take_along_axis(cpv, seg, axis=1)
            r = ypred - self.y[None, :]
            llik = -0.5 * (np.sum(r * r, axis=1) / sig ** 2 + n * np.log(2 * np.pi) + n * np.log(sig ** 2))
            lp = lprior + llik
        return np.where(np.isfinite(lp), lp, -np.inf)


def run(model, theta0, T, seed, L_prop=None, eps=None, nsteps=0):
    """T iterations of Sampler.sample for all chains at once; random walk (L_prop: Cholesky factor of
    the proposal covariance) or HMC / MALA (eps, nsteps).  Returns (seconds, accept rate)."""
    g = np.random.default_rng(seed)
    th = np.array(theta0, dtype=np.float64)
    lp = model.log_posterior(th)
    N, d = th.shape
    acc = 0
    t0 = time.perf_counter()
    for _ in range(T):
        xi = g.standard_normal((N, d))
        if L_prop is not None:
            thp, logq = th + xi @ L_prop.T, 0.0                                   # randomwalk.py:26
        else:
            p = xi + 0.5 * eps * model.grad(th)                                   # hamiltonian.py:27
            q = th + eps * p
            for _ in range(nsteps - 1):
                p = p + eps * model.grad(q)
                q = q + eps * p
            p = p + 0.5 * eps * model.grad(q)
            thp, logq = q, 0.5 * (np.einsum("ij,ij->i", p, p) - np.einsum("ij,ij->i", xi, xi))
        lpp = model.log_posterior(thp)                                            # sampler.py:82
        with np.errstate(invalid="ignore"):
            delta = lpp - lp - logq
            ok = np.log(g.random(N)) < np.where(delta < 0.0, delta, 0.0)          # :83-84, NaN accepts
        th = np.where(ok[:, None], thp, th)
        lp = np.where(ok, lpp, lp)
        acc += int(ok.sum())
    return time.perf_counter() - t0, acc / float(N * max(T, 1))
