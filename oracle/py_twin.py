"""Pure-Python single-chain twin of the reference's hot path (test infrastructure).

The reference's own files never travel to the GPU box, so the "reference-
equivalent" CPU rate of BASELINE.md (section 3, item 2) is measured on this
restatement: the same numpy/LAPACK calls, one chain, one thread, driven by the
build's rmh-rng stream instead of numpy's global generator.  Pinned against the
golden vectors recorded from the reference (tests/test_oracle_golden.py).

Each class cites the reference lines it follows (relative to /root/reference).
"""
import numpy as np


class TwinGaussian:
    """riemann/models/gaussian.py:27-58"""

    def __init__(self, mu, C):
        self.mu = np.atleast_1d(np.asarray(mu, dtype=np.float64))
        self.L = np.linalg.cholesky(np.atleast_2d(C))
        self.logdetC = 2 * np.sum(np.log(np.diag(self.L)))

    def log_posterior(self, theta):                      # models/model.py:43-55 with log_prior = 0
        y = theta - self.mu
        u = np.linalg.solve(self.L, y)                   # gaussian.py:51
        logL = -0.5 * (np.dot(u, u) + len(y) * np.log(2 * np.pi) + self.logdetC)
        return logL if np.isfinite(logL) else -np.inf

    def grad(self, theta):                               # gaussian.py:54-58
        u = np.linalg.solve(self.L, theta - self.mu)
        return -np.linalg.solve(self.L.T, u)


class TwinLogReg:
    """the logistic plugin against models/model.py:4-64"""

    def __init__(self, X, y, tau2):
        self.X, self.y, self.tau2 = X, y, tau2

    def log_posterior(self, theta):
        eta = self.X @ theta
        lp = -np.dot(theta, theta) / (2.0 * self.tau2) + np.sum(self.y * eta - np.logaddexp(0.0, eta))
        return lp if np.isfinite(lp) else -np.inf

    def grad(self, theta):
        eta = self.X @ theta
        e = np.exp(-np.abs(eta))
        sg = np.where(eta >= 0, 1.0 / (1.0 + e), e / (1.0 + e))
        return self.X.T @ (self.y - sg) - theta / self.tau2


class TwinRandomWalk:
    """riemann/proposals/randomwalk.py:17-26"""

    def __init__(self, C, scale=1.0):
        self.scale, self.L = scale, np.linalg.cholesky(np.atleast_2d(C))

    def propose(self, theta, xi):
        return theta + self.scale * np.dot(self.L, xi), 0.0


class TwinHMC:
    """riemann/proposals/hamiltonian.py:13-52, 76-91 (M = None)"""

    def __init__(self, eps, nsteps, grad):
        self.eps, self.nsteps, self.grad = eps, nsteps, grad

    def propose(self, theta, xi):
        eps, g = self.eps, self.grad
        p = xi + 0.5 * eps * g(theta)                    # :27
        q = theta + eps * p                              # :30
        for _ in range(self.nsteps - 1):                 # :34-38
            p = p + eps * g(q)
            q = q + eps * p
        p = p + 0.5 * eps * g(q)                         # :40
        return q, 0.5 * (np.sum(p ** 2) - np.sum(xi ** 2))   # :89


def run_chain(model, proposal, theta0, xi, u):
    """riemann/samplers/sampler.py:44-90 for one chain on a recorded stream
    xi (T,d), u (T,).  Returns thetas (T+1,d), logpost (T+1,), accept (T,)."""
    T = len(u)
    theta, lp = np.array(theta0, dtype=np.float64), None
    lp = model.log_posterior(theta)
    thetas, lps, acc = [theta], [lp], np.zeros(T, dtype=np.uint8)
    for t in range(T):
        thp, logq = proposal.propose(theta, xi[t])       # sampler.py:81
        lpp = model.log_posterior(thp)                   # :82
        mhratio = min(0, lpp - lp - logq)                # :83 (Python min: NaN -> 0)
        if np.log(u[t]) < mhratio:                       # :84
            theta, lp, acc[t] = thp, lpp, 1
        thetas.append(theta)
        lps.append(lp)
    return np.array(thetas), np.array(lps), acc
