#!/usr/bin/env python3
"""The plugin contract of the reference on the MI355X back end: ANY Model / Proposal subclass
samples (riemann/samplers/sampler.py:72-90, proposals/hamiltonian.py:60-70).

1. a user's numpy Model sampled by the device MetropolisRandomWalk;
2. the same Model with VanillaHMC on its own Python gradient (host leapfrog, device-drawn momenta);
3. the same Model with the adaptive proposals it is usually paired with: AdaptCovRandomWalk (Haario; moments
   and Cholesky factor on the host, theta + L xi on the device) and AdaptScaleHMC on the Python gradient;
4. a user's Proposal on a device Model;
5. for comparison, the all-device version of (2): one fused launch.
In 1-4 the Sampler runs the reference's loop on the host with the chains' device stream, so chain
c sees the same random numbers as in the fused path."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import riemann_amd as ra                                              # noqa: E402
from riemann_amd.models import LogisticRegression                     # noqa: E402
from riemann_amd.proposals import AdaptCovRandomWalk, AdaptScaleHMC, MetropolisRandomWalk, VanillaHMC    # noqa: E402
from riemann_amd.workloads import logreg_data                         # noqa: E402


class NumpyLogReg(ra.Model):
    """User code: numpy only."""

    def __init__(self, X, y, tau2):
        self.X, self.y, self.tau2 = X, y, tau2

    def log_prior(self, theta):
        return -np.dot(theta, theta) / (2.0 * self.tau2)

    def log_likelihood(self, theta):
        eta = self.X @ theta
        return np.sum(self.y * eta - np.logaddexp(0.0, eta))

    def grad_log_posterior(self, theta):
        eta = self.X @ theta
        return self.X.T @ (self.y - 1.0 / (1.0 + np.exp(-eta))) - theta / self.tau2


class MyWalk(ra.Proposal):
    """User code: a proposal with its own generator."""

    def __init__(self, step, seed):
        self.step, self.g = step, np.random.default_rng(seed)

    def propose(self, theta):
        return theta + self.step * self.g.standard_normal(theta.shape), 0.0


def main(nchains=64, T=200):
    X, y = logreg_data(512, 8, seed=2)
    user, dev = NumpyLogReg(X, y, 100.0), LogisticRegression(X, y, 100.0)
    th0 = np.zeros((nchains, 8))
    runs = [("numpy Model + device random walk", user, MetropolisRandomWalk(0.02 * np.eye(8))),
            ("numpy Model + HMC on its Python gradient", user, VanillaHMC(0.15, 5, user.grad_log_posterior, Npars=8)),
            ("numpy Model + device AdaptCovRandomWalk", user, AdaptCovRandomWalk(0.02 * np.eye(8), t_adapt=500,
                                                                                    smooth_adapt=True)),
            ("numpy Model + AdaptScaleHMC on its gradient", user, AdaptScaleHMC(0.15, 5, user.grad_log_posterior, Npars=8)),
            ("device Model + user's Proposal", dev, MyWalk(0.12, 3)),
            ("device Model + device HMC (fused)", dev, VanillaHMC(0.15, 5, dev))]
    for name, m, p in runs:
        s = ra.Sampler(m, p, th0, seed=1)
        t0 = time.perf_counter()
        s.run(T, T // 4)
        dt = time.perf_counter() - t0
        post = s._chain_thetas.reshape(-1, 8)
        print("%-44s accept %.2f  mean[0] %+.3f  %.3f s" % (name, s._chain_accept.mean(), post[:, 0].mean(), dt))


if __name__ == "__main__":
    main()
