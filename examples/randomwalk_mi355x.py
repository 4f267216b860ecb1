#!/usr/bin/env python3
"""The reference's examples/test_randomwalk.py on the MI355X back end: build a
proposal, run the sampler, print the autocorrelation time and the accept rate.

Differences from the reference script: 4096 chains instead of one (one fused
kernel launch for all of them), the build's own tau estimator instead of emcee
(not installed), no plots.  Usage:  python examples/randomwalk_mi355x.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from riemann_amd import Sampler                                         # noqa: E402
from riemann_amd import diagnostics                                      # noqa: E402
from riemann_amd.models.benchmarks import benchmark_gauss2d_corr         # noqa: E402
from riemann_amd.proposals.hamiltonian import (AdaptCovHMC, AdaptScaleCovHMC, AdaptScaleHMC,  # noqa: E402
                                                VanillaHMC)
from riemann_amd.proposals.randomwalk import (AdaptCovRandomWalk, AdaptScaleCovRandomWalk,    # noqa: E402
                                              AdaptScalepCN, AdaptScaleRandomWalk,
                                              MetropolisRandomWalk, pCN)


def test_sampling_gauss2d(nchains=4096):
    model = benchmark_gauss2d_corr
    grad = model.grad_log_likelihood
    C0 = 0.0001 * np.eye(2)
    # the nine proposals examples/test_randomwalk.py:23-37 builds (it samples with the last one only)
    for name, proposal in (("MetropolisRandomWalk", MetropolisRandomWalk(np.eye(2))),
                           ("AdaptCovRandomWalk", AdaptCovRandomWalk(C0, t_adapt=100, smooth_adapt=False)),
                           ("AdaptScaleRandomWalk", AdaptScaleRandomWalk(C0)),
                           ("AdaptScaleCovRandomWalk", AdaptScaleCovRandomWalk(C0, t_adapt=0, smooth_adapt=False)),
                           ("pCN(I, 0.5)", pCN(np.eye(2), 0.5)),
                           ("AdaptScalepCN(I, 0.5)", AdaptScalepCN(np.eye(2), 0.5)),
                           ("VanillaHMC(0.1, 5)", VanillaHMC(0.1, 5, grad)),
                           ("AdaptScaleHMC(0.1, 5)", AdaptScaleHMC(0.1, 5, grad)),
                           ("AdaptCovHMC(0.1, 5)", AdaptCovHMC(0.1, 5, grad, np.eye(2), t_adapt=100,
                                                               smooth_adapt=True)),
                           ("AdaptScaleCovHMC(0.1, 5)", AdaptScaleCovHMC(0.1, 5, grad, np.eye(2), t_adapt=100,
                                                                         smooth_adapt=True))):
        sampler = Sampler(model, proposal, np.ones((nchains, 2)), seed=0)
        sampler.run(10000, 1000, 1)
        chain = sampler._chain_thetas                       # (9001, nchains, 2)
        tau = diagnostics.integrated_time(chain[:, :64])
        print("%-24s tau = %s  accept = %.3f  mean = %s  %.1f ms on the GPU"
              % (name, np.round(tau, 1), sampler._chain_accept.mean(),
                 np.round(chain.mean(axis=(0, 1)), 3), sampler.last_run_ms))


def float64_normals(nchains=4096):
    """The same walk drawing its normals at float64 resolution (one per 64-bit stream word instead of two):
    Sampler(..., normals="f64") -- what np.random.normal gives the reference's proposals."""
    model = benchmark_gauss2d_corr
    for mode in ("f32x2", "f64"):
        sampler = Sampler(model, MetropolisRandomWalk(np.eye(2)), np.ones((nchains, 2)), seed=0, normals=mode)
        sampler.run(10000, 1000, 1)
        chain = sampler._chain_thetas
        print("normals=%-6s accept = %.3f  mean = %s  var = %s  %.1f ms on the GPU"
              % (mode, sampler._chain_accept.mean(), np.round(chain.mean(axis=(0, 1)), 3),
                 np.round(chain.var(axis=(0, 1)), 3), sampler.last_run_ms))


if __name__ == "__main__":
    test_sampling_gauss2d()
    float64_normals()
