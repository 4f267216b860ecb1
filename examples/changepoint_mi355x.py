#!/usr/bin/env python3
"""The fixed-k part of the reference's riemann/test_changepoint.py on the MI355X
back end: 5 changepoints, 100 data points, random-walk MH with the block
covariances of test_changepoint.py:36-38, many chains.  No plots."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from riemann_amd import Sampler                                       # noqa: E402
from riemann_amd.models import ChangepointParams, ChangepointRegression1D  # noqa: E402
from riemann_amd.proposals import MetropolisRandomWalk                # noqa: E402
from riemann_amd.workloads import changepoint_data                    # noqa: E402


def main(nchains=8192, k=5):
    x, y, truth, cdiag = changepoint_data(seed=3, k=k)
    model = ChangepointRegression1D(x, y, 1.0, 3.0, 1.0 * k, 2 * k, 1, 1, k=k)
    proposal = MetropolisRandomWalk(np.diag(cdiag))
    proposal.scale = 0.1
    g = np.random.default_rng(0)
    theta0 = truth + 0.01 * g.standard_normal((nchains, 2 * k + 2))
    theta0[:, :k] = np.sort(theta0[:, :k], axis=1)
    sampler = Sampler(model, proposal, theta0, seed=1)
    sampler.run(20000, 5000, 50)
    post = sampler._chain_thetas.reshape(-1, 2 * k + 2)
    est, tru = ChangepointParams.unpack(post.mean(axis=0)), ChangepointParams.unpack(truth)
    print("accept rate %.3f, %.1f ms on the GPU" % (sampler._chain_accept.mean(), sampler.last_run_ms))
    print("changepoints  true", np.round(tru.cpx, 3), " posterior mean", np.round(est.cpx, 3))
    print("levels        true", np.round(tru.cpv, 3), " posterior mean", np.round(est.cpv, 3))
    print("sigma         true %.3f  posterior mean %.3f" % (tru.sig, est.sig))


if __name__ == "__main__":
    main()
