#!/usr/bin/env python3
"""The reference's riemann/test_changepoint.py on the MI355X back end: 5 true changepoints,
100 data points, many chains.  First at a fixed number of changepoints (random-walk MH with
the block covariances of test_changepoint.py:36-38, the BASELINE cfg-5 path), then -- as the
reference script does (test_changepoint.py:137-167) -- with an indeterminate number of
changepoints: ChangepointRegression1DProp with its birth / death moves, every chain started
from ONE changepoint (theta0 of :150).  No plots."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from riemann_amd import Sampler                                       # noqa: E402
from riemann_amd.models import ChangepointParams, ChangepointRegression1D  # noqa: E402
from riemann_amd.proposals import ChangepointRegression1DProp, MetropolisRandomWalk  # noqa: E402
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

    # indeterminate number of changepoints (test_changepoint.py:145-150, 167)
    xmin, xmax, hmin, hmax = 1.0, 3.0, 1.0, 3.0
    model = ChangepointRegression1D(x, y, xmin, xmax, 1.0 * k, 2 * k, 1, 1, variable_k=True)
    proposal = ChangepointRegression1DProp(model, hmax - hmin)
    theta0 = [ChangepointParams([0.5 * (xmin + xmax)], [hmin, hmax], 0.1)] * 2048
    sampler = Sampler(model, proposal, theta0, seed=2)
    sampler.run(20000, 10000, 100)
    th, code = sampler._chain_thetas, sampler._chain_accept
    kk = th[:, :, -1].astype(int).ravel()
    move, acc = code >> 1, code & 1
    print("variable k: %.1f ms on the GPU for %d chains x 20000 iterations" % (sampler.last_run_ms, len(theta0)))
    print("  number of changepoints (true %d): " % k
          + ", ".join("P(k=%d)=%.2f" % (v, (kk == v).mean()) for v in range(kk.max() + 1) if (kk == v).mean() > 0.005))
    for name, mv in (("cpx", 0), ("cpv", 1), ("sig", 2), ("birth", 3), ("death", 4)):
        print("  %-5s proposed %.3f of the time, accepted %.3f" % (name, (move == mv).mean(), acc[move == mv].mean()))


if __name__ == "__main__":
    main()
