"""Random-walk MH on a d-dimensional Gaussian for a sweep of d: ms per 1000 iterations of
65 536 chains (device timer).  Shows what padding rows/k-steps cost at d below a tile boundary.

    python tools/dim_sweep_timing.py [d ...]
"""
import sys

import numpy as np

import riemann_amd as ra


def one(d, chains=65536, its=1000):
    rng = np.random.default_rng(d)
    A = rng.standard_normal((d, d)) / np.sqrt(d)
    C = A @ A.T + np.eye(d)
    model = ra.models.MultiGaussianDist(np.zeros(d), C)
    prop = ra.proposals.MetropolisRandomWalk(2.38 ** 2 / d * C)
    s = ra.Sampler(model, prop, np.zeros((chains, d)), seed=1)
    s.run(its, record=False)  # warm-up
    s.run(its, record=False)
    return s.last_run_ms


if __name__ == "__main__":
    dims = [int(a) for a in sys.argv[1:]] or [8, 16, 20, 32, 36, 48, 52, 64, 70, 84, 96, 100, 112]
    for d in dims:
        print("d=%4d  %8.2f ms / 1000 its" % (d, one(d)), flush=True)
