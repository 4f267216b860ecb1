"""Throughput of the trans-dimensional changepoint sampler on the cfg-5 data:
65 536 chains, every chain started from one changepoint, for kmax = 6 and kmax = 10.

    python tools/rj_timing.py [chains] [iterations]
"""
import sys

import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import riemann_amd as ra
from riemann_amd import workloads as W
from riemann_amd.models import ChangepointParams, ChangepointRegression1D
from riemann_amd.proposals import ChangepointRegression1DProp


def main(chains=65536, its=500):
    x, y, truth, _ = W.changepoint_data()
    xmin, xmax, hmin, hmax = 1.0, 3.0, 1.0, 3.0
    for kmax in (6, 10):
        model = ChangepointRegression1D(x, y, xmin, xmax, 5.0, kmax, 1, 1, variable_k=True)
        prop = ChangepointRegression1DProp(model, hmax - hmin)
        th0 = [ChangepointParams([0.5 * (xmin + xmax)], [hmin, hmax], 0.1)] * chains
        s = ra.Sampler(model, prop, th0, seed=2)
        s.run(its, record=False)
        s.run(its, record=False)
        print("kmax = %2d: %.2f ms per %d iterations of %d chains = %.3g it/s" %
              (kmax, s.last_run_ms, its, chains, chains * its / (s.last_run_ms * 1e-3)), flush=True)


if __name__ == "__main__":
    main(*[int(a) for a in sys.argv[1:]])
