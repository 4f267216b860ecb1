"""Throughput of the block-mixture changepoint proposal at fixed k (ChangepointRegression1DProp,
riemann/test_changepoint.py:18-56) on the cfg-5 data, 65 536 chains: the sufficient-statistics kernel
(k_cp_rw_ss<.., BLOCK>) against the O(n) kernel (kernel policy CP_NO_SS).

    python tools/block_timing.py [chains] [iterations]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import riemann_amd as ra                                              # noqa: E402
from riemann_amd import _lib, workloads as W                                # noqa: E402
from riemann_amd.models import ChangepointRegression1D                # noqa: E402
from riemann_amd.proposals import ChangepointRegression1DProp         # noqa: E402


def main(chains=65536, its=1000):
    x, y, truth, _ = W.changepoint_data()
    m = ChangepointRegression1D(x, y, 1.0, 3.0, 5.0, 10, 1, 1, k=5)
    th0 = truth + 0.01 * np.random.default_rng(1).standard_normal((chains, 12))
    th0[:, :5] = np.sort(th0[:, :5], axis=1)
    for env in (None, "1"):
        with _lib.kernel_policy(*(("CP_NO_SS",) if env else ())):
            p = ChangepointRegression1DProp(m, 2.0)
            p.scale = 0.5
            s = ra.Sampler(m, p, th0, seed=2)
            s.run(its, record=False)
            s.run(its, record=False)
        print("%-28s %.2f ms per %d iterations of %d chains = %.3g it/s" %
              ("O(n) kernel" if env else "sufficient statistics", s.last_run_ms, its, chains,
               chains * its / (s.last_run_ms * 1e-3)), flush=True)


if __name__ == "__main__":
    main(*[int(a) for a in sys.argv[1:]])
