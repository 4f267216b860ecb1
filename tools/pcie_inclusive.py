"""cfg2 with every recorded sample copied to the host: chain-iterations/s including the
device->host transfer (the C ABI hands histories over as host buffers; bench.py's `value` times
the on-device part only).

    python tools/pcie_inclusive.py [chains] [iterations]
"""
import sys
import time

import numpy as np

import riemann_amd as ra
from riemann_amd import _lib


def main(chains=65536, its=200):
    d = 100
    rng = np.random.default_rng(0)
    A = rng.standard_normal((d, d)) / np.sqrt(d)
    C = A @ A.T + np.eye(d)
    s = ra.Sampler(ra.models.MultiGaussianDist(np.zeros(d), C), ra.proposals.MetropolisRandomWalk(2.38 ** 2 / d * C),
                   np.zeros((chains, d)), seed=1)
    s._pool = _lib.PinnedPool(max_bytes=16 << 30)   # the default budget (4 GiB) is below this history
    s.run(its)
    s._chain_thetas              # warm-up: the first download page-locks its host buffer (kept by the
    #                              Sampler's pool and reused once nothing refers to the old history)
    t0 = time.perf_counter()
    s.run(its)
    t1 = time.perf_counter()
    th = s._chain_thetas
    lp = s._chain_logpost
    t2 = time.perf_counter()
    gb = (th.nbytes + lp.nbytes) / 1e9
    print("run %.1f ms (kernel %.1f ms), download of %.2f GB %.1f ms = %.1f GB/s" %
          ((t1 - t0) * 1e3, s.last_run_ms, gb, (t2 - t1) * 1e3, gb / (t2 - t1)))
    print("device only %.3g it/s; with every sample on the host %.3g it/s" %
          (chains * its / (t1 - t0), chains * its / (t2 - t0)))


if __name__ == "__main__":
    a = [int(x) for x in sys.argv[1:]]
    main(*a)
