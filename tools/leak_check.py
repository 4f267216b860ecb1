"""Device-memory leak check: create, run and drop many Samplers (all model / proposal families, histories,
pinned downloads, tempering, adaptation) and compare the free device memory before and after.

    python tools/leak_check.py [rounds]
"""
import ctypes as C
import gc
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import riemann_amd as ra                                                    # noqa: E402
from riemann_amd import workloads as W                                       # noqa: E402
from riemann_amd.models import ChangepointRegression1D, LogisticRegression, MultiGaussianDist  # noqa: E402
from riemann_amd.models.benchmarks import benchmark_gauss2d_mix2_cross       # noqa: E402
from riemann_amd import proposals as P                                       # noqa: E402
from riemann_amd.samplers import PTSampler                                   # noqa: E402

hip = C.CDLL("libamdhip64.so")


def free_bytes():
    f, t = C.c_size_t(), C.c_size_t()
    assert hip.hipMemGetInfo(C.byref(f), C.byref(t)) == 0
    return f.value


def one_round(g):
    d = 20
    A = g.standard_normal((d, d))
    Cov = A @ A.T / d + np.eye(d)
    m = MultiGaussianDist(np.zeros(d), Cov)
    th0 = g.standard_normal((3000, d))
    for p in (P.MetropolisRandomWalk(0.3 * Cov), P.AdaptScaleRandomWalk(0.3 * Cov), P.pCN(Cov, 0.9),
              P.VanillaHMC(0.2, 3, m), P.VanillaHMC(0.2, 2, m, M=Cov),
              P.AdaptCovRandomWalk(0.3 * Cov, t_adapt=10, smooth_adapt=True)):
        s = ra.Sampler(m, p, th0, seed=1)
        s.run(30)
        s.run(20, 5, 2)
        _ = s._chain_thetas.shape, s.history_stats(nlags=32)["tau"]
        del s
    X, y = W.logreg_data(500, 24, seed=1)
    lm = LogisticRegression(X, y, 100.0)
    s = ra.Sampler(lm, P.MALA(0.1, lm), 0.1 * g.standard_normal((2000, 24)), seed=2)
    s.run(10)
    del s
    x, yy, truth, cdiag = W.changepoint_data()
    cm = ChangepointRegression1D(x, yy, 1.0, 3.0, 5.0, 10, 1, 1, k=5)
    t0 = truth + 0.01 * g.standard_normal((2000, 12))
    t0[:, :5] = np.sort(t0[:, :5], axis=1)
    for p in (P.MetropolisRandomWalk(np.diag(cdiag)), P.ChangepointRegression1DProp(cm, 2.0)):
        s = ra.Sampler(cm, p, t0, seed=3)
        s.run(40)
        del s
    pt = PTSampler(benchmark_gauss2d_mix2_cross, P.MetropolisRandomWalk(np.eye(2)), g.standard_normal((500, 2)), seed=4)
    pt.run(30)
    del pt, m, lm, cm
    gc.collect()


def main(rounds=12):
    g = np.random.default_rng(0)
    one_round(g)                       # warm-up: library-owned buffers (staging, pinned pool) are allocated once
    _lib = ra._lib
    _lib.call("rmh_synchronize")
    f0 = free_bytes()
    for r in range(rounds):
        one_round(g)
    _lib.call("rmh_synchronize")
    f1 = free_bytes()
    print("free device memory: %.1f MiB before, %.1f MiB after %d rounds (difference %.2f MiB)" %
          (f0 / 2 ** 20, f1 / 2 ** 20, rounds, (f0 - f1) / 2 ** 20))
    return 0 if f0 - f1 < 64 * 2 ** 20 else 1


if __name__ == "__main__":
    sys.exit(main(*[int(a) for a in sys.argv[1:]]))
