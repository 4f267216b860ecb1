#!/usr/bin/env python3
"""Cost of the pooled Haario adaptation at the cfg2 size (65 536 chains, d = 100):
plain random walk vs AdaptCovRandomWalk with the reference's float update test and with
updates at perfect squares only.  Prints ms per T (default 1000) iterations and launches.

    python tools/adaptcov_timing.py [T] [float|squares|plain|hmc|scalecov ...]

The HMC variant (AdaptCovHMC: the adapted covariance becomes the mass matrix) and the walk with both scale and
covariance adaptation (AdaptScaleCovRandomWalk) take the scratch-history route; their plain counterparts beside them."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import riemann_amd as ra                                                    # noqa: E402
from riemann_amd import _lib, workloads as W                                 # noqa: E402
from riemann_amd.models import MultiGaussianDist                            # noqa: E402
from riemann_amd.proposals import (AdaptCovHMC, AdaptCovRandomWalk, AdaptScaleCovRandomWalk,  # noqa: E402
                                   MetropolisRandomWalk, VanillaHMC)

d, N, T = 100, 65536, int(sys.argv[1]) if len(sys.argv) > 1 else 1000
only = sys.argv[2:]
Ct = W.corr_gauss_cov(d)
m = MultiGaussianDist(np.zeros(d), Ct)
th0 = np.random.default_rng(1).standard_normal((N, d)) @ np.linalg.cholesky(Ct).T
C0 = (2.38 ** 2 / d) * np.eye(d)
for name, p in (("plain", MetropolisRandomWalk(C0)),
                ("adaptcov float test", AdaptCovRandomWalk(C0, t_adapt=1000, smooth_adapt=True)),
                ("adaptcov squares", AdaptCovRandomWalk(C0, t_adapt=1000, smooth_adapt=True, exact_squares=True)),
                ("scalecov float test", AdaptScaleCovRandomWalk(C0, t_adapt=1000, smooth_adapt=True)),
                ("scalecov squares", AdaptScaleCovRandomWalk(C0, t_adapt=1000, smooth_adapt=True, exact_squares=True)),
                ("hmc L=2 mass plain", VanillaHMC(0.1, 2, m, M=np.eye(d))),
                ("hmc L=2 adaptcov float test", AdaptCovHMC(0.1, 2, m, np.eye(d), t_adapt=1000, smooth_adapt=True)),
                ("hmc L=2 adaptcov squares", AdaptCovHMC(0.1, 2, m, np.eye(d), t_adapt=1000, smooth_adapt=True,
                                                         exact_squares=True))):
    if only and not any(o in name for o in only):
        continue
    s = ra.Sampler(m, p, th0, seed=3)
    s.run(20, record=False)
    t0 = time.perf_counter()
    s.run(T, record=False)
    wall = (time.perf_counter() - t0) * 1e3
    n = C.c_int()
    ms = C.c_double()
    _lib.call("rmh_last_run_ms", C.byref(ms), C.byref(n))
    acc = None
    if hasattr(p, "C"):
        acc = float(np.abs(p.C * d ** 0.8 / (2.38 ** 2 / d) - Ct).max()) if False else float(np.trace(p.C))
    print("%-22s wall %8.1f ms  device %8.1f ms  launches %4d  tr(C) %s" % (name, wall, ms.value, n.value, acc))
