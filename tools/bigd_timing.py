#!/usr/bin/env python3
"""Throughput of the d > 112 Gaussian random-walk kernels (rmh_kernels_big.hip), 65 536 chains."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import riemann_amd as ra                                          # noqa: E402
from riemann_amd import _lib                                       # noqa: E402
from riemann_amd.models import MultiGaussianDist                  # noqa: E402
from riemann_amd.proposals import MetropolisRandomWalk            # noqa: E402

N = 65536
for d, T in ((128, 200), (256, 100), (512, 40), (1024, 16)):
    g = np.random.default_rng(d)
    Ct = 0.1 * np.eye(d) + 0.9 * np.ones((d, d))
    m = MultiGaussianDist(np.zeros(d), Ct)
    th0 = g.standard_normal((N, d)) @ np.linalg.cholesky(Ct).T
    s = ra.Sampler(m, MetropolisRandomWalk((2.38 ** 2 / d) * Ct), th0, seed=1)
    s.run(3, record=False)
    for rec in (False, True):
        s.run(T, Nburn=0, Nthin=1, record=rec)
        ms = C.c_double()
        _lib.call("rmh_last_run_ms", C.byref(ms), None)
        fl = (2 * d * (d + 1) + 4 * d) * N * T
        print("d=%4d record=%-5s %8.2f ms / %d its  %.3e chain-it/s  %.1f TFLOP/s (%.2f of 78.6)  accept %.2f"
              % (d, rec, ms.value, T, N * T / ms.value * 1e3, fl / ms.value / 1e9, fl / ms.value / 1e9 / 78.6,
                 s._chain_accept.mean() if rec else float("nan")))
    del s, m
