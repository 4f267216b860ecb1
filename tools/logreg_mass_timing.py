"""Logistic regression with a mass matrix: MALA, MALA with M, HMC (L = 3) with M — ms per iteration, 65 536 chains
(device timer), at cfg3's data size for three dimensions and at 1000 rows (where what the triangular products of the
mass matrix add shows beside the evaluation).

    python tools/logreg_mass_timing.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import riemann_amd as ra                                  # noqa: E402
from riemann_amd import proposals as P                    # noqa: E402
from riemann_amd.workloads import logreg_data             # noqa: E402

N = 65536
for n, d, its in ((10000, 100, 10), (10000, 96, 10), (10000, 36, 20), (1000, 100, 40)):
    X, y = logreg_data(n, d)
    m = ra.models.LogisticRegression(X, y)
    A = np.random.default_rng(d).standard_normal((d, d)) / np.sqrt(d)
    M = A @ A.T + np.eye(d)
    for name, mk in (("mala", lambda: P.MALA(0.02, m)),
                     ("mala_mass", lambda: P.VanillaHMC(0.02, 1, m, M=M)),
                     ("hmc3_mass", lambda: P.VanillaHMC(0.02, 3, m, M=M))):
        s = ra.Sampler(m, mk(), np.zeros((N, d)), seed=1)
        s.run(its, record=False)
        s.run(its, record=False)
        print("n=%d d=%d %-10s %.4f ms/it" % (n, d, name, s.last_run_ms / its), flush=True)
