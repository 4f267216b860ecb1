"""Every proposal kind of the library on a d-dimensional Gaussian, 65 536 chains: ms per iteration
(device timer) — the variants that are not a BASELINE config (pCN, mass-matrix HMC, adapted scales,
tempering) beside the plain random walk / HMC they share their kernels' structure with.

    python tools/variant_timing.py [--json out.json] [--model gauss|logreg|mixture] [d ...]   (default d = 100 96 36)

--model logreg: logistic regression on 1000 x d synthetic rows (a small data term, so that what the
proposal adds shows); mixture: two Gaussian components.
"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import riemann_amd as ra                      # noqa: E402
from riemann_amd import proposals as P        # noqa: E402


def variants(model, C):
    d = C.shape[0]
    s2 = 2.38 ** 2 / d
    g = model.grad_log_posterior
    return [
        ("rw", lambda: P.MetropolisRandomWalk(s2 * C), 400),
        ("rw_adapt_scale", lambda: P.AdaptScaleRandomWalk(s2 * C), 400),
        ("pcn", lambda: P.pCN(C, 0.9), 400),
        ("pcn_adapt_scale", lambda: P.AdaptScalepCN(C, 0.9), 400),
        ("mala", lambda: P.MALA(0.1, g), 200),
        ("hmc_L5", lambda: P.VanillaHMC(0.1, 5, g), 100),
        ("hmc_L5_mass", lambda: P.VanillaHMC(0.1, 5, g, M=np.linalg.inv(C)), 100),
        ("hmc_L5_adapt_scale", lambda: P.AdaptScaleHMC(0.1, 5, g), 100),
        ("hmc_L5_mass_adapt_scale", lambda: P.AdaptScaleHMC(0.1, 5, g, M=np.linalg.inv(C)), 100),
    ]


def one(d, kind="gauss", chains=65536):
    rng = np.random.default_rng(d)
    A = rng.standard_normal((d, d)) / np.sqrt(d)
    C = A @ A.T + np.eye(d)
    div = 1
    if kind == "gauss":
        model = ra.models.MultiGaussianDist(np.zeros(d), C)
    elif kind == "mixture":
        model = ra.models.MixtureDist([[0.3, ra.models.MultiGaussianDist(np.full(d, -0.5), C)],
                                       [0.7, ra.models.MultiGaussianDist(np.full(d, 0.5), C)]])
    else:
        from riemann_amd.workloads import logreg_data
        X, y = logreg_data(1000, d)[:2]
        model = ra.models.LogisticRegression(X, y)
        C, div = 0.01 * C, 4
    rows = []
    for name, make, its in variants(model, C):
        its = max(its // div, 10)
        s = ra.Sampler(model, make(), np.zeros((chains, d)), seed=1)
        s.run(its, record=False)
        s.run(its, record=False)
        ms = s.last_run_ms / its
        rows.append({"d": d, "model": kind, "variant": name, "ms_per_iteration": ms})
        print("d=%4d  %-26s %8.4f ms / iteration" % (d, name, ms), flush=True)
    return rows


if __name__ == "__main__":
    args = sys.argv[1:]
    out = None
    if args and args[0] == "--json":
        out, args = args[1], args[2:]
    kind = "gauss"
    if args and args[0] == "--model":
        kind, args = args[1], args[2:]
    rows = []
    for d in [int(a) for a in args] or [100, 96, 36]:
        rows += one(d, kind)
    if out:
        with open(out, "w") as f:
            json.dump({"chains": 65536, "recorded": False, "rows": rows}, f, indent=1)
