"""What Sampler(..., normals="f64") costs: the BASELINE shapes in both normal modes, 65 536 chains
(ms per iteration, device timer).  In the 64-bit mode the d = 100 shapes run the generic kernels
(the VALU-tail kernels draw a compile-time number of 32-bit normals), so the random walk pays for
both the longer stream and the padded row tile.

    python tools/normals64_timing.py [--json out.json]
"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import riemann_amd as ra                                   # noqa: E402
from riemann_amd import proposals as P                     # noqa: E402
from riemann_amd._lib import kernel_policy                 # noqa: E402
from riemann_amd.workloads import corr_gauss_cov, logreg_data  # noqa: E402

N = 65536


def timed(model, make, d, its, **kw):
    s = ra.Sampler(model, make(), np.zeros((N, d)), seed=1, **kw)
    s.run(its, record=False)
    s.run(its, record=False)
    return s.last_run_ms / its


if __name__ == "__main__":
    rows = []
    for d in (100, 96, 32):
        C = corr_gauss_cov(d)
        m = ra.models.MultiGaussianDist(np.zeros(d), C)
        mk = lambda: P.MetropolisRandomWalk(2.38 ** 2 / d * C)          # noqa: E731
        row = {"workload": "gauss rw d=%d" % d, "f32x2": timed(m, mk, d, 400), "f64": timed(m, mk, d, 400, normals="f64")}
        if d == 100:
            with kernel_policy("NO_TAIL"):
                row["f32x2_generic_kernel"] = timed(m, mk, d, 400)
        rows.append(row)
    X, y = logreg_data(10000, 100)
    m = ra.models.LogisticRegression(X, y)
    mk = lambda: P.MALA(0.02, m)                                        # noqa: E731
    row = {"workload": "logreg 10000x100 mala", "f32x2": timed(m, mk, 100, 20), "f64": timed(m, mk, 100, 20, normals="f64")}
    with kernel_policy("NO_TAIL"):
        row["f32x2_generic_kernel"] = timed(m, mk, 100, 20)
    rows.append(row)
    for r in rows:
        print("  ".join("%s=%s" % (k, v if isinstance(v, str) else "%.4f" % v) for k, v in r.items()), flush=True)
    if "--json" in sys.argv:
        with open(sys.argv[sys.argv.index("--json") + 1], "w") as f:
            json.dump({"chains": N, "unit": "ms per iteration", "rows": rows}, f, indent=1)
