"""Logistic-regression MALA (cfg3's kernel family) for a sweep of the dimension: ms per iteration of
65 536 chains on n = 10 000 data rows and the fraction of the FP64 roof (4 n d flop per
chain-iteration over 78.6 TFLOP/s).  d = 16 NF + 1 .. 16 NF + 4 runs the gradient's last column tile on
the VALU (LogRegModel<NT, true>); each such d is timed a second time on the padded-tile kernel
(kernel policy NO_TAIL).

    python tools/logreg_dim_timing.py [--json out.json] [d ...]
"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import riemann_amd as ra                      # noqa: E402
from riemann_amd._lib import kernel_policy    # noqa: E402
from riemann_amd.workloads import logreg_data  # noqa: E402

PEAK, N, ROWS = 78.6e12, 65536, 10000


def one(d, its=20):
    X, y = logreg_data(ROWS, d, seed=2)
    m = ra.models.LogisticRegression(X, y, 100.0)
    s = ra.Sampler(m, ra.proposals.MALA(0.175 * np.sqrt(100.0 / d) * 0.5, m), np.zeros((N, d)), seed=1)
    s.run(4, record=False)
    s.run(its, record=False)
    return s.last_run_ms / its


if __name__ == "__main__":
    args = sys.argv[1:]
    out = None
    if args and args[0] == "--json":
        out, args = args[1], args[2:]
    dims = [int(a) for a in args] or [16, 20, 32, 36, 48, 52, 64, 68, 80, 84, 96, 100, 112]
    rows = []
    for d in dims:
        ms = one(d)
        row = {"d": d, "ms_per_iteration": ms, "frac_fp64_roof": 4.0 * ROWS * d * N / (ms * 1e-3) / PEAK}
        line = "d=%4d  %8.3f ms / iteration   %.3f of the FP64 roof" % (d, ms, row["frac_fp64_roof"])
        if d > 16 and d <= 100 and (d - 1) % 16 < 4:
            with kernel_policy("NO_TAIL"):
                row["ms_per_iteration_padded_tile_kernel"] = one(d)
            line += "   (padded-tile kernel: %.3f ms)" % row["ms_per_iteration_padded_tile_kernel"]
        rows.append(row)
        print(line, flush=True)
    if out:
        with open(out, "w") as f:
            json.dump({"chains": N, "data_rows": ROWS, "recorded": False, "rows": rows}, f, indent=1)
