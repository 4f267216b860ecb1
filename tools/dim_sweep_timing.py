"""Random-walk MH on a d-dimensional Gaussian for a sweep of d: ms per 1000 iterations of
65 536 chains (device timer) and the fraction of the FP64 roof (2 d (d + 1) + 4 d flop per
chain-iteration over 78.6 TFLOP/s).  Shows what padding rows / k-steps cost below a tile boundary,
and what the VALU tail rows (k_run_rw_tail<NF>, d = 16 NF + 1 .. 16 NF + 4) give back: every such d
is timed a second time on the generic padded-tile kernel (kernel policy NO_TAIL).

    python tools/dim_sweep_timing.py [--json out.json] [d ...]
"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import riemann_amd as ra                      # noqa: E402
from riemann_amd._lib import kernel_policy    # noqa: E402

PEAK = 78.6e12


def one(d, chains=65536, its=1000):
    rng = np.random.default_rng(d)
    A = rng.standard_normal((d, d)) / np.sqrt(d)
    C = A @ A.T + np.eye(d)
    model = ra.models.MultiGaussianDist(np.zeros(d), C)
    prop = ra.proposals.MetropolisRandomWalk(2.38 ** 2 / d * C)
    s = ra.Sampler(model, prop, np.zeros((chains, d)), seed=1)
    s.run(its, record=False)  # warm-up
    s.run(its, record=False)
    return s.last_run_ms


if __name__ == "__main__":
    args = sys.argv[1:]
    out = None
    if args and args[0] == "--json":
        out, args = args[1], args[2:]
    dims = [int(a) for a in args] or [16, 17, 20, 32, 33, 34, 35, 36, 37, 48, 49, 52, 53, 64, 65, 68, 69, 70, 80, 81, 84, 85,
                                      96, 97, 100, 101, 112]
    rows = []
    for d in dims:
        ms = one(d)
        fl = 2 * d * (d + 1) + 4 * d
        row = {"d": d, "ms_per_1000": ms, "frac_fp64_roof": fl * 65536 * 1000 / (ms * 1e-3) / PEAK}
        line = "d=%4d  %8.2f ms / 1000 its   %.3f of the FP64 roof" % (d, ms, row["frac_fp64_roof"])
        if d > 16 and d <= 100 and (d - 1) % 16 < 4:
            with kernel_policy("NO_TAIL"):
                row["ms_per_1000_padded_tile_kernel"] = one(d)
            line += "   (padded-tile kernel: %.2f ms)" % row["ms_per_1000_padded_tile_kernel"]
        rows.append(row)
        print(line, flush=True)
    if out:
        with open(out, "w") as f:
            json.dump({"chains": 65536, "iterations": 1000, "recorded": False, "rows": rows}, f, indent=1)
