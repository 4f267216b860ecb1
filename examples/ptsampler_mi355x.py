#!/usr/bin/env python3
"""Parallel tempering on a bimodal target, many ladders at once on one MI355X.

The counterpart of driving /root/reference/riemann/samplers/ptsampler.py by hand:
the 1-d two-component mixture benchmark (models/benchmarks.py:75-78) has modes at
-5 and +5 that a plain random walk with unit steps practically never crosses; the
beta = 1 rung of a tempered ladder does, through swaps with the hotter rungs.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import riemann_amd as ra                                     # noqa: E402
from riemann_amd.models import benchmarks                    # noqa: E402
from riemann_amd.proposals import MetropolisRandomWalk       # noqa: E402
from riemann_amd.samplers import PTSampler                   # noqa: E402


def main():
    target = benchmarks.benchmark_gauss1d_mix2          # 0.3 N(-5,1) + 0.7 N(+5,1)
    n = 3072                                            # ladders (or plain chains)
    theta0 = np.full((n, 1), 5.0)                       # everything starts in the right-hand mode
    T = 4000

    plain = ra.Sampler(target, MetropolisRandomWalk(np.array([[1.0]])), theta0, seed=1)
    plain.run(T)
    left_plain = (plain._chain_thetas[T // 2:, :, 0] < 0).mean()

    pt = PTSampler(target, MetropolisRandomWalk(np.array([[1.0]])), theta0,
                   betas=0.35 ** np.arange(8), Pswap=0.2, seed=1)
    pt.run(T)
    left_pt = (pt._chain_thetas[T // 2:, :, 0] < 0).mean()
    codes = pt._chain_codes
    swaps = (codes >> 1).mean()
    swap_acc = ((codes == 3).sum() / max(1, (codes >= 2).sum()))
    print("mass of the left mode (exact 0.30): plain RW %.3f, cold rung of PT %.3f" % (left_plain, left_pt))
    print("PT: %d ladders x %d temperatures, %.0f%% of replica-sweeps were swap proposals, %.0f%% accepted, "
          "kernel %.1f ms for %d sweeps" % (n, pt.Ntemps, 100 * swaps, 100 * swap_acc, pt.last_run_ms, T))


if __name__ == "__main__":
    main()
