#!/usr/bin/env python3
"""Several ranks driving the library's own RCCL communicator (rmh_comm_*): all-gather of ragged
state shards, all-reduce of moments, and the covariance adaptation pooled over ranks.

    python tools/cabi_comm_2rank.py --spawn 2          # starts its own ranks (no launcher needed)
    python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 tools/cabi_comm_2rank.py
    python tools/cabi_comm_2rank.py --scale            # the hardware check of an 8-GPU node, one command:
        the correctness check above at every rank count the node has, then the bench lines of BASELINE
        cfg4 (logistic HMC, 65 536 chains per GPU at 1/2/4/8 GPUs = 524 288 at 8) and cfg5 (changepoint
        walk, 65 536 per GPU at 1/2/4 = 262 144 at 4), each with "pool": {"rccl_ranks": N, ...}

No PyTorch is imported: the 128-byte RCCL id travels over riemann_amd.distributed.HostGroup.
Rank r binds device r % RMH_NDEV (default: all visible devices).  On a one-GPU box both ranks
share the device, which RCCL refuses ("duplicate GPU"): the script reports that and exits 0."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if "--spawn" in sys.argv:
    from riemann_amd.distributed import spawn_ranks
    n = int(sys.argv[sys.argv.index("--spawn") + 1])
    sys.exit(spawn_ranks([sys.executable, os.path.abspath(__file__)], n))   # 0 only if every rank succeeded

if "--scale" in sys.argv:
    from riemann_amd.distributed import spawn_ranks, visible_devices
    ndev = visible_devices()
    counts = [n for n in (1, 2, 4, 8) if n <= ndev]
    rc = 0
    for n in counts[1:]:
        print("== correctness, %d ranks" % n, flush=True)
        rc |= spawn_ranks([sys.executable, os.path.abspath(__file__)], n)
    bench = os.path.join(ROOT, "bench.py")
    for wl, most in (("logreg_hmc", 8), ("changepoint_rw", 4)):
        for n in [c for c in counts if c <= most]:
            print("== bench.py --gpus %d --workload %s" % (n, wl), flush=True)
            rc |= spawn_ranks([sys.executable, bench, "--gpus", str(n), "--workload", wl, "--no-cpu-baseline"], n)
    sys.exit(rc)

import riemann_amd as ra                                             # noqa: E402
from riemann_amd import _lib, distributed as D                        # noqa: E402
from riemann_amd.models import MultiGaussianDist                     # noqa: E402
from riemann_amd.proposals import AdaptCovRandomWalk, MetropolisRandomWalk  # noqa: E402

group = D.HostGroup.from_env()
rank, world = group.rank, group.world
nd = C.c_int()
_lib.call("rmh_device_count", C.byref(nd))
ndev = int(os.environ.get("RMH_NDEV", str(nd.value)))
_lib.call("rmh_init", rank % ndev)

try:
    D.comm_init(group)
    ok = True
except RuntimeError as e:
    print("rank %d: RCCL communicator not available here: %s" % (rank, str(e)[:200]), flush=True)
    ok = False
if not all(group.allgather(ok)):
    group.barrier()
    group.close()
    sys.exit(0)

d, N = 12, 481                                   # ragged: the shards differ by one chain
g = np.random.default_rng(0)
th0 = g.standard_normal((N, d))
lo, hi = D.shard_range(N, rank, world)
m = MultiGaussianDist(np.zeros(d), np.eye(d))
s = ra.Sampler(m, MetropolisRandomWalk(0.2 * np.eye(d)), th0[lo:hi], seed=3, chain0=lo)
s.run(20, record=False)
pooled, mean, cov, info = D.pool(s)
ref = ra.Sampler(m, MetropolisRandomWalk(0.2 * np.eye(d)), th0, seed=3)
ref.run(20, record=False)
full = ref.current_state()[0]
assert np.array_equal(pooled, full), "all-gather != single-process run"
assert info["rccl_ranks"] == world and info["pooled_chains"] == N
np.testing.assert_allclose(mean, full.mean(axis=0), rtol=1e-11, atol=1e-13)
np.testing.assert_allclose(cov, np.cov(full.T, bias=True), rtol=1e-9, atol=1e-12)
# pooled R-hat / tau / ESS of a recorded run from per-rank device reductions == one process, bit for bit;
# thinned samples by all-gather
s.run(60, 10, 1)
ph = D.pool_history(s, nlags=32, samples_every=10)
ref.run(60, 10, 1)
hs = ref.history_stats(nlags=32)
for k in ("acf", "tau", "ess", "rhat"):
    assert np.array_equal(ph[k], hs[k]), "pooled %s != single-process value" % k
assert np.array_equal(ph["mean"], hs["pooled_mean"]) and np.array_equal(ph["var"], hs["pooled_var"])
assert np.array_equal(ph["samples"], ref._chain_thetas[::10]), "all-gather of thinned samples"
np.testing.assert_allclose(ph["cov"], ref.history_moments()[2], rtol=1e-9, atol=1e-12)
# covariance adaptation pooled over the ranks == single process over all chains (sums differ in order)
C0 = 0.1 * np.eye(d)
pa = AdaptCovRandomWalk(C0, t_adapt=100, smooth_adapt=True)
sa = ra.Sampler(m, pa, th0[lo:hi], seed=4, chain0=lo)
sa.run(30)
_lib.call("rmh_comm_barrier")
_lib.call("rmh_comm_destroy")
pb = AdaptCovRandomWalk(C0, t_adapt=100, smooth_adapt=True)
sb = ra.Sampler(m, pb, th0, seed=4)
sb.run(30)
np.testing.assert_allclose(pa.C, pb.C, rtol=1e-10)
np.testing.assert_allclose(sa._chain_thetas, sb._chain_thetas[:, lo:hi], rtol=1e-9, atol=1e-12)
print("rank %d of %d: C-ABI RCCL all-gather (%d B/rank, %.3f ms) and rank-pooled covariance adaptation agree "
      "with one process" % (rank, world, info["allgather_bytes_per_rank"], info["allgather_ms"]), flush=True)
group.barrier()
group.close()
