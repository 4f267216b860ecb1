#!/usr/bin/env python3
"""Two ranks driving the library's own RCCL communicator (rmh_comm_*): all-gather of states,
all-reduce of moments, and the covariance adaptation pooled over ranks.  Launch with
  python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 tools/cabi_comm_2rank.py
(gloo only carries the 128-byte RCCL id).  On a one-GPU box both ranks share the device, which
RCCL may refuse ("duplicate GPU"): the script reports that and exits 0."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch.distributed as dist                                     # noqa: E402

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
import riemann_amd as ra                                             # noqa: E402
from riemann_amd import _lib, distributed as D                        # noqa: E402
from riemann_amd.models import MultiGaussianDist                     # noqa: E402
from riemann_amd.proposals import AdaptCovRandomWalk, MetropolisRandomWalk  # noqa: E402

ndev = int(os.environ.get("RMH_NDEV", "1"))
_lib.call("rmh_init", rank % ndev)


def bcast(b):
    obj = [b]
    dist.broadcast_object_list(obj, src=0)
    return obj[0]


try:
    D.cabi_comm_init(rank, world, bcast)
except RuntimeError as e:
    print("rank %d: RCCL communicator not available here: %s" % (rank, str(e)[:200]), flush=True)
    dist.barrier()
    sys.exit(0)

d, N = 12, 480
g = np.random.default_rng(0)
th0 = g.standard_normal((N, d))
lo, hi = D.shard_range(N, rank, world)
m = MultiGaussianDist(np.zeros(d), np.eye(d))
s = ra.Sampler(m, MetropolisRandomWalk(0.2 * np.eye(d)), th0[lo:hi], seed=3, chain0=lo)
s.run(20, record=False)
pooled = np.empty((N, d))
_lib.call("rmh_allgather_samples", s._state, pooled)
ref = ra.Sampler(m, MetropolisRandomWalk(0.2 * np.eye(d)), th0, seed=3)
ref.run(20, record=False)
assert np.array_equal(pooled, ref.current_state()[0]), "all-gather != single-process run"
# covariance adaptation pooled over the ranks == single process over all chains (sums differ in order)
C0 = 0.1 * np.eye(d)
pa = AdaptCovRandomWalk(C0, t_adapt=100, smooth_adapt=True)
sa = ra.Sampler(m, pa, th0[lo:hi], seed=4, chain0=lo)
sa.run(30)
_lib.call("rmh_comm_destroy")
pb = AdaptCovRandomWalk(C0, t_adapt=100, smooth_adapt=True)
sb = ra.Sampler(m, pb, th0, seed=4)
sb.run(30)
np.testing.assert_allclose(pa.C, pb.C, rtol=1e-10)
np.testing.assert_allclose(sa._chain_thetas, sb._chain_thetas[:, lo:hi], rtol=1e-9, atol=1e-12)
print("rank %d of %d: C-ABI RCCL all-gather and rank-pooled covariance adaptation agree with one process" % (rank, world), flush=True)
dist.barrier()
