#!/usr/bin/env python3
"""Chains sharded over the GPUs of one node, pooled through the library's own RCCL communicator.
No launcher is needed -- the script starts one process per GPU itself --

    python examples/multi_gpu_mi355x.py [ngpus]

but any launcher that sets RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT works too
(python -m torch.distributed.run --nproc-per-node N examples/multi_gpu_mi355x.py; nothing of
PyTorch is imported).  Chains keep their random streams whatever the number of GPUs (streams are
keyed by global chain id), so the pooled sample is the single-GPU sample."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if "RANK" not in os.environ:          # start one rank per GPU
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    from riemann_amd.distributed import spawn_ranks, visible_devices
    if n == 0:
        n = visible_devices()
    sys.exit(spawn_ranks([sys.executable, os.path.abspath(__file__)], n))   # 0 only if every rank succeeded

import riemann_amd as ra                                              # noqa: E402
from riemann_amd import _lib, distributed as D                        # noqa: E402
from riemann_amd.models import benchmarks                             # noqa: E402
from riemann_amd.proposals import MetropolisRandomWalk                # noqa: E402
from riemann_amd.workloads import corr_gauss_cov                      # noqa: E402

group = D.HostGroup.from_env()
_lib.call("rmh_init", int(os.environ.get("LOCAL_RANK", "0")))
D.comm_init(group)                                     # rank 0's RCCL id over the socket group
N, d = 65536 * group.world, 100
lo, hi = D.shard_range(N, group.rank, group.world)
Ct = corr_gauss_cov(d)
th0 = np.random.default_rng(1).standard_normal((N, d))[lo:hi] @ np.linalg.cholesky(Ct).T
s = ra.Sampler(benchmarks.benchmark_gauss100d_corr, MetropolisRandomWalk((2.38 ** 2 / d) * Ct), th0,
               seed=1, chain0=lo)
s.run(1000, record=False)
pooled, mean, cov, info = D.pool(s)                    # one ncclAllGather + one ncclAllReduce
if group.rank == 0:
    print("%d ranks, %d chains pooled (%.1f MB per rank in %.2f ms); |mean|max %.3f, mean variance %.3f (target 1.0)"
          % (info["rccl_ranks"], info["pooled_chains"], info["allgather_bytes_per_rank"] / 1e6, info["allgather_ms"],
             np.abs(mean).max(), np.diag(cov).mean()))
# diagnostics of a recorded stretch over the chains of ALL ranks: every rank reduces its device history, the ranks
# exchange a few small all-reduces, and every rank gets the numbers one process with all chains would report
s.run(8000, 0, 80)
st = D.pool_history(s, nlags=64, samples_every=25)
if group.rank == 0:
    print("pooled over %d chains: split R-hat max %.4f, tau_int max %.1f kept states, ESS min %.3g; %d thinned samples "
          "of every chain gathered" % (st["chains"], st["rhat"].max(), st["tau"].max(), st["ess"].min(), st["samples"].shape[0]))
_lib.call("rmh_comm_barrier")
_lib.call("rmh_comm_destroy")
group.close()
