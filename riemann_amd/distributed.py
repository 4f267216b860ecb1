"""Multi-GPU layout of the chains: one process per GPU, chains sharded by
GLOBAL chain id, no collective inside the sampling loop; the only exchange is
an all-gather of samples for pooled diagnostics.

The reference has nothing distributed (README.md:21-23 states the aspiration).
The random stream is keyed by (seed, global chain id), so the samples of chain
c are the same whatever the number of ranks -- sharding is exact, not just
statistically equivalent.

``torch.distributed`` is plumbing here: backend "nccl" is RCCL over xGMI on the
MI355X node, "gloo" is used by the CPU tests.
"""
import numpy as np


def shard_range(n_chains, rank, world):
    """Contiguous block [lo, hi) of global chain ids owned by ``rank``.
    Blocks differ by at most one chain; lo is the ``chain0`` of the rank's Sampler."""
    if not (0 <= rank < world):
        raise ValueError("rank %d outside world of %d" % (rank, world))
    base, extra = divmod(int(n_chains), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def pool_samples(local, n_chains, group=None, device=None):
    """All-gather per-rank sample blocks ``(..., n_local, d)`` (chain axis = -2)
    into the pooled ``(..., n_chains, d)`` array, on every rank.

    Ragged shards are padded to the largest block for the collective (one
    all_gather of equal-sized tensors) and trimmed afterwards."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    sizes = [shard_range(n_chains, r, world)[1] - shard_range(n_chains, r, world)[0]
             for r in range(world)]
    nmax = max(sizes)
    a = np.ascontiguousarray(local, dtype=np.float64)
    if a.shape[-2] != sizes[dist.get_rank(group)]:
        raise ValueError("local block has %d chains, this rank owns %d"
                         % (a.shape[-2], sizes[dist.get_rank(group)]))
    pad = np.zeros(a.shape[:-2] + (nmax, a.shape[-1]))
    pad[..., : a.shape[-2], :] = a
    t = torch.from_numpy(pad)
    if device is not None:
        t = t.to(device)
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t, group=group)
    parts = [o.cpu().numpy()[..., : sizes[r], :] for r, o in enumerate(out)]
    return np.concatenate(parts, axis=-2)


def pooled_moments(local, group=None, device=None):
    """Pooled mean and covariance over all chains of all ranks from ONE
    all-reduce of [count, sum, sum of outer products]."""
    import torch
    import torch.distributed as dist
    x = np.asarray(local, dtype=np.float64).reshape(-1, np.shape(local)[-1])
    d = x.shape[1]
    acc = np.concatenate([[x.shape[0]], x.sum(axis=0), (x.T @ x).ravel()])
    t = torch.from_numpy(acc)
    if device is not None:
        t = t.to(device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    acc = t.cpu().numpy()
    n, s, ss = acc[0], acc[1: d + 1], acc[d + 1:].reshape(d, d)
    mean = s / n
    return mean, ss / n - np.outer(mean, mean)


# ---- the same two collectives through the C ABI (RCCL opened by the library) ----
def cabi_comm_init(rank, world, broadcast_bytes):
    """Create the library's RCCL communicator.  ``broadcast_bytes(b)`` must return
    rank 0's 128-byte id on every rank (e.g. a torch.distributed / file broadcast)."""
    from . import _lib
    uid = np.zeros(128, dtype=np.uint8)
    if rank == 0:
        _lib.call("rmh_comm_unique_id", uid)
    uid = np.frombuffer(broadcast_bytes(uid.tobytes()), dtype=np.uint8).copy()
    _lib.call("rmh_comm_init", uid, int(world), int(rank))


def cabi_pool(sampler, world):
    """(pooled states (world*N, d), pooled mean (d,), pooled covariance (d,d)) of a
    Sampler's current states over all ranks: one ncclAllGather of the device state
    blocks and one ncclAllReduce of the moment accumulators."""
    from . import _lib
    N, d = sampler.Nchains, sampler.Npars
    pooled = np.empty((world * N, d))
    _lib.call("rmh_allgather_samples", sampler._state, pooled)
    mom = np.empty(1 + d + d * d)
    _lib.call("rmh_allreduce_moments", sampler._state, mom)
    mean = mom[1:1 + d] / mom[0]
    return pooled, mean, mom[1 + d:].reshape(d, d) / mom[0] - np.outer(mean, mean)
