"""Test-side twins of the pooling collectives over ``torch.distributed`` (gloo on CPU): the
product pools through the library's own RCCL communicator (riemann_amd/distributed.py,
csrc/rmh_comm.hip), which needs GPUs; these restate the same layout -- ragged shards padded
to the largest block for ONE all-gather, ONE all-reduce of the moment sums -- so that the
N > 1 path is exercised without a GPU.  torch is test plumbing only."""
import numpy as np

from riemann_amd.distributed import shard_range


def pool_samples(local, n_chains, group=None):
    """All-gather per-rank sample blocks ``(..., n_local, d)`` (chain axis = -2) into the
    pooled ``(..., n_chains, d)`` array, on every rank."""
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
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t, group=group)
    parts = [o.numpy()[..., : sizes[r], :] for r, o in enumerate(out)]
    return np.concatenate(parts, axis=-2)


def pooled_moments(local, group=None):
    """Pooled mean and covariance over all chains of all ranks from ONE all-reduce of
    [count, sum, sum of outer products]."""
    import torch
    import torch.distributed as dist
    x = np.asarray(local, dtype=np.float64).reshape(-1, np.shape(local)[-1])
    d = x.shape[1]
    acc = np.concatenate([[x.shape[0]], x.sum(axis=0), (x.T @ x).ravel()])
    t = torch.from_numpy(acc)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    acc = t.numpy()
    n, s, ss = acc[0], acc[1: d + 1], acc[d + 1:].reshape(d, d)
    mean = s / n
    return mean, ss / n - np.outer(mean, mean)
