"""world_size-2 (and 3, ragged) CPU test of the multi-GPU path over gloo: chains
sharded by global chain id, each rank samples its shard (here with the CPU
oracle standing in for the device), the ranks all-gather their samples, and the
pooled result equals a single-process run of all chains bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_chains, out_dir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from oracle import oracle as orc
    from riemann_amd.distributed import pool_samples, pooled_moments, shard_range
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank,
                            world_size=world)
    d, T, seed = 6, 20, 11
    g = np.random.default_rng(0)
    A = g.standard_normal((d, d))
    Cov = A @ A.T + d * np.eye(d)
    th0 = g.standard_normal((n_chains, d))
    lo, hi = shard_range(n_chains, rank, world)
    m, p = orc.Gaussian(np.zeros(d), Cov), orc.RandomWalk(0.5 * Cov / d)
    th = np.ascontiguousarray(th0[lo:hi]).copy()
    st = orc.rng_seed(seed, hi - lo, chain0=lo)          # streams keyed by GLOBAL chain id
    ht, hl, acc = orc.run(m, p, th, m.log_posterior(th), st, T)
    pooled = pool_samples(ht, n_chains)                  # (T+1, n_chains, d) on every rank
    mean, cov = pooled_moments(ht[-1])
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), pooled=pooled, mean=mean, cov=cov)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_chains", [(2, 40), (3, 37)])
def test_sharded_run_pools_to_the_single_process_run(tmp_path, world, n_chains):
    import torch.multiprocessing as mp
    from oracle import oracle as orc
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n_chains, str(tmp_path)), nprocs=world, join=True)
    d, T, seed = 6, 20, 11
    g = np.random.default_rng(0)
    A = g.standard_normal((d, d))
    Cov = A @ A.T + d * np.eye(d)
    th0 = g.standard_normal((n_chains, d))
    m, p = orc.Gaussian(np.zeros(d), Cov), orc.RandomWalk(0.5 * Cov / d)
    th = th0.copy()
    ht, _, _ = orc.run(m, p, th, m.log_posterior(th), orc.rng_seed(seed, n_chains), T)
    for r in range(world):
        z = np.load(os.path.join(str(tmp_path), "rank%d.npz" % r))
        np.testing.assert_array_equal(z["pooled"], ht)
        np.testing.assert_allclose(z["mean"], ht[-1].mean(axis=0), rtol=1e-12, atol=1e-14)
        np.testing.assert_allclose(z["cov"], np.cov(ht[-1].T, bias=True), rtol=1e-10, atol=1e-12)


def test_shard_range_partitions_exactly():
    from riemann_amd.distributed import shard_range
    for n in (1, 7, 16, 65536, 524288, 1000003):
        for w in (1, 2, 3, 4, 8):
            blocks = [shard_range(n, r, w) for r in range(w)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in blocks]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(10, 2, 2)
