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
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from oracle import oracle as orc
    from dist_gloo import pool_samples, pooled_moments
    from riemann_amd.distributed import shard_range
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
    # pooled R-hat / tau / ESS from per-rank reductions (the history never travels): the product's
    # combination code (riemann_amd.diagnostics.pooled_history_stats, used by distributed.pool_history)
    # over a gloo all-reduce, the rank's reductions from the numpy twin of rmh_history_reduce_sums
    import torch
    from riemann_amd import diagnostics as dg

    def allreduce(v):
        t = torch.from_numpy(np.ascontiguousarray(v, dtype=np.float64).copy())
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return t.numpy()
    st = dg.pooled_history_stats(dg.history_pieces_numpy(ht, 32), allreduce, rank, world)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), pooled=pooled, mean=mean, cov=cov,
             **{"st_" + k: np.asarray(v) for k, v in st.items()})
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
    from riemann_amd import diagnostics as dg
    one = dg.pooled_history_stats(dg.history_pieces_numpy(ht, 32))       # one process, all chains
    for r in range(world):
        z = np.load(os.path.join(str(tmp_path), "rank%d.npz" % r))
        np.testing.assert_array_equal(z["pooled"], ht)
        np.testing.assert_allclose(z["mean"], ht[-1].mean(axis=0), rtol=1e-12, atol=1e-14)
        np.testing.assert_allclose(z["cov"], np.cov(ht[-1].T, bias=True), rtol=1e-10, atol=1e-12)
        # pooled diagnostics: bit for bit the single-process values, on every rank
        for k in ("mean", "var", "acf", "tau", "ess", "rhat", "chains"):
            np.testing.assert_array_equal(z["st_" + k], np.asarray(one[k]), err_msg=k)
    np.testing.assert_allclose(one["rhat"], dg.split_rhat(ht), rtol=1e-12)
    np.testing.assert_allclose(one["mean"], ht.mean(axis=(0, 1)), rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(one["var"], ht.reshape(-1, ht.shape[-1]).var(axis=0), rtol=1e-11)


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


# ---- the product's own rendezvous (pure Python, no torch): riemann_amd.distributed.HostGroup ----
def _host_worker(rank, world, key, n_chains, out_dir):
    sys.path.insert(0, ROOT)
    from oracle import oracle as orc
    from riemann_amd.distributed import HostGroup, shard_range
    assert "torch" not in sys.modules
    g = HostGroup(rank, world, key=key, timeout=60.0)
    uid = g.broadcast(bytes(range(128)) if rank == 0 else None)          # the RCCL id hand-off
    assert uid == bytes(range(128))
    assert g.allgather(("r", rank)) == [("r", r) for r in range(world)]
    assert g.max(10.0 + rank) == 10.0 + world - 1
    g.barrier()
    # a sharded run pooled through the host group (small arrays; the data path proper is RCCL)
    d, T, seed = 5, 12, 7
    rg = np.random.default_rng(1)
    th0 = rg.standard_normal((n_chains, d))
    lo, hi = shard_range(n_chains, rank, world)
    m, p = orc.Gaussian(np.zeros(d), np.eye(d)), orc.RandomWalk(0.3 * np.eye(d))
    th = np.ascontiguousarray(th0[lo:hi]).copy()
    ht, _, _ = orc.run(m, p, th, m.log_posterior(th), orc.rng_seed(seed, hi - lo, chain0=lo), T)
    pooled = np.concatenate(g.allgather(ht[-1]), axis=0)
    np.save(os.path.join(out_dir, "host_rank%d.npy" % rank), pooled)
    from riemann_amd import diagnostics as dg
    st = dg.pooled_history_stats(dg.history_pieces_numpy(ht, 32),
                                 lambda v: np.sum(g.allgather(np.asarray(v, dtype=np.float64)), axis=0), rank, world)
    np.savez(os.path.join(out_dir, "host_stats%d.npz" % rank), **{k: np.asarray(v) for k, v in st.items()})
    g.barrier()
    g.close()


@pytest.mark.parametrize("world,n_chains", [(2, 24), (3, 25), (8, 67)])   # 8: the ranks of one node
def test_host_group_rendezvous_and_pooling(tmp_path, world, n_chains):
    import multiprocessing as mp
    from oracle import oracle as orc
    ctx = mp.get_context("spawn")
    key = "test_%d_%d" % (os.getpid(), world)
    procs = [ctx.Process(target=_host_worker, args=(r, world, key, n_chains, str(tmp_path)))
             for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    d, T, seed = 5, 12, 7
    th0 = np.random.default_rng(1).standard_normal((n_chains, d))
    m, p = orc.Gaussian(np.zeros(d), np.eye(d)), orc.RandomWalk(0.3 * np.eye(d))
    th = th0.copy()
    ht, _, _ = orc.run(m, p, th, m.log_posterior(th), orc.rng_seed(seed, n_chains), T)
    from riemann_amd import diagnostics as dg
    one = dg.pooled_history_stats(dg.history_pieces_numpy(ht, 32))
    for r in range(world):
        np.testing.assert_array_equal(np.load(os.path.join(str(tmp_path), "host_rank%d.npy" % r)), ht[-1])
        z = np.load(os.path.join(str(tmp_path), "host_stats%d.npz" % r))
        for k in ("mean", "var", "acf", "tau", "ess", "rhat", "chains"):
            np.testing.assert_array_equal(z[k], np.asarray(one[k]), err_msg=k)


def test_host_group_wire_format_and_rendezvous_hardening(tmp_path):
    """no pickle on the wire, loopback bind by default, hello with rank check and token;
    a stray or foreign connection is dropped and the group still forms."""
    import struct
    import threading
    import time
    import riemann_amd.distributed as D
    src = open(D.__file__).read()
    assert "import pickle" not in src and "pickle.loads" not in src
    # round trip of every value kind that travels; unknown kinds are refused, not pickled
    vals = (None, True, 7, -2.5, b"\x00\x01", "id", np.arange(6, dtype=np.int64).reshape(2, 3),
            [np.float64(1.5), ("r", 3)])
    out = []
    D._enc(vals, out)
    back, pos = D._dec(b"".join(out), 0)
    assert pos == len(b"".join(out)) and back[:6] == vals[:6]
    np.testing.assert_array_equal(back[6], vals[6])
    assert back[7] == [1.5, ("r", 3)]
    with pytest.raises(TypeError):
        D._enc(object(), [])
    with pytest.raises(TypeError):
        D._enc(np.array(["a"], dtype=object), [])
    with pytest.raises(ValueError):
        D._dec(b"X", 0)
    assert D._is_loopback("127.0.0.1") and D._is_loopback("localhost") and D._is_loopback("::1")
    assert not D._is_loopback("10.0.0.5") and not D._is_loopback("192.168.1.2")
    # rendezvous of two ranks in threads; a stranger connects first with garbage and with a wrong token
    key = "harden_%d" % os.getpid()
    res = {}

    def run(rank):
        g = D.HostGroup(rank, 2, key=key, timeout=30.0)
        res[rank] = (g.bound, g.allgather(("rank", rank)))
        g.barrier()
        g.close()
    t0 = threading.Thread(target=run, args=(0,))
    t0.start()
    path = os.path.join(__import__("tempfile").gettempdir(), "rmh_rdzv_%s" % key)
    for _ in range(400):
        if os.path.exists(path):
            break
        time.sleep(0.01)
    port, tok = open(path).read().split()
    assert oct(os.stat(path).st_mode & 0o777) == "0o600"          # the token is the user's alone
    for hello in (b"GET / HTTP/1.0\r\n\r\n" + b"\0" * 32,
                  D._MAGIC + struct.pack("<ii", 1, 2) + b"\0" * 16,           # wrong token
                  D._MAGIC + struct.pack("<ii", 5, 2) + bytes.fromhex(tok)):    # rank out of range
        c = socket.create_connection(("127.0.0.1", int(port)), timeout=5.0)
        c.sendall(hello)
        c.settimeout(5.0)
        try:
            assert c.recv(1) == b""                               # dropped without an answer
        except (ConnectionError, socket.timeout):
            pass
        c.close()
    t1 = threading.Thread(target=run, args=(1,))
    t1.start()
    t0.join(40)
    t1.join(40)
    assert res[0][1] == res[1][1] == [("rank", 0), ("rank", 1)]
    assert res[0][0][0] == "127.0.0.1"                            # rank 0 listened on loopback only
    assert not os.path.exists(path)


def test_product_imports_no_torch():
    """north_star: pure Python + ctypes host, no PyTorch in the product package."""
    import re
    import subprocess
    pkg = os.path.join(ROOT, "riemann_amd")
    pat = re.compile(r"^\s*(import|from)\s+torch\b")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                with open(os.path.join(dirpath, f)) as fh:
                    for i, line in enumerate(fh, 1):
                        assert not pat.match(line), "%s:%d imports torch" % (os.path.join(dirpath, f), i)
    code = ("import sys; sys.path.insert(0, %r); import riemann_amd, riemann_amd.distributed, "
            "riemann_amd.samplers, riemann_amd.proposals, riemann_amd.models, riemann_amd.diagnostics; "
            "assert 'torch' not in sys.modules" % ROOT)
    subprocess.check_call([sys.executable, "-c", code])
