"""Multi-GPU layout of the chains: one process per GPU, chains sharded by
GLOBAL chain id, no collective inside the sampling loop; the only exchange is
an all-gather of samples (and an all-reduce of moment sums) for pooled
diagnostics.

The reference has nothing distributed (README.md:21-23 states the aspiration).
The random stream is keyed by (seed, global chain id), so the samples of chain
c are the same whatever the number of ranks -- sharding is exact, not just
statistically equivalent.

Pure Python + the library's own RCCL communicator (``rmh_comm_*``, RCCL over
xGMI; csrc/rmh_comm.hip): no PyTorch anywhere in this package.  The only thing
the ranks exchange outside RCCL is the 128-byte communicator id, which
:class:`HostGroup` carries over a loopback/TCP socket -- any launcher that sets
RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (``torch.distributed.run``,
``mpirun`` wrappers, a shell loop) works, and nothing of the launcher is
imported here.
"""
import os
import pickle
import socket
import struct
import tempfile
import time

import numpy as np


def shard_range(n_chains, rank, world):
    """Contiguous block [lo, hi) of global chain ids owned by ``rank``.
    Blocks differ by at most one chain; lo is the ``chain0`` of the rank's Sampler."""
    if not (0 <= rank < world):
        raise ValueError("rank %d outside world of %d" % (rank, world))
    base, extra = divmod(int(n_chains), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


# ---------------------------------------------------------------- host group --
def _send(sock, obj):
    b = pickle.dumps(obj, protocol=4)
    sock.sendall(struct.pack("<Q", len(b)) + b)


def _recv(sock):
    def exactly(n):
        buf = bytearray()
        while len(buf) < n:
            chunk = sock.recv(n - len(buf))
            if not chunk:
                raise ConnectionError("riemann_amd.distributed: peer closed the rendezvous socket")
            buf += chunk
        return bytes(buf)
    (n,) = struct.unpack("<Q", exactly(8))
    return pickle.loads(exactly(n))


class HostGroup(object):
    """The ranks of one job as a star over TCP sockets (rank 0 = hub): small host-side
    collectives -- broadcast / all-gather of Python objects, barrier, max -- used to hand out
    the RCCL id, and as the fallback where RCCL cannot form a communicator (several ranks on
    one GPU).  Never on the data path: samples move over RCCL (:func:`pool`).

    Rendezvous: rank 0 listens on ``port`` of ``addr``.  With ``port=None`` it takes an
    ephemeral port and publishes it in a file named after the launcher's process id and
    MASTER_PORT under the temp directory (all ranks of a node share both), so no second
    well-known port is needed next to the launcher's own MASTER_PORT store."""

    def __init__(self, rank, world, addr="127.0.0.1", port=None, key=None, timeout=300.0):
        self.rank, self.world = int(rank), int(world)
        self._peers, self._hub, self._file = [], None, None
        if self.world == 1:
            return
        if key is None:
            key = "%d_%s" % (os.getppid(), os.environ.get("MASTER_PORT", "0"))
        path = os.path.join(tempfile.gettempdir(), "rmh_rdzv_%s" % key)
        deadline = time.time() + timeout
        if self.rank == 0:
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind((addr if port is not None else "", int(port or 0)))
            srv.listen(self.world)
            srv.settimeout(timeout)
            if port is None:
                tmp = "%s.%d" % (path, os.getpid())
                with open(tmp, "w") as f:
                    f.write("%d" % srv.getsockname()[1])
                os.replace(tmp, path)     # atomic: readers see the whole number or nothing
                self._file = path
            peers = {}
            while len(peers) < self.world - 1:
                c, _ = srv.accept()
                c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                c.settimeout(timeout)
                peers[_recv(c)] = c
            srv.close()
            self._peers = [peers[r] for r in range(1, self.world)]
        else:
            while True:
                try:
                    p = port
                    if p is None:
                        with open(path) as f:
                            p = int(f.read())
                    s = socket.create_connection((addr, int(p)), timeout=5.0)
                    break
                except (OSError, ValueError):
                    if time.time() > deadline:
                        raise TimeoutError("riemann_amd.distributed: no rendezvous with rank 0 (%s)" % path)
                    time.sleep(0.05)
            s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            s.settimeout(timeout)
            _send(s, self.rank)
            self._hub = s

    @classmethod
    def from_env(cls, **kw):
        """From the launcher's environment (RANK, WORLD_SIZE, MASTER_ADDR; the rendezvous port is
        RMH_RDZV_PORT if set -- needed across nodes -- else published through the temp directory)."""
        port = os.environ.get("RMH_RDZV_PORT")
        return cls(int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
                   addr=os.environ.get("MASTER_ADDR", "127.0.0.1"),
                   port=int(port) if port else None, **kw)

    def allgather(self, obj):
        """[obj of rank 0, obj of rank 1, ...] on every rank."""
        if self.world == 1:
            return [obj]
        if self.rank == 0:
            out = [obj] + [_recv(c) for c in self._peers]
            for c in self._peers:
                _send(c, out)
            return out
        _send(self._hub, obj)
        return _recv(self._hub)

    def broadcast(self, obj, src=0):
        return self.allgather(obj if self.rank == src else None)[src]

    def barrier(self):
        self.allgather(None)

    def max(self, x):
        return max(self.allgather(float(x)))

    def close(self):
        for c in self._peers + ([self._hub] if self._hub else []):
            try:
                c.close()
            except OSError:
                pass
        self._peers, self._hub = [], None
        if self._file:
            try:
                os.unlink(self._file)
            except OSError:
                pass
            self._file = None


# ------------------------------------------------- RCCL through the C ABI ----
def comm_init(group):
    """Create the library's RCCL communicator over the ranks of ``group``: rank 0 draws the
    128-byte id (``rmh_comm_unique_id``), the group broadcasts it, every rank joins
    (``rmh_comm_init``).  The device must be bound (``rmh_init``) first."""
    from . import _lib
    uid = np.zeros(128, dtype=np.uint8)
    if group.rank == 0:
        _lib.call("rmh_comm_unique_id", uid)
    uid = np.frombuffer(group.broadcast(uid.tobytes()), dtype=np.uint8).copy()
    _lib.call("rmh_comm_init", uid, int(group.world), int(group.rank))


def comm_ranks():
    """(ranks, rank) of the library's communicator; (0, 0) when there is none."""
    import ctypes as C
    from . import _lib
    n, r = C.c_int(), C.c_int()
    _lib.call("rmh_comm_info", C.byref(n), C.byref(r))
    return n.value, r.value


def pool(sampler):
    """Pooled diagnostics of a Sampler's current states over all ranks, device resident until
    the final copy: ``(states (sum N_r, d), mean (d,), covariance (d, d), info)`` from ONE
    ncclAllGather of the ranks' state blocks (ragged shards are padded to the largest) and ONE
    ncclAllReduce of the moment sums ``[count, sum theta, sum theta theta^T]``."""
    import ctypes as C
    from . import _lib
    d = sampler.Npars
    total = C.c_int64()
    _lib.call("rmh_comm_pooled_chains", sampler._state, C.byref(total), None)
    pooled = np.empty((total.value, d))
    _lib.call("rmh_allgather_samples", sampler._state, pooled)
    ms, nb = C.c_double(), C.c_int64()
    _lib.call("rmh_comm_last_ms", C.byref(ms), C.byref(nb))
    mom = np.empty(1 + d + d * d)
    _lib.call("rmh_allreduce_moments", sampler._state, mom)
    mean = mom[1:1 + d] / mom[0]
    cov = mom[1 + d:].reshape(d, d) / mom[0] - np.outer(mean, mean)
    info = {"rccl_ranks": comm_ranks()[0], "allgather_bytes_per_rank": int(nb.value),
            "allgather_ms": float(ms.value), "pooled_chains": int(total.value)}
    return pooled, mean, cov, info
