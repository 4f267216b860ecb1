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
import hashlib
import os
import secrets
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


def spawn_ranks(argv, n, env=None, poll=0.1):
    """Start ``n`` ranks of ``argv`` (a command line: list of str) on this node with RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_ADDR / MASTER_PORT set, and wait.  Returns 0 iff every rank exited 0.  A rank
    that fails -- including one killed by a signal, which ``Popen`` reports as a negative code that
    ``max()`` over the ranks' codes would hide behind the others' zeros -- is reported with its code,
    and the remaining ranks are terminated so that they do not wait for it in a collective."""
    import subprocess
    import sys
    base = dict(os.environ if env is None else env)
    base.update(WORLD_SIZE=str(n), MASTER_ADDR=base.get("MASTER_ADDR", "127.0.0.1"),
                MASTER_PORT=base.get("MASTER_PORT", "0"))
    procs = [subprocess.Popen(list(argv), env=dict(base, RANK=str(r), LOCAL_RANK=str(r))) for r in range(n)]
    codes = [None] * n
    failed = False
    while any(c is None for c in codes):
        for r, p in enumerate(procs):
            if codes[r] is None:
                codes[r] = p.poll()
                if codes[r] not in (None, 0) and not failed:
                    failed = True
                    print("rank %d exited with code %d; stopping the other ranks" % (r, codes[r]), file=sys.stderr)
                    for q in procs:
                        if q.poll() is None:
                            q.terminate()
        if any(c is None for c in codes):
            time.sleep(poll)
    return 0 if all(c == 0 for c in codes) else 1


def visible_devices():
    """Number of HIP devices, counted in a child process: a launcher that goes on to start its ranks
    must not have initialised the GPU itself (fork + exec from a process that has is refused on the
    GPU boxes of this pool)."""
    import subprocess
    import sys
    code = ("import ctypes as C, sys; sys.path.insert(0, %r); from riemann_amd import _lib; n = C.c_int(); "
            "_lib.call('rmh_device_count', C.byref(n)); print(n.value)"
            % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    return int(subprocess.check_output([sys.executable, "-c", code]).split()[-1])


# ---------------------------------------------------------------- host group --
# Wire format: fixed struct frames, no pickle (nothing a peer sends is ever executed).  A value is
#   N                      None            b <u8>             bool
#   i <i64>                int             f <f64>            float
#   B <u64 n> bytes        bytes           s <u64 n> utf-8    str
#   A <u8 len> dtype.str <u8 ndim> <u64 shape...> raw C-order data      numpy array (numeric dtypes)
#   T / L <u32 n> values   tuple / list    (nesting <= 4)
# A message is <u64 length> + one encoded value, at most _MAX_FRAME bytes.
_MAX_FRAME = 1 << 30
_MAX_DEPTH = 4
_MAGIC = b"RMHRDZV2"


def _enc(o, out, depth=0):
    if depth > _MAX_DEPTH:
        raise ValueError("riemann_amd.distributed: value nested too deeply for the wire format")
    if o is None:
        out.append(b"N")
    elif isinstance(o, (bool, np.bool_)):
        out.append(b"b" + struct.pack("<B", 1 if o else 0))
    elif isinstance(o, (int, np.integer)):
        out.append(b"i" + struct.pack("<q", int(o)))
    elif isinstance(o, (float, np.floating)):
        out.append(b"f" + struct.pack("<d", float(o)))
    elif isinstance(o, (bytes, bytearray)):
        out.append(b"B" + struct.pack("<Q", len(o)))
        out.append(bytes(o))
    elif isinstance(o, str):
        b = o.encode("utf-8")
        out.append(b"s" + struct.pack("<Q", len(b)))
        out.append(b)
    elif isinstance(o, np.ndarray):
        if o.dtype.kind not in "biuf":
            raise TypeError("riemann_amd.distributed: only numeric arrays travel (got %s)" % o.dtype)
        a = np.ascontiguousarray(o)
        ds = a.dtype.str.encode("ascii")
        out.append(b"A" + struct.pack("<B", len(ds)) + ds + struct.pack("<B", a.ndim) +
                   struct.pack("<%dQ" % a.ndim, *a.shape))
        out.append(a.tobytes())
    elif isinstance(o, (tuple, list)):
        out.append((b"T" if isinstance(o, tuple) else b"L") + struct.pack("<I", len(o)))
        for x in o:
            _enc(x, out, depth + 1)
    else:
        raise TypeError("riemann_amd.distributed: %s does not travel over the host group" % type(o).__name__)


def _dec(buf, pos, depth=0):
    if depth > _MAX_DEPTH:
        raise ValueError("bad frame: nesting")
    tag = buf[pos:pos + 1]
    pos += 1
    if tag == b"N":
        return None, pos
    if tag == b"b":
        return buf[pos] != 0, pos + 1
    if tag == b"i":
        return struct.unpack_from("<q", buf, pos)[0], pos + 8
    if tag == b"f":
        return struct.unpack_from("<d", buf, pos)[0], pos + 8
    if tag in (b"B", b"s"):
        (n,) = struct.unpack_from("<Q", buf, pos)
        pos += 8
        if n > len(buf) - pos:
            raise ValueError("bad frame: length")
        b = bytes(buf[pos:pos + n])
        return (b if tag == b"B" else b.decode("utf-8")), pos + n
    if tag == b"A":
        nd = buf[pos]
        dt = np.dtype(bytes(buf[pos + 1:pos + 1 + nd]).decode("ascii"))
        pos += 1 + nd
        if dt.kind not in "biuf":
            raise ValueError("bad frame: dtype")
        ndim = buf[pos]
        pos += 1
        shape = struct.unpack_from("<%dQ" % ndim, buf, pos)
        pos += 8 * ndim
        n = int(np.prod(shape, dtype=np.int64)) * dt.itemsize if ndim else dt.itemsize
        if n > len(buf) - pos:
            raise ValueError("bad frame: array length")
        a = np.frombuffer(bytes(buf[pos:pos + n]), dtype=dt).reshape(shape).copy()
        return a, pos + n
    if tag in (b"T", b"L"):
        (n,) = struct.unpack_from("<I", buf, pos)
        pos += 4
        if n > len(buf):
            raise ValueError("bad frame: count")
        items = []
        for _ in range(n):
            x, pos = _dec(buf, pos, depth + 1)
            items.append(x)
        return (tuple(items) if tag == b"T" else items), pos
    raise ValueError("bad frame: tag %r" % tag)


def _send(sock, obj):
    out = []
    _enc(obj, out)
    b = b"".join(out)
    if len(b) > _MAX_FRAME:
        raise ValueError("riemann_amd.distributed: message of %d bytes exceeds the frame limit" % len(b))
    sock.sendall(struct.pack("<Q", len(b)) + b)


def _exactly(sock, n):
    buf = bytearray()
    while len(buf) < n:
        chunk = sock.recv(min(n - len(buf), 1 << 20))
        if not chunk:
            raise ConnectionError("riemann_amd.distributed: peer closed the rendezvous socket")
        buf += chunk
    return bytes(buf)


def _recv(sock):
    (n,) = struct.unpack("<Q", _exactly(sock, 8))
    if n > _MAX_FRAME:
        raise ConnectionError("riemann_amd.distributed: frame of %d bytes refused" % n)
    obj, pos = _dec(_exactly(sock, n), 0)
    if pos != n:
        raise ConnectionError("riemann_amd.distributed: trailing bytes in a frame")
    return obj


def _is_loopback(host):
    """True for 127.0.0.0/8, ::1 and ``localhost``."""
    h = str(host).strip().lower()
    return h == "localhost" or h == "::1" or h.startswith("127.") or h == "::ffff:127.0.0.1"


def _token(key, extra=b""):
    return hashlib.sha256(b"rmh-rdzv|" + str(key).encode("utf-8") + b"|" + extra).digest()[:16]


class HostGroup(object):
    """The ranks of one job as a star over TCP sockets (rank 0 = hub): small host-side
    collectives -- broadcast / all-gather of plain values (None, bool, int, float, bytes, str,
    numeric numpy arrays, tuples and lists of those), barrier, max -- used to hand out the RCCL id,
    and as the fallback where RCCL cannot form a communicator (several ranks on one GPU).  Never on
    the data path: samples move over RCCL (:func:`pool`).

    Rendezvous.  Single node (``port=None``, the default): rank 0 listens on an ephemeral port of
    **127.0.0.1 only** and publishes ``port token`` in a file readable by its user alone, named
    after the launcher's process id and MASTER_PORT under the temp directory (all ranks of a node
    share both) -- no second well-known port next to the launcher's own MASTER_PORT store.
    Across nodes: ``port`` (``RMH_RDZV_PORT``) on ``addr`` (``MASTER_ADDR``); peers that are not
    on the loopback interface are accepted only then.  Every peer opens with a fixed hello frame
    -- magic, rank, 16-byte token (the file's random token, or one derived from the job key and
    ``RMH_RDZV_TOKEN``) -- and a connection with a wrong hello, an out-of-range or duplicate rank,
    or from a non-loopback address in single-node mode is dropped without disturbing the group.
    Nothing received is ever unpickled or executed (fixed struct frames, see ``_enc``)."""

    def __init__(self, rank, world, addr="127.0.0.1", port=None, key=None, timeout=300.0):
        self.rank, self.world = int(rank), int(world)
        self._peers, self._hub, self._file = [], None, None
        self.bound = None
        if self.world == 1:
            return
        if not (0 <= self.rank < self.world):
            raise ValueError("rank %d outside world of %d" % (self.rank, self.world))
        if key is None:
            key = "%d_%s" % (os.getppid(), os.environ.get("MASTER_PORT", "0"))
        path = os.path.join(tempfile.gettempdir(), "rmh_rdzv_%s" % key)
        deadline = time.time() + timeout
        env_tok = os.environ.get("RMH_RDZV_TOKEN", "").encode("utf-8")
        if self.rank == 0:
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            # loopback unless a multi-node rendezvous was asked for explicitly
            bind_addr = addr if port is not None else "127.0.0.1"
            remote_ok = port is not None and not _is_loopback(bind_addr)
            srv.bind((bind_addr, int(port or 0)))
            srv.listen(self.world + 4)
            srv.settimeout(1.0)
            self.bound = srv.getsockname()
            if port is None:
                token = secrets.token_bytes(16)
                tmp = "%s.%d" % (path, os.getpid())
                fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o600)
                with os.fdopen(fd, "w") as f:
                    f.write("%d %s" % (self.bound[1], token.hex()))
                os.replace(tmp, path)     # atomic: readers see the whole line or nothing
                self._file = path
            else:
                token = _token(key, env_tok)
            peers = {}
            while len(peers) < self.world - 1:
                if time.time() > deadline:
                    srv.close()
                    raise TimeoutError("riemann_amd.distributed: %d of %d ranks joined the rendezvous"
                                       % (len(peers) + 1, self.world))
                try:
                    c, peer = srv.accept()
                except socket.timeout:
                    continue
                r = self._hello(c, peer, token, remote_ok)
                if r is None or r in peers:
                    c.close()             # stray, foreign or duplicate connection: dropped
                    continue
                c.settimeout(timeout)
                peers[r] = c
            srv.close()
            self._peers = [peers[r] for r in range(1, self.world)]
        else:
            while True:
                try:
                    p, token = port, _token(key, env_tok)
                    if p is None:
                        with open(path) as f:
                            ps, ts = f.read().split()
                        p, token = int(ps), bytes.fromhex(ts)
                    s = socket.create_connection((addr, int(p)), timeout=5.0)
                    break
                except (OSError, ValueError):
                    if time.time() > deadline:
                        raise TimeoutError("riemann_amd.distributed: no rendezvous with rank 0 (%s)" % path)
                    time.sleep(0.05)
            s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            s.settimeout(timeout)
            s.sendall(_MAGIC + struct.pack("<ii", self.rank, self.world) + token)
            if _exactly(s, 1) != b"\x01":
                raise ConnectionError("riemann_amd.distributed: rank 0 refused the rendezvous")
            self._hub = s

    def _hello(self, c, peer, token, remote_ok):
        """Rank of a connecting peer, or None if the connection is not one of ours."""
        try:
            if not remote_ok and not _is_loopback(peer[0]):
                return None
            c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            c.settimeout(5.0)
            h = _exactly(c, len(_MAGIC) + 8 + 16)
            r, w = struct.unpack_from("<ii", h, len(_MAGIC))
            if (h[:len(_MAGIC)] != _MAGIC or w != self.world or not (1 <= r < self.world) or
                    not secrets.compare_digest(h[len(_MAGIC) + 8:], token)):
                return None
            c.sendall(b"\x01")
            return r
        except (OSError, ConnectionError, struct.error):
            return None

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


def _allreduce_vec(v):
    """Sum of a float64 vector over the ranks of the library's communicator (RCCL all-reduce)."""
    from . import _lib
    a = np.ascontiguousarray(v, dtype=np.float64).copy()
    _lib.call("rmh_comm_allreduce_host", a, int(a.size), 0)
    return a


def pool_history(sampler, nlags=128, c=5.0, samples_every=None, allreduce=None, rank=None, world=None):
    """Diagnostics of the last recorded ``run()`` pooled over the chains of ALL ranks without any
    history leaving a GPU: every rank reduces its device history to per-chain
    moments and integer autocorrelation sums (``Sampler.history_pieces``), the ranks exchange a few
    small all-reduces, and every rank gets the same dictionary --

    ``mean``, ``var`` (d,) pooled moments; ``cov`` (d, d) pooled covariance over all chains and kept
    states (one all-reduce of the device-formed moment sums; d <= 111, else absent); ``acf`` (nlags,
    d), ``tau``, ``closed``, ``ess`` (d,); ``rhat`` (d,) the split R-hat over the 2 M half chains of
    all M chains; ``chains``.  mean / var / acf / tau / ess / rhat are bit for bit what one process
    holding all chains reports (``Sampler.history_stats``): integer autocorrelation sums and
    exactly rounded sums of the per-chain moments (``diagnostics.pooled_history_stats``).

    ``samples_every=k`` adds ``samples``: kept states 0, k, 2k, ... of all chains, (n_sel, M, d), by
    ncclAllGather (``rmh_allgather_history``) -- thinned samples, not only the final state.
    ``allreduce`` / ``rank`` / ``world`` default to the library's RCCL communicator."""
    import ctypes as C
    from . import _lib, diagnostics as dg
    if allreduce is None:
        nr, rk = comm_ranks()
        if nr == 0:
            raise RuntimeError("pool_history: no communicator (comm_init first)")
        allreduce, rank, world = _allreduce_vec, rk, nr
    piece = sampler.history_pieces(nlags)
    out = dg.pooled_history_stats(piece, allreduce, rank, world, c)
    d = sampler.Npars
    if d <= 111:
        mom = np.empty(1 + d + d * d)
        _lib.call("rmh_history_moments", sampler._hist, 0, sampler._kept, mom)
        mom = np.asarray(allreduce(mom))
        m1 = mom[1:1 + d] / mom[0]
        out["cov"] = mom[1 + d:].reshape(d, d) / mom[0] - np.outer(m1, m1)
    if samples_every:
        k = int(samples_every)
        nsel = (sampler._kept + k - 1) // k
        pooled = np.empty((nsel, out["chains"], d))
        _lib.call("rmh_allgather_history", sampler._hist, 0, sampler._kept, k, pooled)
        out["samples"] = pooled
    return out
