"""ctypes binding of csrc/libriemann_hip.so (C ABI: include/riemann_hip.h).

There is no CPU implementation behind this module: if the library has not
been built, or no MI355X is visible when a compute entry point is called, the
call raises.  Build with ``python -m riemann_amd.build`` (hipcc, gfx950).
"""
import ctypes as C
import os

import numpy as np

from .sampling_errors import ParameterError

# Sampler(..., normals=...): resolution of the stream's standard normals (rmh_state_set_normal_mode)
NORMAL_MODES = {"f32x2": 0, "f64": 1}


def normal_mode(name):
    if name not in NORMAL_MODES:
        raise ParameterError("normals must be 'f32x2' (two normals per stream word, the default) or 'f64' "
                             "(one per word), not {!r}".format(name))
    return NORMAL_MODES[name]

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("RIEMANN_HIP_LIB") or os.path.join(_HERE, "csrc", "libriemann_hip.so")

RMH_OK, RMH_ERR_PARAM, RMH_ERR_HIP, RMH_ERR_STATE = 0, 1, 2, 3

_f64 = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_u64 = np.ctypeslib.ndpointer(dtype=np.uint64, flags="C_CONTIGUOUS")
_u8 = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")
_i32 = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
_vp, _i64, _u64s, _dbl, _int = C.c_void_p, C.c_int64, C.c_uint64, C.c_double, C.c_int
_pvp = C.POINTER(C.c_void_p)

# name -> argtypes; every function returns int except rmh_last_error
SIGNATURES = {
    "rmh_init": [_int],
    "rmh_device_count": [C.POINTER(_int)],
    "rmh_device_name": [C.c_char_p, _int],
    "rmh_synchronize": [],
    "rmh_last_run_ms": [C.POINTER(_dbl), C.POINTER(_int)],
    "rmh_model_gaussian_create": [_f64, _f64, _int, _pvp],
    "rmh_model_logreg_create": [_f64, _f64, _i64, _int, _dbl, _pvp],
    "rmh_model_changepoint_create": [_f64, _f64, _int, _int, _dbl, _dbl, _dbl, _dbl, _dbl, _pvp],
    "rmh_model_mixture_create": [_f64, _f64, _f64, _int, _int, _dbl, _pvp],
    "rmh_model_changepoint_rj_create": [_f64, _f64, _int, _int, _dbl, _dbl, _dbl, _dbl, _dbl, _pvp],
    "rmh_model_dim": [_vp, C.POINTER(_int)],
    "rmh_model_free": [_vp],
    "rmh_log_prior": [_vp, _f64, _i64, _f64],
    "rmh_log_likelihood": [_vp, _f64, _i64, _f64],
    "rmh_log_posterior": [_vp, _f64, _i64, _f64],
    "rmh_grad_log_posterior": [_vp, _f64, _i64, _f64],
    "rmh_proposal_rw_create": [_f64, _int, _dbl, _pvp],
    "rmh_proposal_rw_set_blocks": [_vp, _int, _i32, _f64],
    "rmh_proposal_cprj_create": [_dbl, _f64, _int, _pvp],
    "rmh_proposal_cprj_set_probs": [_vp, _f64],
    "rmh_proposal_set_adapt_cov": [_vp, _vp, _dbl, _int, _int, _int],
    "rmh_proposal_adapt_cov_get": [_vp, _f64],
    "rmh_proposal_adapt_cov_set": [_vp, _f64, _dbl],
    "rmh_proposal_pcn_create": [_f64, _int, _dbl, _pvp],
    "rmh_proposal_hmc_create": [_dbl, _int, _int, _pvp],
    "rmh_proposal_hmc_set_mass": [_vp, _vp],
    "rmh_proposal_set_scale": [_vp, _dbl],
    "rmh_proposal_set_eps": [_vp, _dbl],
    "rmh_proposal_set_adapt": [_vp, _dbl],
    "rmh_proposal_free": [_vp],
    "rmh_propose": [_vp, _vp, _f64, _i64, _u64, _f64, _f64],
    "rmh_propose_mode": [_vp, _vp, _f64, _i64, _u64, _int, _f64, _f64],
    "rmh_rng_draw": [_u64, _i64, _int, _int, _vp, _vp],
    "rmh_rng_draw_mode": [_u64, _i64, _int, _int, _int, _vp, _vp],
    "rmh_state_set_normal_mode": [_vp, _int],
    "rmh_state_get_normal_mode": [_vp, _vp],
    "rmh_state_create": [_vp, _i64, _u64s, _u64s, _pvp],
    "rmh_state_set_theta": [_vp, _f64],
    "rmh_state_get": [_vp, _vp, _vp],
    "rmh_state_get_rng": [_vp, _u64],
    "rmh_state_set_rng": [_vp, _u64],
    "rmh_state_get_adapt": [_vp, _f64],
    "rmh_state_set_adapt": [_vp, _f64],
    "rmh_state_get_adapt_aux": [_vp, _f64],
    "rmh_state_set_adapt_aux": [_vp, _f64],
    "rmh_state_device_ptrs": [_vp, _pvp, _pvp, C.POINTER(_i64)],
    "rmh_state_free": [_vp],
    "rmh_history_create": [_vp, _i64, _i64, _pvp],
    "rmh_history_entries": [_vp, C.POINTER(_i64), C.POINTER(_i64)],
    "rmh_host_alloc": [_pvp, _u64s],
    "rmh_host_free": [_vp],
    "rmh_history_download": [_vp, _i64, _i64, _vp, _vp],
    "rmh_history_reduce": [_vp, _i64, _i64, _int, _vp, _vp, _vp],
    "rmh_history_accept": [_vp, _i64, _i64, _u8],
    "rmh_history_device_ptrs": [_vp, _pvp, _pvp, _pvp, C.POINTER(_i64)],
    "rmh_history_free": [_vp],
    "rmh_run": [_vp, _vp, _vp, _i64, _i64, _i64, _vp],
    "rmh_run_pt": [_vp, _vp, _vp, _f64, _int, _dbl, _i64, _i64, _i64, _vp],
    "rmh_comm_unique_id": [_u8],
    "rmh_comm_init": [_u8, _int, _int],
    "rmh_comm_destroy": [],
    "rmh_comm_info": [C.POINTER(_int), C.POINTER(_int)],
    "rmh_comm_barrier": [],
    "rmh_comm_allreduce_host": [_f64, _i64, _int],
    "rmh_comm_pooled_chains": [_vp, C.POINTER(_i64), _vp],
    "rmh_comm_last_ms": [C.POINTER(_dbl), C.POINTER(_i64)],
    "rmh_allgather_samples": [_vp, _f64],
    "rmh_allreduce_moments": [_vp, _f64],
    "rmh_debug_rng_stream": [_u64s, _u64s, _i64, _int, _int, _f64, _f64],
    "rmh_debug_rng_stream_mode": [_u64s, _u64s, _i64, _int, _int, _int, _f64, _f64],
    "rmh_history_reduce_sums": [_vp, _i64, _i64, _int, _vp, _vp, _vp],
    "rmh_history_moments": [_vp, _i64, _i64, _f64],
    "rmh_allgather_history": [_vp, _i64, _i64, _i64, _f64],
    "rmh_debug_mfma": [_f64, _f64, _f64, _f64],
    "rmh_debug_set_kernel_policy": [_int, C.POINTER(_int)],
}

_lib = None


def load():
    """The CDLL, loaded once.  Loading needs no GPU; calling compute entries does."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise ImportError(
            "riemann_amd: %s is missing -- build it with `python -m riemann_amd.build` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback." % SO_PATH)
    lib = C.CDLL(SO_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = _int
    lib.rmh_last_error.argtypes = []
    lib.rmh_last_error.restype = C.c_char_p
    _lib = lib
    return lib


def check(rc):
    """Map a C-ABI return code to the exceptions of the plugin API."""
    if rc == RMH_OK:
        return
    msg = load().rmh_last_error().decode("utf-8", "replace")
    if rc == RMH_ERR_PARAM:
        raise ParameterError(msg)  # riemann/proposals/randomwalk.py:23-24, models/gaussian.py:35-39
    if rc == RMH_ERR_STATE:
        raise IndexError(msg)      # e.g. current_state() of an empty history, sampler.py:60
    raise RuntimeError("riemann_amd (HIP): " + msg)


def call(name, *args):
    check(getattr(load(), name)(*args))


def as_f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


# ---- kernel policy (A/B measurements, cross-checks) ---------------------------------------------
POLICY = {"NO_TAIL": 0x01, "CP_NO_SS": 0x02, "CP_SS64": 0x04, "CP_SS16": 0x08, "CP_NO_BUCKETS": 0x10,
          "CPRJ_NO_SS": 0x20, "ACOV_HOST": 0x40, "ACOV_UNFUSED": 0x80}


class kernel_policy(object):
    """``with kernel_policy("CP_SS64", "CP_NO_BUCKETS"): ...`` -- rmh_debug_set_kernel_policy for the
    block (include/riemann_hip.h, RMH_POLICY_*).  The library reads no environment variable."""

    def __init__(self, *names):
        self.mask = 0
        for n in names:
            self.mask |= POLICY[n]

    def __enter__(self):
        old = _int()
        call("rmh_debug_set_kernel_policy", self.mask, C.byref(old))
        self.old = old.value
        return self

    def __exit__(self, *exc):
        call("rmh_debug_set_kernel_policy", self.old, None)
        return False


# ---- pinned host arrays (history downloads at the PCIe rate, rmh_host_alloc) --------------------
class _PinnedAlloc(object):
    """One page-locked allocation; freed when the pool and every lease of it are gone."""

    def __init__(self, nbytes):
        p = C.c_void_p()
        call("rmh_host_alloc", C.byref(p), nbytes)
        self.addr, self.nbytes, self.busy = p.value, nbytes, False

    def release(self):
        self.busy = False

    def __del__(self):
        try:
            load().rmh_host_free(self.addr)
        except Exception:
            pass


class PinnedPool(object):
    """Page-locked numpy arrays for one owner (a Sampler's history downloads).  Pinning costs about
    as much as a pageable transfer, so allocations are kept and handed out again -- but only when
    no array made from the previous hand-out is alive any more (the user may still hold the
    previous run's history).  Liveness is explicit: every hand-out wraps the allocation in a fresh
    lease object that is the ``base`` of the array and of all its views; a ``weakref.finalize`` on
    the lease marks the allocation free."""

    def __init__(self, max_bytes=None):
        self.max_bytes = int(os.environ.get("RMH_PINNED_MAX_BYTES", 4 << 30)) if max_bytes is None else max_bytes
        self._allocs = []

    def empty(self, shape, dtype=np.float64):
        """A pinned array of the shape, or a pageable one when the pool's budget does not allow it."""
        import weakref
        n = int(np.prod(shape, dtype=np.int64)) * np.dtype(dtype).itemsize
        if n == 0 or n > self.max_bytes:
            return np.empty(shape, dtype=dtype)
        pick = None
        for a in self._allocs:
            if a.nbytes >= n and not a.busy and (pick is None or a.nbytes < pick.nbytes):
                pick = a
        if pick is None:
            self._allocs = [a for a in self._allocs if a.busy]     # drop idle smaller ones
            if sum(a.nbytes for a in self._allocs) + n > self.max_bytes:
                return np.empty(shape, dtype=dtype)
            try:
                pick = _PinnedAlloc(n)
            except RuntimeError:
                return np.empty(shape, dtype=dtype)
            self._allocs.append(pick)
        lease = type("PinnedLease", (C.c_char * n,), {}).from_address(pick.addr)
        pick.busy = True
        weakref.finalize(lease, pick.release)      # (the bound method keeps the allocation alive)
        return np.frombuffer(lease, dtype=dtype, count=n // np.dtype(dtype).itemsize).reshape(shape)
