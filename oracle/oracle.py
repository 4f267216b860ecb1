"""ctypes front-end of the CPU ORACLE (test infrastructure, NOT the product).

Loads ``oracle/librmh_oracle.so`` (built from ``rmh_oracle.c`` by
``oracle/Makefile``; rebuilt on demand with gcc).  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import this module; nothing under ``riemann_amd/`` does.

Array conventions here are the reference's: theta is ``(N, d)`` row-major,
histories are ``(T+1, N, d)`` with entry 0 the state on entry to ``run``
(/root/reference/riemann/samplers/sampler.py:49-54).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# RMH_ORACLE_SO: another build of the same source (the sanitizer build of oracle/Makefile)
_SO = os.environ.get("RMH_ORACLE_SO") or os.path.join(_HERE, "librmh_oracle.so")
_lib = None

_f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_u64p = np.ctypeslib.ndpointer(dtype=np.uint64, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")


def build(force=False):
    src = os.path.join(_HERE, "rmh_oracle.c")
    if os.environ.get("RMH_ORACLE_SO"):
        return _SO
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "librmh_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_SO)
    vp, i64, u64, dbl, i32 = C.c_void_p, C.c_int64, C.c_uint64, C.c_double, C.c_int
    L.orc_log.restype = dbl
    L.orc_log.argtypes = [dbl]
    L.orc_normal_from_half.restype = dbl
    L.orc_normal_from_half.argtypes = [C.c_uint32]
    L.orc_normal_from_word.restype = dbl
    L.orc_normal_from_word.argtypes = [u64]
    L.orc_set_normal_mode.argtypes = [i32]
    L.orc_get_normal_mode.restype = i32
    L.orc_rng_seed.argtypes = [u64, u64, i64, _u64p]
    L.orc_rng_stream.argtypes = [_u64p, i64, i32, i32, _f64p, _f64p]
    L.orc_model_gaussian.restype = vp
    L.orc_model_gaussian.argtypes = [_f64p, _f64p, i32]
    L.orc_model_logreg.restype = vp
    L.orc_model_logreg.argtypes = [_f64p, _f64p, i64, i32, dbl]
    L.orc_model_changepoint.restype = vp
    L.orc_model_changepoint.argtypes = [_f64p, _f64p, i32, i32, dbl, dbl, dbl, dbl, dbl]
    L.orc_model_free.argtypes = [vp]
    L.orc_model_dim.argtypes = [vp]
    for name in ("orc_log_prior", "orc_log_likelihood", "orc_log_posterior",
                 "orc_grad_log_posterior"):
        getattr(L, name).argtypes = [vp, _f64p, i64, _f64p]
    L.orc_changepoint_indices.argtypes = [vp, _f64p, i64, _i32p]
    L.orc_proposal_rw.restype = vp
    L.orc_proposal_rw.argtypes = [_f64p, i32, dbl]
    L.orc_proposal_pcn.restype = vp
    L.orc_proposal_pcn.argtypes = [_f64p, i32, dbl]
    L.orc_proposal_hmc.restype = vp
    L.orc_proposal_hmc.argtypes = [dbl, i32, i32]
    L.orc_proposal_free.argtypes = [vp]
    L.orc_propose.argtypes = [vp, vp, _f64p, i64, _u64p, _f64p, _f64p]
    L.orc_proposal_set_adapt.argtypes = [vp, dbl]
    L.orc_proposal_set_mass.argtypes = [vp, vp]
    L.orc_run_adapt.restype = i32
    L.orc_run_adapt.argtypes = [vp, vp, i64, _f64p, _f64p, _u64p, i32, vp, vp, vp, i32, vp, vp]
    L.orc_run.restype = i32
    L.orc_run.argtypes = [vp, vp, i64, _f64p, _f64p, _u64p, i32, vp, vp, vp, i32]
    L.orc_rng_normals.argtypes = [_u64p, i32, _f64p]
    L.orc_rng_words.argtypes = [_u64p, _f64p]
    L.orc_rng_pt_stream.argtypes = [_u64p, i64, i32, i32, _f64p, _f64p, _f64p, _f64p]
    L.orc_model_mixture.restype = vp
    L.orc_model_mixture.argtypes = [_f64p, _f64p, _f64p, i32, i32, dbl]
    L.orc_run_pt.restype = i32
    L.orc_run_pt.argtypes = [vp, vp, i64, _f64p, _f64p, _u64p, i32, _f64p, i32, dbl, vp, vp, vp]
    L.orc_proposal_set_blocks.argtypes = [vp, i32, _i32p, _f64p]
    L.orc_run_adaptcov.restype = i32
    L.orc_run_adaptcov.argtypes = [vp, vp, i64, _f64p, _f64p, _u64p, i32, _f64p, _f64p, dbl, i32, i32,
                                   i32, vp, vp, vp, vp]
    L.orc_cprj_log_posterior.argtypes = [vp, i32, _f64p, i64, _f64p]
    L.orc_run_cprj.restype = i32
    L.orc_run_cprj.argtypes = [vp, i32, dbl, _f64p, i64, _f64p, _f64p, _u64p, i32, vp, vp, vp, vp]
    L.orc_max_threads.restype = i32
    _lib = L
    return L


def _c(a, dtype=np.float64):
    return np.ascontiguousarray(a, dtype=dtype)


# --------------------------------------------------------------------- RNG --
def rng_seed(seed, N, chain0=0):
    """(N, 16) uint64 state: 4 xoshiro256++ sub-streams per chain."""
    st = np.empty((N, 16), dtype=np.uint64)
    lib().orc_rng_seed(seed, chain0, N, st)
    return st


def rng_stream(state, d, T):
    """Advance ``state`` by T iterations; returns xi (T,N,d), u (T,N)."""
    N = state.shape[0]
    xi = np.empty((T, N, d))
    u = np.empty((T, N))
    lib().orc_rng_stream(state, N, d, T, xi, u)
    return xi, u


def rng_normals(state, d):
    """The next d normals of ONE chain (state (1,16), advanced in place)."""
    xi = np.empty(d)
    lib().orc_rng_normals(state, d, xi)
    return xi


def rng_words(state):
    """One more word from each of the four sub-streams of ONE chain, as U[0,1) doubles."""
    u = np.empty(4)
    lib().orc_rng_words(state, u)
    return u


def rng_pt_stream(state, d, T):
    """Advance ``state`` (S,16) by T parallel-tempering sweeps; returns the selector
    uniforms usel (T,S), swap uniforms vswap (T,S), normals xi (T,S,d), accept uniforms ua (T,S)."""
    S = state.shape[0]
    usel, vswp, ua = np.empty((T, S)), np.empty((T, S)), np.empty((T, S))
    xi = np.empty((T, S, d))
    lib().orc_rng_pt_stream(state, S, d, T, usel, vswp, xi, ua)
    return usel, vswp, xi, ua


def olog(x):
    return lib().orc_log(float(x))


def normal_from_word(w):
    """one normal from one 64-bit stream word (the 64-bit-resolution normal mode)"""
    return lib().orc_normal_from_word(int(w))


class normal_mode:
    """with oracle.normal_mode("f64"): every draw of normals inside uses ONE 64-bit word per normal
    (Sampler(..., normals="f64") of the product); "f32x2" is the default stream."""

    def __init__(self, mode):
        if mode not in ("f32x2", "f64"):
            raise ValueError("normal mode is 'f32x2' or 'f64'")
        self.mode = 1 if mode == "f64" else 0

    def __enter__(self):
        self.prev = lib().orc_get_normal_mode()
        lib().orc_set_normal_mode(self.mode)
        return self

    def __exit__(self, *exc):
        lib().orc_set_normal_mode(self.prev)
        return False


def normal_from_half(h):
    """The standard normal that the 32-bit half word ``h`` of a stream word maps to (rmh-rng v4)."""
    return lib().orc_normal_from_half(int(h))


# ------------------------------------------------------------------ models --
class _Model:
    def __init__(self, handle):
        self._h = handle
        self.d = lib().orc_model_dim(handle)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_model_free(self._h)
            self._h = None

    def _eval(self, fn, theta, grad=False):
        th = _c(np.atleast_2d(theta))
        N = th.shape[0]
        out = np.empty((N, self.d) if grad else (N,))
        getattr(lib(), fn)(self._h, th, N, out)
        return out

    def log_prior(self, theta):
        return self._eval("orc_log_prior", theta)

    def log_likelihood(self, theta):
        return self._eval("orc_log_likelihood", theta)

    def log_posterior(self, theta):
        return self._eval("orc_log_posterior", theta)

    def grad_log_posterior(self, theta):
        return self._eval("orc_grad_log_posterior", theta, grad=True)


class Gaussian(_Model):
    """MultiGaussianDist(mu, C): /root/reference/riemann/models/gaussian.py:21-58."""

    def __init__(self, mu, Cov):
        mu = _c(np.atleast_1d(mu))
        Lc = _c(np.linalg.cholesky(np.atleast_2d(Cov)))
        super().__init__(lib().orc_model_gaussian(mu, Lc, len(mu)))


class Mixture(_Model):
    """benchmarks.MixtureDist of MultiGaussianDist components:
    /root/reference/riemann/models/benchmarks.py:31-62.  components = [(weight, mu, C), ...]."""

    def __init__(self, components):
        w = np.array([c[0] for c in components], dtype=np.float64)
        w = w / np.sum(w)                                  # benchmarks.py:40
        mus = _c(np.array([np.atleast_1d(c[1]) for c in components], dtype=np.float64))
        Ls = _c(np.array([np.linalg.cholesky(np.atleast_2d(c[2])) for c in components]))
        logP0 = float(np.log(np.sum(w * np.exp(np.zeros(len(w))))))   # :54-62, component priors are 0
        self.weights = w
        super().__init__(lib().orc_model_mixture(_c(w), mus, Ls, len(w), mus.shape[1], logP0))


class LogReg(_Model):
    """Logistic regression with N(0, tau2 I) prior (new plugin)."""

    def __init__(self, X, y, tau2):
        X = _c(X)
        y = _c(y)
        super().__init__(lib().orc_model_logreg(X, y, X.shape[0], X.shape[1], float(tau2)))


class Changepoint(_Model):
    """ChangepointRegression1D, fixed k: /root/reference/riemann/models/changepoint.py:81-181.
    theta packs (cpx[k], cpv[k+1], sig)."""

    def __init__(self, x, y, k, xmin, xmax, lamb, alpha, beta):
        x = _c(x)
        y = _c(y)
        self.n = len(x)
        super().__init__(lib().orc_model_changepoint(
            x, y, len(x), k, float(xmin), float(xmax), float(lamb), float(alpha), float(beta)))

    def indices(self, theta):
        th = _c(np.atleast_2d(theta))
        idx = np.empty((th.shape[0], self.n), dtype=np.int32)
        lib().orc_changepoint_indices(self._h, th, th.shape[0], idx)
        return idx


# --------------------------------------------------------------- proposals --
class _Proposal:
    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_proposal_free(self._h)
            self._h = None

    def propose(self, model, theta, rng_state):
        th = _c(np.atleast_2d(theta))
        N, d = th.shape
        thp = np.empty_like(th)
        logq = np.empty(N)
        lib().orc_propose(self._h, model._h, th, N, rng_state, thp, logq)
        return thp, logq


class RandomWalk(_Proposal):
    """MetropolisRandomWalk(C): /root/reference/riemann/proposals/randomwalk.py:12-26."""

    def __init__(self, Cov, scale=1.0):
        Lc = _c(np.linalg.cholesky(np.atleast_2d(Cov)))
        self.d = Lc.shape[0]
        self._h = lib().orc_proposal_rw(Lc, self.d, float(scale))


class BlockRandomWalk(RandomWalk):
    """ChangepointRegression1DProp at fixed k (/root/reference/riemann/test_changepoint.py:18-56):
    blocks of consecutive coordinates, block b picked iff u_b < pcum[b] and none before it."""

    def __init__(self, Cov, sizes, pcum, scale=1.0):
        super().__init__(Cov, scale)
        lib().orc_proposal_set_blocks(self._h, len(sizes), _c(sizes, np.int32), _c(pcum))


class PCN(_Proposal):
    """pCN(C, rho): /root/reference/riemann/proposals/randomwalk.py:78-101."""

    def __init__(self, Cov, rho):
        Lc = _c(np.linalg.cholesky(np.atleast_2d(Cov)))
        self.d = Lc.shape[0]
        self._h = lib().orc_proposal_pcn(Lc, self.d, float(rho))


class HMC(_Proposal):
    """VanillaHMC(eps, Nsteps, grad, M=None): /root/reference/riemann/proposals/hamiltonian.py:55-91.
    Nsteps=1 is MALA."""

    def __init__(self, eps, nsteps, d, M=None):
        self.d = d
        self._h = lib().orc_proposal_hmc(float(eps), int(nsteps), int(d))
        if M is not None:   # hamiltonian.py:71-74
            chM = _c(np.linalg.cholesky(np.atleast_2d(M)))
            lib().orc_proposal_set_mass(self._h, chM.ctypes.data)


# ----------------------------------------------------------------- sampler --
def set_adapt(proposal, target):
    """Turn a RandomWalk/HMC into its AdaptScale* variant (proposals/adaptive.py:11-35)."""
    lib().orc_proposal_set_adapt(proposal._h, float(target))
    return proposal


def new_adapt_state(N):
    """(N,3) [Nsamples, Naccepts, scale] as AdaptScaleProposal.__init__ (adaptive.py:18-23)."""
    a = np.zeros((N, 3))
    a[:, 2] = 1.0
    return a


def run(model, proposal, theta, logpost, rng_state, T, history=True, nthreads=0, adapt=None, aux=None):
    """Sampler.run over N chains (/root/reference/riemann/samplers/sampler.py:44-90).

    ``theta`` (N,d), ``logpost`` (N,), ``rng_state`` (N,16) are advanced in
    place.  Returns (hist_theta (T+1,N,d), hist_logpost (T+1,N), accept (T,N))
    or (None, None, accept) with ``history=False``.
    """
    N, d = theta.shape
    assert theta.flags.c_contiguous and logpost.flags.c_contiguous
    ht = np.empty((T + 1, N, d)) if history else None
    hl = np.empty((T + 1, N)) if history else None
    acc = np.empty((T, N), dtype=np.uint8)
    rc = lib().orc_run_adapt(model._h, proposal._h, N, theta, logpost, rng_state, T,
                             ht.ctypes.data if history else None,
                             hl.ctypes.data if history else None,
                             acc.ctypes.data, nthreads,
                             adapt.ctypes.data if adapt is not None else None,
                             aux.ctypes.data if aux is not None else None)
    if rc != 0:
        raise ValueError("oracle: model and proposal dimensions differ")
    return ht, hl, acc


def new_adaptcov_state(C0):
    """[n, SX, SX2, C, L] of AdaptCovProposal.__init__ (proposals/adaptive.py:59-67)."""
    C0 = _c(np.atleast_2d(C0))
    d = C0.shape[0]
    st = np.zeros(2 + d + 3 * d * d)
    st[1 + d + d * d:1 + d + 2 * d * d] = C0.ravel()
    st[1 + d + 2 * d * d:1 + d + 3 * d * d] = np.linalg.cholesky(C0).ravel()
    return st


def adaptcov_CL(st, d):
    """The adapted (C, L) held in an adaptcov state vector."""
    o = 1 + d + d * d
    return st[o:o + d * d].reshape(d, d).copy(), st[o + d * d:o + 2 * d * d].reshape(d, d).copy()


def run_adaptcov(model, proposal, theta, logpost, rng_state, T, st, C0, t_adapt=1,
                 marginalize=False, smooth_adapt=False, history=True, exact_squares=False, adapt=None):
    """Sampler.run with AdaptCovRandomWalk pooled over the N chains
    (/root/reference/riemann/proposals/adaptive.py:38-103).  ``proposal`` is a RandomWalk
    (its scale is used; its L is replaced by the adapted one in ``st``)."""
    N, d = theta.shape
    ht = np.empty((T + 1, N, d)) if history else None
    hl = np.empty((T + 1, N)) if history else None
    acc = np.empty((T, N), dtype=np.uint8)
    rc = lib().orc_run_adaptcov(model._h, proposal._h, N, theta, logpost, rng_state, T, st,
                                _c(np.atleast_2d(C0)), float(t_adapt), int(marginalize),
                                int(smooth_adapt), int(exact_squares), ht.ctypes.data if history else None,
                                hl.ctypes.data if history else None, acc.ctypes.data,
                                adapt.ctypes.data if adapt is not None else None)
    if rc == -2:
        raise np.linalg.LinAlgError("adapted covariance is not positive definite")
    if rc != 0:
        raise ValueError("oracle: bad AdaptCov arguments")
    return ht, hl, acc


def pt_slots(n_ensembles, K):
    """Slot layout of parallel tempering (rmh_oracle.c section 5): returns (S, index)
    with index[e, i] the slot of rung i of ensemble e."""
    epw = 16 // K
    e = np.arange(n_ensembles)
    idx = ((e // epw) * 16 + (e % epw) * K)[:, None] + np.arange(K)[None, :]
    S = -(-n_ensembles // epw) * 16
    return S, idx


def run_pt(model, proposal, theta, rng_state, T, betas, pswap, history=True):
    """PTSampler.run over slots (/root/reference/riemann/samplers/ptsampler.py:84-127).
    theta (S,d), rng_state (S,16) advanced in place.  Returns (hist_theta (T+1,S,d),
    hist_logpost (T+1,S), code (T,S)), and the final logpost (S,)."""
    S, d = theta.shape
    betas = _c(betas)
    logpost = np.zeros(S)
    ht = np.zeros((T + 1, S, d)) if history else None
    hl = np.zeros((T + 1, S)) if history else None
    code = np.zeros((T, S), dtype=np.uint8)
    rc = lib().orc_run_pt(model._h, proposal._h, S, theta, logpost, rng_state, T, betas,
                          len(betas), float(pswap),
                          ht.ctypes.data if history else None,
                          hl.ctypes.data if history else None, code.ctypes.data)
    if rc != 0:
        raise ValueError("oracle: bad parallel-tempering arguments")
    return ht, hl, code, logpost


def cprj_pack(params, KM):
    """[(cpx, cpv, sig), ...] -> padded (N, 2 KM + 3) states of the trans-dimensional
    changepoint sampler (rmh_oracle.c section 6)."""
    out = np.zeros((len(params), 2 * KM + 3))
    out[:, :KM] = np.inf
    out[:, KM:2 * KM + 1] = 1.0
    for r, (cpx, cpv, sig) in zip(out, params):
        k = len(cpx)
        r[:k], r[KM:KM + k + 1], r[2 * KM + 1], r[2 * KM + 2] = cpx, cpv, sig, k
    return out


def cprj_unpack(row, KM):
    k = int(row[2 * KM + 2])
    return row[:k].copy(), row[KM:KM + k + 1].copy(), float(row[2 * KM + 1])


def cprj_log_posterior(model, KM, theta):
    th = _c(np.atleast_2d(theta))
    out = np.empty(th.shape[0])
    lib().orc_cprj_log_posterior(model._h, KM, th, th.shape[0], out)
    return out


def run_cprj(model, KM, hscale, pcum, theta, logpost, rng_state, T):
    """Trans-dimensional changepoint sampling over N chains (rmh_oracle.c section 6).
    Returns (hist_theta (T+1,N,2KM+3), hist_logpost (T+1,N), accept (T,N), move (T,N))."""
    N, D = theta.shape
    ht, hl = np.empty((T + 1, N, D)), np.empty((T + 1, N))
    acc, mv = np.empty((T, N), dtype=np.uint8), np.empty((T, N), dtype=np.uint8)
    lib().orc_run_cprj(model._h, KM, float(hscale), _c(pcum), N, theta, logpost, rng_state, T,
                       ht.ctypes.data, hl.ctypes.data, acc.ctypes.data, mv.ctypes.data)
    return ht, hl, acc, mv


def max_threads():
    return lib().orc_max_threads()
