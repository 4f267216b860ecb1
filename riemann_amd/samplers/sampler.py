"""The MH sampler of /root/reference/riemann/samplers/sampler.py:28-90, for N
chains at once on one MI355X.

Same constructor, methods and attributes as the reference.  Differences, all
additive:
  * ``theta0`` may be ``(N, d)``: N independent chains (the reference runs one);
  * the random stream is explicit (``seed``, ``chain0``) instead of numpy's
    process-global generator (randomwalk.py:25, sampler.py:84);
  * ``run`` is ONE fused kernel launch for all chains and all iterations; the
    history stays on the device until ``_chain_thetas`` / ``_chain_logpost``
    are read.
"""
import ctypes as C

import numpy as np

from .. import _lib
from ..models.model import DeviceModel
from ..proposals.proposal import DeviceProposal


class Sampler(object):
    """Given a Model and a Proposal, sample the Model using the Proposal."""

    def __init__(self, model, proposal, theta0, seed=0, chain0=0):
        """
        :param model: a device Model (riemann_amd.models)
        :param proposal: a device Proposal (riemann_amd.proposals)
        :param theta0: starting parameters, ``(d,)`` or ``(N, d)``
        :param seed: seed of the rmh-rng stream
        :param chain0: global id of the first chain (chains of a multi-GPU job
            keep their streams whatever the number of GPUs)
        """
        if not isinstance(model, DeviceModel) or not isinstance(proposal, DeviceProposal):
            raise TypeError("riemann_amd.Sampler runs on the GPU: model and proposal must be "
                            "device plugins (riemann_amd.models / riemann_amd.proposals); "
                            "there is no CPU sampling path in this package")
        self.model = model
        self.proposal = proposal
        th = model._pack_theta(theta0)
        th = _lib.as_f64(th)
        self._single = th.ndim <= 1
        th = np.atleast_2d(np.atleast_1d(th))
        if th.shape[1] != model.Npars:
            raise _lib.ParameterError("theta0 has shape {}, model has {} parameters"
                                      .format(np.shape(theta0), model.Npars))
        self.Nchains, self.Npars = th.shape
        self._state = C.c_void_p()
        _lib.call("rmh_state_create", model._h, self.Nchains, int(seed), int(chain0),
                  C.byref(self._state))
        _lib.call("rmh_state_set_theta", self._state, th)   # sampler.py:42
        lp = np.empty(self.Nchains)
        _lib.call("rmh_state_get", self._state, None, lp.ctypes.data)
        self._hist = None
        self._host_thetas = th[None].copy()
        self._host_logpost = lp[None].copy()
        self._host_accept = np.zeros((0, self.Nchains), dtype=np.uint8)
        self._kept = 1
        self.last_run_ms = None

    def __del__(self):
        for name, free in (("_hist", "rmh_history_free"), ("_state", "rmh_state_free")):
            h = getattr(self, name, None)
            setattr(self, name, None)
            if h:
                try:
                    getattr(_lib.load(), free)(h)
                except Exception:
                    pass

    # ------------------------------------------------------------- history --
    def _drop_history(self):
        if self._hist:
            _lib.call("rmh_history_free", self._hist)
        self._hist = None

    def _materialise(self):
        """Download the device history of the last run() on first use."""
        if self._hist is not None and self._host_thetas is None:
            n, N, d = self._kept, self.Nchains, self.Npars
            th, lp = np.empty((n, N, d)), np.empty((n, N))
            acc = np.empty((self._iters, N), dtype=np.uint8)
            if n:
                _lib.call("rmh_history_download", self._hist, 0, n, th.ctypes.data, lp.ctypes.data)
            if self._iters:
                _lib.call("rmh_history_accept", self._hist, 0, self._iters, acc)
            self._host_thetas, self._host_logpost, self._host_accept = th, lp, acc

    def _view(self, a):
        return a[:, 0] if self._single else a

    @property
    def _chain_thetas(self):
        """History of states, ``(n, N, d)`` (``(n, d)`` for a single chain):
        the list of sampler.py:41,69 as one array."""
        self._materialise()
        return self._view(self._host_thetas)

    @property
    def _chain_logpost(self):
        self._materialise()
        return self._view(self._host_logpost)

    @property
    def _chain_accept(self):
        """uint8 accept flags of the iterations of the last run(), ``(T, N)``."""
        self._materialise()
        return self._view(self._host_accept)

    # ----------------------------------------------------------------- API --
    def run(self, Nsamples, Nburn=0, Nthin=1, record=True):
        """Run the Sampler (sampler.py:44-54): keep the last state, take
        ``Nsamples`` MH steps, keep ``[Nburn::Nthin]`` of the Nsamples+1 states.

        ``record=False`` skips the history (throughput runs); the chain state
        still advances and ``current_state()`` stays valid."""
        T, nb, nt = int(Nsamples), int(Nburn), int(Nthin)
        if T < 0 or nb < 0 or nt < 1:
            raise _lib.ParameterError("need Nsamples >= 0, Nburn >= 0, Nthin >= 1")
        self.proposal._sync()
        self._drop_history()
        kept = (T - nb) // nt + 1 if T >= nb else 0
        hist = None
        if record:
            self._hist = C.c_void_p()
            _lib.call("rmh_history_create", self._state, kept, T, C.byref(self._hist))
            hist = self._hist
        _lib.call("rmh_run", self.model._h, self.proposal._h, self._state, T, nb, nt, hist)
        _lib.call("rmh_synchronize")
        ms = C.c_double()
        _lib.call("rmh_last_run_ms", C.byref(ms), None)
        self.last_run_ms = ms.value
        self._kept, self._iters = kept, T
        if hasattr(self.proposal, "_pull_adapt"):   # AdaptScale*: publish the adapted scales
            self.proposal._pull_adapt({"handle": self._state, "N": self.Nchains}, self._single)
        if record:
            self._host_thetas = self._host_logpost = self._host_accept = None
        else:
            self._kept = 0
            self._host_thetas = np.empty((0, self.Nchains, self.Npars))
            self._host_logpost = np.empty((0, self.Nchains))
            self._host_accept = np.empty((0, self.Nchains), dtype=np.uint8)

    def run_to_disk(self, prefix, Nsamples, Nburn=0, Nthin=1, chunk=1000):
        """``run(Nsamples, Nburn, Nthin)`` with the kept states streamed to disk instead of
        held in memory: ``prefix + "_thetas.npy"`` (n, N, d),
        ``"_logpost.npy"`` (n, N) and ``"_accept.npy"`` (Nsamples, N) are standard .npy files,
        written chunk by chunk through memory maps (``numpy.load(..., mmap_mode="r")`` reads
        them back lazily).  The iterations are launched ``chunk`` at a time; the chains and
        their random streams are the same as for one ``run`` call.  Returns the number of kept
        states.  The in-memory history afterwards holds only the last kept state."""
        T, nb, nt, chunk = int(Nsamples), int(Nburn), int(Nthin), int(chunk)
        if T < 0 or nb < 0 or nt < 1 or chunk < 1:
            raise _lib.ParameterError("need Nsamples >= 0, Nburn >= 0, Nthin >= 1, chunk >= 1")
        kept = (T - nb) // nt + 1 if T >= nb else 0
        N, d = self.Nchains, self.Npars
        shp = (kept, d) if self._single else (kept, N, d)
        fth = np.lib.format.open_memmap(prefix + "_thetas.npy", mode="w+", dtype=np.float64, shape=shp)
        flp = np.lib.format.open_memmap(prefix + "_logpost.npy", mode="w+", dtype=np.float64, shape=shp[:-1])
        fac = np.lib.format.open_memmap(prefix + "_accept.npy", mode="w+", dtype=np.uint8,
                                        shape=(T,) if self._single else (T, N))
        done, out = 0, 0
        while True:
            step = min(chunk, T - done)
            # global state numbers done .. done+step; kept iff e >= nb and (e - nb) % nt == 0.
            # Within the chunk the local state 0 is the global state `done` (kept by the previous
            # chunk unless this is the first one).
            first = nb if done == 0 and nb <= 0 else max(nb, done + (1 if done else 0))
            first += (-(first - nb)) % nt                     # next kept global state number
            self.run(step, 0, 1)                              # every state of the chunk, then slice
            th, lp, ac = self._chain_thetas, self._chain_logpost, self._chain_accept
            loc = np.arange(first - done, step + 1, nt) if first - done <= step else np.arange(0)
            n = len(loc)
            if n:
                fth[out:out + n], flp[out:out + n] = th[loc], lp[loc]
                out += n
            fac[done:done + step] = ac
            done += step
            if done >= T:
                break
        for f in (fth, flp, fac):
            f.flush()
        assert out == kept, (out, kept)
        # as after run(): the current state is the LAST KEPT one (sampler.py:49-50, 53-54)
        if kept == 0:
            self.run(0, 1, 1)                                   # keeps nothing: no current state
        elif nb + (kept - 1) * nt != T:
            last = np.asarray(fth[kept - 1])
            _lib.call("rmh_state_set_theta", self._state, _lib.as_f64(np.atleast_2d(last)))
            self._host_thetas = np.atleast_2d(last)[None].copy()
            self._host_logpost = np.atleast_1d(np.asarray(flp[kept - 1]))[None].copy()
            self._host_accept = np.zeros((0, self.Nchains), dtype=np.uint8)
            self._drop_history()
            self._kept = 1
        return kept

    # ------------------------------------------------------ checkpoint / resume --
    def checkpoint(self, path):
        """Write the chains' resumable state to ``path`` (.npz): current states,
        random-stream states and, for adaptive proposals, the adaptation state.
        The reference only resumes in-process (run() continues from the last state,
        sampler.py:49-50)."""
        th, lp = np.empty((self.Nchains, self.Npars)), np.empty(self.Nchains)
        _lib.call("rmh_state_get", self._state, th.ctypes.data, lp.ctypes.data)
        rng = np.empty((self.Nchains, 16), dtype=np.uint64)
        _lib.call("rmh_state_get_rng", self._state, rng)
        ad = np.empty((self.Nchains, 3))
        _lib.call("rmh_state_get_adapt", self._state, ad)
        extra = {}
        aux = np.empty(self.Nchains)
        _lib.call("rmh_state_get_adapt_aux", self._state, aux)
        extra["adapt_aux"] = aux
        if hasattr(self.proposal, "_adapt_cov_state"):      # AdaptCov*: the pooled accumulators
            extra["adapt_cov"] = self.proposal._adapt_cov_state()
        np.savez(path, theta=th, logpost=lp, rng=rng, adapt=ad, single=np.bool_(self._single), **extra)

    def restore(self, path):
        """Load a checkpoint written by :meth:`checkpoint` into this Sampler (same
        model, proposal and number of chains); the next run() continues the same
        chains bit for bit."""
        z = np.load(path)
        if z["theta"].shape != (self.Nchains, self.Npars):
            raise _lib.ParameterError("checkpoint holds states of shape {}, sampler has {}"
                                      .format(z["theta"].shape, (self.Nchains, self.Npars)))
        _lib.call("rmh_state_set_theta", self._state, _lib.as_f64(z["theta"]))
        _lib.call("rmh_state_set_rng", self._state, np.ascontiguousarray(z["rng"], dtype=np.uint64))
        _lib.call("rmh_state_set_adapt", self._state, _lib.as_f64(z["adapt"]))
        if "adapt_aux" in z:
            _lib.call("rmh_state_set_adapt_aux", self._state, _lib.as_f64(z["adapt_aux"]))
        if "adapt_cov" in z and hasattr(self.proposal, "_adapt_cov_state"):
            st = _lib.as_f64(z["adapt_cov"])
            _lib.call("rmh_proposal_adapt_cov_set", self.proposal._h, st, float(st[0]) * self.Nchains)
            self.proposal._pull_adapt(None, self._single)
        self._drop_history()
        lp = np.empty(self.Nchains)
        _lib.call("rmh_state_get", self._state, None, lp.ctypes.data)
        self._host_thetas = _lib.as_f64(z["theta"])[None].copy()
        self._host_logpost = lp[None].copy()
        self._host_accept = np.zeros((0, self.Nchains), dtype=np.uint8)
        self._kept = 1

    def current_state(self):
        """(theta, log_posterior) of the current state (sampler.py:56-60)."""
        th, lp = np.empty((self.Nchains, self.Npars)), np.empty(self.Nchains)
        _lib.call("rmh_state_get", self._state, th.ctypes.data, lp.ctypes.data)
        if self._single:
            return th[0], float(lp[0])
        return th, lp

    def _add_state(self, theta, logpost):
        """Append an explicitly given state (sampler.py:62-70; hook for
        meta-samplers).  The state becomes the chain's current state."""
        self._materialise()
        th = np.atleast_2d(_lib.as_f64(self.model._pack_theta(theta)))
        _lib.call("rmh_state_set_theta", self._state, th)
        lp = np.atleast_1d(np.asarray(logpost, dtype=np.float64))
        self._host_thetas = np.concatenate([self._host_thetas, th[None]])
        self._host_logpost = np.concatenate([self._host_logpost, lp[None]])
        self._drop_history()

    def sample(self):
        """One MH step appended to the history (sampler.py:72-90)."""
        self._materialise()
        th_hist, lp_hist, acc_hist = self._host_thetas, self._host_logpost, self._host_accept
        self.run(1)
        self._materialise()
        self._host_thetas = np.concatenate([th_hist, self._host_thetas[1:]])
        self._host_logpost = np.concatenate([lp_hist, self._host_logpost[1:]])
        self._host_accept = np.concatenate([acc_hist, self._host_accept])
        self._drop_history()
        return self._view(self._host_thetas)[-1], self._view(self._host_logpost)[-1]
