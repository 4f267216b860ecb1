"""The MH sampler of /root/reference/riemann/samplers/sampler.py:28-90, for N
chains at once on one MI355X.

Same constructor, methods and attributes as the reference.  Differences, all
additive:
  * ``theta0`` may be ``(N, d)``: N independent chains (the reference runs one);
  * the random stream is explicit (``seed``, ``chain0``) instead of numpy's
    process-global generator (randomwalk.py:25, sampler.py:84);
  * with a device Model AND a device Proposal ``run`` is ONE fused kernel launch for all
    chains and all iterations; the history stays on the device until ``_chain_thetas`` /
    ``_chain_logpost`` are read.

Mixed plugins ("Batched extension"): the reference samples ANY ``Model`` /
``Proposal`` subclass (sampler.py:72-90).  When either plugin is the user's own Python class --
a numpy Model, a Proposal with its own ``propose``, a ``VanillaHMC`` built on a Python gradient
callable -- the Sampler runs sampler.py:72-90 itself, per iteration: the Proposal's ``propose``
(device proposals through ``rmh_propose`` on the chains' stream), the Model's ``log_posterior``
(device Models through ``rmh_log_posterior``), and the accept test with the stream's own uniform
(``rmh_rng_draw``), so the chain consumes the same stream as the fused path would.  That is the
plugin API doing what it says, not a CPU fallback: device plugins never take it.
"""
import ctypes as C

import numpy as np

from .. import _lib
from ..models.model import DeviceModel
from ..proposals.proposal import DeviceProposal


class Sampler(object):
    """Given a Model and a Proposal, sample the Model using the Proposal."""

    def __init__(self, model, proposal, theta0, seed=0, chain0=0, normals="f32x2"):
        """
        :param model: a device Model (riemann_amd.models)
        :param proposal: a device Proposal (riemann_amd.proposals)
        :param theta0: starting parameters, ``(d,)`` or ``(N, d)``
        :param seed: seed of the rmh-rng stream
        :param chain0: global id of the first chain (chains of a multi-GPU job
            keep their streams whatever the number of GPUs)
        :param normals: ``"f32x2"`` (default): two standard normals per 64-bit stream word, each the
            quantile of a 32-bit half word (|z| <= 6.4, a lattice of 2^-27 in p); ``"f64"``: one
            normal per word at float64 resolution (52 bits inside a cell, |z| <= 8.6, the map good
            to 2e-16) -- what ``np.random.normal`` gives the reference (randomwalk.py:25) -- on
            the generic kernels (not d > 112, the block-mixture and the variable-k changepoint
            proposals) and for a device Proposal next to a Python Model
        """
        if not hasattr(model, "log_posterior") or not hasattr(proposal, "propose"):
            raise TypeError("Sampler needs a Model (log_posterior) and a Proposal (propose)")
        self.model = model
        self.proposal = proposal
        self._seed, self._chain0 = int(seed), int(chain0)
        self.normals, nmode = normals, _lib.normal_mode(normals)
        self._state = self._hist = None
        self.pin_history, self._pool = True, None
        self.last_run_ms = None
        if not self._fused():
            # mixed plugins: a device Proposal (or the momenta of a host-gradient HMC) draws the chains' normals
            # through rmh_propose_mode / rmh_rng_draw_mode in this Sampler's mode
            if nmode and not isinstance(proposal, DeviceProposal):
                raise _lib.ParameterError("normals=\"f64\" concerns the device stream's normals: this proposal is a "
                                          "Python plugin and draws its own")
            if isinstance(proposal, DeviceProposal):
                proposal.normals = normals
            self._init_mixed(theta0)
            return
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
        if nmode:
            _lib.call("rmh_state_set_normal_mode", self._state, nmode)
        _lib.call("rmh_state_set_theta", self._state, th)   # sampler.py:42
        lp = np.empty(self.Nchains)
        _lib.call("rmh_state_get", self._state, None, lp.ctypes.data)
        self._hist = None
        self._host_thetas = th[None].copy()
        self._host_logpost = lp[None].copy()
        self._host_accept = np.zeros((0, self.Nchains), dtype=np.uint8)
        self._kept = 1

    # ------------------------------------------------------- mixed plugins --
    def _fused(self):
        """Both plugins device-backed (and the proposal not a host-gradient HMC): the fused path."""
        return (isinstance(self.model, DeviceModel) and isinstance(self.proposal, DeviceProposal)
                and getattr(self.proposal, "_host_grad", None) is None)

    def _init_mixed(self, theta0):
        from ..rng import rng_seed_host
        pack = getattr(self.model, "_pack_theta", None)
        th = _lib.as_f64(pack(theta0) if pack else theta0)
        self._single = th.ndim <= 1
        th = np.atleast_2d(np.atleast_1d(th)).copy()
        self.Nchains, self.Npars = th.shape
        self._rng = rng_seed_host(self._seed, self.Nchains, self._chain0)
        self._th, self._lp = th, self._model_lp(th)                      # sampler.py:42
        self._host_thetas, self._host_logpost = th[None].copy(), self._lp[None].copy()
        self._host_accept = np.zeros((0, self.Nchains), dtype=np.uint8)
        self._kept = 1

    def _model_lp(self, th):
        if isinstance(self.model, DeviceModel):
            return np.atleast_1d(np.asarray(self.model.log_posterior(th), dtype=np.float64))
        return np.array([float(self.model.log_posterior(row)) for row in th])

    def _step_mixed(self):
        """Sampler.sample (sampler.py:72-90) for all chains, plugins called as the reference calls them."""
        N = self.Nchains
        ada = None
        if isinstance(self.proposal, DeviceProposal):
            if self._mixed_adapt_hmc_host():
                # AdaptScaleHMC on a Python gradient: the trajectory and the chains' scales on the host
                ada = self._mx_adapt
                self.proposal._host_scale = ada[:, 2]
                thp, logq = self.proposal.propose(self._th, rng=self._rng)
                thp, logq = np.atleast_2d(thp), np.atleast_1d(logq)
            elif self._mixed_adapt_scale():
                # AdaptScaleRandomWalk next to a Python Model: the chains' scales live here; the device
                # forms L xi (the proposal at theta = 0, unscaled image) and theta' = theta + scale_c (L xi)
                # is formed as the fused kernel forms it (a product, then a sum)
                ada = self._mx_adapt
                lxi, logq = DeviceProposal.propose(self.proposal, np.zeros_like(self._th), rng=self._rng)
                thp = self._th + ada[:, 2:3] * np.atleast_2d(lxi)
                logq = np.atleast_1d(logq)
            elif self._mixed_adapt_cov():
                # AdaptCovRandomWalk next to a Python Model: theta + L xi on the device with the current factor
                # (as the fused kernel forms it); moments, covariance and factor on the host (below)
                thp, logq = DeviceProposal.propose(self.proposal, self._th, rng=self._rng)
                thp, logq = np.atleast_2d(thp), np.atleast_1d(logq)
            elif hasattr(self.proposal, "_pull_adapt") or hasattr(self.proposal, "_init_adapt_cov"):
                raise TypeError("this adaptive device proposal needs a device Model (next to a Python Model "
                                "AdaptScaleRandomWalk, AdaptCovRandomWalk, AdaptScaleCovRandomWalk and "
                                "AdaptScaleHMC on a Python gradient are built)")
            else:
                thp, logq = self.proposal.propose(self._th, rng=self._rng)        # :79
                thp, logq = np.atleast_2d(thp), np.atleast_1d(logq)
        else:
            out = [self.proposal.propose(row) for row in self._th]
            thp = np.stack([np.asarray(o[0], dtype=np.float64) for o in out])
            logq = np.array([float(o[1]) for o in out])
        lpp = self._model_lp(thp)                                                  # :82
        u = np.empty((1, N))
        _lib.call("rmh_rng_draw", self._rng, N, 0, 1, None, u.ctypes.data)        # the stream's uniform, :84
        with np.errstate(divide="ignore", invalid="ignore"):
            delta = lpp - self._lp - logq
            mh = np.where(delta < 0.0, delta, 0.0)                                 # Python min(0, D): NaN accepts
            ok = np.log(u[0]) < mh
        self._th = np.where(ok[:, None], thp, self._th)
        self._lp = np.where(ok, lpp, self._lp)
        if not isinstance(self.proposal, DeviceProposal):
            for row in self._th:
                self.proposal.adapt(row)                                           # :88
        elif ada is not None:   # AdaptScaleProposal.adapt per chain (adaptive.py:25-35)
            from ..proposals.adaptive import adapt_scale_step
            adapt_scale_step(ada, ok, self.proposal.target_accept_rate)
            self._publish_mixed_adapt()
            if self._mixed_adapt_cov():          # AdaptScaleCovRandomWalk: scale per chain AND pooled covariance
                self._adapt_cov_mixed()
        elif self._mixed_adapt_cov():
            self._adapt_cov_mixed()
        return ok.astype(np.uint8)

    def _mixed_adapt_cov(self):
        """AdaptCovRandomWalk (Haario, pooled over the chains) can run next to a Python Model."""
        from ..proposals.randomwalk import AdaptCovRandomWalk, AdaptScaleCovRandomWalk
        if not isinstance(self.proposal, (AdaptCovRandomWalk, AdaptScaleCovRandomWalk)):
            return False
        if getattr(self, "_mx_cov", None) is None:
            d = self.Npars
            self._mx_cov = {"n": 0.0, "S": 0.0, "SX": np.zeros(d), "SX2": np.zeros((d, d))}
        return True

    def _adapt_cov_mixed(self):
        """AdaptCovProposal.adapt for every chain (adaptive.py:73-102), pooled as the device path pools it
        (rmh_run_api.hip, acov_update): sums over all chains, update when the per-chain count passes the
        reference's test, C and L pushed to the device proposal."""
        from ..proposals.adaptive import adapt_cov_due, adapt_cov_update
        p, st, th = self.proposal, self._mx_cov, self._th
        st["n"] += 1.0
        st["S"] += float(th.shape[0])
        st["SX"] += th.sum(axis=0)
        st["SX2"] += th.T @ th
        n, nt = st["n"], st["S"]
        if not adapt_cov_due(n, p.exact_squares):                                    # adaptive.py:81-82
            return
        try:
            Cn, L = adapt_cov_update(nt, st["SX"], st["SX2"], p.C0, p.t_adapt, p.smooth_adapt, p.marginalize)
        except np.linalg.LinAlgError:
            raise RuntimeError("AdaptCov: the adapted covariance is not positive definite")
        p.C, p.L, p._S, p._SX, p._SX2 = Cn, L, n, st["SX"].copy(), st["SX2"].copy()
        pack = np.concatenate([[n], st["SX"], st["SX2"].ravel(), Cn.ravel(), L.ravel(), [0.0]])
        _lib.call("rmh_proposal_adapt_cov_set", p._h, _lib.as_f64(pack), float(nt))

    def _mixed_adapt_hmc_host(self):
        """AdaptScaleHMC on a Python gradient callable (hamiltonian.py:94-104 with a user's model)."""
        from ..proposals.adaptive import AdaptScaleProposal
        p = self.proposal
        if not (isinstance(p, AdaptScaleProposal) and getattr(p, "_host_grad", None) is not None):
            return False
        if getattr(self, "_mx_adapt", None) is None:
            a = np.zeros((self.Nchains, 3))
            a[:, 2] = 1.0
            self._mx_adapt = a
        return True

    def _mixed_adapt_scale(self):
        """AdaptScaleRandomWalk can run next to a Python Model (the other adaptive device proposals cannot)."""
        from ..proposals.randomwalk import AdaptScaleCovRandomWalk, AdaptScaleRandomWalk
        if not isinstance(self.proposal, (AdaptScaleRandomWalk, AdaptScaleCovRandomWalk)):
            return False
        if getattr(self, "_mx_adapt", None) is None:
            a = np.zeros((self.Nchains, 3))
            a[:, 2] = 1.0                                                          # adaptive.py:18-23
            self._mx_adapt = a
        return True

    def _publish_mixed_adapt(self):
        a, p = self._mx_adapt, self.proposal
        ns, na = a[:, 0].astype(np.int64), a[:, 1].astype(np.int64)
        rate = np.where(ns > 0, a[:, 1] / np.maximum(a[:, 0], 1.0), 0.0)
        if self._single:
            p.Nsamples, p.Naccepts, p.scale, p.accept_rate = int(ns[0]), int(na[0]), float(a[0, 2]), float(rate[0])
        else:
            p.Nsamples, p.Naccepts, p.scale, p.accept_rate = ns, na, a[:, 2].copy(), rate

    def _run_mixed(self, T, nb, nt, record):
        ths, lps, acc = [self._th.copy()], [self._lp.copy()], []
        for _ in range(T):
            acc.append(self._step_mixed())
            if record:
                ths.append(self._th.copy())
                lps.append(self._lp.copy())
        kept = (T - nb) // nt + 1 if T >= nb else 0
        self._kept, self._iters = (kept if record else 0), T
        N, d = self.Nchains, self.Npars
        self._host_accept = np.array(acc, dtype=np.uint8).reshape(T, N)
        if record:
            self._host_thetas = np.array(ths)[nb::nt].reshape(kept, N, d)
            self._host_logpost = np.array(lps)[nb::nt].reshape(kept, N)
            if kept:   # the current state is the LAST KEPT one (sampler.py:49-50, 53-54)
                self._th, self._lp = self._host_thetas[-1].copy(), self._host_logpost[-1].copy()
        else:
            self._host_thetas, self._host_logpost = np.empty((0, N, d)), np.empty((0, N))
        if kept == 0:
            self._th = None

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
            # histories of 64 MB and more land in pinned memory (DMA at the PCIe rate, transposes
            # overlapped): _lib.PinnedPool; smaller ones are not worth the page-locking
            if n * N * d * 8 >= (64 << 20) and self.pin_history:
                if self._pool is None:
                    self._pool = _lib.PinnedPool()
                th = self._pool.empty((n, N, d))
            else:
                th = np.empty((n, N, d))
            lp = np.empty((n, N))
            acc = np.empty((self._iters, N), dtype=np.uint8)
            if n:
                _lib.call("rmh_history_download", self._hist, 0, n, th.ctypes.data, lp.ctypes.data)
            if self._iters:
                _lib.call("rmh_history_accept", self._hist, 0, self._iters, acc)
            self._host_thetas, self._host_logpost, self._host_accept = th, lp, acc

    def _view(self, a):
        return a[:, 0] if self._single else a

    # sample() and _add_state() GROW the history one state at a time (sampler.py:69-70, :89; PTSampler
    # calls them once per sweep): earlier pieces are parked in a list and joined on the next read,
    # so n calls cost O(n), not O(n^2) array copies.
    def _park(self):
        self._materialise()
        if not hasattr(self, "_arch"):
            self._arch = []
        self._arch.append((self._host_thetas, self._host_logpost, self._host_accept))

    def _join(self):
        self._materialise()
        arch = getattr(self, "_arch", None)
        if arch:
            arch.append((self._host_thetas, self._host_logpost, self._host_accept))
            self._host_thetas = np.concatenate([a[0] for a in arch])
            self._host_logpost = np.concatenate([a[1] for a in arch])
            self._host_accept = np.concatenate([a[2] for a in arch])
            self._arch = []

    @property
    def _chain_thetas(self):
        """History of states, ``(n, N, d)`` (``(n, d)`` for a single chain):
        the list of sampler.py:41,69 as one array."""
        self._join()
        return self._view(self._host_thetas)

    @property
    def _chain_logpost(self):
        self._join()
        return self._view(self._host_logpost)

    @property
    def _chain_accept(self):
        """uint8 accept flags of the iterations of the last run(), ``(T, N)``."""
        self._join()
        return self._view(self._host_accept)

    # ----------------------------------------------------------------- API --
    def run(self, Nsamples, Nburn=0, Nthin=1, record=True):
        """Run the Sampler (sampler.py:44-54): keep the last state, take
        ``Nsamples`` MH steps, keep ``[Nburn::Nthin]`` of the Nsamples+1 states.

        ``record=False`` skips the history (throughput runs); the chain state
        still advances and ``current_state()`` stays valid."""
        self._arch = []          # run() restarts the history from the current state (sampler.py:49-50)
        return self._run(Nsamples, Nburn, Nthin, record)

    def _run(self, Nsamples, Nburn=0, Nthin=1, record=True):
        T, nb, nt = int(Nsamples), int(Nburn), int(Nthin)
        if T < 0 or nb < 0 or nt < 1:
            raise _lib.ParameterError("need Nsamples >= 0, Nburn >= 0, Nthin >= 1")
        if self._state is None:
            if self._th is None:
                raise IndexError("run: no current state (the previous run kept nothing)")
            if not record and T >= nb and nb + ((T - nb) // nt) * nt != T:
                # the state to end in is the last KEPT one: it has to be seen
                return self._run_mixed_unrecorded(T, nb, nt)
            return self._run_mixed(T, nb, nt, record)
        self.proposal._sync()
        kept = (T - nb) // nt + 1 if T >= nb else 0
        hist = None
        # a device history of the same shape is written again rather than freed and re-allocated
        # (chunked runs, run_to_disk: freeing / mapping 10 GB per call halves the download rate that
        # follows, tools/dl_probe*.py)
        if not (record and self._hist and getattr(self, "_hist_shape", None) == (kept, T)):
            self._drop_history()
        if record:
            if not self._hist:
                self._hist = C.c_void_p()
                _lib.call("rmh_history_create", self._state, kept, T, C.byref(self._hist))
                self._hist_shape = (kept, T)
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

    def history_pieces(self, nlags=128):
        """The reductions of the last recorded run() formed ON THE DEVICE (the history itself stays
        there), in the form ``diagnostics.pooled_history_stats`` combines over ranks: ``n``,
        ``mean`` / ``var`` (d, N) per chain and parameter, ``acf_sums`` (nlags, d) int64 integer
        autocorrelation sums (rmh_history_reduce_sums), and the same moments of the two half
        chains (``mean_h``, ``var_h`` (2, d, N), ``h``) for the split R-hat.  ``nlags``: 32, 64, ... 320."""
        if self._state is None or not self._hist or self._kept < 2:
            raise IndexError("history_stats: no device history (run() with at least two kept states first)")
        N, d, n = self.Nchains, self.Npars, self._kept
        mean, var, sums = np.empty((d, N)), np.empty((d, N)), np.empty((nlags, d), dtype=np.int64)
        _lib.call("rmh_history_reduce_sums", self._hist, 0, n, int(nlags), mean.ctypes.data, var.ctypes.data,
                  sums.ctypes.data)
        piece = {"n": n, "mean": mean, "var": var, "acf_sums": sums}
        h = n // 2
        if h >= 2:
            mh, vh = np.empty((2, d, N)), np.empty((2, d, N))
            for k in range(2):
                _lib.call("rmh_history_reduce_sums", self._hist, k * h, (k + 1) * h, 32, mh[k].ctypes.data,
                          vh[k].ctypes.data, None)
            piece.update(h=h, mean_h=mh, var_h=vh)
        return piece

    def history_stats(self, nlags=128, c=5.0):
        """Diagnostics of the last recorded run() reduced ON THE DEVICE (the history itself stays
        there): ``mean``, ``var`` (N, d) per chain and parameter, ``acf`` (nlags, d) the
        chain-averaged autocorrelation functions, ``tau`` (d,) the integrated autocorrelation
        times (Sokal window ``c``), ``closed`` whether the window closed within ``nlags`` lags,
        ``ess`` = N * n / tau, and ``rhat`` (d,) the split R-hat over the 2N half-chains.
        ``nlags``: 32, 64, ... 320.  The one-rank case of ``distributed.pool_history`` (same code:
        a sharded job reports the same numbers bit for bit); ``pooled_mean`` / ``pooled_var`` are
        the moments over all chains and kept states."""
        from .. import diagnostics as dg
        piece = self.history_pieces(nlags)
        st = dg.pooled_history_stats(piece, c=c)
        out = {"mean": piece["mean"].T, "var": piece["var"].T, "acf": st["acf"], "tau": st["tau"],
               "closed": st["closed"], "ess": st["ess"], "pooled_mean": st["mean"], "pooled_var": st["var"]}
        if "rhat" in st:
            out["rhat"] = st["rhat"]
        return out

    def history_moments(self):
        """``(count, mean (d,), covariance (d, d))`` over all chains and kept states of the last
        recorded run(), from one pass of the pooled moment kernel over the device history
        (rmh_history_moments; d <= 111)."""
        if self._state is None or not self._hist or self._kept < 1:
            raise IndexError("history_moments: no device history")
        d = self.Npars
        mom = np.empty(1 + d + d * d)
        _lib.call("rmh_history_moments", self._hist, 0, self._kept, mom)
        mean = mom[1:1 + d] / mom[0]
        return mom[0], mean, mom[1 + d:].reshape(d, d) / mom[0] - np.outer(mean, mean)

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
            last = _lib.as_f64(np.atleast_2d(np.asarray(fth[kept - 1]))).copy()
            last_lp = _lib.as_f64(np.atleast_1d(np.asarray(flp[kept - 1]))).copy()
            if self._state is None:      # mixed plugins: the chains live on the host
                self._th, self._lp = last.copy(), last_lp.copy()
            else:
                _lib.call("rmh_state_set_theta", self._state, last)
                self._drop_history()
            self._host_thetas = last[None].copy()
            self._host_logpost = last_lp[None].copy()
            self._host_accept = np.zeros((0, self.Nchains), dtype=np.uint8)
            self._kept = 1
        return kept

    # ------------------------------------------------------ checkpoint / resume --
    def checkpoint(self, path):
        """Write the chains' resumable state to ``path`` (.npz): current states,
        random-stream states and, for adaptive proposals, the adaptation state.
        The reference only resumes in-process (run() continues from the last state,
        sampler.py:49-50)."""
        if self._state is None:      # mixed plugins: the chains live on the host
            extra = {"adapt": self._mx_adapt} if getattr(self, "_mx_adapt", None) is not None else {}
            if getattr(self, "_mx_cov", None) is not None:
                c = self._mx_cov
                extra.update(cov_n=c["n"], cov_S=c["S"], cov_SX=c["SX"], cov_SX2=c["SX2"],
                             cov_C=self.proposal.C, cov_L=self.proposal.L)
            np.savez(path, theta=self._th, logpost=self._lp, rng=self._rng, single=np.bool_(self._single), **extra)
            return
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
        np.savez(path, theta=th, logpost=lp, rng=rng, adapt=ad, single=np.bool_(self._single),
                 normals=np.str_(self.normals), **extra)

    def restore(self, path):
        """Load a checkpoint written by :meth:`checkpoint` into this Sampler (same
        model, proposal and number of chains); the next run() continues the same
        chains bit for bit."""
        z = np.load(path)
        if z["theta"].shape != (self.Nchains, self.Npars):
            raise _lib.ParameterError("checkpoint holds states of shape {}, sampler has {}"
                                      .format(z["theta"].shape, (self.Nchains, self.Npars)))
        saved = str(z["normals"]) if "normals" in z.files else "f32x2"
        if saved != self.normals:
            raise _lib.ParameterError("checkpoint was written with normals={!r}, this sampler draws {!r}: the "
                                      "chains would not continue their stream".format(saved, self.normals))
        self._arch = []
        if self._state is None:
            self._th, self._lp = _lib.as_f64(z["theta"]).copy(), _lib.as_f64(z["logpost"]).copy()
            self._rng = np.ascontiguousarray(z["rng"], dtype=np.uint64)
            if "adapt" in z.files:
                self._mx_adapt = _lib.as_f64(z["adapt"]).copy()
                self._publish_mixed_adapt()
            if "cov_n" in z.files and self._mixed_adapt_cov():
                c, pr = self._mx_cov, self.proposal
                c["n"], c["S"] = float(z["cov_n"]), float(z["cov_S"])
                c["SX"], c["SX2"] = _lib.as_f64(z["cov_SX"]).copy(), _lib.as_f64(z["cov_SX2"]).copy()
                pr.C, pr.L = _lib.as_f64(z["cov_C"]).copy(), _lib.as_f64(z["cov_L"]).copy()
                pack = np.concatenate([[c["n"]], c["SX"], c["SX2"].ravel(), pr.C.ravel(), pr.L.ravel(), [0.0]])
                _lib.call("rmh_proposal_adapt_cov_set", pr._h, _lib.as_f64(pack), c["S"])
            self._host_thetas, self._host_logpost = self._th[None].copy(), self._lp[None].copy()
            self._host_accept = np.zeros((0, self.Nchains), dtype=np.uint8)
            self._kept = 1
            return
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

    def _run_mixed_unrecorded(self, T, nb, nt):
        self._run_mixed(T, nb, nt, True)
        N, d = self.Nchains, self.Npars
        self._kept = 0
        self._host_thetas, self._host_logpost = np.empty((0, N, d)), np.empty((0, N))

    def current_state(self):
        """(theta, log_posterior) of the current state (sampler.py:56-60)."""
        if self._state is None:
            if self._th is None:
                raise IndexError("state: no current state (empty history)")
            return (self._th[0].copy(), float(self._lp[0])) if self._single else (self._th.copy(), self._lp.copy())
        th, lp = np.empty((self.Nchains, self.Npars)), np.empty(self.Nchains)
        _lib.call("rmh_state_get", self._state, th.ctypes.data, lp.ctypes.data)
        if self._single:
            return th[0], float(lp[0])
        return th, lp

    def _add_state(self, theta, logpost):
        """Append an explicitly given state (sampler.py:62-70; hook for
        meta-samplers).  The state becomes the chain's current state."""
        self._park()
        pack = getattr(self.model, "_pack_theta", None)
        th = np.atleast_2d(_lib.as_f64(pack(theta) if pack else theta))
        lp = np.atleast_1d(np.asarray(logpost, dtype=np.float64))
        self._host_thetas, self._host_logpost = th[None].copy(), lp[None].copy()
        self._host_accept = np.zeros((0, self.Nchains), dtype=np.uint8)
        self._kept = 1
        if self._state is None:
            self._th, self._lp = th.copy(), lp.copy()
            return
        _lib.call("rmh_state_set_theta", self._state, th)
        self._drop_history()

    def sample(self):
        """One MH step appended to the history (sampler.py:72-90)."""
        self._park()
        self._run(1)
        self._materialise()
        self._host_thetas, self._host_logpost = self._host_thetas[1:], self._host_logpost[1:]
        self._drop_history()
        return self._view(self._host_thetas)[-1], self._view(self._host_logpost)[-1]
