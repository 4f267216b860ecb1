"""Parallel tempering on the device: /root/reference/riemann/samplers/ptsampler.py:11-127.

``PTSampler(model, proposal, theta0, betas=None, Pswap=0.1)`` as in the reference, for
many ensembles at once: ``theta0`` may be ``(Ne, d)`` (Ne independent ladders; the
reference runs one).  A ladder of K temperatures lives in K neighbouring chain slots
of one wavefront, so a swap move exchanges registers between lanes; the whole of
``run(Nsamples)`` is one kernel launch (include/riemann_hip.h, rmh_run_pt).

Differences from the reference, all in the direction of what it evidently means:
  * ``betas`` may be any 1-d sequence (the reference's ``isinstance(betas, np.array)``
    test, ptsampler.py:62, raises TypeError for every array);
  * ``run`` honours ``Nburn`` / ``Nthin`` (accepted and ignored at :84-90); the
    defaults give the reference's behaviour;
  * ``sample()`` returns the list of the rungs' current states (the reference's
    ``np.array(zip(...))``, :127, is a Python-2 idiom that yields a 0-d object array);
  * the random stream is explicit (``seed``, ``chain0``), as for ``Sampler``.
"""
import ctypes as C

import numpy as np

from .. import _lib
from ..models.model import DeviceModel, Model
from ..proposals.proposal import DeviceProposal
from ..sampling_errors import ParameterError


class TemperedModel(Model):
    """A Model at inverse temperature beta (ptsampler.py:11-39): the likelihood is
    raised to the power beta, the prior is unchanged.  Host-side view of what the
    device kernel samples for one rung; evaluations go through the base model's
    kernels."""

    def __init__(self, base_model, beta):
        if not (beta >= 0 and beta <= 1):
            raise ParameterError("beta = {} must be a number between 0 and 1".format(beta))
        self.base_model = base_model
        self.beta = beta

    def set_beta(self, beta):
        self.beta = beta

    def log_likelihood(self, theta):
        return self.base_model.log_likelihood(theta) * self.beta

    def log_prior(self, theta):
        return self.base_model.log_prior(theta)


def pt_slots(n_ensembles, K):
    """Slot layout of rmh_run_pt: returns (S, index) with ``index[e, i]`` the chain
    slot of rung i of ensemble e; a 16-slot tile holds 16 // K ensembles."""
    if not 1 <= K <= 16:
        raise ParameterError("the device PTSampler takes 1 to 16 temperatures, not {}".format(K))
    epw = 16 // K
    e = np.arange(n_ensembles)
    idx = ((e // epw) * 16 + (e % epw) * K)[:, None] + np.arange(K)[None, :]
    return -(-n_ensembles // epw) * 16, idx


class _Rung(object):
    """One rung of the ladder with the attributes PTSampler users read on the
    reference's inner Samplers (ptsampler.py:108-124): model, histories, current_state."""

    def __init__(self, pt, i):
        self._pt, self._i = pt, i
        self.model = TemperedModel(pt.model, pt.betas[i])
        self.proposal = pt.proposal

    @property
    def _chain_thetas(self):
        return self._pt._rung_history(self._i)[0]

    @property
    def _chain_logpost(self):
        return self._pt._rung_history(self._i)[1]

    def current_state(self):
        return self._pt._rung_state(self._i)


class PTSampler(object):
    """A parallel-tempered Sampler (ptsampler.py:42-127) for Ne ensembles at once."""

    def __init__(self, model, proposal, theta0, betas=None, Pswap=0.1, seed=0, chain0=0, normals="f32x2"):
        """
        :param model: device Model to sample (MultiGaussianDist or MixtureDist)
        :param proposal: the proposal used within every rung (MetropolisRandomWalk, pCN,
            VanillaHMC / MALA with or without a mass matrix; adaptive proposals are refused)
        :param theta0: initial parameters, ``(d,)`` or ``(Ne, d)``; every rung of an
            ensemble starts there (ptsampler.py:82)
        :param betas: temperature ladder, default ``0.5**arange(5)`` (:60)
        :param Pswap: probability of proposing a swap instead of a within-chain step
        :param seed, chain0: rmh-rng stream; ``chain0`` = global id of the first SLOT
            (a multiple of 16 when the ensembles of a job are split over GPUs)
        :param normals: resolution of the stream's normals, as ``Sampler``
        """
        if not isinstance(model, DeviceModel) or not isinstance(proposal, DeviceProposal):
            raise TypeError("riemann_amd.PTSampler runs on the GPU: model and proposal must be "
                            "device plugins; there is no CPU sampling path in this package")
        if betas is None:
            self.betas = 0.5 ** np.arange(5)
        else:
            self.betas = np.array(betas, dtype=np.float64)
            if self.betas.ndim != 1:
                raise ParameterError("betas must have shape (Ntemps,)")
            if self.betas.shape[0] > 20:
                print("PTSampler: warning -- more than 20 temperatures"
                      " -- watch out, this could be *really* slow")
        if not (Pswap > 0 and Pswap < 1):
            raise ParameterError("Pswap must be a number between 0 and 1")
        self.Pswap = Pswap
        self.model, self.proposal = model, proposal
        th = _lib.as_f64(model._pack_theta(theta0))
        self._single = th.ndim <= 1
        th = np.atleast_2d(np.atleast_1d(th))
        if th.shape[1] != model.Npars:
            raise ParameterError("theta0 has shape {}, model has {} parameters"
                                 .format(np.shape(theta0), model.Npars))
        self.Nensembles, self.Npars = th.shape
        self.Ntemps = K = len(self.betas)
        self._S, self._idx = pt_slots(self.Nensembles, K)
        slots = np.zeros((self._S, self.Npars))
        slots[:] = th[0]                       # idle slots: any finite state
        slots[self._idx] = th[:, None, :]
        self._state = C.c_void_p()
        _lib.call("rmh_state_create", model._h, self._S, int(seed), int(chain0), C.byref(self._state))
        self.normals = normals
        if _lib.normal_mode(normals):
            _lib.call("rmh_state_set_normal_mode", self._state, _lib.normal_mode(normals))
        _lib.call("rmh_state_set_theta", self._state, slots)
        self._hist = None
        # T = 0 run: evaluates the tempered posteriors of the starting states
        _lib.call("rmh_run_pt", model._h, proposal._h, self._state, _lib.as_f64(self.betas), K,
                  float(Pswap), 0, 0, 1, None)
        th0, lp0 = self._pull_state()
        self._host_thetas, self._host_logpost = th0[None], lp0[None]   # (n, Ne, K, d), (n, Ne, K)
        self._host_codes = np.zeros((0, self.Nensembles, K), dtype=np.uint8)
        self.samplers = [_Rung(self, i) for i in range(K)]
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

    # ------------------------------------------------------------ internals --
    def _pull_state(self):
        th, lp = np.empty((self._S, self.Npars)), np.empty(self._S)
        _lib.call("rmh_state_get", self._state, th.ctypes.data, lp.ctypes.data)
        return th[self._idx], lp[self._idx]

    def _rung_state(self, i):
        th, lp = self._pull_state()
        if self._single:
            return th[0, i], float(lp[0, i])
        return th[:, i], lp[:, i]

    def _rung_history(self, i):
        th, lp = self._host_thetas[:, :, i], self._host_logpost[:, :, i]
        if self._single:
            return th[:, 0], lp[:, 0]
        return th, lp

    # ------------------------------------------------------------------ API --
    def run(self, Nsamples, Nburn=0, Nthin=1):
        """Nsamples sweeps of ``sample`` for every ensemble (ptsampler.py:84-90), one
        kernel launch.  The states are APPENDED to the rungs' histories, as the
        reference's inner Samplers do."""
        T, nb, nt = int(Nsamples), int(Nburn), int(Nthin)
        if T < 0 or nb < 0 or nt < 1:
            raise ParameterError("need Nsamples >= 0, Nburn >= 0, Nthin >= 1")
        self.proposal._sync()
        kept = (T - nb) // nt + 1 if T >= nb else 0
        hist = C.c_void_p()
        _lib.call("rmh_history_create", self._state, kept, T, C.byref(hist))
        try:
            _lib.call("rmh_run_pt", self.model._h, self.proposal._h, self._state,
                      _lib.as_f64(self.betas), self.Ntemps, float(self.Pswap), T, nb, nt, hist)
            _lib.call("rmh_synchronize")
            ms = C.c_double()
            _lib.call("rmh_last_run_ms", C.byref(ms), None)
            self.last_run_ms = ms.value
            th, lp = np.empty((kept, self._S, self.Npars)), np.empty((kept, self._S))
            codes = np.empty((T, self._S), dtype=np.uint8)
            if kept:
                _lib.call("rmh_history_download", hist, 0, kept, th.ctypes.data, lp.ctypes.data)
            if T:
                _lib.call("rmh_history_accept", hist, 0, T, codes)
        finally:
            _lib.call("rmh_history_free", hist)
        # entry 0 of a run is the state on entry, already the last entry of the history
        first = 1 if nb == 0 else 0
        self._host_thetas = np.concatenate([self._host_thetas, th[first:, self._idx]])
        self._host_logpost = np.concatenate([self._host_logpost, lp[first:, self._idx]])
        self._host_codes = np.concatenate([self._host_codes, codes[:, self._idx]])

    def sample(self):
        """One sweep (ptsampler.py:92-127); returns the rungs' (theta, logpost)."""
        self.run(1)
        return [s.current_state() for s in self.samplers]

    @property
    def _chain_thetas(self):
        """History of the beta = betas[0] rung (ptsampler.py:89)."""
        return self.samplers[0]._chain_thetas

    @property
    def _chain_logpost(self):
        return self.samplers[0]._chain_logpost

    @property
    def _chain_codes(self):
        """Per sweep, ensemble and rung: bit 0 = state changed, bit 1 = swap proposal."""
        c = self._host_codes
        return c[:, 0] if self._single else c

    def current_state(self):
        return self.samplers[0].current_state()
