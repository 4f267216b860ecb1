"""VanillaHMC / MALA: /root/reference/riemann/proposals/hamiltonian.py:13-91.

``gradlogpost`` is (i) a device Model or one of its bound gradient methods
(``model.grad_log_posterior`` / ``model.grad_log_likelihood``: how the reference's own example
passes it, examples/test_randomwalk.py:32-37) -- then the whole trajectory is fused with the
sampler in one kernel -- or (ii) any Python callable ``theta (d,) -> gradient (d,)``
(hamiltonian.py:60-70), the plugin contract of the reference: the trajectory then runs on the
host, one callback per leapfrog step, with the momenta drawn from the rmh-rng stream on the
device (``rmh_rng_draw``).  (ii) is the plugin API for user code, not a CPU twin of the device
models: a device Model never takes that route.
Unlike the reference module, importing this one does not import matplotlib."""
import ctypes as C

import numpy as np

from .. import _lib
from ..models.model import DeviceModel
from .adaptive import AdaptCovProposal, AdaptScaleProposal
from .proposal import DeviceProposal


class VanillaHMC(DeviceProposal):
    """Leapfrog proposal with identity mass matrix; logqratio = dK
    (hamiltonian.py:76-91).  ``.eps``, ``.Nsteps``, ``.M``, ``.chM`` as in the reference."""

    def __init__(self, eps, Nsteps, gradlogpost, M=None, Npars=None, batched=False):
        """``Npars``: the dimension, needed only with a Python gradient callable when no ``M``
        gives it; ``batched``: the callable takes (N, d) and returns (N, d)."""
        model = getattr(gradlogpost, "__self__", gradlogpost)
        if int(Nsteps) != Nsteps or Nsteps < 1:
            raise _lib.ParameterError("Nsteps must be an integer >= 1")
        self.Nsteps, self.eps = int(Nsteps), eps
        self._mgradU = gradlogpost
        if not isinstance(model, DeviceModel):
            if not callable(gradlogpost):
                raise TypeError("VanillaHMC needs a device Model or a callable gradient (hamiltonian.py:60-70)")
            self._init_host(gradlogpost, M, Npars, batched)
            return
        self._model = model
        self.Npars = model.Npars
        h = C.c_void_p()
        _lib.call("rmh_proposal_hmc_create", float(eps), self.Nsteps, self.Npars, C.byref(h))
        self._h = h
        if M is None:                                         # hamiltonian.py:71-74
            self.M, self.chM = None, None
        else:
            self.M = np.atleast_2d(np.asarray(M, dtype=np.float64))
            self.chM = np.linalg.cholesky(self.M)
            if self.chM.shape != (self.Npars, self.Npars):
                raise _lib.ParameterError("M has shape {}, model has {} parameters"
                                          .format(self.M.shape, self.Npars))
            _lib.call("rmh_proposal_hmc_set_mass", self._h, _lib.as_f64(self.chM).ctypes.data)

    def _sync(self):
        if self._h:
            _lib.call("rmh_proposal_set_eps", self._h, float(self.eps))

    # ---- (ii) Python gradient callable: the trajectory on the host (hamiltonian.py:13-52, 76-91)
    _host_grad = None

    def _init_host(self, grad, M, Npars, batched):
        self._host_grad, self._batched = grad, bool(batched)
        if M is None:
            self.M = self.chM = None
            if Npars is not None:
                self.Npars = int(Npars)
        else:
            self.M = np.atleast_2d(np.asarray(M, dtype=np.float64))
            self.chM = np.linalg.cholesky(self.M)
            self.Npars = self.M.shape[0]

    def _grad_rows(self, q):
        if self._batched:
            return np.asarray(self._host_grad(q), dtype=np.float64)
        return np.stack([np.asarray(self._host_grad(row), dtype=np.float64) for row in q])

    def _propose_host(self, theta, rng):
        th = _lib.as_f64(theta)
        single = th.ndim <= 1
        q = np.atleast_2d(np.atleast_1d(th)).copy()
        N, d = q.shape
        if self.Npars is None:
            self.Npars = d
        if d != self.Npars:
            raise _lib.ParameterError("theta and M have incompatible shapes")
        p = np.empty((1, N, d))
        _lib.call("rmh_rng_draw_mode", self._stream(N) if rng is None else rng, N, d, 1,
                  _lib.normal_mode(self.normals), p.ctypes.data, None)
        p = p[0]                                                   # p ~ N(0, I), hamiltonian.py:79
        k0 = np.sum(p * p, axis=1)
        # AdaptScaleHMC on a Python gradient: eps = scale * eps0 per chain (hamiltonian.py:101-102); the Sampler
        # hands the chains' scales through _host_scale
        sc = getattr(self, "_host_scale", None)
        eps = float(self.eps) if sc is None else (np.asarray(sc, dtype=np.float64) * float(self.eps0)).reshape(-1, 1)
        if self.chM is not None:
            p = p @ self.chM.T                                     # :80-81
            minv = lambda v: np.linalg.solve(self.M, v.T).T        # noqa: E731   :29, :36
        else:
            minv = lambda v: v                                     # noqa: E731
        p = p + 0.5 * eps * self._grad_rows(q)                     # leapfrog, :27
        q = q + eps * minv(p)                                      # :29-30
        for _ in range(self.Nsteps - 1):                           # :33-37
            p = p + eps * self._grad_rows(q)
            q = q + eps * minv(p)
        p = p + 0.5 * eps * self._grad_rows(q)                     # :40
        if self.chM is not None:
            p = np.linalg.solve(self.chM, p.T).T                   # :85-87
        logq = 0.5 * (np.sum(p * p, axis=1) - k0)                  # :89
        return (q[0], float(logq[0])) if single else (q, logq)

    def propose(self, theta, rng=None):
        if self._host_grad is not None:
            return self._propose_host(theta, rng)
        return DeviceProposal.propose(self, theta, rng)


class MALA(VanillaHMC):
    """Metropolis-adjusted Langevin: exactly VanillaHMC(eps, 1, grad) -- theta' =
    theta + eps^2/2 grad + eps z, and dK equals the Langevin proposal ratio
    (SURVEY.md 8a-7).  The reference has no such class."""

    def __init__(self, eps, gradlogpost, Npars=None, batched=False):
        VanillaHMC.__init__(self, eps, 1, gradlogpost, Npars=Npars, batched=batched)


class AdaptScaleHMC(AdaptScaleProposal, VanillaHMC):
    """HMC whose step size is eps = scale * eps0 with the scale adapted towards an
    accept rate of 0.75 (hamiltonian.py:94-104), per chain."""

    def __init__(self, eps, Nsteps, gradlogpost, M=None, Npars=None, batched=False):
        VanillaHMC.__init__(self, eps, Nsteps, gradlogpost, M=M, Npars=Npars, batched=batched)
        self._init_adapt(0.75)
        self.eps0 = self.eps

    def _sync(self):
        if self._h:
            _lib.call("rmh_proposal_set_eps", self._h, float(self.eps0))


class AdaptCovHMC(AdaptCovProposal, VanillaHMC):
    """HMC whose mass matrix follows the covariance of the chain history
    (hamiltonian.py:107-120): before every proposal ``M, chM = C, L`` of the covariance
    adaptation (AdaptCovProposal, pooled over the chains here).  As in the reference, after
    the first update ``chM`` is not the Cholesky factor of ``M`` but ``M = d^0.4 chM chM^T``
    (adaptive.py:101-102): momenta are drawn and weighed with ``chM``, positions drift with
    ``M^-1``."""

    def __init__(self, eps, Nsteps, gradlogpost, M0, t_adapt=1, marginalize=False, smooth_adapt=False,
                 exact_squares=False):
        VanillaHMC.__init__(self, eps, Nsteps, gradlogpost, M=None)
        self._init_adapt_cov(M0, t_adapt=t_adapt, marginalize=marginalize, smooth_adapt=smooth_adapt,
                             exact_squares=exact_squares)
        self.M, self.chM = self.C, self.L

    def _pull_adapt(self, state, single):
        AdaptCovProposal._pull_adapt(self, state, single)
        self.M, self.chM = self.C, self.L


class AdaptScaleCovHMC(AdaptScaleHMC):
    """hamiltonian.py:123-134.  The reference class defines no ``adapt`` of its own, so
    Python's method resolution order finds ``AdaptScaleProposal.adapt`` and the covariance
    part never runs: the sampler is AdaptScaleHMC with the constant mass matrix ``M0``
    (``C``, ``L`` stay at their initial values).  Reproduced as that."""

    def __init__(self, eps, Nsteps, gradlogpost, M0, t_adapt=1, marginalize=False, smooth_adapt=False):
        AdaptScaleHMC.__init__(self, eps, Nsteps, gradlogpost, M=M0)
        self.C0 = self.C = self.M
        self.L = self.chM
        self.t_adapt, self.marginalize, self.smooth_adapt = t_adapt, marginalize, smooth_adapt
