"""VanillaHMC / MALA on the device: /root/reference/riemann/proposals/hamiltonian.py:13-91.

The reference takes an arbitrary Python callable for the gradient
(hamiltonian.py:60-70).  Here the gradient must come from a device Model, so
``gradlogpost`` is either that Model or one of its bound gradient methods
(``model.grad_log_posterior`` / ``model.grad_log_likelihood``), which is how
the reference's own example passes it (examples/test_randomwalk.py:32-37).
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

    def __init__(self, eps, Nsteps, gradlogpost, M=None):
        model = getattr(gradlogpost, "__self__", gradlogpost)
        if not isinstance(model, DeviceModel):
            raise TypeError("VanillaHMC on the MI355X back end needs the gradient of a device "
                            "Model: pass the Model or its grad_log_posterior method")
        if int(Nsteps) != Nsteps or Nsteps < 1:
            raise _lib.ParameterError("Nsteps must be an integer >= 1")
        self.Nsteps, self.eps = int(Nsteps), eps
        self._mgradU = gradlogpost
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
        _lib.call("rmh_proposal_set_eps", self._h, float(self.eps))


class MALA(VanillaHMC):
    """Metropolis-adjusted Langevin: exactly VanillaHMC(eps, 1, grad) -- theta' =
    theta + eps^2/2 grad + eps z, and dK equals the Langevin proposal ratio
    (SURVEY.md 8a-7).  The reference has no such class."""

    def __init__(self, eps, gradlogpost):
        VanillaHMC.__init__(self, eps, 1, gradlogpost)


class AdaptScaleHMC(AdaptScaleProposal, VanillaHMC):
    """HMC whose step size is eps = scale * eps0 with the scale adapted towards an
    accept rate of 0.75 (hamiltonian.py:94-104), per chain."""

    def __init__(self, eps, Nsteps, gradlogpost, M=None):
        VanillaHMC.__init__(self, eps, Nsteps, gradlogpost, M=M)
        self._init_adapt(0.75)
        self.eps0 = self.eps

    def _sync(self):
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
