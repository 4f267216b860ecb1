"""MetropolisRandomWalk on the device: /root/reference/riemann/proposals/randomwalk.py:12-26."""
import ctypes as C

import numpy as np

from .. import _lib
from .adaptive import AdaptCovProposal, AdaptScaleProposal
from .proposal import DeviceProposal


class MetropolisRandomWalk(DeviceProposal):
    """theta' = theta + scale * L xi, xi ~ N(0, I), L = cholesky(C); logqratio = 0
    (randomwalk.py:17-26).  ``.scale`` and ``.L`` as in the reference."""

    def __init__(self, C_):
        self.scale = 1.0
        self.L = np.linalg.cholesky(np.atleast_2d(np.asarray(C_, dtype=np.float64)))
        self.Npars = self.L.shape[0]
        h = C.c_void_p()
        _lib.call("rmh_proposal_rw_create", _lib.as_f64(self.L), self.Npars, float(self.scale),
                  C.byref(h))
        self._h = h

    def _sync(self):
        _lib.call("rmh_proposal_set_scale", self._h, float(self.scale))


class AdaptScaleRandomWalk(AdaptScaleProposal, MetropolisRandomWalk):
    """A Metropolis random walk that adapts only the overall scale, target accept
    rate 0.25 (randomwalk.py:29-38).  ``.scale`` is adapted per chain by the kernel."""

    def __init__(self, C_):
        MetropolisRandomWalk.__init__(self, C_)
        self._init_adapt(0.25)

    def _sync(self):
        pass  # the scale is the chains' state, not the proposal's


class AdaptCovRandomWalk(AdaptCovProposal, MetropolisRandomWalk):
    """An adaptive Metropolis random walk after Haario et al. (2001) that updates the
    proposal shape from the chain covariance (randomwalk.py:41-56); options as in the
    reference plus ``exact_squares`` (see AdaptCovProposal)."""

    def __init__(self, C0, t_adapt=1, marginalize=False, smooth_adapt=False, exact_squares=False):
        MetropolisRandomWalk.__init__(self, C0)
        self._init_adapt_cov(C0, t_adapt=t_adapt, marginalize=marginalize, smooth_adapt=smooth_adapt,
                             exact_squares=exact_squares)


# aliases for backward compatibility, randomwalk.py:59
AdaptiveMetropolisRandomWalk = HaarioRandomWalk = AdaptCovRandomWalk


class AdaptScaleCovRandomWalk(AdaptScaleProposal, AdaptCovProposal, MetropolisRandomWalk):
    """A Metropolis random walk that adapts both scale and covariance (randomwalk.py:62-75;
    "experimental and not obviously correct" there): every chain adapts its own scale
    towards an acceptance rate of 0.25 while the covariance adapts from the pooled history."""

    def __init__(self, C0, t_adapt=1, marginalize=False, smooth_adapt=False, exact_squares=False):
        MetropolisRandomWalk.__init__(self, C0)
        self._init_adapt(0.25)
        self._init_adapt_cov(C0, t_adapt=t_adapt, marginalize=marginalize, smooth_adapt=smooth_adapt,
                             exact_squares=exact_squares)

    def _sync(self):
        pass  # the scale is the chains' state

    def _pull_adapt(self, state, single):
        AdaptScaleProposal._pull_adapt(self, state, single)
        AdaptCovProposal._pull_adapt(self, state, single)


class pCN(DeviceProposal):
    """A preconditioned Crank-Nicolson proposal (randomwalk.py:78-101):
    theta' = rho theta + sqrt(1-rho^2) L xi with its asymmetric logqratio.
    ``.scale``, ``.rho``, ``.rho_c``, ``.L`` as in the reference."""

    def __init__(self, C_, rho):
        self.scale = 1.0
        self.rho = rho
        self.rho_c = np.sqrt(1 - self.rho ** 2)
        self.L = np.linalg.cholesky(np.atleast_2d(np.asarray(C_, dtype=np.float64)))
        self.Npars = self.L.shape[0]
        h = C.c_void_p()
        _lib.call("rmh_proposal_pcn_create", _lib.as_f64(self.L), self.Npars, float(rho), C.byref(h))
        self._h = h


class AdaptScalepCN(AdaptScaleProposal, pCN):
    """A preconditioned Crank-Nicolson proposal with an adaptive scaling parameter targeting
    an acceptance rate of 0.25 (randomwalk.py:104-119).  As in the reference, before every
    proposal ``rho = tanh(rho / scale)`` -- a recursion on the current ``rho``, not on
    ``rho0`` -- while ``rho_c`` keeps its initial value.  Every chain carries its own ``rho``
    and ``scale`` (arrays after ``run`` when the Sampler has several chains)."""

    def __init__(self, C_, rho):
        pCN.__init__(self, C_, rho)
        self._init_adapt(0.25)
        self.rho0 = self.rho

    def _pull_adapt(self, state, single):
        AdaptScaleProposal._pull_adapt(self, state, single)
        aux = np.empty(state["N"])
        _lib.call("rmh_state_get_adapt_aux", state["handle"], aux)
        aux = np.where(np.isnan(aux), self.rho0, aux)
        self.rho = float(aux[0]) if single else aux
