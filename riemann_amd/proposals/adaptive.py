"""Adaptive mix-ins on the device: /root/reference/riemann/proposals/adaptive.py:11-35.

Only the scale adaptation (AdaptScaleProposal) is built so far;
the Haario covariance adaptation of adaptive.py:38-103 is not (it also needs
``np.float``, removed from numpy >= 1.24)."""
import numpy as np

from .. import _lib
from .proposal import DeviceProposal


class AdaptScaleProposal(DeviceProposal):
    """Adapts an overall scale towards a target acceptance rate (adaptive.py:11-35).

    In the reference the state (Nsamples, Naccepts, scale) belongs to the one chain
    the proposal serves; here every chain of the Sampler adapts its own scale in the
    fused kernel (the ``adapt`` hook of samplers/sampler.py:88).  After ``run`` the
    attributes ``scale``, ``accept_rate``, ``Nsamples``, ``Naccepts`` hold one value
    per chain (scalars for a single chain)."""

    def _init_adapt(self, target_accept_rate):
        self.Nsamples = 0
        self.Naccepts = 0
        self.last_theta = None
        self.accept_rate = 0.0
        self.target_accept_rate = target_accept_rate
        self.scale = 1.0
        _lib.call("rmh_proposal_set_adapt", self._h, float(target_accept_rate))

    def _pull_adapt(self, state, single):
        """Refresh the public attributes from the chains' adaptation state."""
        N = state["N"]
        a = np.empty((N, 3))
        _lib.call("rmh_state_get_adapt", state["handle"], a)
        ns, na, sc = a[:, 0].astype(np.int64), a[:, 1].astype(np.int64), a[:, 2].copy()
        rate = np.where(ns > 0, na / np.maximum(ns, 1), 0.0)
        if single:
            self.Nsamples, self.Naccepts = int(ns[0]), int(na[0])
            self.scale, self.accept_rate = float(sc[0]), float(rate[0])
        else:
            self.Nsamples, self.Naccepts, self.scale, self.accept_rate = ns, na, sc, rate

    def propose(self, theta):
        raise NotImplementedError("stand-alone propose() of an adaptive proposal is not built: "
                                  "the per-chain scale lives with the Sampler's chains")
