"""The block-mixture proposal of the reference's changepoint smoke script on the device:
/root/reference/riemann/test_changepoint.py:18-73 (class ChangepointRegression1DProp), at
fixed k (variant B)."""
import numpy as np

from .. import _lib
from .randomwalk import MetropolisRandomWalk


class ChangepointRegression1DProp(MetropolisRandomWalk):
    """Proposal for changepoint regression with a fixed number of changepoints.

    Same constructor and attributes as the reference (``P_cumprop_cpx``, ``_cpv``, ``_sig``,
    ``_dim``; ``cpx_prop`` / ``cpv_prop`` / ``cps_prop`` covariances of
    test_changepoint.py:36-41).  Each iteration chained independent uniforms pick ONE block
    (:47-55): cpx with probability P_cumprop_cpx, else cpv with P_cumprop_cpv, else sig with
    P_cumprop_sig; the picked block makes a MetropolisRandomWalk step, the others stay.
    When no block is picked the reference makes a trans-dimensional move (:56-71), which
    needs a variable number of parameters: here the proposal is then theta itself
    (set ``P_cumprop_sig = 1.0`` for a pure fixed-k sampler)."""

    def __init__(self, model, hscale):
        if getattr(model, "variable_k", False):
            self._init_trans_dimensional(model, hscale)
            return
        self.model = model
        self.Ndata = len(model.x)
        self.hscale = hscale
        self.P_cumprop_cpx = 0.20
        self.P_cumprop_cpv = 0.40
        self.P_cumprop_sig = 0.60
        self.P_cumprop_dim = 1.00
        if getattr(model, "k", None) is None:
            raise _lib.ParameterError("the device proposal works at a fixed number of changepoints: "
                                      "construct the model with k=...")
        self.k = k = model.k
        xmin, xmax = model.xmin, model.xmax
        # _reset_proposal_covariances, test_changepoint.py:34-41
        self.C_cpx = 0.01 * (xmax - xmin) / (k + 1) * np.eye(k)
        self.C_cpv = 0.01 * self.hscale ** 2 / self.Ndata * np.eye(k + 1)
        self.C_sig = np.atleast_2d(0.01 * self.hscale)
        Cdiag = np.concatenate([np.diag(self.C_cpx), np.diag(self.C_cpv), np.diag(self.C_sig)])
        MetropolisRandomWalk.__init__(self, np.diag(Cdiag))
        self._sizes = np.array([k, k + 1, 1], dtype=np.int32)
        self._sync()

    def _init_trans_dimensional(self, model, hscale):
        """On a ``variable_k=True`` model: the whole proposal of test_changepoint.py:43-73,
        birth / death moves included (``P_cumprop_dim`` is what remains after the three
        blocks); the block covariances follow the chain's current k (:34-41)."""
        import ctypes as C
        self.model, self.Ndata, self.hscale = model, len(model.x), hscale
        self.P_cumprop_cpx, self.P_cumprop_cpv, self.P_cumprop_sig, self.P_cumprop_dim = 0.20, 0.40, 0.60, 1.00
        self.k = None
        self.scale = 1.0
        self.Npars = model.Npars
        self._rj = True
        h = C.c_void_p()
        _lib.call("rmh_proposal_cprj_create", float(hscale), _lib.as_f64(
            [self.P_cumprop_cpx, self.P_cumprop_cpv, self.P_cumprop_sig]), self.Npars, C.byref(h))
        self._h = h

    def _sync(self):
        pcum = _lib.as_f64([self.P_cumprop_cpx, self.P_cumprop_cpv, self.P_cumprop_sig])
        if getattr(self, "_rj", False):
            _lib.call("rmh_proposal_cprj_set_probs", self._h, pcum)
            return
        MetropolisRandomWalk._sync(self)
        _lib.call("rmh_proposal_rw_set_blocks", self._h, 3, self._sizes, pcum)
