"""Changepoint regression on the device:
/root/reference/riemann/models/changepoint.py:22-45 (ChangepointParams) and
:81-181 (ChangepointRegression1D: log_likelihood, log_prior, predict).

Two modes.  With ``k=`` the number of changepoints is fixed (BASELINE cfg 5, the fast path).
With ``variable_k=True`` the state carries its own number of changepoints k <= kmax in a
padded vector and the reference's trans-dimensional proposal
(``ChangepointRegression1DProp`` with its birth / death branch, :48-78 and :193-240) samples
over k on the device."""
import ctypes as C

import numpy as np

from .. import _lib
from .model import DeviceModel


class ChangepointParams(object):
    """State of a ChangepointRegression1D (changepoint.py:22-45)."""

    def __init__(self, cpx, cpv, sig):
        if len(cpv) != len(cpx) + 1:
            raise ValueError("number of constant pieces ({}) must be "
                             "1 more than number of changepoints ({})"
                             .format(len(cpv), len(cpx)))
        self.cpx = np.array(cpx)
        self.cpv = np.array(cpv)
        self.sig = np.array(sig)

    def pack(self):
        """(cpx[k], cpv[k+1], sig) -> parameter vector of length 2k+2."""
        return np.concatenate([self.cpx, self.cpv, np.atleast_1d(self.sig)]).astype(np.float64)

    @classmethod
    def unpack(cls, v):
        v = np.asarray(v)
        k = (v.shape[-1] - 2) // 2
        return cls(v[:k], v[k:2 * k + 1], v[2 * k + 1])

    def __str__(self):
        return ("ChangepointParams instance with sig = {}, "
                "cpx = {}, cpv = {}".format(self.sig, self.cpx, self.cpv))


class ChangepointRegression1D(DeviceModel):
    """1-D piecewise constant regression (changepoint.py:81-181), k fixed.

    ``theta`` is a packed vector ``(2k+2,)`` / batch ``(N, 2k+2)``, or a
    ChangepointParams instance as in the reference."""

    def __init__(self, x, y, xmin, xmax, lamb, kmax, alpha, beta, k=None, debug=False, variable_k=False):
        if len(x) != len(y):
            raise ValueError("length of predictor array ({}) must be"
                             "same as length of response array ({})".format(len(x), len(y)))
        self.x = _lib.as_f64(x)
        self.y = _lib.as_f64(y)
        self.xmin, self.xmax, self.kmax = xmin, xmax, kmax
        self.lamb, self.alpha, self.beta, self.debug = lamb, alpha, beta, debug
        self.k = None
        self.variable_k = bool(variable_k)
        if self.variable_k:
            if k is not None:
                raise _lib.ParameterError("give either k (fixed) or variable_k=True")
            h = C.c_void_p()
            _lib.call("rmh_model_changepoint_rj_create", self.x, self.y, len(self.x), int(kmax),
                      float(xmin), float(xmax), float(lamb), float(alpha), float(beta), C.byref(h))
            self._h, self.Npars = h, 2 * int(kmax) + 3
        elif k is not None:
            self._bind(int(k))

    # ---- variable number of changepoints: padded states ----
    def pack_padded(self, params):
        """ChangepointParams (or a list of them, one per chain) -> padded states
        ``[cpx (kmax, +inf beyond k), cpv (kmax+1, 1.0 beyond k+1), sig, k]``."""
        single = isinstance(params, ChangepointParams)
        plist = [params] if single else list(params)
        KM = int(self.kmax)
        out = np.zeros((len(plist), 2 * KM + 3))
        out[:, :KM] = np.inf
        out[:, KM:2 * KM + 1] = 1.0
        for r, t in zip(out, plist):
            k = len(t.cpx)
            if k > KM:
                raise _lib.ParameterError("{} changepoints exceed kmax = {}".format(k, KM))
            r[:k], r[KM:KM + k + 1], r[2 * KM + 1], r[2 * KM + 2] = t.cpx, t.cpv, float(t.sig), k
        return out[0] if single else out

    def unpack_padded(self, v):
        """Padded state(s) -> ChangepointParams (a list for a batch)."""
        v = np.asarray(v, dtype=np.float64)
        KM = int(self.kmax)

        def one(r):
            k = int(r[2 * KM + 2])
            return ChangepointParams(r[:k].copy(), r[KM:KM + k + 1].copy(), r[2 * KM + 1])
        return one(v) if v.ndim == 1 else [one(r) for r in v.reshape(-1, v.shape[-1])]

    def _bind(self, k):
        """(Re)create the device model for k changepoints."""
        if self.k == k and self._h:
            return
        DeviceModel.__del__(self)
        h = C.c_void_p()
        _lib.call("rmh_model_changepoint_create", self.x, self.y, len(self.x), k,
                  float(self.xmin), float(self.xmax), float(self.lamb), float(self.alpha),
                  float(self.beta), C.byref(h))
        self._h, self.k, self.Npars = h, k, 2 * k + 2

    def _pack_theta(self, theta):
        if self.variable_k:
            if isinstance(theta, ChangepointParams) or (
                    isinstance(theta, (list, tuple)) and len(theta) and isinstance(theta[0], ChangepointParams)):
                return self.pack_padded(theta)
            return np.asarray(theta, dtype=np.float64)
        if isinstance(theta, ChangepointParams):
            theta = theta.pack()
        theta = np.asarray(theta, dtype=np.float64)
        self._bind((theta.shape[-1] - 2) // 2)
        return theta

    def predict(self, theta, x):
        """cpv[searchsorted(cpx, x)] (changepoint.py:175-181), host-side helper."""
        if not isinstance(theta, ChangepointParams):
            theta = ChangepointParams.unpack(theta)
        return theta.cpv[np.searchsorted(theta.cpx, x)]
