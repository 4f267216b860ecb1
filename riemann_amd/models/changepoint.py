"""Changepoint regression at a fixed number of changepoints, on the device:
/root/reference/riemann/models/changepoint.py:22-45 (ChangepointParams) and
:81-181 (ChangepointRegression1D: log_likelihood, log_prior, predict).

The reversible-jump moves of :48-78, :193-240 change the dimension of the
state and are out of scope of this back end."""
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

    def __init__(self, x, y, xmin, xmax, lamb, kmax, alpha, beta, k=None, debug=False):
        if len(x) != len(y):
            raise ValueError("length of predictor array ({}) must be"
                             "same as length of response array ({})".format(len(x), len(y)))
        self.x = _lib.as_f64(x)
        self.y = _lib.as_f64(y)
        self.xmin, self.xmax, self.kmax = xmin, xmax, kmax
        self.lamb, self.alpha, self.beta, self.debug = lamb, alpha, beta, debug
        self.k = None
        if k is not None:
            self._bind(int(k))

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
