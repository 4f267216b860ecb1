"""Logistic regression as a Model plugin (not in the reference).

    logL(beta) = sum_i y_i eta_i - log(1 + exp(eta_i)),  eta = X beta
    logP(beta) = -beta.beta / (2 tau2)
    grad       = X^T (y - sigmoid(eta)) - beta / tau2

Written against /root/reference/riemann/models/model.py:4-64; the dense X beta
and X^T r contractions run on v_mfma_f64_16x16x4_f64."""
import ctypes as C

import numpy as np

from .. import _lib
from ..sampling_errors import ParameterError
from .model import DeviceModel


class LogisticRegression(DeviceModel):
    def __init__(self, X, y, tau2=100.0):
        X = _lib.as_f64(X)
        y = _lib.as_f64(y)
        if X.ndim != 2 or y.ndim != 1 or X.shape[0] != y.shape[0]:
            raise ParameterError("X and y have incompatible shapes {}, {}".format(X.shape, y.shape))
        self.X, self.y, self.tau2 = X, y, float(tau2)
        self.Ndata, self.Npars = X.shape
        h = C.c_void_p()
        _lib.call("rmh_model_logreg_create", X, y, X.shape[0], X.shape[1], self.tau2, C.byref(h))
        self._h = h
