"""Model plugin contract: /root/reference/riemann/models/model.py:4-64, plus the
device-backed base class the MI355X kernels need."""
import numpy as np

from .. import _lib


class Model(object):
    """A statistical model: prior, likelihood, posterior (model.py:4-64).

    Same methods as the reference.  ``theta`` may be one state ``(Npars,)`` or
    a batch ``(N, Npars)``; batched calls return ``(N,)`` arrays."""

    def __init__(self):
        pass

    def pack(self):
        pass

    def unpack(self, theta):
        pass

    def log_likelihood(self, theta):
        raise NotImplementedError("Non-overloaded abstract method!")

    def log_prior(self, theta):
        raise NotImplementedError("Non-overloaded abstract method!")

    def log_posterior(self, theta):
        """logP + logL, or -inf when either is inf or nan (model.py:49-55)."""
        logP, logL = self.log_prior(theta), self.log_likelihood(theta)
        logP, logL = np.asarray(logP, dtype=np.float64), np.asarray(logL, dtype=np.float64)
        bad = np.isinf(logP) | np.isnan(logP) | np.isinf(logL) | np.isnan(logL)
        with np.errstate(invalid="ignore"):
            out = np.where(bad, -np.inf, logP + logL)
        return float(out) if out.ndim == 0 else out

    def logL(self, theta):
        return self.log_likelihood(theta)

    def logP(self, theta):
        return self.log_prior(theta)

    def __call__(self, theta):
        return self.log_posterior(theta)


class DeviceModel(Model):
    """A Model whose densities (and gradient) are HIP kernels.

    Subclasses create ``self._h`` (an ``rmh_model_t``) and set ``self.Npars``.
    Every evaluation goes through the C ABI; nothing is computed in Python."""

    _h = None
    Npars = None

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                _lib.load().rmh_model_free(h)
            except Exception:
                pass

    def _batch(self, theta):
        th = _lib.as_f64(theta)
        single = th.ndim == 1
        th = np.atleast_2d(th)
        if th.ndim != 2 or th.shape[1] != self.Npars:
            raise _lib.ParameterError("theta has shape {}, model has {} parameters"
                                      .format(np.shape(theta), self.Npars))
        return th, single

    def _eval(self, fn, theta, grad=False):
        th, single = self._batch(self._pack_theta(theta))
        out = np.empty(th.shape if grad else th.shape[0])
        _lib.call(fn, self._h, th, th.shape[0], out)
        if single:
            return out[0] if grad else float(out[0])
        return out

    def _pack_theta(self, theta):
        return theta

    def log_prior(self, theta):
        return self._eval("rmh_log_prior", theta)

    def log_likelihood(self, theta):
        return self._eval("rmh_log_likelihood", theta)

    def log_posterior(self, theta):
        return self._eval("rmh_log_posterior", theta)

    def grad_log_posterior(self, theta):
        """Gradient of the log posterior: the callable VanillaHMC expects
        (/root/reference/riemann/proposals/hamiltonian.py:60-70)."""
        return self._eval("rmh_grad_log_posterior", theta, grad=True)
