"""Adaptive mix-ins on the device: /root/reference/riemann/proposals/adaptive.py:11-103
(SURVEY.md 8f-1): AdaptScaleProposal (per-chain scale, inside the fused kernel) and
AdaptCovProposal (Haario covariance adaptation, pooled over the chains)."""
import numpy as np

from .. import _lib
from .proposal import DeviceProposal


# ---- the two adaptation rules, once, for every host-side user ---------------------------------------
# (the mixed-plugin loop of samplers/sampler.py; tests replay the reference's fixtures through them:
# tests/test_abi_and_host.py.  The device states them in csrc/rmh_run.hpp AdaptScale and
# csrc/rmh_kernels_mfma.hip k_acov_update; the CPU checker under tests keeps its own restatement.)
def adapt_scale_step(ada, moved, target):
    """``AdaptScaleProposal.adapt`` (adaptive.py:25-35) for N chains at once, in place: ``ada`` (N, 3)
    holds Nsamples, Naccepts, scale; ``moved`` (N,) whether the chain's state changed in this step.
    The first call compares with ``last_theta = None`` and therefore always counts as a move
    (adaptive.py:27 with :22)."""
    m = np.where((ada[:, 0] == 0.0) | np.asarray(moved, dtype=bool), 1.0, 0.0)
    ada[:, 0] += 1.0
    ada[:, 1] += m
    ratio = np.exp(1.0 / ada[:, 0])
    up = ada[:, 1] / ada[:, 0] > target
    ada[:, 2] = np.where(up, ada[:, 2] * ratio, ada[:, 2] / ratio)
    return ada


def adapt_cov_due(n, exact_squares=False):
    """Is a covariance update due after the n-th ``adapt`` call?  adaptive.py:81-82: ``np.sqrt(n)**2
    == n and n > 2`` -- literally (the float test also passes for about half of the non-squares),
    or at perfect squares only."""
    if not n > 2:
        return False
    r = np.sqrt(float(n))
    return bool(np.floor(r + 0.5) ** 2 == n) if exact_squares else bool(r * r == n)


def adapt_cov_update(count, SX, SX2, C0, t_adapt, smooth_adapt=False, marginalize=False):
    """The covariance update of ``AdaptCovProposal.adapt`` (adaptive.py:83-102) from the moment sums
    over ``count`` samples: returns (C, L) with C = cov / d^0.4 and L = cholesky(C) / d^0.2.  Raises
    numpy's LinAlgError when the adapted covariance is not positive definite, as the reference does."""
    d = C0.shape[0]
    cs = (SX2 - np.outer(SX, SX) / count) / (count - 1.0)
    if smooth_adapt:
        cv = (count * cs + t_adapt * C0) / (count + t_adapt)
    elif count < t_adapt:
        cv = C0.copy()                                    # (:90 assigns this to an unused _cov; read as written)
    else:
        cv = cs + 1e-12 * (np.trace(C0) / d) * np.eye(d)
    if marginalize:
        cv = np.diag(np.diag(cv))
    C = cv / float(d) ** 0.4
    return C, np.linalg.cholesky(C) / float(d) ** 0.2


class AdaptScaleProposal(DeviceProposal):
    """Adapts an overall scale towards a target acceptance rate (adaptive.py:11-35).

    In the reference the state (Nsamples, Naccepts, scale) belongs to the one chain
    the proposal serves; here every chain of the Sampler adapts its own scale in the
    fused kernel (the ``adapt`` hook of samplers/sampler.py:88).  After ``run`` the
    attributes ``scale``, ``accept_rate``, ``Nsamples``, ``Naccepts`` hold one value
    per chain (scalars for a single chain)."""

    def _init_adapt(self, target_accept_rate):
        if not self._h and getattr(self, "_host_grad", None) is None:
            raise TypeError("the adaptive device proposals need a device Model (their adaptation state "
                            "lives with the chains on the GPU)")
        self.Nsamples = 0
        self.Naccepts = 0
        self.last_theta = None
        self.accept_rate = 0.0
        self.target_accept_rate = target_accept_rate
        self.scale = 1.0
        if self._h:   # (an HMC on a Python gradient callable adapts on the host, inside the Sampler's mixed loop)
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

    def propose(self, theta, rng=None):
        """Stand-alone form for the random walk: theta + scale * (L xi) with the current ``scale`` (a
        scalar, or one value per row of ``theta`` after a many-chain ``run``), as randomwalk.py:26
        with the adapted scale.  The pCN- and HMC-based variants are built inside ``Sampler.run`` only."""
        from .randomwalk import MetropolisRandomWalk, pCN
        if getattr(self, "_host_grad", None) is not None:     # AdaptScaleHMC on a Python gradient callable
            if getattr(self, "_host_scale", None) is None:
                self._host_scale = np.atleast_1d(np.asarray(self.scale, dtype=np.float64))
            return self._propose_host(theta, rng)
        if not isinstance(self, MetropolisRandomWalk) or isinstance(self, pCN):
            raise NotImplementedError("stand-alone propose() of this adaptive proposal is not built: "
                                      "its state lives with the Sampler's chains")
        th = _lib.as_f64(theta)
        single = th.ndim <= 1
        th2 = np.atleast_2d(np.atleast_1d(th))
        sc = np.asarray(self.scale, dtype=np.float64)
        if sc.ndim and sc.shape[0] != th2.shape[0]:
            raise _lib.ParameterError("scale has one value per chain of the last run: {} values, {} rows of theta"
                                      .format(sc.shape[0], th2.shape[0]))
        lxi, logq = DeviceProposal.propose(self, np.zeros_like(th2), rng=rng)
        thp = th2 + (sc[:, None] if sc.ndim else sc) * np.atleast_2d(lxi)
        if single:
            return thp[0], float(np.atleast_1d(logq)[0])
        return thp, np.atleast_1d(logq)


class AdaptCovProposal(DeviceProposal):
    """Adapts the proposal covariance to the covariance of the chain history, after
    Haario et al. (2001) (adaptive.py:38-103).  ``self.C`` and ``self.L`` hold the adapted
    covariance and its Cholesky factor after ``run``, as in the reference.

    The reference adapts ONE chain's proposal from that chain's history.  Here one
    proposal serves all N chains of a Sampler and the accumulators of ``adapt`` are pooled
    over the chains (over the ranks too once ``distributed.comm_init`` was called); with
    one chain this is the reference's algorithm.  Two statements of the reference are read
    as written rather than as executed: ``np.float`` (:81, removed from numpy) is ``float``,
    and ``C = C0 while n < t_adapt`` is applied (:90 assigns it to an unused ``_cov``).
    The update test ``np.sqrt(n)**2 == n`` (:82) is kept literally -- in double arithmetic
    it also passes for about half of the non-squares; ``exact_squares=True`` adapts at
    perfect squares only (far fewer kernel launches)."""

    def _init_adapt_cov(self, C0, t_adapt=1, marginalize=False, smooth_adapt=False, exact_squares=False):
        self.C0 = np.array(np.atleast_2d(C0), dtype=np.float64)
        self.C = self.C0
        self.L = np.linalg.cholesky(self.C0)
        self.t_adapt = t_adapt
        self.marginalize = marginalize
        self.smooth_adapt = smooth_adapt
        self.exact_squares = exact_squares
        self._S = 0.0
        _lib.call("rmh_proposal_set_adapt_cov", self._h, _lib.as_f64(self.C0).ctypes.data, float(t_adapt),
                  int(bool(marginalize)), int(bool(smooth_adapt)), int(bool(exact_squares)))

    def _adapt_cov_state(self):
        d = self.C0.shape[0]
        st = np.empty(2 + d + 3 * d * d)
        _lib.call("rmh_proposal_adapt_cov_get", self._h, st)
        return st

    def _pull_adapt(self, state, single):
        """Refresh ``C``, ``L`` and the accumulators from the device after a run."""
        d = self.C0.shape[0]
        st = self._adapt_cov_state()
        self._S = st[0]
        self._SX = st[1:1 + d].copy()
        self._SX2 = st[1 + d:1 + d + d * d].reshape(d, d).copy()
        self.C = st[1 + d + d * d:1 + d + 2 * d * d].reshape(d, d).copy()
        self.L = st[1 + d + 2 * d * d:1 + d + 3 * d * d].reshape(d, d).copy()

    def propose(self, theta, rng=None):
        """Stand-alone form with the CURRENT adapted factor: theta + L xi (randomwalk.py:26 with the L of
        adaptive.py:101-102), or the trajectory with the current mass matrix for the HMC variant
        (hamiltonian.py:117-119) -- what the reference's propose() does between two adapt() calls."""
        return DeviceProposal.propose(self, theta, rng=rng)
