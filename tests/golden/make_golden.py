#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running THE REFERENCE.

Run in the build container only (the reference never travels):

    PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python3 tests/golden/make_golden.py

What it does: imports rscalzo/riemann from /root/reference, replaces
``np.random.normal`` / ``np.random.uniform`` (looked up at call time at
riemann/proposals/randomwalk.py:25, riemann/proposals/hamiltonian.py:79 and
riemann/samplers/sampler.py:84) by functions that replay the build's own
"rmh-rng v4" stream (oracle/rmh_oracle.c section 1), runs the reference's
``Sampler`` once per chain and records inputs and outputs as small ``.npz``
files.  Only data is written: no reference source text is stored.

Fixture ids follow SURVEY.md section 8c (G1..G7).
"""
import hashlib
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
sys.dont_write_bytecode = True
os.environ.setdefault("MPLBACKEND", "Agg")

from oracle import oracle as orc  # noqa: E402  (stream source only)
from riemann_amd.workloads import changepoint_data, logreg_data  # noqa: E402

import riemann  # noqa: E402  (the reference)
from riemann import Model, ParameterError, Sampler  # noqa: E402
from riemann.models.gaussian import MultiGaussianDist  # noqa: E402
from riemann.models import benchmarks as rbench  # noqa: E402
from riemann.proposals.hamiltonian import VanillaHMC  # noqa: E402
from riemann.proposals.randomwalk import MetropolisRandomWalk  # noqa: E402


# ----------------------------------------------------------- stream replay --
class Replay:
    """Feeds one chain's (xi[t], u[t]) to the reference in consumption order."""

    def __init__(self, xi, u):
        self.xi, self.u, self.tn, self.tu = xi, u, 0, 0

    def normal(self, loc=0.0, scale=1.0, size=None):
        v = self.xi[self.tn]
        self.tn += 1
        shape = (size,) if np.isscalar(size) else tuple(size)
        assert v.shape == shape, (v.shape, shape)
        return v.copy()

    def uniform(self, low=0.0, high=1.0, size=None):
        assert size is None
        v = self.u[self.tu]
        self.tu += 1
        return float(v)


def run_reference(model, make_proposal, theta0, seed, T, runs=None, capture=None):
    """Run the reference Sampler for every chain on the injected stream.

    theta0 (N,d).  ``runs`` = list of (Nsamples, Nburn, Nthin) executed back to
    back on the same Sampler (default one run of T).  Returns dict with the
    stream and, per run, thetas/logpost histories and accept flags."""
    theta0 = np.atleast_2d(np.asarray(theta0, dtype=np.float64))
    N, d = theta0.shape
    runs = runs or [(T, 0, 1)]
    Ttot = sum(r[0] for r in runs)
    st = orc.rng_seed(seed, N)
    xi, u = orc.rng_stream(st, d, Ttot)
    out = {"seed": np.uint64(seed), "theta0": theta0, "xi": xi, "u": u}
    per_run = [dict(thetas=[], logpost=[], accept=[]) for _ in runs]
    real_normal, real_uniform = np.random.normal, np.random.uniform
    try:
        for c in range(N):
            rp = Replay(xi[:, c], u[:, c])
            np.random.normal, np.random.uniform = rp.normal, rp.uniform
            s = Sampler(model, make_proposal(), theta0[c].copy())
            for k, (ns, nb, nt) in enumerate(runs):
                first = s._chain_thetas[-1]
                s.run(ns, nb, nt)
                per_run[k]["thetas"].append(np.array(s._chain_thetas))
                per_run[k]["logpost"].append(np.array(s._chain_logpost, dtype=np.float64))
                if nb == 0 and nt == 1:
                    th = s._chain_thetas
                    assert th[0] is first
                    # rejected steps re-append the same object (sampler.py:86-89)
                    per_run[k]["accept"].append(
                        np.array([th[t + 1] is not th[t] for t in range(ns)], dtype=np.uint8))
            assert rp.tn == Ttot and rp.tu == Ttot
            if capture is not None:
                for key, val in capture(s.proposal).items():
                    out.setdefault(key, []).append(val)
    finally:
        np.random.normal, np.random.uniform = real_normal, real_uniform
    if capture is not None:
        for key in [k for k in out if isinstance(out[k], list)]:
            out[key] = np.array(out[key])
    for k, pr in enumerate(per_run):
        sfx = "" if len(runs) == 1 else "_run%d" % k
        out["thetas" + sfx] = np.stack(pr["thetas"], axis=1)    # (Tk, N, d)
        out["logpost" + sfx] = np.stack(pr["logpost"], axis=1)  # (Tk, N)
        if pr["accept"]:
            out["accept" + sfx] = np.stack(pr["accept"], axis=1)  # (T, N)
    return out


def save(name, **arrs):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrs)
    print("%-28s %8.1f KB" % (name + ".npz", os.path.getsize(path) / 1024))


# -------------------------------------------------- G0: the stream itself --
def g0_rng():
    arrs = {}
    for d in (1, 2, 5, 12, 100):
        st = orc.rng_seed(12345, 3, chain0=7)
        arrs["state0_d%d" % d] = st.copy()
        xi, u = orc.rng_stream(st, d, 3)
        arrs["xi_d%d" % d], arrs["u_d%d" % d], arrs["state3_d%d" % d] = xi, u, st.copy()
    save("g0_rng", seed=np.uint64(12345), chain0=np.uint64(7), **arrs)


def g18_f64_normals():
    """The 64-bit-resolution normal mode (Sampler(..., normals="f64")): the stream itself (as G0) and the
    reference's Sampler on it -- cfg2's shape and a 100-d HMC trajectory -- with every normal drawn from
    one whole stream word."""
    with orc.normal_mode("f64"):
        arrs = {}
        for d in (1, 2, 5, 12, 100):
            st = orc.rng_seed(12345, 3, chain0=7)
            xi, u = orc.rng_stream(st, d, 3)
            arrs["xi_d%d" % d], arrs["u_d%d" % d], arrs["state3_d%d" % d] = xi, u, st.copy()
        save("g18_rng_f64", seed=np.uint64(12345), chain0=np.uint64(7), **arrs)
        Ct = corr100()
        Cp = (2.38 ** 2 / 100.0) * Ct
        th0 = np.random.default_rng(1).standard_normal((6, 100)) @ np.linalg.cholesky(Ct).T
        r = run_reference(rbench.benchmark_gauss100d_corr, lambda: MetropolisRandomWalk(Cp), th0, seed=18, T=32)
        save("g18_gauss100d_rw_f64", mu=np.zeros(100), prop_factor=np.float64(2.38 ** 2 / 100.0), **r)
        m = rbench.benchmark_gauss100d_corr
        r = run_reference(m, lambda: VanillaHMC(0.28, 5, m.grad_log_likelihood), th0[:3], seed=19, T=12)
        save("g18_gauss100d_hmc5_f64", eps=np.float64(0.28), nsteps=np.int64(5), **r)


# ----------------------------------------------- G1/G2: random-walk Gaussian --
def g1_gauss2d_rw():
    """cfg1: benchmark_gauss2d_iso (models/benchmarks.py:15) + MetropolisRandomWalk(I2)."""
    r = run_reference(rbench.benchmark_gauss2d_iso, lambda: MetropolisRandomWalk(np.eye(2)),
                      np.ones((4, 2)), seed=0, T=64)
    save("g1_gauss2d_rw", mu=np.zeros(2), C=np.eye(2), Cprop=np.eye(2), **r)


def corr100():
    return 0.1 * np.eye(100) + 0.9 * np.ones((100, 100))  # models/benchmarks.py:25-26


def g2_gauss100d_rw():
    """cfg2 at fixture size: benchmark_gauss100d_corr + RW with C=(2.38^2/100) C_target."""
    Ct = corr100()
    Cp = (2.38 ** 2 / 100.0) * Ct
    th0 = np.random.default_rng(1).standard_normal((8, 100)) @ np.linalg.cholesky(Ct).T
    r = run_reference(rbench.benchmark_gauss100d_corr, lambda: MetropolisRandomWalk(Cp),
                      th0, seed=1, T=32)
    save("g2_gauss100d_rw", mu=np.zeros(100), prop_factor=np.float64(2.38 ** 2 / 100.0), **r)


# -------------------------------------------------- G3: HMC / MALA, Gaussian --
def g3_gauss_hmc():
    Ct = corr100()
    th0 = np.random.default_rng(3).standard_normal((4, 100)) @ np.linalg.cholesky(Ct).T
    m = rbench.benchmark_gauss100d_corr
    for tag, eps, L in (("mala", 0.25, 1), ("hmc10", 0.28, 10)):
        r = run_reference(m, lambda: VanillaHMC(eps, L, m.grad_log_likelihood), th0,
                          seed=30 + L, T=16)
        save("g3_gauss100d_" + tag, eps=np.float64(eps), nsteps=np.int64(L), **r)
    # the model of examples/test_randomwalk.py:32-37, non-adaptive HMC
    m2 = rbench.benchmark_gauss2d_corr
    r = run_reference(m2, lambda: VanillaHMC(0.6, 5, m2.grad_log_likelihood),
                      np.ones((4, 2)), seed=35, T=48)
    save("g3_gauss2d_hmc5", eps=np.float64(0.6), nsteps=np.int64(5), mu=np.zeros(2),
         C=0.1 * np.eye(2) + 0.9 * np.ones((2, 2)), **r)


# ------------------------------------------ G4: logistic regression plugin --
class LogRegModel(Model):
    """The build's logistic-regression plugin, written against the reference's
    Model ABC (models/model.py:4-64).  NOT in the reference:
    logL = sum y*eta - log(1+exp(eta)), eta = X beta; logP = -|beta|^2/(2 tau2)."""

    def __init__(self, X, y, tau2):
        self.X, self.y, self.tau2 = X, y, tau2

    def log_prior(self, theta):
        return -np.dot(theta, theta) / (2.0 * self.tau2)

    def log_likelihood(self, theta):
        eta = self.X @ theta
        return np.sum(self.y * eta - np.logaddexp(0.0, eta))

    def grad_log_posterior(self, theta):
        eta = self.X @ theta
        sg = np.where(eta >= 0, 1.0 / (1.0 + np.exp(-np.abs(eta))),
                      np.exp(-np.abs(eta)) / (1.0 + np.exp(-np.abs(eta))))
        return self.X.T @ (self.y - sg) - theta / self.tau2


def digest(a):
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), dtype=np.uint8)


def g4_logreg():
    X, y = logreg_data(256, 16, seed=2)
    m = LogRegModel(X, y, 100.0)
    th0 = 0.1 * np.random.default_rng(4).standard_normal((4, 16))
    for tag, eps, L in (("mala", 0.6, 1), ("hmc10", 0.6, 10)):
        r = run_reference(m, lambda: VanillaHMC(eps, L, m.grad_log_posterior), th0,
                          seed=40 + L, T=32)
        save("g4_logreg256x16_" + tag, X=X, y=y, tau2=np.float64(100.0), eps=np.float64(eps),
             nsteps=np.int64(L), **r)
    # full size (cfg3/cfg4 data): data regenerated from the seed
    Xf, yf = logreg_data(10000, 100, seed=2)
    mf = LogRegModel(Xf, yf, 100.0)
    # start near stationarity: 300 MALA sweeps of the oracle from 0 (theta0 is stored)
    om = orc.LogReg(Xf, yf, 100.0)
    th0 = np.zeros((2, 100))
    orc.run(om, orc.HMC(0.1, 1, 100), th0, om.log_posterior(th0), orc.rng_seed(9, 2), 300,
            history=False)
    for tag, eps, L in (("mala", 0.16, 1), ("hmc10", 0.15, 10)):
        r = run_reference(mf, lambda: VanillaHMC(eps, L, mf.grad_log_posterior), th0,
                          seed=50 + L, T=6)
        r.pop("xi")
        save("g4_logreg10000x100_" + tag, data_seed=np.int64(2), n=np.int64(10000),
             d=np.int64(100), X_sha256=digest(Xf), y_sha256=digest(yf), tau2=np.float64(100.0),
             eps=np.float64(eps), nsteps=np.int64(L), **r)


# ----------------------------------------------------------- G5: changepoint --
def import_changepoint():
    """riemann/models/changepoint.py needs `autograd` (absent) and does
    `from ..riemann import Model` (module does not exist): SURVEY.md 8c.
    Container-local stand-ins for those two IMPORTS only; every line of the
    model's arithmetic that runs is the reference's own."""
    ag = types.ModuleType("autograd")
    ag.numpy = np
    ag.jacobian = lambda f: (lambda p: (_ for _ in ()).throw(NotImplementedError()))
    sys.modules["autograd"] = ag
    sys.modules["autograd.numpy"] = np
    sys.modules["riemann.riemann"] = riemann
    from riemann.models import changepoint as cp
    return cp


class PackedChangepoint(Model):
    """Adapter: a packed vector (cpx[k], cpv[k+1], sig) in front of the
    reference's ChangepointRegression1D (theta there is a ChangepointParams)."""

    def __init__(self, cp, model, k):
        self.cp, self.m, self.k = cp, model, k

    def unpack(self, v):
        k = self.k
        return self.cp.ChangepointParams(v[:k], v[k:2 * k + 1], v[2 * k + 1])

    def log_prior(self, v):
        return self.m.log_prior(self.unpack(v))

    def log_likelihood(self, v):
        return self.m.log_likelihood(self.unpack(v))


def scaled_rw(Cdiag, scale):
    p = MetropolisRandomWalk(np.diag(Cdiag))
    p.scale = scale  # public attribute, proposals/randomwalk.py:18
    return p


def changepoint_setup(cp, k=5):
    """riemann/test_changepoint.py:137-150,167 on the seeded cfg5 data."""
    x, y, truth, Cdiag = changepoint_data(seed=3, k=k)
    model = cp.ChangepointRegression1D(x, y, 1.0, 3.0, 1.0 * k, 2 * k, 1, 1)
    return model, x, y, truth, Cdiag


def g5_changepoint():
    cp = import_changepoint()
    k = 5
    model, x, y, truth, Cdiag = changepoint_setup(cp)
    pm = PackedChangepoint(cp, model, k)
    g = np.random.default_rng(55)
    th = truth + 0.05 * g.standard_normal((64, 2 * k + 2))
    th[:, :k] = np.sort(th[:, :k], axis=1)
    th[40:44, 0], th[40:44, 1] = th[40:44, 1].copy(), th[40:44, 0].copy()  # unsorted cpx
    th[44:48, k + 2] = [-0.3, 0.0, -1e-9, -2.0]                           # cpv <= 0
    th[48:52, -1] = [-0.1, -1.0, 0.0, -1e-12]                             # sig <= 0
    th[52, 0] = 0.5                                                       # cpx < xmin
    th[53, k - 1] = 3.5                                                   # cpx > xmax
    th[54, 1] = th[54, 0]                                                 # coincident changepoints
    logL = np.array([pm.log_likelihood(v) for v in th], dtype=np.float64)
    logP = np.array([pm.log_prior(v) for v in th], dtype=np.float64)
    logpost = np.array([pm.log_posterior(v) for v in th], dtype=np.float64)
    idx = np.array([np.searchsorted(v[:k], x) for v in th], dtype=np.int32)
    common = dict(x=x, y=y, k=np.int64(k), xmin=np.float64(1.0), xmax=np.float64(3.0),
                  lamb=np.float64(k), alpha=np.float64(1.0), beta=np.float64(1.0))
    save("g5_changepoint_eval", theta=th, logL=logL, logP=logP, logpost=logpost, idx=idx,
         truth=truth, **common)
    # alpha != 1 exercises (alpha-1)*log(cpv)
    model2 = cp.ChangepointRegression1D(x, y, 1.0, 3.0, 4.0, 2 * k, 2.5, 1.5)
    pm2 = PackedChangepoint(cp, model2, k)
    save("g5_changepoint_eval_a25", theta=th,
         logL=np.array([pm2.log_likelihood(v) for v in th], dtype=np.float64),
         logP=np.array([pm2.log_prior(v) for v in th], dtype=np.float64),
         logpost=np.array([pm2.log_posterior(v) for v in th], dtype=np.float64),
         **dict(common, lamb=np.float64(4.0), alpha=np.float64(2.5), beta=np.float64(1.5)))
    th0 = truth + 0.01 * g.standard_normal((4, 2 * k + 2))
    th0[:, :k] = np.sort(th0[:, :k], axis=1)
    r = run_reference(pm, lambda: scaled_rw(Cdiag, 0.1), th0, seed=5, T=64)
    save("g5_changepoint_rw", Cdiag=Cdiag, scale=np.float64(0.1), **common, **r)
    return cp, pm, Cdiag, th0, common


# ------------------------------------------------------------ G6: edge cases --
def g6_edges(cp, pm, Cdiag, th0, common):
    # (a) chain started where logpost = -inf: D = NaN -> min(0, NaN) = 0 -> accepts
    #     every move until it lands on a finite state (sampler.py:83-84)
    bad = th0.copy()
    bad[:, -1] = -0.05  # sig < 0 -> log_prior nan -> -inf
    bad[2:, -1] = 0.0005  # sig tiny positive: next proposals often cross zero
    r = run_reference(pm, lambda: scaled_rw(Cdiag, 0.1), bad, seed=6, T=48)
    save("g6_neginf_start", Cdiag=Cdiag, scale=np.float64(0.1), **common, **r)
    # (b) Nburn/Nthin slicing and run() continuation (sampler.py:49-54):
    #     run(10,4,3) then run(5) on the same Sampler
    r = run_reference(rbench.benchmark_gauss2d_iso, lambda: MetropolisRandomWalk(np.eye(2)),
                      np.ones((3, 2)), seed=61, T=None, runs=[(10, 4, 3), (5, 0, 1)])
    save("g6_burn_thin_resume", mu=np.zeros(2), C=np.eye(2), Cprop=np.eye(2),
         runs=np.array([[10, 4, 3], [5, 0, 1]]), **r)


# -------------------------------------------------------- G7: error messages --
def g7_errors():
    msgs = {}
    try:
        MetropolisRandomWalk(np.eye(3)).propose(np.ones(2))
    except ParameterError as e:
        msgs["rw_shape"] = str(e)
    try:
        MultiGaussianDist(np.zeros(2), np.ones((2, 3)))
    except ParameterError as e:
        msgs["gauss_nonsquare"] = str(e)
    try:
        MultiGaussianDist(np.zeros(3), np.eye(2))
    except ParameterError as e:
        msgs["gauss_mismatch"] = str(e)
    save("g7_errors", **{k: np.array(v) for k, v in msgs.items()})


# ------------------------------------------- G8: scale adaptation --
def g8_adapt_scale():
    """AdaptScaleRandomWalk (proposals/randomwalk.py:29-38) and AdaptScaleHMC
    (proposals/hamiltonian.py:94-104): the adapt() hook of sampler.py:88."""
    from riemann.proposals.hamiltonian import AdaptScaleHMC
    from riemann.proposals.randomwalk import AdaptScaleRandomWalk

    def cap(p):
        return dict(ad_n=p.Nsamples, ad_acc=p.Naccepts, ad_scale=p.scale, ad_rate=p.accept_rate)

    m2 = rbench.benchmark_gauss2d_corr
    r = run_reference(m2, lambda: AdaptScaleRandomWalk(np.eye(2)), np.ones((4, 2)), seed=80, T=200,
                      capture=cap)
    save("g8_adapt_rw_gauss2d", mu=np.zeros(2), C=0.1 * np.eye(2) + 0.9 * np.ones((2, 2)),
         Cprop=np.eye(2), target=np.float64(0.25), **r)
    Ct = corr100()
    th0 = np.random.default_rng(8).standard_normal((4, 100)) @ np.linalg.cholesky(Ct).T
    m = rbench.benchmark_gauss100d_corr
    r = run_reference(m, lambda: AdaptScaleRandomWalk(0.05 * Ct), th0, seed=81, T=64, capture=cap)
    r.pop("xi")
    save("g8_adapt_rw_gauss100d", prop_factor=np.float64(0.05), target=np.float64(0.25), **r)
    r = run_reference(m, lambda: AdaptScaleHMC(0.2, 5, m.grad_log_likelihood), th0, seed=82, T=40,
                      capture=cap)
    r.pop("xi")
    save("g8_adapt_hmc_gauss100d", eps0=np.float64(0.2), nsteps=np.int64(5),
         target=np.float64(0.75), **r)


# ---------------------------------------------- G9: HMC with a mass matrix M --
def mass_matrix(d, seed):
    g = np.random.default_rng(seed)
    v = g.standard_normal(d) / np.sqrt(d)
    return np.diag(np.linspace(0.5, 2.0, d)) + 0.8 * np.outer(v, v)


def g9_hmc_mass():
    """VanillaHMC(eps, Nsteps, grad, M): hamiltonian.py:71-74, 80-81, 29, 36, 85-87."""
    Ct = corr100()
    m = rbench.benchmark_gauss100d_corr
    th0 = np.random.default_rng(9).standard_normal((4, 100)) @ np.linalg.cholesky(Ct).T
    M = mass_matrix(100, 90)
    r = run_reference(m, lambda: VanillaHMC(0.2, 4, m.grad_log_likelihood, M=M), th0, seed=91, T=16)
    r.pop("xi")
    save("g9_gauss100d_hmc_mass", M=M, eps=np.float64(0.2), nsteps=np.int64(4), **r)
    X, y = logreg_data(256, 16, seed=2)
    lm = LogRegModel(X, y, 100.0)
    M2 = 0.25 * X.T @ X + np.eye(16) / 100.0
    th0 = 0.1 * np.random.default_rng(4).standard_normal((4, 16))
    r = run_reference(lm, lambda: VanillaHMC(0.5, 6, lm.grad_log_posterior, M=M2), th0, seed=92, T=32)
    save("g9_logreg256x16_hmc_mass", X=X, y=y, tau2=np.float64(100.0), M=M2, eps=np.float64(0.5),
         nsteps=np.int64(6), **r)


# ----------------------------------------------------- G10: pCN --
def g10_pcn():
    """pCN(C, rho), proposals/randomwalk.py:78-101: asymmetric logqratio."""
    from riemann.proposals.randomwalk import pCN
    Ct = corr100()
    m = rbench.benchmark_gauss100d_corr
    th0 = np.random.default_rng(10).standard_normal((4, 100)) @ np.linalg.cholesky(Ct).T
    r = run_reference(m, lambda: pCN(0.7 * Ct + 0.3 * np.eye(100), 0.995), th0, seed=100, T=32)
    r.pop("xi")
    save("g10_pcn_gauss100d", rho=np.float64(0.995), **r)
    m2 = rbench.benchmark_gauss2d_corr
    r = run_reference(m2, lambda: pCN(np.eye(2), 0.8), np.ones((4, 2)), seed=101, T=64)
    save("g10_pcn_gauss2d", mu=np.zeros(2), C=0.1 * np.eye(2) + 0.9 * np.ones((2, 2)), Cprop=np.eye(2),
         rho=np.float64(0.8), **r)
    # AdaptScalepCN (randomwalk.py:104-119): rho <- tanh(rho/scale) recursively, rho_c stale
    from riemann.proposals.randomwalk import AdaptScalepCN
    r = run_reference(m2, lambda: AdaptScalepCN(np.eye(2), 0.8), np.ones((4, 2)), seed=102, T=64,
                      capture=lambda p: {"rho_final": np.float64(p.rho), "scale_final": np.float64(p.scale)})
    save("g10_adaptscalepcn_gauss2d", mu=np.zeros(2), C=0.1 * np.eye(2) + 0.9 * np.ones((2, 2)),
         Cprop=np.eye(2), rho=np.float64(0.8), **r)


def g11_mixture():
    """benchmarks.MixtureDist (models/benchmarks.py:31-85): density values and a
    random-walk run on the two mixture benchmarks."""
    rng = np.random.default_rng(11)
    out = {}
    for name, m, d in (("mix1d", rbench.benchmark_gauss1d_mix2, 1),
                       ("mix2d", rbench.benchmark_gauss2d_mix2_cross, 2)):
        pts = np.concatenate([rng.standard_normal((40, d)) * 4.0,
                              rng.standard_normal((8, d)) * 60.0])   # far tail: exp underflows -> -inf
        with np.errstate(divide="ignore"):
            out[name + "_theta"] = pts
            out[name + "_logL"] = np.array([m.log_likelihood(t) for t in pts])
            out[name + "_logP"] = np.array([m.log_prior(t) for t in pts])
            out[name + "_logpost"] = np.array([m.log_posterior(t) for t in pts])
        out[name + "_weights"] = np.array(m._mixweights)
    save("g11_mixture_values", **out)
    m = rbench.benchmark_gauss2d_mix2_cross
    th0 = rng.standard_normal((4, 2))
    r = run_reference(m, lambda: MetropolisRandomWalk(1.5 * np.eye(2)), th0, seed=110, T=96)
    save("g11_mixture_rw", Cprop=1.5 * np.eye(2), **r)
    # HMC / MALA on the mixture: the reference's MixtureDist has no gradient, so the callable the
    # reference's VanillaHMC is given here is the build's own statement of it (as for the logistic
    # plugin): grad log sum_k w_k N_k = sum_k w_k N_k grad(log N_k) / sum_k w_k N_k
    def mix_grad(th):
        e = np.array([w * np.exp(c.log_likelihood(th)) for w, c in zip(m._mixweights, m._model_list)])
        gk = np.array([c.grad_log_likelihood(th) for c in m._model_list])
        return (e[:, None] * gk).sum(axis=0) / e.sum()
    for tag, eps, L in (("mala", 0.5, 1), ("hmc4", 0.3, 4)):
        r = run_reference(m, lambda: VanillaHMC(eps, L, mix_grad), th0, seed=112 + L, T=64)
        save("g11_mixture_" + tag, eps=np.float64(eps), nsteps=np.int64(L), **r)
    m1 = rbench.benchmark_gauss1d_mix2
    r = run_reference(m1, lambda: MetropolisRandomWalk(np.array([[9.0]])), np.array([[5.0], [-5.0], [0.0], [1.0]]),
                      seed=111, T=96)
    save("g11_mixture1d_rw", Cprop=np.array([[9.0]]), **r)


class PTReplay:
    """Feeds the draws of one parallel-tempering ensemble to the reference's PTSampler.
    The call sites tell the draws apart: the selector `u` (ptsampler.py:103) opens a
    rung, the swap uniform is the other draw made from PTSampler.sample (:119), the
    accept uniform comes from Sampler.sample (sampler.py:84), the normals from the
    proposal (randomwalk.py:25)."""

    def __init__(self, usel, vswp, xi, ua):
        self.usel, self.vswp, self.xi, self.ua = usel, vswp, xi, ua   # (T,K), (T,K), (T,K,d), (T,K)
        self.t, self.rung = -1, -1
        self.swaps = np.zeros(usel.shape, dtype=np.uint8)

    def next_sweep(self):
        self.t += 1
        self.rung = -1

    def normal(self, loc=0.0, scale=1.0, size=None):
        v = self.xi[self.t, self.rung]
        shape = (size,) if np.isscalar(size) else tuple(size)
        assert v.shape == shape, (v.shape, shape)
        return v.copy()

    def uniform(self, low=0.0, high=1.0, size=None):
        import linecache
        assert size is None
        fr = sys._getframe(1)
        fn, line = fr.f_code.co_filename, linecache.getline(fr.f_code.co_filename, fr.f_lineno)
        if fn.endswith("ptsampler.py"):
            if line.strip().startswith("u = "):
                self.rung += 1
                return float(self.usel[self.t, self.rung])
            assert "mhratio" in line, line
            self.swaps[self.t, self.rung] = 1
            return float(self.vswp[self.t, self.rung])
        assert fn.endswith("sampler.py"), fn
        return float(self.ua[self.t, self.rung])


def run_reference_pt(model, make_proposal, theta0, seed, T, pswap):
    """The reference's PTSampler (default ladder betas = 0.5**arange(5)) for every
    ensemble, on the slot layout of oracle/rmh_oracle.c section 5."""
    import contextlib
    import io
    from riemann.samplers.ptsampler import PTSampler
    theta0 = np.atleast_2d(np.asarray(theta0, dtype=np.float64))
    Ne, d = theta0.shape
    K = 5
    S, idx = orc.pt_slots(Ne, K)
    st = orc.rng_seed(seed, S)
    usel, vswp, xi, ua = orc.rng_pt_stream(st, d, T)
    thetas = np.zeros((T + 1, S, d))
    logpost = np.zeros((T + 1, S))
    within_acc = np.zeros((T, S), dtype=np.uint8)
    swap_prop = np.zeros((T, S), dtype=np.uint8)
    swap_acc = np.zeros((T, S), dtype=np.int8)      # -1: cannot be told (identical objects)
    real_normal, real_uniform = np.random.normal, np.random.uniform
    try:
        for e in range(Ne):
            sl = idx[e]
            rp = PTReplay(usel[:, sl], vswp[:, sl], xi[:, sl], ua[:, sl])
            np.random.normal, np.random.uniform = rp.normal, rp.uniform
            with contextlib.redirect_stdout(io.StringIO()):
                pt = PTSampler(model, make_proposal(), theta0[e].copy(), Pswap=pswap)
            betas = np.array(pt.betas, dtype=np.float64)
            for i, smp in enumerate(pt.samplers):
                thetas[0, sl[i]] = smp._chain_thetas[-1]
                logpost[0, sl[i]] = smp._chain_logpost[-1]
            for t in range(T):
                before = [smp._chain_thetas[-1] for smp in pt.samplers]
                rp.next_sweep()
                pt.sample()
                assert rp.rung == K - 1
                for i, smp in enumerate(pt.samplers):
                    assert len(smp._chain_thetas) == t + 2          # one state per sweep and rung
                    thetas[t + 1, sl[i]] = smp._chain_thetas[-1]
                    logpost[t + 1, sl[i]] = smp._chain_logpost[-1]
                for i in range(K):
                    if rp.swaps[t, i]:
                        swap_prop[t, sl[i]] = swap_prop[t, sl[i + 1]] = 1
                        if before[i] is before[i + 1]:
                            ok = -1
                        else:
                            ok = int(pt.samplers[i]._chain_thetas[-1] is before[i + 1])
                        swap_acc[t, sl[i]] = swap_acc[t, sl[i + 1]] = ok
                for i, smp in enumerate(pt.samplers):
                    if not swap_prop[t, sl[i]]:
                        within_acc[t, sl[i]] = smp._chain_thetas[-1] is not before[i]
    finally:
        np.random.normal, np.random.uniform = real_normal, real_uniform
    return dict(seed=np.uint64(seed), theta0=theta0, betas=betas, pswap=np.float64(pswap),
                slots=idx, thetas=thetas, logpost=logpost, within_acc=within_acc,
                swap_prop=swap_prop, swap_acc=swap_acc, usel=usel, vswp=vswp, ua=ua)


def g12_pt():
    """PTSampler (samplers/ptsampler.py:42-127), default ladder, on a Gaussian and on the
    crossed 2-d mixture; Pswap large enough that swaps are frequent."""
    rng = np.random.default_rng(12)
    m = rbench.benchmark_gauss2d_corr
    r = run_reference_pt(m, lambda: MetropolisRandomWalk(0.5 * np.eye(2)),
                         rng.standard_normal((4, 2)), seed=120, T=96, pswap=0.3)
    save("g12_pt_gauss2d", mu=np.zeros(2), C=0.1 * np.eye(2) + 0.9 * np.ones((2, 2)),
         Cprop=0.5 * np.eye(2), **r)
    # HMC within the rungs: the one proposal object serves every rung (ptsampler.py:82), so the
    # trajectories use the BASE model's gradient and only the accept test is tempered
    r = run_reference_pt(m, lambda: VanillaHMC(0.35, 3, m.grad_log_likelihood),
                         rng.standard_normal((3, 2)), seed=122, T=96, pswap=0.3)
    save("g12_pt_hmc_gauss2d", mu=np.zeros(2), C=0.1 * np.eye(2) + 0.9 * np.ones((2, 2)),
         eps=np.float64(0.35), nsteps=np.int64(3), **r)
    mx = rbench.benchmark_gauss2d_mix2_cross
    r = run_reference_pt(mx, lambda: MetropolisRandomWalk(1.5 * np.eye(2)),
                         rng.standard_normal((3, 2)) * 2.0, seed=121, T=96, pswap=0.1)
    save("g12_pt_mix2d", Cprop=1.5 * np.eye(2), **r)


class RJReplay:
    """Feeds one chain's draws to the reference's ChangepointRegression1DProp WITH its
    trans-dimensional branch (riemann/test_changepoint.py:43-73): inside ``propose`` the
    argument-less uniforms are U1, U2, U3 (block selectors) and U4 (birth/death, :58), the
    two uniforms with bounds are the new location (:59) and the height split (:60), randint
    is the changepoint to remove (:66); the block's normals are the first k (k+1) of its
    padded block; Sampler.sample's uniform is the accept draw and closes the iteration."""

    def __init__(self, xi, w1, w2, KM):
        self.xi, self.w1, self.w2, self.KM = xi, w1, w2, KM      # (T,Dn), (T,4), (T,4)
        self.t, self.nsel, self.nbnd = 0, 0, 0
        self.move = np.full(len(w1), -1, dtype=np.int8)

    def normal(self, loc=0.0, scale=1.0, size=None):
        shape = (size,) if np.isscalar(size) else tuple(size)
        b, KM = self.nsel - 1, self.KM
        lo = {0: 0, 1: KM, 2: 2 * KM + 1}[b]
        self.move[self.t] = b
        return self.xi[self.t, lo:lo + shape[0]].copy()

    def uniform(self, low=None, high=None, size=None):
        assert size is None
        fr = sys._getframe(1)
        if fr.f_code.co_name == "propose":
            if low is None:
                self.nsel += 1
                return float(self.w1[self.t, self.nsel]) if self.nsel <= 3 else float(self.w2[self.t, 0])
            self.nbnd += 1
            self.move[self.t] = 3
            return low + (high - low) * float(self.w2[self.t, self.nbnd])
        assert fr.f_code.co_filename.endswith("sampler.py"), fr.f_code.co_filename
        v = float(self.w1[self.t, 0])
        self.t += 1
        self.nsel = self.nbnd = 0
        return v

    def randint(self, n):
        self.move[self.t] = 4
        return min(int(float(self.w2[self.t, 3]) * n), n - 1)


def g17_changepoint_rj():
    """The reference's trans-dimensional changepoint sampler: ChangepointRegression1DProp
    (riemann/test_changepoint.py:18-73) with its birth/death branch, on
    ChangepointRegression1D.add_changepoint / subtract_changepoint
    (models/changepoint.py:193-240).  P_cumprop_sig = 0 keeps the sig block out (it raises
    under numpy >= 1.24, see g13): cpx 0.2, cpv 0.32, birth/death 0.48.  The autograd
    stand-in differentiates the height map by the complex-step method (exact to rounding,
    like the automatic differentiation it replaces)."""
    import contextlib
    import importlib.util
    import io
    cp = import_changepoint()

    def jacobian(f):
        def jac(p):
            p = np.asarray(p, dtype=np.float64)
            cols = []
            for j in range(len(p)):
                z = p.astype(np.complex128)
                z[j] += 1e-30j
                cols.append(np.imag(f(z)) / 1e-30)
            return np.array(cols).T
        return jac
    cp.jacobian = jacobian
    import riemann.models as rmodels
    import riemann.proposals as rprops
    rmodels.ChangepointParams, rmodels.ChangepointRegression1D = cp.ChangepointParams, cp.ChangepointRegression1D
    rprops.MetropolisRandomWalk = MetropolisRandomWalk
    spec = importlib.util.spec_from_file_location("riemann_ref_test_changepoint_rj",
                                                  "/root/reference/riemann/test_changepoint.py")
    tc = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tc)
    k0, KM = 5, 10
    model, x, y, truth, Cdiag = changepoint_setup(cp)
    hscale = 2.0
    Dn = 2 * KM + 2
    g = np.random.default_rng(17)
    N, T = 4, 300
    starts = []
    for c in range(N):
        kc = [5, 1, 0, 8][c]
        cpx = np.sort(g.uniform(1.1, 2.9, size=kc))
        starts.append((cpx, g.uniform(1.0, 3.0, size=kc + 1), 0.5))
    th0 = orc.cprj_pack(starts, KM)
    st = orc.rng_seed(170, N)
    xi = np.empty((T, N, Dn)); w1 = np.empty((T, N, 4)); w2 = np.empty((T, N, 4))
    for t in range(T):
        for c in range(N):
            s1 = st[c:c + 1]
            xi[t, c] = orc.rng_normals(s1, Dn)
            w1[t, c] = orc.rng_words(s1)
            w2[t, c] = orc.rng_words(s1)
    thetas = np.zeros((T + 1, N, 2 * KM + 3)); logpost = np.zeros((T + 1, N))
    accept = np.zeros((T, N), dtype=np.uint8); move = np.zeros((T, N), dtype=np.int8)
    real = np.random.normal, np.random.uniform, np.random.randint
    try:
        for c in range(N):
            rp = RJReplay(xi[:, c], w1[:, c], w2[:, c], KM)
            np.random.normal, np.random.uniform, np.random.randint = rp.normal, rp.uniform, rp.randint
            prop = tc.ChangepointRegression1DProp(model, hscale)
            prop.P_cumprop_sig = 0.0
            t0 = cp.ChangepointParams(*starts[c])
            s = Sampler(model, prop, t0)
            with contextlib.redirect_stdout(io.StringIO()), np.errstate(all="ignore"):
                s.run(T)
            assert rp.t == T
            for e, (tt, lp) in enumerate(zip(s._chain_thetas, s._chain_logpost)):
                thetas[e, c] = orc.cprj_pack([(np.ravel(tt.cpx), np.ravel(tt.cpv), float(tt.sig))], KM)[0]
                logpost[e, c] = lp
            accept[:, c] = [s._chain_thetas[t + 1] is not s._chain_thetas[t] for t in range(T)]
            move[:, c] = rp.move
    finally:
        np.random.normal, np.random.uniform, np.random.randint = real
    ks = thetas[:, :, -1]
    assert ks.max() <= KM and set(np.unique(move)) == {0, 1, 3, 4}
    assert (accept[move == 3] == 1).sum() > 3 and (accept[move == 4] == 1).sum() > 3, "no accepted births/deaths"
    save("g17_changepoint_rj", x=x, y=y, KM=np.int64(KM), xmin=np.float64(1.0), xmax=np.float64(3.0),
         lamb=np.float64(k0), alpha=np.float64(1.0), beta=np.float64(1.0), hscale=np.float64(hscale),
         pcum=np.array([0.2, 0.4, 0.0]), seed=np.uint64(170), theta0=th0, thetas=thetas, logpost=logpost,
         accept=accept, move=move)


def g14_adapt_cov():
    """AdaptCovRandomWalk (Haario et al. 2001), proposals/adaptive.py:38-103 with
    randomwalk.py:41-56: strict, smooth_adapt and marginalize variants, one chain each
    (the reference adapts per chain; the build pools over chains, identical for N = 1).
    adaptive.py:81 calls ``np.float``, an alias numpy removed in 1.24: it is restored for
    this run (``np.float = float``, what the alias always was)."""
    from riemann.proposals.randomwalk import AdaptCovRandomWalk
    had = hasattr(np, "float")
    if not had:
        np.float = float
    try:
        m2 = rbench.benchmark_gauss2d_corr
        C0 = np.array([[0.5, 0.1], [0.1, 0.3]])
        T = 150
        for tag, kw in (("strict", {}), ("smooth", dict(t_adapt=20, smooth_adapt=True)),
                        ("marg", dict(marginalize=True))):
            outs = []
            for c in range(3):      # three independent single-chain runs, seeds 140+c
                r = run_reference(m2, lambda: AdaptCovRandomWalk(C0.copy(), **kw), np.ones((1, 2)) * (c + 1),
                                  seed=140 + c, T=T,
                                  capture=lambda p: {"C_final": np.array(p.C), "L_final": np.array(p.L)})
                outs.append(r)
            save("g14_adaptcov_%s" % tag, C0=C0, mu=np.zeros(2), C=0.1 * np.eye(2) + 0.9 * np.ones((2, 2)),
                 t_adapt=np.float64(kw.get("t_adapt", 1)), smooth=np.bool_(kw.get("smooth_adapt", False)),
                 marginalize=np.bool_(kw.get("marginalize", False)),
                 seeds=np.array([140, 141, 142], dtype=np.uint64),
                 theta0=np.stack([o["theta0"][0] for o in outs]),
                 thetas=np.stack([o["thetas"][:, 0] for o in outs], axis=1),
                 logpost=np.stack([o["logpost"][:, 0] for o in outs], axis=1),
                 accept=np.stack([o["accept"][:, 0] for o in outs], axis=1),
                 C_final=np.stack([o["C_final"][0] for o in outs]),
                 L_final=np.stack([o["L_final"][0] for o in outs]))
        # 10-d: the first updates (n = 4, 9) see a rank-deficient sample covariance + 1e-12 ridge
        d = 10
        g = np.random.default_rng(14)
        A = g.standard_normal((d, d))
        Ct = A @ A.T / d + 0.3 * np.eye(d)
        m10 = MultiGaussianDist(np.zeros(d), Ct)
        r = run_reference(m10, lambda: AdaptCovRandomWalk(0.1 * np.eye(d), t_adapt=30, smooth_adapt=True),
                          g.standard_normal((1, d)), seed=145, T=120,
                          capture=lambda p: {"C_final": np.array(p.C), "L_final": np.array(p.L)})
        r.pop("xi")
        save("g14_adaptcov_10d", Ct=Ct, C0=0.1 * np.eye(d), t_adapt=np.float64(30), **r)
        # AdaptScaleCovRandomWalk (randomwalk.py:62-75): scale and covariance adapted together
        from riemann.proposals.randomwalk import AdaptScaleCovRandomWalk
        r = run_reference(m10, lambda: AdaptScaleCovRandomWalk(0.1 * np.eye(d), t_adapt=30, smooth_adapt=True),
                          g.standard_normal((1, d)), seed=146, T=120,
                          capture=lambda p: {"C_final": np.array(p.C), "L_final": np.array(p.L),
                                             "scale_final": np.float64(p.scale),
                                             "Naccepts": np.int64(p.Naccepts)})
        r.pop("xi")
        save("g14_adaptscalecov_10d", Ct=Ct, C0=0.1 * np.eye(d), t_adapt=np.float64(30), **r)
        # AdaptCovHMC / AdaptScaleCovHMC (hamiltonian.py:107-134): the adapted covariance is the mass matrix
        from riemann.proposals.hamiltonian import AdaptCovHMC, AdaptScaleCovHMC
        M0 = 0.5 * np.eye(d)
        cap = lambda p: {"C_final": np.array(p.C), "L_final": np.array(p.L)}        # noqa: E731
        r = run_reference(m10, lambda: AdaptCovHMC(0.25, 4, m10.grad_log_likelihood, M0, t_adapt=40,
                                                   smooth_adapt=True),
                          g.standard_normal((2, d)), seed=147, T=60, capture=cap)
        r.pop("xi")
        save("g14_adaptcovhmc_10d", Ct=Ct, M0=M0, eps=np.float64(0.25), nsteps=np.int64(4),
             t_adapt=np.float64(40), **r)
        cap2 = lambda p: {"C_final": np.array(p.C), "L_final": np.array(p.L),       # noqa: E731
                          "scale_final": np.float64(p.scale)}
        r = run_reference(m10, lambda: AdaptScaleCovHMC(0.25, 3, m10.grad_log_likelihood, M0, t_adapt=40,
                                                        smooth_adapt=True),
                          g.standard_normal((2, d)), seed=148, T=60, capture=cap2)
        r.pop("xi")
        save("g14_adaptscalecovhmc_10d", Ct=Ct, M0=M0, eps=np.float64(0.25), nsteps=np.int64(3),
             t_adapt=np.float64(40), **r)
    finally:
        if not had:
            del np.float


def g15_gauss_big_d():
    """MultiGaussianDist + MetropolisRandomWalk beyond the register-tile size (d = 200):
    the workgroup-per-16-chains kernels of rmh_kernels_big.hip."""
    d = 200
    g = np.random.default_rng(15)
    A = g.standard_normal((d, d))
    Ct = A @ A.T / d + 0.2 * np.eye(d)
    mu = g.standard_normal(d)
    m = MultiGaussianDist(mu, Ct)
    Cp = (2.38 ** 2 / d) * Ct
    th0 = mu + g.standard_normal((3, d)) @ np.linalg.cholesky(Ct).T
    r = run_reference(m, lambda: MetropolisRandomWalk(Cp), th0, seed=150, T=24)
    r.pop("xi")
    save("g15_gauss200d_rw", rng_seed_for_model=np.int64(15), **r)


def g16_reference_example():
    """The run of the reference's own smoke script, examples/test_randomwalk.py:34-40:
    AdaptScaleCovHMC(0.1, 5, grad, I, t_adapt=100, smooth_adapt=True) with scale = 1.0 on
    benchmark_gauss2d_corr from theta0 = ones(2), run(10000, 1000, 1) -- here on the injected
    stream (the script itself is unseeded)."""
    from riemann.proposals.hamiltonian import AdaptScaleCovHMC
    m2 = rbench.benchmark_gauss2d_corr

    def make():
        p = AdaptScaleCovHMC(0.1, 5, m2.grad_log_likelihood, np.eye(2), t_adapt=100, smooth_adapt=True)
        p.scale = 1.0
        return p
    r = run_reference(m2, make, np.ones((1, 2)), seed=160, T=10000, runs=[(10000, 1000, 1)],
                      capture=lambda p: {"scale_final": np.float64(p.scale), "Naccepts": np.int64(p.Naccepts)})
    r.pop("xi")
    r.pop("u")
    save("g16_example_test_randomwalk", mu=np.zeros(2), C=0.1 * np.eye(2) + 0.9 * np.ones((2, 2)), **r)


class BlockReplay:
    """Feeds one chain's draws to the reference's ChangepointRegression1DProp + Sampler:
    the uniforms asked for inside ``propose`` (riemann/test_changepoint.py:47-51) are the
    block selectors u_0, u_1, u_2 of the iteration, in order; the block's normals are the
    iteration's normals of that block's coordinates; the uniform asked for by
    Sampler.sample (sampler.py:84) is the accept uniform and closes the iteration."""

    def __init__(self, xi, usel, ua, ends):
        self.xi, self.usel, self.ua, self.ends = xi, usel, ua, ends   # (T,d), (T,3), (T,), block ends
        self.t, self.nsel = 0, 0
        self.picked = np.full(len(ua), -1, dtype=np.int8)

    def normal(self, loc=0.0, scale=1.0, size=None):
        b = self.nsel - 1
        lo, hi = (0 if b == 0 else self.ends[b - 1]), self.ends[b]
        v = self.xi[self.t, lo:hi]
        shape = (size,) if np.isscalar(size) else tuple(size)
        assert v.shape == shape, (v.shape, shape, b)
        self.picked[self.t] = b
        return v.copy()

    def uniform(self, low=0.0, high=1.0, size=None):
        assert size is None
        fr = sys._getframe(1)
        if fr.f_code.co_name == "propose":
            assert fr.f_code.co_filename.endswith("test_changepoint.py") and (low, high) == (0.0, 1.0)
            self.nsel += 1
            return float(self.usel[self.t, self.nsel - 1])
        assert fr.f_code.co_filename.endswith("sampler.py"), fr.f_code.co_filename
        v = float(self.ua[self.t])
        self.t += 1
        self.nsel = 0
        return v


def g13_changepoint_block_proposal():
    """a-11 variant (B): the reference's own ChangepointRegression1DProp
    (riemann/test_changepoint.py:18-73) driving its Sampler at fixed k.  P_cumprop_cpv is
    set to 1.0 (a public attribute, :30): blocks are picked with P(cpx) = 0.2, P(cpv) = 0.8.
    The sig block cannot be recorded: its MetropolisRandomWalk returns a (1,)-array for the
    0-d sig (randomwalk.py:22,26), after which Model.log_posterior's np.any([...])
    (model.py:50) raises "inhomogeneous shape" under numpy >= 1.24 -- an ordinary failure of
    the reference in this environment; that block is covered by the oracle, whose block
    logic this fixture pins.  The trans-dimensional branch (:56-71) does not exist at fixed k."""
    import importlib.util
    cp = import_changepoint()
    import riemann.models as rmodels
    import riemann.proposals as rprops
    # the script's `from riemann.models import ...` / `from riemann.proposals import ...`
    # (test_changepoint.py:14-15) expect names the (empty) package __init__ files never bind
    rmodels.ChangepointParams, rmodels.ChangepointRegression1D = cp.ChangepointParams, cp.ChangepointRegression1D
    rprops.MetropolisRandomWalk = MetropolisRandomWalk
    spec = importlib.util.spec_from_file_location("riemann_ref_test_changepoint",
                                                  "/root/reference/riemann/test_changepoint.py")
    tc = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tc)
    k = 5
    model, x, y, truth, Cdiag = changepoint_setup(cp)
    hscale = 2.0                                   # hmax - hmin, test_changepoint.py:148
    d = 2 * k + 2
    ends = [k, 2 * k + 1, d]
    g = np.random.default_rng(13)
    N, T = 4, 160
    th0 = truth + 0.01 * g.standard_normal((N, d))
    th0[:, :k] = np.sort(th0[:, :k], axis=1)
    st = orc.rng_seed(130, N)
    xi = np.empty((T, N, d)); usel = np.empty((T, N, 3)); ua = np.empty((T, N))
    for t in range(T):                             # normals, then one word per sub-stream
        x1, _ = None, None
        for c in range(N):
            s1 = st[c:c + 1]
            xi[t, c] = orc.rng_normals(s1, d)
            w = orc.rng_words(s1)
            ua[t, c], usel[t, c] = w[0], w[1:]
    thetas = np.zeros((T + 1, N, d)); logpost = np.zeros((T + 1, N))
    accept = np.zeros((T, N), dtype=np.uint8); picked = np.zeros((T, N), dtype=np.int8)
    real_normal, real_uniform = np.random.normal, np.random.uniform
    import contextlib, io
    try:
        for c in range(N):
            rp = BlockReplay(xi[:, c], usel[:, c], ua[:, c], ends)
            np.random.normal, np.random.uniform = rp.normal, rp.uniform
            prop = tc.ChangepointRegression1DProp(model, hscale)
            prop.P_cumprop_cpv = 1.0
            t0 = cp.ChangepointParams(th0[c, :k], th0[c, k:2 * k + 1], th0[c, 2 * k + 1])
            s = Sampler(model, prop, t0)
            with contextlib.redirect_stdout(io.StringIO()):
                s.run(T)
            assert rp.t == T
            for e, (tt, lp) in enumerate(zip(s._chain_thetas, s._chain_logpost)):
                thetas[e, c] = np.concatenate([np.ravel(tt.cpx), np.ravel(tt.cpv), np.ravel(tt.sig)])
                logpost[e, c] = lp
            accept[:, c] = [s._chain_thetas[t + 1] is not s._chain_thetas[t] for t in range(T)]
            picked[:, c] = rp.picked
            Cd = np.concatenate([np.diag(prop.cpx_prop.L) ** 2, np.diag(prop.cpv_prop.L) ** 2,
                                 np.diag(prop.cps_prop.L) ** 2])
    finally:
        np.random.normal, np.random.uniform = real_normal, real_uniform
    np.testing.assert_allclose(Cd, Cdiag, rtol=1e-14)
    assert set(np.unique(picked)) == {0, 1}
    save("g13_changepoint_block", x=x, y=y, k=np.int64(k), xmin=np.float64(1.0), xmax=np.float64(3.0),
         lamb=np.float64(k), alpha=np.float64(1.0), beta=np.float64(1.0), hscale=np.float64(hscale),
         Cdiag=Cd, sizes=np.array([k, k + 1, 1], dtype=np.int32), pcum=np.array([0.2, 1.0, 0.6]),
         seed=np.uint64(130), theta0=th0, thetas=thetas, logpost=logpost, accept=accept,
         picked=picked, usel=usel, ua=ua)


if __name__ == "__main__":
    if len(sys.argv) > 1:          # regenerate selected fixtures: make_golden.py g11_mixture g12_pt
        for name in sys.argv[1:]:
            globals()[name]()
        sys.exit(0)
    g0_rng()
    g1_gauss2d_rw()
    g2_gauss100d_rw()
    g3_gauss_hmc()
    g4_logreg()
    args = g5_changepoint()
    g6_edges(*args)
    g7_errors()
    g8_adapt_scale()
    g9_hmc_mass()
    g10_pcn()
    g11_mixture()
    g12_pt()
    g13_changepoint_block_proposal()
    g14_adapt_cov()
    g15_gauss_big_d()
    g16_reference_example()
    g17_changepoint_rj()
    g18_f64_normals()
