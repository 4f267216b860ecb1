"""The mixed-plugin path (SURVEY.md 8a-8 ii, 8b "Batched extension"): the reference samples ANY
Model / Proposal subclass (riemann/samplers/sampler.py:72-90, proposals/hamiltonian.py:60-70).
A user's own Python Model next to a device Proposal, a Python Proposal next to a device Model,
and VanillaHMC on a Python gradient callable run sampler.py:72-90 on the host with the chains'
rmh-rng stream -- pinned on the reference's own runs: the numpy plugins below are the classes
tests/golden/make_golden.py drove through the reference's Sampler to record g1 and g4."""
import numpy as np
import pytest

from gpu_common import RTOL, check_run, load

pytestmark = pytest.mark.gpu


def _numpy_logreg(ra, X, y, tau2):
    class LogRegModel(ra.Model):          # the plugin of make_golden.py:166-190, on this package's Model ABC
        def log_prior(self, theta):
            return -np.dot(theta, theta) / (2.0 * tau2)

        def log_likelihood(self, theta):
            eta = X @ theta
            return np.sum(y * eta - np.logaddexp(0.0, eta))

        def grad_log_posterior(self, theta):
            eta = X @ theta
            sg = np.where(eta >= 0, 1.0 / (1.0 + np.exp(-np.abs(eta))),
                          np.exp(-np.abs(eta)) / (1.0 + np.exp(-np.abs(eta))))
            return X.T @ (y - sg) - theta / tau2
    return LogRegModel()


@pytest.mark.parametrize("tag", ["mala", "hmc10"])
def test_python_model_and_python_gradient_reproduce_the_reference_run(ra, tag):
    """g4: numpy logistic Model + VanillaHMC(eps, L, model.grad_log_posterior), every byte of
    arithmetic in the user's numpy code; stream, leapfrog order and accept rule are the package's."""
    from riemann_amd.proposals import VanillaHMC
    z = load("g4_logreg256x16_" + tag)
    m = _numpy_logreg(ra, z["X"], z["y"], float(z["tau2"]))
    p = VanillaHMC(float(z["eps"]), int(z["nsteps"]), m.grad_log_posterior, Npars=16)
    s = ra.Sampler(m, p, z["theta0"], seed=int(z["seed"]))
    s.run(32)
    check_run(z, s)
    th, lp = s.current_state()
    np.testing.assert_array_equal(th, s._chain_thetas[-1])


def test_python_model_with_device_random_walk(ra):
    """g1: a numpy Gaussian Model sampled by the device MetropolisRandomWalk (rmh_propose on the
    chains' stream, host log_posterior, host accept with the stream's uniform)."""
    from riemann_amd.proposals import MetropolisRandomWalk
    z = load("g1_gauss2d_rw")
    Li = np.linalg.inv(np.linalg.cholesky(z["C"]))
    logdet = np.log(np.linalg.det(z["C"]))

    class Gauss(ra.Model):
        def log_prior(self, theta):
            return 0.0

        def log_likelihood(self, theta):
            u = Li @ (theta - z["mu"])
            return -0.5 * (u @ u + 2 * np.log(2 * np.pi) + logdet)

    s = ra.Sampler(Gauss(), MetropolisRandomWalk(z["Cprop"]), z["theta0"], seed=int(z["seed"]))
    s.run(64)
    check_run(z, s)
    # thinning, resume and the one-step API on the host path
    a = ra.Sampler(Gauss(), MetropolisRandomWalk(z["Cprop"]), z["theta0"], seed=int(z["seed"]))
    a.run(10, 4, 3)                       # states 4, 7, 10
    np.testing.assert_allclose(a._chain_thetas, z["thetas"][4:11:3], rtol=RTOL, atol=1e-12)
    for t in range(11, 15):
        th, lp = a.sample()
        np.testing.assert_allclose(th, z["thetas"][t], rtol=RTOL, atol=1e-12)
    assert a._chain_thetas.shape == (3 + 4, 4, 2)


def test_run_to_disk_with_a_python_model(ra, tmp_path):
    """run_to_disk on the mixed-plugin path: the chains live on the host there; with
    (T - Nburn) % Nthin != 0 the current state must become the last KEPT one, and the run continues
    like an in-memory run()."""
    from riemann_amd.proposals import MetropolisRandomWalk

    class Gauss(ra.Model):
        def log_prior(self, th):
            return np.zeros(np.atleast_2d(th).shape[0])

        def log_likelihood(self, th):
            t = np.atleast_2d(th)
            return -0.5 * np.sum(t * t, axis=1)

        def pack(self):
            pass

        def unpack(self):
            pass
    th0 = np.random.default_rng(3).standard_normal((21, 3))
    for T, nb, nt in ((23, 4, 5), (20, 0, 4)):
        a = ra.Sampler(Gauss(), MetropolisRandomWalk(0.4 * np.eye(3)), th0, seed=11)
        b = ra.Sampler(Gauss(), MetropolisRandomWalk(0.4 * np.eye(3)), th0, seed=11)
        a.run(T, nb, nt)
        kept = b.run_to_disk(str(tmp_path / "m"), T, nb, nt, chunk=6)
        assert kept == a._chain_thetas.shape[0]
        np.testing.assert_array_equal(np.load(str(tmp_path / "m_thetas.npy")), a._chain_thetas)
        np.testing.assert_array_equal(np.load(str(tmp_path / "m_accept.npy")), a._chain_accept)
        np.testing.assert_array_equal(b.current_state()[0], a.current_state()[0])
        np.testing.assert_array_equal(b.current_state()[1], a.current_state()[1])
        a.run(4)
        b.run(4)
        np.testing.assert_array_equal(b._chain_thetas, a._chain_thetas)


def test_device_model_with_python_proposal(ra):
    """A user's Proposal (its own numpy generator) on a device Model: the Sampler supplies the
    accept uniforms from the chains' stream and the device evaluates the densities."""
    from riemann_amd import _lib
    from riemann_amd.models import MultiGaussianDist
    from riemann_amd.rng import rng_seed_host
    Cov = np.array([[2.0, 0.6], [0.6, 1.0]])
    m = MultiGaussianDist(np.zeros(2), Cov)
    th0 = np.array([[0.3, -0.2], [1.0, 1.0], [-2.0, 0.5]])

    class MyWalk(ra.Proposal):
        def __init__(self, seed):
            self.g, self.calls, self.adapted = np.random.default_rng(seed), 0, 0

        def propose(self, theta):
            self.calls += 1
            return theta + 0.7 * self.g.standard_normal(theta.shape), 0.0

        def adapt(self, theta):
            self.adapted += 1

    p = MyWalk(5)
    s = ra.Sampler(m, p, th0, seed=21)
    T = 25
    s.run(T)
    assert p.calls == 3 * T and p.adapted == 3 * T
    # the same loop written out: sampler.py:72-90 with the stream's uniforms
    q, rng = MyWalk(5), rng_seed_host(21, 3)
    th, lp = th0.copy(), m.log_posterior(th0)
    ths, acc = [th.copy()], []
    for _ in range(T):
        thp = np.stack([q.propose(r)[0] for r in th])
        lpp = m.log_posterior(thp)
        u = np.empty((1, 3))
        _lib.call("rmh_rng_draw", rng, 3, 0, 1, None, u.ctypes.data)
        ok = np.log(u[0]) < np.minimum(0.0, lpp - lp)
        th, lp = np.where(ok[:, None], thp, th), np.where(ok, lpp, lp)
        ths.append(th.copy())
        acc.append(ok)
    np.testing.assert_array_equal(s._chain_accept, np.array(acc, dtype=np.uint8))
    np.testing.assert_allclose(s._chain_thetas, np.array(ths), rtol=1e-12)
    assert 0.1 < s._chain_accept.mean() < 0.9


def test_rng_draw_is_the_fused_kernels_stream(ra):
    """rmh_rng_draw (normals + uniform, normals only, uniform only) against the recorded stream."""
    from riemann_amd import _lib
    from riemann_amd.rng import rng_seed_host
    z = load("g1_gauss2d_rw")
    rng = rng_seed_host(int(z["seed"]), 4)
    xi, u = np.empty((5, 4, 2)), np.empty((5, 4))
    _lib.call("rmh_rng_draw", rng, 4, 2, 5, xi.ctypes.data, u.ctypes.data)
    np.testing.assert_array_equal(xi, z["xi"][:5])
    np.testing.assert_array_equal(u, z["u"][:5])
    x1, u1 = np.empty((1, 4, 2)), np.empty((1, 4))
    _lib.call("rmh_rng_draw", rng, 4, 2, 1, x1.ctypes.data, None)       # normals only ...
    _lib.call("rmh_rng_draw", rng, 4, 0, 1, None, u1.ctypes.data)       # ... then the uniform
    np.testing.assert_array_equal(x1[0], z["xi"][5])
    np.testing.assert_array_equal(u1[0], z["u"][5])


def test_python_model_with_device_adaptive_scale_random_walk(ra, tmp_path):
    """AdaptScaleRandomWalk next to a user's numpy Model: the chains' scales adapt on the host
    (adaptive.py:25-35), the device forms L xi on the chains' stream.  Same chains as the fused run of
    the device Gaussian with the same proposal (states to rounding: the host takes exp from numpy,
    the kernel from the device library), adapted scales included; checkpoint / restore carries the
    adaptation state.  The other adaptive device proposals still need a device Model."""
    from riemann_amd.models import MultiGaussianDist
    from riemann_amd.proposals import AdaptScaleRandomWalk
    g = np.random.default_rng(5)
    d, N, T = 6, 40, 120
    A = g.standard_normal((d, d))
    Cov, mu = A @ A.T + d * np.eye(d), g.standard_normal(d)
    Li = np.linalg.inv(np.linalg.cholesky(Cov))
    logdet = np.log(np.linalg.det(Cov))

    class Gauss(ra.Model):
        def log_prior(self, theta):
            return 0.0

        def log_likelihood(self, theta):
            u = Li @ (theta - mu)
            return -0.5 * (u @ u + logdet + d * np.log(2 * np.pi))

    th0 = g.standard_normal((N, d))
    fused = ra.Sampler(MultiGaussianDist(mu, Cov), AdaptScaleRandomWalk(0.5 * Cov), th0, seed=8)
    fused.run(T)
    pm = AdaptScaleRandomWalk(0.5 * Cov)
    mixed = ra.Sampler(Gauss(), pm, th0, seed=8)
    mixed.run(70)
    mixed.checkpoint(str(tmp_path / "mx.npz"))
    mixed.run(T - 70)
    np.testing.assert_array_equal(mixed._chain_accept, fused._chain_accept[70:])
    np.testing.assert_allclose(mixed._chain_thetas, fused._chain_thetas[70:], rtol=1e-11, atol=1e-13)
    np.testing.assert_allclose(pm.scale, fused.proposal.scale, rtol=1e-12)
    np.testing.assert_array_equal(pm.Naccepts, fused.proposal.Naccepts)
    assert np.all(pm.Nsamples == T) and pm.scale.std() > 0.01
    again = ra.Sampler(Gauss(), AdaptScaleRandomWalk(0.5 * Cov), th0, seed=8)
    again.restore(str(tmp_path / "mx.npz"))
    again.run(T - 70)
    np.testing.assert_array_equal(again._chain_thetas, mixed._chain_thetas)
    np.testing.assert_array_equal(again.proposal.scale, pm.scale)
    from riemann_amd.proposals import AdaptScalepCN
    s = ra.Sampler(Gauss(), AdaptScalepCN(np.eye(d), 0.9), np.zeros(d), seed=1)
    with pytest.raises(TypeError):
        s.run(2)


def test_stand_alone_propose_of_the_adaptive_scale_random_walk(ra):
    """AdaptScaleRandomWalk.propose(theta): theta + scale (L xi) with the adapted scale(s), against the
    plain random walk carrying the same scale; the other adaptive proposals say what is not built."""
    from riemann_amd.models import MultiGaussianDist
    from riemann_amd.proposals import AdaptScaleRandomWalk, AdaptScalepCN, MetropolisRandomWalk
    from riemann_amd.rng import rng_seed_host
    g = np.random.default_rng(9)
    d, N = 5, 33
    A = g.standard_normal((d, d))
    Cov = A @ A.T + d * np.eye(d)
    p = AdaptScaleRandomWalk(Cov)
    th = g.standard_normal((N, d))
    q = MetropolisRandomWalk(Cov)
    a, la = p.propose(th, rng=rng_seed_host(3, N, 0))                 # scale = 1.0 before any run
    b, lb = q.propose(th, rng=rng_seed_host(3, N, 0))
    np.testing.assert_allclose(a, b, rtol=1e-14, atol=1e-14)
    s = ra.Sampler(MultiGaussianDist(np.zeros(d), Cov), p, th, seed=2)
    s.run(60)
    assert np.shape(p.scale) == (N,) and np.std(p.scale) > 0
    a, _ = p.propose(th, rng=rng_seed_host(4, N, 0))
    step, _ = q.propose(np.zeros((N, d)), rng=rng_seed_host(4, N, 0))
    np.testing.assert_allclose(a, th + p.scale[:, None] * step, rtol=1e-14, atol=1e-14)
    with pytest.raises(ra.ParameterError):
        p.propose(th[:5])
    with pytest.raises(NotImplementedError):
        AdaptScalepCN(Cov, 0.9).propose(th)


def test_adaptive_scale_hmc_on_a_python_gradient(ra):
    """AdaptScaleHMC(eps, L, user_gradient) with a user's numpy Model: trajectory on the host with
    device-drawn momenta, eps = scale * eps0 adapted per chain on the host (hamiltonian.py:94-104).
    Same chains as the fused device run of the same target (to rounding: numpy's matrix products and
    exp against the kernels')."""
    from riemann_amd.models import MultiGaussianDist
    from riemann_amd.proposals import AdaptScaleHMC
    g = np.random.default_rng(11)
    d, N, T = 5, 24, 60
    A = g.standard_normal((d, d))
    Cov, mu = A @ A.T + d * np.eye(d), g.standard_normal(d)
    P = np.linalg.inv(Cov)
    Li = np.linalg.inv(np.linalg.cholesky(Cov))
    logdet = np.log(np.linalg.det(Cov))

    class Gauss(ra.Model):
        def log_prior(self, theta):
            return 0.0

        def log_likelihood(self, theta):
            u = Li @ (theta - mu)
            return -0.5 * (u @ u + logdet + d * np.log(2 * np.pi))

    dm = MultiGaussianDist(mu, Cov)
    th0 = mu + g.standard_normal((N, d))
    fused = ra.Sampler(dm, AdaptScaleHMC(0.6, 3, dm), th0, seed=5)
    fused.run(T)
    ph = AdaptScaleHMC(0.6, 3, lambda th: -(P @ (th - mu)), Npars=d)
    mixed = ra.Sampler(Gauss(), ph, th0, seed=5)
    mixed.run(T)
    np.testing.assert_array_equal(mixed._chain_accept, fused._chain_accept)
    np.testing.assert_allclose(mixed._chain_thetas, fused._chain_thetas, rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(ph.scale, fused.proposal.scale, rtol=1e-12)
    assert np.all(ph.Nsamples == T) and np.std(ph.scale) > 0


@pytest.mark.parametrize("smooth,squares,scaled", [(True, False, False), (False, True, False), (True, False, True)])
def test_python_model_with_device_adaptive_covariance_random_walk(ra, tmp_path, smooth, squares, scaled):
    """AdaptCovRandomWalk (Haario) next to a user's numpy Model -- the reference's everyday combination:
    theta + L xi on the device with the current factor, moments / covariance / Cholesky factor on the host
    as adaptive.py:73-102 computes them (pooled over the chains as the fused path pools them).  Same
    chains as the fused device run to rounding (numpy's sums and LAPACK's factor against the kernels'),
    adapted C and L included; checkpoint / restore carries the adaptation."""
    from riemann_amd.models import MultiGaussianDist
    from riemann_amd import proposals as P
    AdaptCovRandomWalk = P.AdaptScaleCovRandomWalk if scaled else P.AdaptCovRandomWalk   # scaled: + per-chain scale
    g = np.random.default_rng(17)
    d, N, T = 4, 32, 90
    A = g.standard_normal((d, d))
    Cov, mu = A @ A.T + d * np.eye(d), g.standard_normal(d)
    Li = np.linalg.inv(np.linalg.cholesky(Cov))
    logdet = np.log(np.linalg.det(Cov))

    class Gauss(ra.Model):
        def log_prior(self, theta):
            return 0.0

        def log_likelihood(self, theta):
            u = Li @ (theta - mu)
            return -0.5 * (u @ u + logdet + d * np.log(2 * np.pi))

    kw = dict(t_adapt=200, smooth_adapt=smooth, exact_squares=squares)
    th0 = mu + g.standard_normal((N, d))
    fused = ra.Sampler(MultiGaussianDist(mu, Cov), AdaptCovRandomWalk(0.5 * np.eye(d), **kw), th0, seed=6)
    fused.run(T)
    pm = AdaptCovRandomWalk(0.5 * np.eye(d), **kw)
    mixed = ra.Sampler(Gauss(), pm, th0, seed=6)
    mixed.run(50)
    mixed.checkpoint(str(tmp_path / "cov.npz"))
    mixed.run(T - 50)
    np.testing.assert_array_equal(mixed._chain_accept, fused._chain_accept[50:])
    np.testing.assert_allclose(mixed._chain_thetas, fused._chain_thetas[50:], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(pm.C, fused.proposal.C, rtol=1e-9)
    np.testing.assert_allclose(pm.L, fused.proposal.L, rtol=1e-9, atol=1e-12)
    assert not np.allclose(pm.C, 0.5 * np.eye(d) / d ** 0.4)          # it did adapt
    if scaled:
        np.testing.assert_allclose(pm.scale, fused.proposal.scale, rtol=1e-12)
    again = ra.Sampler(Gauss(), AdaptCovRandomWalk(0.5 * np.eye(d), **kw), th0, seed=6)
    again.restore(str(tmp_path / "cov.npz"))
    again.run(T - 50)
    np.testing.assert_array_equal(again._chain_thetas, mixed._chain_thetas)
