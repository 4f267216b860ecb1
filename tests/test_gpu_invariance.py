"""Distributional checks on the GPU at the full chain count (65 536): started from the
target, a correct MH kernel leaves the target invariant, so pooled moments over chains must
stay at the target's within Monte-Carlo error.  Bit parity with the oracle cannot see a rule
that oracle and kernel share by mistake; this can.  (The property does not depend on the
number of iterations, which keeps the runs short.)"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N, D = 65536, 20


@pytest.fixture(scope="module")
def target():
    import riemann_amd
    from riemann_amd import _lib
    from riemann_amd.models import MultiGaussianDist
    _lib.call("rmh_init", 0)
    g = np.random.default_rng(100)
    A = g.standard_normal((D, D))
    Ct = A @ A.T / D + 0.2 * np.eye(D)
    mu = g.standard_normal(D)
    return riemann_amd, MultiGaussianDist(mu, Ct), mu, Ct, np.linalg.cholesky(Ct), g


def moments_ok(x, mu, C, nsig=6.0):
    """mean and covariance of the rows of x against (mu, C), entries within nsig standard errors"""
    n = x.shape[0]
    sd = np.sqrt(np.diag(C))
    assert np.all(np.abs(x.mean(0) - mu) < nsig * sd / np.sqrt(n))
    S = np.cov(x.T)
    se = np.sqrt((np.outer(np.diag(C), np.diag(C)) + C ** 2) / n)     # Wishart standard errors
    assert np.all(np.abs(S - C) < nsig * se), float(np.abs((S - C) / se).max())


@pytest.mark.parametrize("kind", ["rw", "pcn", "mala", "hmc_mass", "adapt_rw"])
def test_target_is_invariant(target, kind):
    ra, m, mu, Ct, Lt, g = target
    from riemann_amd.proposals import AdaptScaleRandomWalk, MALA, MetropolisRandomWalk, VanillaHMC, pCN
    th0 = mu + g.standard_normal((N, D)) @ Lt.T
    p = {"rw": lambda: MetropolisRandomWalk((2.38 ** 2 / D) * Ct),
         "pcn": lambda: pCN(0.5 * np.eye(D), 0.9),
         "mala": lambda: MALA(0.5, m),
         "hmc_mass": lambda: VanillaHMC(0.35, 4, m, M=np.linalg.inv(Ct)),
         "adapt_rw": lambda: AdaptScaleRandomWalk((2.38 ** 2 / D) * Ct)}[kind]()
    s = ra.Sampler(m, p, th0, seed=7)
    s.run(120, record=False)
    th, lp = s.current_state()
    moments_ok(th, mu, Ct)
    np.testing.assert_allclose(lp, m.log_posterior(th), rtol=1e-9)


def test_tempered_ladder_is_invariant(target):
    """Every rung started from ITS tempered target N(mu, C / beta): within-chain moves and
    swaps must leave all of them where they are."""
    ra, m, mu, Ct, Lt, g = target
    from riemann_amd.proposals import MetropolisRandomWalk
    from riemann_amd.samplers import PTSampler
    K, Ne = 4, N // 4
    betas = 0.5 ** np.arange(K)
    pt = PTSampler(m, MetropolisRandomWalk((2.38 ** 2 / D) * Ct), np.tile(mu, (Ne, 1)), betas=betas,
                   Pswap=0.3, seed=3)
    slots = np.empty((pt._S, D))
    for i, b in enumerate(betas):
        slots[pt._idx[:, i]] = mu + g.standard_normal((Ne, D)) @ Lt.T / np.sqrt(b)
    from riemann_amd import _lib
    _lib.call("rmh_state_set_theta", pt._state, slots)
    pt.run(100, Nburn=100)
    codes = pt._chain_codes
    assert (codes == 3).mean() > 0.02          # swaps do happen (d = 20: about a tenth of the proposals)
    for i, b in enumerate(betas):
        th, _ = pt.samplers[i].current_state()
        moments_ok(th, mu, Ct / b)


def same_moments(x, y, nsig=6.0):
    """two-sample z-test on every first and second moment (x_i, x_i x_j) of two samples"""
    def feats(a):
        iu = np.triu_indices(a.shape[1])
        return np.concatenate([a, (a[:, :, None] * a[:, None, :])[:, iu[0], iu[1]]], axis=1)
    fx, fy = feats(x), feats(y)
    z = (fx.mean(0) - fy.mean(0)) / np.sqrt(fx.var(0) / len(fx) + fy.var(0) / len(fy))
    assert np.abs(z).max() < nsig, float(np.abs(z).max())


def test_mixture_target_is_invariant_under_random_walk_and_tempering():
    """benchmark_gauss2d_mix2_cross: chains started from exact draws of the mixture stay
    distributed as the mixture (compared with an independent exact sample)."""
    import riemann_amd as ra
    from riemann_amd.models import benchmarks as b
    from riemann_amd.proposals import MetropolisRandomWalk
    from riemann_amd.samplers import PTSampler
    g = np.random.default_rng(5)
    C1, C2 = np.array([[1.0, 0.9], [0.9, 1.0]]), np.array([[1.0, -0.9], [-0.9, 1.0]])

    def exact(n):
        pick = g.uniform(size=n) < 0.3
        z = g.standard_normal((n, 2))
        return np.where(pick[:, None], z @ np.linalg.cholesky(C1).T, z @ np.linalg.cholesky(C2).T)

    m = b.benchmark_gauss2d_mix2_cross
    s = ra.Sampler(m, MetropolisRandomWalk(0.8 * np.eye(2)), exact(N), seed=2)
    s.run(150, record=False)
    same_moments(s.current_state()[0], exact(N))
    # cold rung of a ladder whose hotter rungs start from the cold target too: the cold rung is
    # invariant only if swaps are accepted with the right ratio (hot rungs relax meanwhile)
    Ne = N // 4
    pt = PTSampler(m, MetropolisRandomWalk(0.8 * np.eye(2)), exact(Ne), betas=[1.0, 1.0, 1.0, 1.0], Pswap=0.3, seed=4)
    pt.run(100, Nburn=100)
    for i in range(4):                      # equal temperatures: every rung keeps the target
        same_moments(pt.samplers[i].current_state()[0], exact(Ne))
    assert (pt._chain_codes == 3).mean() > 0.2   # and every swap between equals is accepted
    assert (pt._chain_codes == 2).sum() == 0


def test_pooled_covariance_adaptation_finds_the_target_covariance(target):
    """AdaptCovRandomWalk from a poor isotropic start: the pooled history covariance (65 536
    chains x 100 states, matrix-core moment kernel) reproduces the target's, and the chains,
    started from the target, are still distributed as the target afterwards."""
    ra, m, mu, Ct, Lt, g = target
    from riemann_amd.proposals import AdaptCovRandomWalk
    th0 = mu + g.standard_normal((N, D)) @ Lt.T
    p = AdaptCovRandomWalk(0.01 * np.eye(D), t_adapt=10, smooth_adapt=True, exact_squares=True)
    s = ra.Sampler(m, p, th0, seed=21)
    s.run(100, record=False)
    est = p.C * D ** 0.4                        # C = Cs / d^0.4 (adaptive.py:101), prior weight negligible
    assert np.abs(est - Ct).max() < 0.02 * np.abs(Ct).max()
    np.testing.assert_allclose(p._SX / (100 * N), mu, atol=0.02)
    moments_ok(s.current_state()[0], mu, Ct)


@pytest.mark.parametrize("kind", ["rw", "adapt_rw", "pt", "pcn", "mala", "hmc", "hmc_mass"])
def test_cfg2_size_target_is_invariant(kind):
    """The same property at the headline shape (d = 100, 65 536 chains): the kernels that run
    there (four tail rows on the VALU through DPP row broadcasts, row-swap reductions, the
    fixed-length stream) differ from the generic ones the d = 20 cases above exercise."""
    import riemann_amd as ra
    from riemann_amd.models import MultiGaussianDist
    from riemann_amd.proposals import MALA, AdaptScaleRandomWalk, MetropolisRandomWalk, VanillaHMC, pCN
    from riemann_amd.samplers import PTSampler
    d = 100
    g = np.random.default_rng(2024)
    A = g.standard_normal((d, d))
    Ct = A @ A.T / d + 0.3 * np.eye(d)
    mu = g.standard_normal(d)
    Lt = np.linalg.cholesky(Ct)
    m = MultiGaussianDist(mu, Ct)
    Cp = (2.38 ** 2 / d) * Ct
    if kind == "pt":    # every rung started from ITS tempered target, as in the d = 20 case
        K, Ne = 4, N // 4
        betas = 0.8 ** np.arange(K)
        pt = PTSampler(m, MetropolisRandomWalk(Cp), np.tile(mu, (Ne, 1)), betas=betas, Pswap=0.3, seed=11)
        slots = np.empty((pt._S, d))
        for i, b in enumerate(betas):
            slots[pt._idx[:, i]] = mu + g.standard_normal((Ne, d)) @ Lt.T / np.sqrt(b)
        from riemann_amd import _lib
        _lib.call("rmh_state_set_theta", pt._state, slots)
        pt.run(40, Nburn=40)
        assert (pt._chain_codes == 3).mean() > 0.005
        for i, b in enumerate(betas):
            th, _ = pt.samplers[i].current_state()
            moments_ok(th, mu, Ct / b)
        return
    p = {"rw": lambda: MetropolisRandomWalk(Cp), "adapt_rw": lambda: AdaptScaleRandomWalk(Cp),
         "pcn": lambda: pCN(0.5 * np.eye(d), 0.95), "mala": lambda: MALA(0.35, m),
         "hmc": lambda: VanillaHMC(0.25, 5, m),
         "hmc_mass": lambda: VanillaHMC(0.3, 3, m, M=np.linalg.inv(Ct))}[kind]()
    th0 = mu + g.standard_normal((N, d)) @ Lt.T
    s = ra.Sampler(m, p, th0, seed=13)
    s.run(60 if kind in ("rw", "adapt_rw", "pcn") else 25, record=False)
    th, lp = s.current_state()
    moments_ok(th, mu, Ct)
    np.testing.assert_allclose(lp, m.log_posterior(th), rtol=1e-9)
