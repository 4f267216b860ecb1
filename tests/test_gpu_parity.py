"""GPU parity tests: the HIP path (through the C ABI / the plugin API) against the golden
vectors recorded from the reference and against the CPU oracle on the same seeded inputs:
building blocks, fixtures G0-G4, G6, G7, G9-G11, G15, G16, sizes and edge cases, pooling,
checkpoints, streaming to disk.  (Adaptive proposals, tempering and the changepoint model have
their own modules.)"""
import numpy as np
import pytest

from gpu_common import MIX2D, RTOL, check_run, corr100, cp_model, load, oracle_run, scaled_rw  # noqa: F401

pytestmark = pytest.mark.gpu


def test_mfma_f64_layout(ra):
    """A (16x4) B (4x16) + C through one v_mfma_f64_16x16x4_f64, asymmetric data."""
    from riemann_amd import _lib
    g = np.random.default_rng(0)
    A, B, Cm = g.standard_normal((16, 4)), g.standard_normal((4, 16)), g.standard_normal((16, 16))
    D = np.empty((16, 16))
    _lib.call("rmh_debug_mfma", A, B, Cm, D)
    np.testing.assert_allclose(D, A @ B + Cm, rtol=1e-14, atol=1e-14)
    I4 = np.zeros((16, 4))
    I4[:4] = np.eye(4)
    _lib.call("rmh_debug_mfma", I4, B, np.zeros((16, 16)), D)
    np.testing.assert_array_equal(D[:4], B)   # exact: catches a transposed C/D map


def test_rng_stream_bit_exact_vs_fixture(ra):
    from riemann_amd import _lib
    z = load("g0_rng")
    for d in (1, 2, 5, 12, 100):
        xi, u = np.empty((3, 3, d)), np.empty((3, 3))
        _lib.call("rmh_debug_rng_stream", int(z["seed"]), int(z["chain0"]), 3, d, 3, xi, u)
        np.testing.assert_array_equal(xi, z["xi_d%d" % d])
        np.testing.assert_array_equal(u, z["u_d%d" % d])


def test_rng_stream_bit_exact_vs_oracle(ra, orc):
    from riemann_amd import _lib
    N, d, T = 1000, 100, 4
    xi, u = np.empty((T, N, d)), np.empty((T, N))
    _lib.call("rmh_debug_rng_stream", 77, 123456789, N, d, T, xi, u)
    exi, eu = orc.rng_stream(orc.rng_seed(77, N, chain0=123456789), d, T)
    np.testing.assert_array_equal(xi, exi)
    np.testing.assert_array_equal(u, eu)


def test_state_rng_roundtrip_and_seed(ra, orc):
    from riemann_amd.rng import rng_seed_host
    from riemann_amd.models import MultiGaussianDist
    from riemann_amd.proposals import MetropolisRandomWalk
    s = ra.Sampler(MultiGaussianDist(np.zeros(2), np.eye(2)), MetropolisRandomWalk(np.eye(2)),
                   np.ones((37, 2)), seed=5, chain0=11)
    from riemann_amd import _lib
    st = np.empty((37, 16), dtype=np.uint64)
    _lib.call("rmh_state_get_rng", s._state, st)
    np.testing.assert_array_equal(st, orc.rng_seed(5, 37, chain0=11))
    np.testing.assert_array_equal(st, rng_seed_host(5, 37, 11))


def test_gaussian_eval_and_grad(ra, orc):
    from riemann_amd.models import MultiGaussianDist
    g = np.random.default_rng(1)
    for d in (1, 2, 17, 40, 100):
        A = g.standard_normal((d, d))
        Cov = A @ A.T + d * np.eye(d)
        mu = g.standard_normal(d)
        m, o = MultiGaussianDist(mu, Cov), orc.Gaussian(mu, Cov)
        th = mu + 3 * g.standard_normal((50, d))
        np.testing.assert_allclose(m.log_posterior(th), o.log_posterior(th), rtol=RTOL)
        np.testing.assert_allclose(m.log_likelihood(th), o.log_likelihood(th), rtol=RTOL)
        np.testing.assert_array_equal(m.log_prior(th), 0.0)
        np.testing.assert_allclose(m.grad_log_likelihood(th), o.grad_log_posterior(th),
                                   rtol=1e-8, atol=1e-10)
        assert isinstance(m.log_posterior(th[0]), float)


def test_logreg_eval_and_grad(ra, orc):
    from riemann_amd.models import LogisticRegression
    from riemann_amd.workloads import logreg_data
    g = np.random.default_rng(2)
    for n, d in ((256, 16), (1000, 7), (37, 33), (2000, 100)):
        X, y = logreg_data(n, d, seed=n)
        m, o = LogisticRegression(X, y, 100.0), orc.LogReg(X, y, 100.0)
        # (|eta| from the table's first cell to far beyond its edge at 37.5, where the link function saturates)
        th = g.standard_normal((60, d)) * np.array([0.1, 1.0, 5.0, 30.0, 300.0, 1e4])[g.integers(0, 6, 60)][:, None]
        np.testing.assert_allclose(m.log_posterior(th), o.log_posterior(th), rtol=RTOL)
        np.testing.assert_allclose(m.log_prior(th), o.log_prior(th), rtol=RTOL)
        np.testing.assert_allclose(m.log_likelihood(th), o.log_likelihood(th), rtol=RTOL)
        np.testing.assert_allclose(m.grad_log_posterior(th), o.grad_log_posterior(th),
                                   rtol=1e-8, atol=1e-9)


def test_propose_vs_oracle(ra, orc):
    from riemann_amd.models import MultiGaussianDist
    from riemann_amd.proposals import MetropolisRandomWalk, VanillaHMC
    g = np.random.default_rng(3)
    d, N = 100, 70
    Ct = corr100()
    th = g.standard_normal((N, d))
    p = MetropolisRandomWalk(0.05 * Ct)
    p.scale = 0.7
    p.seed(9, chain0=4)
    thp, logq = p.propose(th)
    st = orc.rng_seed(9, N, chain0=4)
    ethp, elogq = orc.RandomWalk(0.05 * Ct, 0.7).propose(orc.Gaussian(np.zeros(d), Ct), th, st)
    np.testing.assert_allclose(thp, ethp, rtol=RTOL, atol=1e-12)
    np.testing.assert_array_equal(logq, 0.0)
    np.testing.assert_array_equal(p._rng, st)          # both streams advanced identically
    m = MultiGaussianDist(np.zeros(d), Ct)
    for L in (1, 4):
        h = VanillaHMC(0.2, L, m.grad_log_likelihood)
        h.seed(10)
        thp, logq = h.propose(th)
        ethp, elogq = orc.HMC(0.2, L, d).propose(orc.Gaussian(np.zeros(d), Ct), th, orc.rng_seed(10, N))
        np.testing.assert_allclose(thp, ethp, rtol=1e-8, atol=1e-10)
        np.testing.assert_allclose(logq, elogq, rtol=1e-7, atol=1e-9)
    t1, q1 = p.propose(th[0])                           # single state in, single state out
    assert t1.shape == (d,) and q1 == 0.0
    # pCN (asymmetric logqratio) and HMC with a mass matrix, stand-alone
    from riemann_amd.proposals import pCN
    pc = pCN(0.05 * Ct, 0.9)
    pc.seed(11)
    thp, logq = pc.propose(th)
    ethp, elogq = orc.PCN(0.05 * Ct, 0.9).propose(orc.Gaussian(np.zeros(d), Ct), th, orc.rng_seed(11, N))
    np.testing.assert_allclose(thp, ethp, rtol=RTOL, atol=1e-12)
    np.testing.assert_allclose(logq, elogq, rtol=1e-7, atol=1e-8)
    assert np.abs(logq).max() > 1.0
    A = g.standard_normal((d, d))
    Mm = A @ A.T / d + 0.5 * np.eye(d)
    h = VanillaHMC(0.15, 3, m, M=Mm)
    h.seed(12)
    thp, logq = h.propose(th)
    ethp, elogq = orc.HMC(0.15, 3, d, M=Mm).propose(orc.Gaussian(np.zeros(d), Ct), th, orc.rng_seed(12, N))
    np.testing.assert_allclose(thp, ethp, rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(logq, elogq, rtol=1e-7, atol=1e-8)


def test_g1_gauss2d_rw(ra):
    from riemann_amd.models import MultiGaussianDist
    from riemann_amd.proposals import MetropolisRandomWalk
    z = load("g1_gauss2d_rw")
    s = ra.Sampler(MultiGaussianDist(z["mu"], z["C"]), MetropolisRandomWalk(z["Cprop"]),
                   z["theta0"], seed=int(z["seed"]))
    s.run(64)
    check_run(z, s)


def test_g2_gauss100d_rw(ra):
    from riemann_amd.models import benchmarks
    from riemann_amd.proposals import MetropolisRandomWalk
    z = load("g2_gauss100d_rw")
    s = ra.Sampler(benchmarks.benchmark_gauss100d_corr,
                   MetropolisRandomWalk(float(z["prop_factor"]) * corr100()),
                   z["theta0"], seed=int(z["seed"]))
    s.run(32)
    check_run(z, s)


@pytest.mark.parametrize("tag", ["mala", "hmc10"])
def test_g3_gauss100d_hmc(ra, tag):
    from riemann_amd.models import benchmarks
    from riemann_amd.proposals import VanillaHMC
    z = load("g3_gauss100d_" + tag)
    m = benchmarks.benchmark_gauss100d_corr
    s = ra.Sampler(m, VanillaHMC(float(z["eps"]), int(z["nsteps"]), m.grad_log_likelihood),
                   z["theta0"], seed=int(z["seed"]))
    s.run(16)
    check_run(z, s)


def test_g3_gauss2d_hmc5(ra):
    from riemann_amd.models import MultiGaussianDist
    from riemann_amd.proposals import VanillaHMC
    z = load("g3_gauss2d_hmc5")
    m = MultiGaussianDist(z["mu"], z["C"])
    s = ra.Sampler(m, VanillaHMC(float(z["eps"]), int(z["nsteps"]), m.grad_log_likelihood),
                   z["theta0"], seed=int(z["seed"]))
    s.run(48)
    check_run(z, s)


@pytest.mark.parametrize("tag", ["mala", "hmc10"])
def test_g4_logreg_small(ra, tag):
    from riemann_amd.models import LogisticRegression
    from riemann_amd.proposals import VanillaHMC
    z = load("g4_logreg256x16_" + tag)
    m = LogisticRegression(z["X"], z["y"], float(z["tau2"]))
    s = ra.Sampler(m, VanillaHMC(float(z["eps"]), int(z["nsteps"]), m), z["theta0"],
                   seed=int(z["seed"]))
    s.run(32)
    check_run(z, s)


@pytest.mark.parametrize("tag", ["mala", "hmc10"])
def test_g4_logreg_full_size(ra, tag):
    from riemann_amd.models import LogisticRegression
    from riemann_amd.proposals import MALA, VanillaHMC
    from riemann_amd.workloads import logreg_data
    z = load("g4_logreg10000x100_" + tag)
    X, y = logreg_data(int(z["n"]), int(z["d"]), int(z["data_seed"]))
    m = LogisticRegression(X, y, float(z["tau2"]))
    p = MALA(float(z["eps"]), m) if tag == "mala" else VanillaHMC(float(z["eps"]), 10, m)
    s = ra.Sampler(m, p, z["theta0"], seed=int(z["seed"]))
    s.run(z["accept"].shape[0])
    check_run(z, s)


def test_g6_neginf_start_accepts_nan_ratio(ra):
    z = load("g6_neginf_start")
    s = ra.Sampler(cp_model(z), scaled_rw(z), z["theta0"], seed=int(z["seed"]))
    s.run(48)
    check_run(z, s)
    assert np.isinf(s._chain_logpost[0, :2]).all() and s._chain_accept[0, :2].all()


def test_g6_burn_thin_resume(ra):
    from riemann_amd.models import MultiGaussianDist
    from riemann_amd.proposals import MetropolisRandomWalk
    z = load("g6_burn_thin_resume")
    s = ra.Sampler(MultiGaussianDist(z["mu"], z["C"]), MetropolisRandomWalk(z["Cprop"]),
                   z["theta0"], seed=int(z["seed"]))
    (n0, b0, t0), (n1, b1, t1) = z["runs"]
    s.run(int(n0), int(b0), int(t0))
    assert s._chain_thetas.shape[0] == 3                  # run(10,4,3) keeps states 4,7,10
    np.testing.assert_allclose(s._chain_thetas, z["thetas_run0"], rtol=RTOL)
    np.testing.assert_allclose(s._chain_logpost, z["logpost_run0"], rtol=RTOL)
    s.run(int(n1), int(b1), int(t1))                      # continues from the last state
    assert s._chain_thetas.shape[0] == 6
    np.testing.assert_allclose(s._chain_thetas, z["thetas_run1"], rtol=RTOL)
    np.testing.assert_array_equal(s._chain_accept, z["accept_run1"])


def test_g7_error_conventions(ra):
    from riemann_amd import ParameterError
    from riemann_amd.models import MultiGaussianDist
    from riemann_amd.proposals import MetropolisRandomWalk
    z = load("g7_errors")
    with pytest.raises(ParameterError) as e:
        MetropolisRandomWalk(np.eye(3)).propose(np.ones(2))
    assert str(e.value) == str(z["rw_shape"])
    with pytest.raises(ParameterError) as e:
        MultiGaussianDist(np.zeros(2), np.ones((2, 3)))
    assert str(e.value) == str(z["gauss_nonsquare"])
    with pytest.raises(ParameterError) as e:
        MultiGaussianDist(np.zeros(3), np.eye(2))
    assert str(e.value) == str(z["gauss_mismatch"])
    with pytest.raises(ParameterError):   # model and proposal of different dimension
        ra.Sampler(MultiGaussianDist(np.zeros(2), np.eye(2)), MetropolisRandomWalk(np.eye(3)),
                   np.ones(2)).run(1)


def test_gauss100d_rw_2048_chains(ra, orc):
    from riemann_amd.models import benchmarks
    from riemann_amd.proposals import MetropolisRandomWalk
    Ct = corr100()
    Cp = (2.38 ** 2 / 100) * Ct
    th0 = np.random.default_rng(1).standard_normal((2048, 100)) @ np.linalg.cholesky(Ct).T
    s = ra.Sampler(benchmarks.benchmark_gauss100d_corr, MetropolisRandomWalk(Cp), th0, seed=1)
    s.run(40)
    ht, hl, acc = oracle_run(orc, orc.Gaussian(np.zeros(100), Ct), orc.RandomWalk(Cp), th0, 1, 40)
    np.testing.assert_array_equal(s._chain_accept, acc)
    np.testing.assert_allclose(s._chain_thetas, ht, rtol=RTOL, atol=1e-12)
    np.testing.assert_allclose(s._chain_logpost, hl, rtol=RTOL)
    assert 0.1 < acc.mean() < 0.4


def test_ragged_chain_counts_and_single_chain(ra, orc):
    """N not a multiple of 16 (partial wavefront tile) and the reference's
    single-chain calling convention."""
    from riemann_amd.models import MultiGaussianDist
    from riemann_amd.proposals import MetropolisRandomWalk
    g = np.random.default_rng(4)
    for N, d in ((1, 3), (5, 3), (17, 20), (63, 33), (100, 64)):
        A = g.standard_normal((d, d))
        Cov = A @ A.T + d * np.eye(d)
        th0 = g.standard_normal((N, d))
        s = ra.Sampler(MultiGaussianDist(np.zeros(d), Cov), MetropolisRandomWalk(0.3 * Cov / d),
                       th0, seed=N)
        s.run(25)
        ht, hl, acc = oracle_run(orc, orc.Gaussian(np.zeros(d), Cov), orc.RandomWalk(0.3 * Cov / d),
                                 th0, N, 25)
        np.testing.assert_array_equal(s._chain_accept, acc)
        np.testing.assert_allclose(s._chain_thetas, ht, rtol=RTOL, atol=1e-12)
    s = ra.Sampler(MultiGaussianDist(np.zeros(2), np.eye(2)), MetropolisRandomWalk(np.eye(2)),
                   np.ones(2), seed=0)
    s.run(10)
    assert s._chain_thetas.shape == (11, 2) and s._chain_logpost.shape == (11,)
    th, lp = s.current_state()
    assert th.shape == (2,) and isinstance(lp, float)
    np.testing.assert_array_equal(th, s._chain_thetas[-1])
    t2, l2 = s.sample()
    assert s._chain_thetas.shape == (12, 2) and l2 == s._chain_logpost[-1]


def test_logreg_mala_256_chains(ra, orc):
    from riemann_amd.models import LogisticRegression
    from riemann_amd.proposals import MALA
    from riemann_amd.workloads import logreg_data
    X, y = logreg_data(2000, 100, seed=2)
    th0 = 0.05 * np.random.default_rng(6).standard_normal((256, 100))
    m = LogisticRegression(X, y, 100.0)
    s = ra.Sampler(m, MALA(0.08, m), th0, seed=8)
    s.run(12)
    ht, hl, acc = oracle_run(orc, orc.LogReg(X, y, 100.0), orc.HMC(0.08, 1, 100), th0, 8, 12)
    np.testing.assert_array_equal(s._chain_accept, acc)
    np.testing.assert_allclose(s._chain_thetas, ht, rtol=RTOL, atol=1e-12)
    np.testing.assert_allclose(s._chain_logpost, hl, rtol=RTOL)


@pytest.mark.parametrize("d", [18, 36, 49, 68, 84])
def test_logreg_gradient_tail_for_every_tile_count_vs_oracle(ra, orc, d):
    """d = 16 NF + 1 .. 16 NF + 4: the last column tile of the logistic gradient on the VALU
    (LogRegModel<NT, true>, round 2: d = 97..100 only) -- MALA, HMC and the adapted-scale HMC against
    the oracle, and against the padded-tile kernel (kernel policy NO_TAIL)."""
    from riemann_amd._lib import kernel_policy
    from riemann_amd.models import LogisticRegression
    from riemann_amd.proposals import MALA, AdaptScaleHMC, VanillaHMC
    from riemann_amd.workloads import logreg_data
    X, y = logreg_data(700, d, seed=d)
    th0 = 0.05 * np.random.default_rng(d).standard_normal((200, d))
    m = LogisticRegression(X, y, 50.0)
    om = orc.LogReg(X, y, 50.0)
    out = {}
    for pol in ((), ("NO_TAIL",)):
        with kernel_policy(*pol):
            a = ra.Sampler(m, MALA(0.1, m), th0, seed=3)
            a.run(8)
            b = ra.Sampler(m, VanillaHMC(0.06, 4, m), th0, seed=4, chain0=16)
            b.run(5)
            c = ra.Sampler(m, AdaptScaleHMC(0.08, 2, m), th0, seed=5)
            c.run(10)
            out[pol] = [(x._chain_accept.copy(), x._chain_thetas.copy(), x._chain_logpost.copy()) for x in (a, b, c)]
    for t, gnr in zip(out[()], out[("NO_TAIL",)]):
        np.testing.assert_array_equal(t[0], gnr[0])
        np.testing.assert_allclose(t[1], gnr[1], rtol=1e-9, atol=1e-12)
    ht, hl, acc = oracle_run(orc, om, orc.HMC(0.1, 1, d), th0, 3, 8)
    np.testing.assert_array_equal(out[()][0][0], acc)
    np.testing.assert_allclose(out[()][0][1], ht, rtol=RTOL, atol=1e-12)
    np.testing.assert_allclose(out[()][0][2], hl, rtol=RTOL)
    th, st = th0.copy(), orc.rng_seed(4, 200, chain0=16)
    ht, hl, acc = orc.run(om, orc.HMC(0.06, 4, d), th, om.log_posterior(th), st, 5)
    np.testing.assert_array_equal(out[()][1][0], acc)
    np.testing.assert_allclose(out[()][1][1], ht, rtol=RTOL, atol=1e-12)
    th, st, ad = th0.copy(), orc.rng_seed(5, 200), orc.new_adapt_state(200)
    ht, hl, acc = orc.run(om, orc.set_adapt(orc.HMC(0.08, 2, d), 0.75), th, om.log_posterior(th), st, 10, adapt=ad)
    np.testing.assert_array_equal(out[()][2][0], acc)
    np.testing.assert_allclose(out[()][2][1], ht, rtol=RTOL, atol=1e-12)


def test_full_size_sharding_and_resume_invariance(ra):
    """65 536 chains (cfg2): (i) the chains of a shard started with chain0 = a
    equal chains [a, b) of the whole job -- the multi-GPU partition is exact;
    (ii) run(6) + run(6) == run(12); (iii) thinning keeps every 3rd state."""
    from riemann_amd.models import benchmarks
    from riemann_amd.proposals import MetropolisRandomWalk
    N = 65536
    Ct = corr100()
    Cp = (2.38 ** 2 / 100) * Ct
    th0 = np.random.default_rng(1).standard_normal((N, 100)) @ np.linalg.cholesky(Ct).T
    m, p = benchmarks.benchmark_gauss100d_corr, MetropolisRandomWalk(Cp)
    whole = ra.Sampler(m, p, th0, seed=1)
    whole.run(12)
    th_w, lp_w = whole.current_state()
    acc_w = whole._chain_accept.copy()
    a, b = 16384, 32768
    shard = ra.Sampler(m, p, th0[a:b], seed=1, chain0=a)
    shard.run(12)
    th_s, lp_s = shard.current_state()
    np.testing.assert_array_equal(th_s, th_w[a:b])
    np.testing.assert_array_equal(lp_s, lp_w[a:b])
    np.testing.assert_array_equal(shard._chain_accept, acc_w[:, a:b])
    two = ra.Sampler(m, p, th0, seed=1)
    two.run(6, record=False)
    two.run(6, 0, 3)
    th_t, lp_t = two.current_state()
    np.testing.assert_array_equal(th_t, th_w)
    np.testing.assert_array_equal(lp_t, lp_w)
    np.testing.assert_array_equal(two._chain_thetas, whole._chain_thetas[6::3])
    assert 0.15 < acc_w.mean() < 0.35


def test_g9_hmc_with_mass_matrix(ra):
    from riemann_amd.models import LogisticRegression, benchmarks
    from riemann_amd.proposals import VanillaHMC
    z = load("g9_gauss100d_hmc_mass")
    m = benchmarks.benchmark_gauss100d_corr
    p = VanillaHMC(float(z["eps"]), int(z["nsteps"]), m.grad_log_likelihood, M=z["M"])
    np.testing.assert_allclose(p.chM @ p.chM.T, z["M"], rtol=1e-12)
    s = ra.Sampler(m, p, z["theta0"], seed=int(z["seed"]))
    s.run(16)
    check_run(z, s)
    z = load("g9_logreg256x16_hmc_mass")
    m = LogisticRegression(z["X"], z["y"], float(z["tau2"]))
    s = ra.Sampler(m, VanillaHMC(float(z["eps"]), int(z["nsteps"]), m, M=z["M"]), z["theta0"],
                   seed=int(z["seed"]))
    s.run(32)
    check_run(z, s)


def test_hmc_mass_matrix_many_chains_vs_oracle(ra, orc):
    from riemann_amd.models import MultiGaussianDist
    from riemann_amd.proposals import VanillaHMC
    g = np.random.default_rng(11)
    for d, N in ((7, 50), (40, 130)):
        A = g.standard_normal((d, d))
        Cov = A @ A.T + d * np.eye(d)
        B = g.standard_normal((d, d))
        M = B @ B.T / d + np.eye(d)
        th0 = g.standard_normal((N, d))
        m = MultiGaussianDist(np.zeros(d), Cov)
        s = ra.Sampler(m, VanillaHMC(0.7, 3, m.grad_log_likelihood, M=M), th0, seed=d)
        s.run(20)
        om = orc.Gaussian(np.zeros(d), Cov)
        ht, hl, acc = oracle_run(orc, om, orc.HMC(0.7, 3, d, M=M), th0, d, 20)
        np.testing.assert_array_equal(s._chain_accept, acc)
        np.testing.assert_allclose(s._chain_thetas, ht, rtol=1e-8, atol=1e-10)
        assert 0.05 < acc.mean() < 1.0


def test_g10_pcn(ra, orc):
    from riemann_amd.models import MultiGaussianDist, benchmarks
    from riemann_amd.proposals import pCN
    z = load("g10_pcn_gauss100d")
    Ct = corr100()
    p = pCN(0.7 * Ct + 0.3 * np.eye(100), float(z["rho"]))
    assert p.rho_c == np.sqrt(1 - p.rho ** 2) and p.scale == 1.0
    s = ra.Sampler(benchmarks.benchmark_gauss100d_corr, p, z["theta0"], seed=int(z["seed"]))
    s.run(32)
    check_run(z, s)
    z = load("g10_pcn_gauss2d")
    s = ra.Sampler(MultiGaussianDist(z["mu"], z["C"]), pCN(z["Cprop"], float(z["rho"])), z["theta0"],
                   seed=int(z["seed"]))
    s.run(64)
    check_run(z, s)
    g = np.random.default_rng(12)                     # many chains vs the oracle
    d, N = 33, 200
    A = g.standard_normal((d, d))
    Cov = A @ A.T + d * np.eye(d)
    th0 = g.standard_normal((N, d)) * 3
    s = ra.Sampler(MultiGaussianDist(np.zeros(d), Cov), pCN(Cov, 0.9), th0, seed=5)
    s.run(25)
    ht, hl, acc = oracle_run(orc, orc.Gaussian(np.zeros(d), Cov), orc.PCN(Cov, 0.9), th0, 5, 25)
    np.testing.assert_array_equal(s._chain_accept, acc)
    np.testing.assert_allclose(s._chain_thetas, ht, rtol=RTOL, atol=1e-12)


def test_edge_cases_run_lengths_and_sizes(ra, orc):
    """Nsamples = 0, Nburn > Nsamples (empty history, then no current state, as the
    reference's empty lists), d = 1, d = 112 (largest tile), Nthin > Nsamples."""
    from riemann_amd.models import MultiGaussianDist
    from riemann_amd.proposals import MetropolisRandomWalk
    m, p = MultiGaussianDist(np.zeros(2), np.eye(2)), MetropolisRandomWalk(np.eye(2))
    s = ra.Sampler(m, p, np.ones((3, 2)), seed=1)
    s.run(0)
    assert s._chain_thetas.shape == (1, 3, 2) and s._chain_accept.shape == (0, 3)
    np.testing.assert_array_equal(s._chain_thetas[0], np.ones((3, 2)))
    s.run(5, 0, 7)                                   # [0::7] of 6 states keeps state 0 only ...
    assert s._chain_thetas.shape == (1, 3, 2)
    np.testing.assert_array_equal(s.current_state()[0], np.ones((3, 2)))   # ... which stays current
    s.run(3, 10, 1)                                  # Nburn > Nsamples: nothing kept
    assert s._chain_thetas.shape == (0, 3, 2)
    with pytest.raises(IndexError):
        s.current_state()
    with pytest.raises(IndexError):
        s.run(1)
    for d in (1, 112):
        g = np.random.default_rng(d)
        A = g.standard_normal((d, d))
        Cov = A @ A.T + d * np.eye(d)
        th0 = g.standard_normal((20, d))
        s = ra.Sampler(MultiGaussianDist(np.zeros(d), Cov), MetropolisRandomWalk(0.1 * Cov), th0, seed=d)
        s.run(15)
        ht, hl, acc = oracle_run(orc, orc.Gaussian(np.zeros(d), Cov), orc.RandomWalk(0.1 * Cov), th0, d, 15)
        np.testing.assert_array_equal(s._chain_accept, acc)
        np.testing.assert_allclose(s._chain_thetas, ht, rtol=RTOL, atol=1e-12)
    # beyond the largest register tile the logistic model is refused loudly (the Gaussian
    # random walk continues on the workgroup kernels, tested further down)
    from riemann_amd.models import LogisticRegression
    with pytest.raises(ra.ParameterError):
        LogisticRegression(np.zeros((8, 113)), np.zeros(8), 1.0)


def test_non_finite_start_states(ra, orc):
    """theta0 with inf / nan: log_posterior -> -inf (model.py:49-52), NaN ratio accepts."""
    from riemann_amd.models import MultiGaussianDist
    from riemann_amd.proposals import MetropolisRandomWalk
    Cov = np.array([[2.0, 0.3], [0.3, 1.0]])
    th0 = np.array([[np.inf, 0.0], [np.nan, 1.0], [1e200, -1e200], [0.5, 0.5]])
    s = ra.Sampler(MultiGaussianDist(np.zeros(2), Cov), MetropolisRandomWalk(np.eye(2)), th0, seed=3)
    assert list(np.isinf(s._chain_logpost[0])) == [True, True, True, False]
    s.run(6)
    ht, hl, acc = oracle_run(orc, orc.Gaussian(np.zeros(2), Cov), orc.RandomWalk(np.eye(2)), th0, 3, 6)
    np.testing.assert_array_equal(s._chain_accept, acc)
    np.testing.assert_array_equal(np.isfinite(s._chain_logpost), np.isfinite(hl))
    np.testing.assert_array_equal(np.isnan(s._chain_thetas), np.isnan(ht))


def test_rccl_pooling_entry_points_single_rank(ra):
    """rmh_comm_* / rmh_allgather_samples / rmh_allreduce_moments with nranks = 1:
    the all-gather returns this rank's states, the moments match numpy."""
    from riemann_amd import _lib
    from riemann_amd.models import MultiGaussianDist
    from riemann_amd.proposals import MetropolisRandomWalk
    g = np.random.default_rng(13)
    N, d = 1000, 20
    th0 = g.standard_normal((N, d))
    s = ra.Sampler(MultiGaussianDist(np.zeros(d), np.eye(d)), MetropolisRandomWalk(0.1 * np.eye(d)),
                   th0, seed=2)
    s.run(5, record=False)
    uid = np.zeros(128, dtype=np.uint8)
    _lib.call("rmh_comm_unique_id", uid)
    _lib.call("rmh_comm_init", uid, 1, 0)
    try:
        pooled = np.empty((N, d))
        _lib.call("rmh_allgather_samples", s._state, pooled)
        th, _ = s.current_state()
        np.testing.assert_array_equal(pooled, th)
        # the rank-count / barrier / host-reduction entries bench.py drives, and the pooled view
        from riemann_amd import distributed as D
        assert D.comm_ranks() == (1, 0)
        _lib.call("rmh_comm_barrier")
        buf = np.array([3.0, -1.0])
        _lib.call("rmh_comm_allreduce_host", buf, 2, 1)
        np.testing.assert_array_equal(buf, [3.0, -1.0])
        pl, mean, cov, info = D.pool(s)
        np.testing.assert_array_equal(pl, th)
        assert info["rccl_ranks"] == 1 and info["pooled_chains"] == N
        assert info["allgather_bytes_per_rank"] == N * d * 8 and info["allgather_ms"] >= 0.0
        np.testing.assert_allclose(mean, th.mean(axis=0), rtol=1e-11, atol=1e-13)
        np.testing.assert_allclose(cov, np.cov(th.T, bias=True), rtol=1e-9, atol=1e-12)
        mom = np.empty(1 + d + d * d)
        _lib.call("rmh_allreduce_moments", s._state, mom)
        assert mom[0] == N
        np.testing.assert_allclose(mom[1:1 + d], th.sum(axis=0), rtol=1e-12, atol=1e-10)
        np.testing.assert_allclose(mom[1 + d:].reshape(d, d), th.T @ th, rtol=1e-12, atol=1e-10)
        # with a communicator the AdaptCov updates all-reduce their moment sums (one rank: same result)
        from riemann_amd.proposals import AdaptCovRandomWalk
        m = MultiGaussianDist(np.zeros(d), np.eye(d))
        pa = AdaptCovRandomWalk(0.05 * np.eye(d), t_adapt=100, smooth_adapt=True)
        sa = ra.Sampler(m, pa, th0, seed=3)
        sa.run(30)
        with_comm = (sa._chain_thetas.copy(), pa.C.copy())
        # pooled diagnostics of a recorded run over the communicator (one rank): the device
        # reductions through RCCL == history_stats == the numpy twin of the reductions; thinned
        # samples by all-gather; pooled covariance from the device-formed moment sums
        from riemann_amd import diagnostics as dg
        s.run(120, 10, 1)
        x = s._chain_thetas
        ph = D.pool_history(s, nlags=64, samples_every=25)
        hs = s.history_stats(nlags=64)
        one = dg.pooled_history_stats(dg.history_pieces_numpy(x, 64))
        assert ph["chains"] == N
        for k in ("acf", "tau", "ess", "rhat"):
            np.testing.assert_array_equal(ph[k], hs[k], err_msg=k)
            np.testing.assert_allclose(ph[k], one[k], rtol=1e-8, atol=1e-11, err_msg=k)
        np.testing.assert_array_equal(ph["mean"], hs["pooled_mean"])
        np.testing.assert_allclose(ph["mean"], x.mean(axis=(0, 1)), rtol=1e-11, atol=1e-13)
        np.testing.assert_allclose(ph["var"], x.reshape(-1, d).var(axis=0), rtol=1e-10)
        np.testing.assert_allclose(ph["cov"], np.cov(x.reshape(-1, d).T, bias=True), rtol=1e-8, atol=1e-11)
        np.testing.assert_array_equal(ph["samples"], x[::25])
        cnt, hm, hc = s.history_moments()
        assert cnt == N * x.shape[0]
        np.testing.assert_allclose(hc, ph["cov"], rtol=1e-12, atol=1e-14)
    finally:
        _lib.call("rmh_comm_destroy")
    pb = AdaptCovRandomWalk(0.05 * np.eye(d), t_adapt=100, smooth_adapt=True)
    sb = ra.Sampler(m, pb, th0, seed=3)
    sb.run(30)
    np.testing.assert_array_equal(sb._chain_thetas, with_comm[0])
    np.testing.assert_array_equal(pb.C, with_comm[1])


def test_checkpoint_restore_continues_bit_for_bit(ra, tmp_path):
    from riemann_amd.models import benchmarks
    from riemann_amd.proposals import AdaptScaleHMC
    Ct = corr100()
    m = benchmarks.benchmark_gauss100d_corr
    th0 = np.random.default_rng(14).standard_normal((40, 100)) @ np.linalg.cholesky(Ct).T
    a = ra.Sampler(m, AdaptScaleHMC(0.2, 3, m.grad_log_likelihood), th0, seed=9)
    a.run(12)
    ref_th, ref_lp = a.current_state()
    b = ra.Sampler(m, AdaptScaleHMC(0.2, 3, m.grad_log_likelihood), th0, seed=9)
    b.run(5)
    b.checkpoint(str(tmp_path / "chains.npz"))
    c = ra.Sampler(m, AdaptScaleHMC(0.2, 3, m.grad_log_likelihood), np.zeros((40, 100)), seed=12345)
    c.restore(str(tmp_path / "chains.npz"))
    c.run(7)
    th, lp = c.current_state()
    np.testing.assert_array_equal(th, ref_th)
    np.testing.assert_array_equal(lp, ref_lp)
    np.testing.assert_array_equal(c.proposal.scale, a.proposal.scale)


def test_g11_mixture_values_and_random_walk(ra):
    from riemann_amd.models import benchmarks as b
    from riemann_amd.proposals import MetropolisRandomWalk
    z = load("g11_mixture_values")
    for name, m in (("mix1d", b.benchmark_gauss1d_mix2), ("mix2d", b.benchmark_gauss2d_mix2_cross)):
        np.testing.assert_array_equal(m._mixweights, z[name + "_weights"])
        th = z[name + "_theta"]
        for fn, key in ((m.log_likelihood, "_logL"), (m.log_prior, "_logP"), (m.log_posterior, "_logpost")):
            got, ref = fn(th), z[name + key]
            assert np.array_equal(np.isinf(got), np.isinf(ref))
            fin = np.isfinite(ref)
            np.testing.assert_allclose(got[fin], ref[fin], rtol=RTOL, atol=1e-15)
    z = load("g11_mixture_rw")
    s = ra.Sampler(b.benchmark_gauss2d_mix2_cross, MetropolisRandomWalk(z["Cprop"]), z["theta0"],
                   seed=int(z["seed"]))
    s.run(96)
    check_run(z, s)
    z = load("g11_mixture1d_rw")
    s = ra.Sampler(b.benchmark_gauss1d_mix2, MetropolisRandomWalk(z["Cprop"]), z["theta0"],
                   seed=int(z["seed"]))
    s.run(96)
    check_run(z, s)
    # gradient (the build's addition) and HMC / MALA on the mixture
    from oracle import oracle as orc
    from riemann_amd.proposals import MALA, VanillaHMC
    mx = b.benchmark_gauss2d_mix2_cross
    pts = np.random.default_rng(0).standard_normal((50, 2)) * 2.0
    np.testing.assert_allclose(mx.grad_log_posterior(pts), orc.Mixture(MIX2D).grad_log_posterior(pts), rtol=1e-9)
    for tag in ("mala", "hmc4"):
        z = load("g11_mixture_" + tag)
        L = int(z["nsteps"])
        prop = MALA(float(z["eps"]), mx) if L == 1 else VanillaHMC(float(z["eps"]), L, mx)
        s = ra.Sampler(mx, prop, z["theta0"], seed=int(z["seed"]))
        s.run(64)
        check_run(z, s)


def test_g15_gauss200d_vs_reference_fixture(ra):
    from riemann_amd.models import MultiGaussianDist
    from riemann_amd.proposals import MetropolisRandomWalk
    z = load("g15_gauss200d_rw")
    g = np.random.default_rng(15)
    A = g.standard_normal((200, 200))
    Ct = A @ A.T / 200 + 0.2 * np.eye(200)
    mu = g.standard_normal(200)
    s = ra.Sampler(MultiGaussianDist(mu, Ct), MetropolisRandomWalk((2.38 ** 2 / 200) * Ct), z["theta0"],
                   seed=int(z["seed"]))
    s.run(24)
    check_run(z, s)


@pytest.mark.parametrize("d,N", [(113, 50), (128, 33), (256, 70), (257, 16), (500, 40), (513, 20), (1024, 17)])
def test_gauss_random_walk_big_d_vs_oracle(ra, orc, d, N):
    """Sizes either side of every tile-count bucket of rmh_kernels_big.hip, ragged chain counts,
    burn/thin/resume bookkeeping, log densities through the eval kernel."""
    from riemann_amd.models import MultiGaussianDist
    from riemann_amd.proposals import MetropolisRandomWalk
    g = np.random.default_rng(d)
    A = g.standard_normal((d, d))
    Ct = A @ A.T / d + 0.2 * np.eye(d)
    mu = g.standard_normal(d)
    Cp = (2.38 ** 2 / d) * Ct
    th0 = mu + g.standard_normal((N, d))
    m = MultiGaussianDist(mu, Ct)
    om = orc.Gaussian(mu, Ct)
    np.testing.assert_allclose(m.log_posterior(th0), om.log_posterior(th0), rtol=RTOL)
    np.testing.assert_allclose(m.log_likelihood(th0[0]), om.log_likelihood(th0[:1])[0], rtol=RTOL)
    p = MetropolisRandomWalk(Cp)
    p.scale = 0.7
    s = ra.Sampler(m, p, th0, seed=9, chain0=1000)
    T = 12
    s.run(T)
    ht, hl, acc = oracle_run(orc, om, orc.RandomWalk(Cp, 0.7), th0, 9, T, chain0=1000)
    np.testing.assert_array_equal(s._chain_accept, acc)
    np.testing.assert_allclose(s._chain_thetas, ht, rtol=RTOL, atol=1e-12)
    np.testing.assert_allclose(s._chain_logpost, hl, rtol=RTOL)
    assert 0.02 < acc.mean() < 0.98
    # continuation with burn-in and thinning == the same chains run straight through
    s.run(9, 2, 3)
    ht2, hl2, _ = oracle_run(orc, om, orc.RandomWalk(Cp, 0.7), th0, 9, T + 9, chain0=1000)
    np.testing.assert_allclose(s._chain_thetas, ht2[T + 2::3], rtol=RTOL, atol=1e-12)
    th, lp = s.current_state()
    np.testing.assert_allclose(lp, hl2[T + 8], rtol=RTOL)
    with pytest.raises(ra.ParameterError):
        m.grad_log_posterior(th0)


def test_dimension_limit_is_reported(ra):
    from riemann_amd.models import MultiGaussianDist
    with pytest.raises(ra.ParameterError):
        MultiGaussianDist(np.zeros(1025), np.eye(1025))


def test_g16_the_reference_example_run(ra):
    """The reference's smoke script, examples/test_randomwalk.py:34-40, on the device: the same
    10 000 steps of AdaptScaleCovHMC with burn-in 1000, state for state."""
    from riemann_amd.models.benchmarks import benchmark_gauss2d_corr
    from riemann_amd.proposals import AdaptScaleCovHMC
    z = load("g16_example_test_randomwalk")
    proposal = AdaptScaleCovHMC(0.1, 5, benchmark_gauss2d_corr.grad_log_likelihood, np.eye(2),
                                t_adapt=100, smooth_adapt=True)
    proposal.scale = 1.0
    sampler = ra.Sampler(benchmark_gauss2d_corr, proposal, np.ones(2), seed=int(z["seed"]))
    sampler.run(10000, 1000, 1)
    chain = np.array(sampler._chain_thetas)
    assert chain.shape == (9001, 2)
    np.testing.assert_allclose(chain, z["thetas"][:, 0], rtol=1e-7, atol=1e-10)
    np.testing.assert_allclose(sampler._chain_logpost, z["logpost"][:, 0], rtol=1e-7, atol=1e-10)
    assert proposal.Naccepts == z["Naccepts"][0]
    np.testing.assert_allclose(proposal.scale, z["scale_final"][0], rtol=1e-10)


def test_chain_count_limit_is_reported(ra):
    """32-bit buffer offsets bound rows x chains: beyond that the state is refused, not corrupted."""
    from riemann_amd.models import MultiGaussianDist
    from riemann_amd import _lib
    import ctypes as C
    m = MultiGaussianDist(np.zeros(100), np.eye(100))        # 112 rows
    h = C.c_void_p()
    with pytest.raises(ra.ParameterError):
        _lib.call("rmh_state_create", m._h, 5_000_000, 0, 0, C.byref(h))


def test_run_to_disk_streams_the_same_samples(ra, tmp_path):
    """Thinned samples streamed to .npy files chunk by chunk == one in-memory run()."""
    from riemann_amd.models import MultiGaussianDist
    from riemann_amd.proposals import MetropolisRandomWalk
    g = np.random.default_rng(0)
    th0 = g.standard_normal((37, 5))
    m = MultiGaussianDist(np.zeros(5), np.eye(5))
    for T, nb, nt, chunk in ((50, 0, 1, 16), (53, 7, 4, 10), (20, 25, 2, 8), (31, 31, 3, 31), (0, 0, 1, 5)):
        a = ra.Sampler(m, MetropolisRandomWalk(0.3 * np.eye(5)), th0, seed=4)
        a.run(T, nb, nt)
        b = ra.Sampler(m, MetropolisRandomWalk(0.3 * np.eye(5)), th0, seed=4)
        kept = b.run_to_disk(str(tmp_path / "s"), T, nb, nt, chunk=chunk)
        th = np.load(str(tmp_path / "s_thetas.npy"), mmap_mode="r")
        assert kept == a._chain_thetas.shape[0] == th.shape[0]
        np.testing.assert_array_equal(th, a._chain_thetas)
        np.testing.assert_array_equal(np.load(str(tmp_path / "s_logpost.npy")), a._chain_logpost)
        np.testing.assert_array_equal(np.load(str(tmp_path / "s_accept.npy")), a._chain_accept)
        if kept:                                    # and the chains continue from the same state
            np.testing.assert_array_equal(b.current_state()[0], a.current_state()[0])
            a.run(3)
            b.run(3)
            np.testing.assert_array_equal(b._chain_thetas, a._chain_thetas)
        else:
            with pytest.raises(IndexError):
                b.current_state()


@pytest.mark.parametrize("d", [96, 97, 98, 99, 100, 101, 16, 17, 20, 21, 33, 36, 50, 52, 65, 67, 81, 84, 85])
def test_gauss_random_walk_around_the_tail_kernel_sizes(ra, orc, d):
    """d = 16 NF + 1 .. 16 NF + 4 (NF = 1..6) run on k_run_rw_tail<NF> (the four rows of the last tile
    on the VALU), the neighbours on the generic kernel: all against the oracle, with a changed
    .scale re-packing both images."""
    from riemann_amd.models import MultiGaussianDist
    from riemann_amd.proposals import MetropolisRandomWalk
    g = np.random.default_rng(d)
    A = g.standard_normal((d, d))
    Ct = A @ A.T / d + 0.3 * np.eye(d)
    mu = g.standard_normal(d)
    Cp = (2.38 ** 2 / d) * Ct
    th0 = mu + g.standard_normal((70, d))
    m, om = MultiGaussianDist(mu, Ct), orc.Gaussian(mu, Ct)
    p = MetropolisRandomWalk(Cp)
    s = ra.Sampler(m, p, th0, seed=2)
    s.run(10)
    ht, hl, acc = oracle_run(orc, om, orc.RandomWalk(Cp), th0, 2, 10)
    np.testing.assert_array_equal(s._chain_accept, acc)
    np.testing.assert_allclose(s._chain_thetas, ht, rtol=RTOL, atol=1e-12)
    np.testing.assert_allclose(s._chain_logpost, hl, rtol=RTOL)
    p.scale = 0.5
    s2 = ra.Sampler(m, p, th0, seed=3)
    s2.run(10)
    ht, hl, acc = oracle_run(orc, om, orc.RandomWalk(Cp, 0.5), th0, 3, 10)
    np.testing.assert_array_equal(s2._chain_accept, acc)
    np.testing.assert_allclose(s2._chain_thetas, ht, rtol=RTOL, atol=1e-12)


@pytest.mark.parametrize("d", [18, 35, 49, 68, 83, 99])
def test_tail_kernel_variants_vs_oracle_and_generic_kernel(ra, orc, d):
    """Every k_run_rw_tail<NF> instantiation in its three variants: the per-chain adapted scale and
    parallel tempering against the oracle, and all three against the generic padded-tile kernel
    (kernel policy NO_TAIL) -- same stream, same accept decisions, states to the rounding of the
    four tail rows' summation order."""
    from riemann_amd._lib import kernel_policy
    from riemann_amd.models import MultiGaussianDist
    from riemann_amd.proposals import AdaptScaleRandomWalk, MetropolisRandomWalk
    from riemann_amd.samplers import PTSampler
    g = np.random.default_rng(1000 + d)
    A = g.standard_normal((d, d))
    Ct = A @ A.T / d + 0.3 * np.eye(d)
    mu = g.standard_normal(d)
    Cp = (2.38 ** 2 / d) * Ct
    th0 = mu + g.standard_normal((150, d))
    m, om = MultiGaussianDist(mu, Ct), orc.Gaussian(mu, Ct)
    out = {}
    for pol in ((), ("NO_TAIL",)):
        with kernel_policy(*pol):
            a = ra.Sampler(m, MetropolisRandomWalk(Cp), th0, seed=5, chain0=32)
            a.run(12, 2, 2)
            b = ra.Sampler(m, AdaptScaleRandomWalk(Cp), th0, seed=6)
            b.run(25)
            pt = PTSampler(m, MetropolisRandomWalk(Cp), th0[:40], betas=0.6 ** np.arange(4), Pswap=0.3, seed=7)
            pt.run(20)
            out[pol] = (a._chain_accept.copy(), a._chain_thetas.copy(), b._chain_accept.copy(), b._chain_thetas.copy(),
                        np.asarray(b.proposal.scale).copy(), pt._host_codes.copy(), pt._host_thetas.copy())
    t, gnr = out[()], out[("NO_TAIL",)]
    for k in (0, 2, 5):
        np.testing.assert_array_equal(t[k], gnr[k])
    for k in (1, 3, 4, 6):
        np.testing.assert_allclose(t[k], gnr[k], rtol=1e-9, atol=1e-12)
    # adapted scale against the oracle
    th, st, ad = th0.copy(), orc.rng_seed(6, 150), orc.new_adapt_state(150)
    ht, hl, acc = orc.run(om, orc.set_adapt(orc.RandomWalk(Cp), 0.25), th, om.log_posterior(th), st, 25, adapt=ad)
    np.testing.assert_array_equal(t[2], acc)
    np.testing.assert_allclose(t[3], ht, rtol=RTOL, atol=1e-12)


@pytest.mark.parametrize("d", [16, 32, 48, 64, 80, 96, 112])
def test_exact_multiple_of_16_kernels_vs_oracle_and_generic_kernel(ra, orc, d):
    """d = 16 NT: k_run_rw<..., FULL> (Gaussian random walk, plain and adapted scale) and
    LogRegModel<NT, false, FULL> (MALA, HMC with and without a mass matrix, adapted scale) carry the k-step
    and word counts as compile-time constants.  Same arithmetic in the same order as the generic kernels
    (kernel policy NO_TAIL): bit-identical to them, and against the oracle."""
    from riemann_amd._lib import kernel_policy
    from riemann_amd.models import LogisticRegression, MultiGaussianDist
    from riemann_amd.proposals import MALA, AdaptScaleHMC, AdaptScaleRandomWalk, MetropolisRandomWalk, VanillaHMC, pCN
    from riemann_amd.samplers import PTSampler
    from riemann_amd.workloads import logreg_data
    g = np.random.default_rng(2000 + d)
    A = g.standard_normal((d, d))
    Ct = A @ A.T / d + 0.3 * np.eye(d)
    mu = g.standard_normal(d)
    Cp = (2.38 ** 2 / d) * Ct
    B = g.standard_normal((d, d))
    Mm = B @ B.T / d + np.eye(d)
    th0 = mu + g.standard_normal((150, d))
    gm, om = MultiGaussianDist(mu, Ct), orc.Gaussian(mu, Ct)
    X, y = logreg_data(300, d, seed=d)
    lm, olm = LogisticRegression(X, y, 50.0), orc.LogReg(X, y, 50.0)
    tl0 = 0.05 * g.standard_normal((150, d))
    out = {}
    for pol in ((), ("NO_TAIL",)):
        with kernel_policy(*pol):
            runs = [ra.Sampler(gm, MetropolisRandomWalk(Cp), th0, seed=5, chain0=32),
                    ra.Sampler(gm, AdaptScaleRandomWalk(Cp), th0, seed=6),
                    ra.Sampler(lm, MALA(0.1, lm), tl0, seed=7),
                    ra.Sampler(lm, VanillaHMC(0.06, 3, lm, M=Mm), tl0, seed=8),
                    ra.Sampler(lm, AdaptScaleHMC(0.08, 2, lm), tl0, seed=9)]
            for s, T in zip(runs, (12, 25, 8, 5, 10)):
                s.run(T)
            out[pol] = [(s._chain_accept.copy(), s._chain_thetas.copy(), s._chain_logpost.copy()) for s in runs]
            # pCN and the tempered walk on the Gaussian model (k_run_rw<..., PCN / PT, ..., FULL>)
            pc = ra.Sampler(gm, pCN(1.5 * Ct, 0.95), th0, seed=10)
            pc.run(10)
            pt = PTSampler(gm, MetropolisRandomWalk(Cp), th0[:40], betas=0.6 ** np.arange(4), Pswap=0.3, seed=11)
            pt.run(15)
            out[pol].append((pc._chain_accept.copy(), pc._chain_thetas.copy(), pc._chain_logpost.copy()))
            out[pol].append((pt._host_codes.copy(), pt._host_thetas.copy(), pt._host_logpost.copy()))
    for t, gnr in zip(out[()], out[("NO_TAIL",)]):
        for a, b in zip(t, gnr):
            np.testing.assert_array_equal(a, b)
    ht, hl, acc = oracle_run(orc, om, orc.RandomWalk(Cp), th0, 5, 12, chain0=32)
    np.testing.assert_array_equal(out[()][0][0], acc)
    np.testing.assert_allclose(out[()][0][1], ht, rtol=RTOL, atol=1e-12)
    np.testing.assert_allclose(out[()][0][2], hl, rtol=RTOL)
    th, st, ad = th0.copy(), orc.rng_seed(6, 150), orc.new_adapt_state(150)
    ht, hl, acc = orc.run(om, orc.set_adapt(orc.RandomWalk(Cp), 0.25), th, om.log_posterior(th), st, 25, adapt=ad)
    np.testing.assert_array_equal(out[()][1][0], acc)
    np.testing.assert_allclose(out[()][1][1], ht, rtol=RTOL, atol=1e-12)
    ht, hl, acc = oracle_run(orc, olm, orc.HMC(0.1, 1, d), tl0, 7, 8)
    np.testing.assert_array_equal(out[()][2][0], acc)
    np.testing.assert_allclose(out[()][2][1], ht, rtol=RTOL, atol=1e-12)
    np.testing.assert_allclose(out[()][2][2], hl, rtol=RTOL)
    ht, hl, acc = oracle_run(orc, olm, orc.HMC(0.06, 3, d, M=Mm), tl0, 8, 5)
    np.testing.assert_array_equal(out[()][3][0], acc)
    np.testing.assert_allclose(out[()][3][1], ht, rtol=1e-8, atol=1e-10)
    ht, hl, acc = oracle_run(orc, om, orc.PCN(1.5 * Ct, 0.95), th0, 10, 10)
    np.testing.assert_array_equal(out[()][5][0], acc)
    np.testing.assert_allclose(out[()][5][1], ht, rtol=RTOL, atol=1e-12)


@pytest.mark.parametrize("d", [16, 20, 36, 48, 52, 68, 84, 96, 100, 112])
def test_hmc_kernels_with_compile_time_kstep_counts_vs_generic_kernel_and_oracle(ra, orc, d):
    """k_run_hmc<..., KSC> (launch_hmc_ks): Gaussian MALA / HMC with a mass matrix / adapted-scale HMC at d = 16 NT and
    d = 16 (NT - 1) + 1 .. 4, mixture MALA at the latter -- bit-identical to the generic kernels (kernel policy NO_TAIL),
    MALA against the oracle."""
    from riemann_amd._lib import kernel_policy
    from riemann_amd.models import MixtureDist, MultiGaussianDist
    from riemann_amd.proposals import MALA, AdaptScaleHMC, VanillaHMC
    g = np.random.default_rng(3000 + d)
    A = g.standard_normal((d, d))
    Ct = A @ A.T / d + 0.3 * np.eye(d)
    B = g.standard_normal((d, d))
    Mm = B @ B.T / d + np.eye(d)
    mu = g.standard_normal(d)
    th0 = mu + g.standard_normal((150, d))
    gm, om = MultiGaussianDist(mu, Ct), orc.Gaussian(mu, Ct)
    comps = [(0.4, mu - 0.3, Ct), (0.6, mu + 0.3, 1.5 * Ct)]
    mm = MixtureDist([[w, MultiGaussianDist(m_, C_)] for w, m_, C_ in comps])
    out = {}
    for pol in ((), ("NO_TAIL",)):
        with kernel_policy(*pol):
            runs = [ra.Sampler(gm, MALA(0.3, gm), th0, seed=5, chain0=16),
                    ra.Sampler(gm, VanillaHMC(0.2, 3, gm, M=Mm), th0, seed=6),
                    ra.Sampler(gm, AdaptScaleHMC(0.2, 2, gm), th0, seed=7),
                    ra.Sampler(mm, MALA(0.3, mm), th0, seed=8)]
            for s, T in zip(runs, (10, 6, 12, 8)):
                s.run(T)
            out[pol] = [(s._chain_accept.copy(), s._chain_thetas.copy(), s._chain_logpost.copy()) for s in runs]
    for t, gnr in zip(out[()], out[("NO_TAIL",)]):
        for a, b in zip(t, gnr):
            np.testing.assert_array_equal(a, b)
    ht, hl, acc = oracle_run(orc, om, orc.HMC(0.3, 1, d), th0, 5, 10, chain0=16)
    np.testing.assert_array_equal(out[()][0][0], acc)
    np.testing.assert_allclose(out[()][0][1], ht, rtol=1e-8, atol=1e-10)
    assert 0.05 < acc.mean() <= 1.0
    ht, hl, acc = oracle_run(orc, orc.Mixture(comps), orc.HMC(0.3, 1, d), th0, 8, 8)
    np.testing.assert_array_equal(out[()][3][0], acc)
    np.testing.assert_allclose(out[()][3][1], ht, rtol=1e-8, atol=1e-10)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["rw", "mala", "cp", "adaptcov", "bigd"])
def test_unrecorded_run_ends_in_the_last_kept_state(ra, kind):
    """Sampler.run leaves the chain at the LAST KEPT entry (sampler.py:49-54) whether or not a
    history is recorded: run(T, Nburn, Nthin, record=False) with (T - Nburn) % Nthin != 0 must
    end where the recorded run ends, and continue identically."""
    from riemann_amd.models import LogisticRegression, MultiGaussianDist
    from riemann_amd.proposals import MALA, AdaptCovRandomWalk, MetropolisRandomWalk
    from riemann_amd.workloads import logreg_data
    g = np.random.default_rng(21)

    def make():
        if kind == "cp":
            z = load("g5_changepoint_rw")
            return cp_model(z), scaled_rw(z), np.repeat(z["theta0"], 5, axis=0)
        if kind == "mala":
            X, y = logreg_data(300, 10, seed=4)
            m = LogisticRegression(X, y, 10.0)
            return m, MALA(0.2, m), np.zeros((50, 10))
        d = 130 if kind == "bigd" else 10
        m = MultiGaussianDist(np.zeros(d), np.eye(d))
        th0 = np.random.default_rng(22).standard_normal((40, d))
        if kind == "adaptcov":
            return m, AdaptCovRandomWalk(0.1 * np.eye(d), t_adapt=10, smooth_adapt=True), th0
        return m, MetropolisRandomWalk(0.2 * np.eye(d)), th0

    del g
    out = []
    for record in (True, False):
        m, p, th0 = make()
        s = ra.Sampler(m, p, th0, seed=9)
        s.run(11, 2, 4, record=record)          # states 2, 6, 10 kept: the chain ends at state 10 of 11
        th, lp = s.current_state()
        if record:
            np.testing.assert_array_equal(th, s._chain_thetas[-1])
        s.run(5)                                 # and continues from there
        out.append((th, lp, s._chain_thetas.copy(), s._chain_accept.copy()))
    for a, b in zip(out[0], out[1]):
        np.testing.assert_array_equal(a, b)


@pytest.mark.gpu
def test_cached_gradient_is_dropped_when_another_proposal_moves_the_chain(ra, orc):
    """HMC caches the gradient at the current state; a random-walk run in between moves the state,
    so the next HMC run must start from a fresh gradient (hamiltonian.py:27 evaluates it anew)."""
    from riemann_amd.models import MultiGaussianDist
    from riemann_amd.proposals import MetropolisRandomWalk, VanillaHMC
    d, N = 12, 48
    g = np.random.default_rng(5)
    A = g.standard_normal((d, d))
    Cov = A @ A.T + d * np.eye(d)
    th0 = g.standard_normal((N, d))
    m = MultiGaussianDist(np.zeros(d), Cov)
    hmc, rw = VanillaHMC(0.3, 3, m), MetropolisRandomWalk(0.3 * np.eye(d))
    s = ra.Sampler(m, hmc, th0, seed=17)
    om = orc.Gaussian(np.zeros(d), Cov)
    oh, orw = orc.HMC(0.3, 3, d), orc.RandomWalk(0.3 * np.eye(d))
    th, st = th0.copy(), orc.rng_seed(17, N)
    lp = om.log_posterior(th)
    for prop, oprop, T in ((hmc, oh, 4), (rw, orw, 3), (hmc, oh, 4), (rw, orw, 2), (hmc, oh, 3)):
        s.proposal = prop
        s.run(T)
        ht, hl, acc = orc.run(om, oprop, th, lp, st, T)
        np.testing.assert_array_equal(s._chain_accept, acc)
        np.testing.assert_allclose(s._chain_thetas, ht, rtol=RTOL, atol=1e-12)
        np.testing.assert_allclose(s._chain_logpost, hl, rtol=RTOL)


def test_history_stats_on_the_device_vs_numpy(ra):
    """rmh_history_reduce (per-chain mean / variance, chain-averaged autocorrelation functions)
    against numpy on the downloaded history, and the tau_int derived from it against the FFT
    estimator of riemann_amd/diagnostics.py."""
    from riemann_amd import diagnostics as dg
    from riemann_amd.models import MultiGaussianDist
    from riemann_amd.proposals import MetropolisRandomWalk
    g = np.random.default_rng(3)
    d, N, T = 7, 333, 1500
    A = g.standard_normal((d, d))
    Cov = A @ A.T + d * np.eye(d)
    s = ra.Sampler(MultiGaussianDist(np.zeros(d), Cov), MetropolisRandomWalk(2.38 ** 2 * Cov / d),
                   g.standard_normal((N, d)), seed=12)
    s.run(T, 50, 1)
    st = s.history_stats(nlags=320)
    x = s._chain_thetas                                   # (n, N, d)
    n = x.shape[0]
    np.testing.assert_allclose(st["mean"], x.mean(axis=0), rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(st["var"], x.var(axis=0), rtol=1e-10)
    xc = x - x.mean(axis=0)
    ref = np.empty((320, d))
    a0 = np.sum(xc * xc, axis=0)
    for k in range(320):
        ref[k] = np.mean(np.sum(xc[: n - k] * xc[k:], axis=0) / a0, axis=0)
    np.testing.assert_allclose(st["acf"], ref, rtol=1e-9, atol=1e-12)
    tau = dg.integrated_time(x)
    assert st["closed"].all()
    np.testing.assert_allclose(st["tau"], tau, rtol=1e-8)
    assert np.all(st["ess"] > 0)
    np.testing.assert_allclose(st["rhat"], dg.split_rhat(x), rtol=1e-9)
    assert np.all(np.abs(st["rhat"] - 1.0) < 0.05)
    s.run(5)                       # six kept states: halves of three, shorter than any lag group
    st = s.history_stats(nlags=32)
    np.testing.assert_allclose(st["rhat"], dg.split_rhat(s._chain_thetas), rtol=1e-9)
    np.testing.assert_allclose(st["mean"], s._chain_thetas.mean(axis=0), rtol=1e-12, atol=1e-13)
