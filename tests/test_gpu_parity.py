"""GPU parity tests: the HIP path (through the C ABI / the plugin API) against
the golden vectors recorded from the reference and against the CPU oracle on
the same seeded inputs.

Bars (BASELINE.json north_star): accept masks bit-exact; log-densities and
states within 1e-6 relative -- asserted here at RTOL = 1e-9.  The random
stream is asserted BIT-exact.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(__file__), "golden")
RTOL = 1e-9


def load(name):
    return np.load(os.path.join(G, name + ".npz"))


@pytest.fixture(scope="module")
def ra():
    import riemann_amd
    from riemann_amd import _lib
    _lib.call("rmh_init", 0)
    return riemann_amd


@pytest.fixture(scope="module")
def orc():
    from oracle import oracle
    return oracle


def corr100():
    return 0.1 * np.eye(100) + 0.9 * np.ones((100, 100))


def check_run(z, s, sfx=""):
    np.testing.assert_array_equal(s._chain_accept, z["accept" + sfx])
    np.testing.assert_allclose(s._chain_thetas, z["thetas" + sfx], rtol=RTOL, atol=1e-12)
    ref, got = z["logpost" + sfx], s._chain_logpost
    assert np.array_equal(np.isinf(got), np.isinf(ref))
    fin = np.isfinite(ref)
    np.testing.assert_allclose(got[fin], ref[fin], rtol=RTOL)


# ------------------------------------------------------------ building blocks
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


# ------------------------------------------------ model evaluations vs oracle
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
        th = g.standard_normal((40, d)) * np.array([0.1, 1.0, 5.0, 30.0])[g.integers(0, 4, 40)][:, None]
        np.testing.assert_allclose(m.log_posterior(th), o.log_posterior(th), rtol=RTOL)
        np.testing.assert_allclose(m.log_prior(th), o.log_prior(th), rtol=RTOL)
        np.testing.assert_allclose(m.log_likelihood(th), o.log_likelihood(th), rtol=RTOL)
        np.testing.assert_allclose(m.grad_log_posterior(th), o.grad_log_posterior(th),
                                   rtol=1e-8, atol=1e-9)


@pytest.mark.parametrize("name", ["g5_changepoint_eval", "g5_changepoint_eval_a25"])
def test_changepoint_eval_vs_reference_fixture(ra, name):
    from riemann_amd.models import ChangepointRegression1D
    z = load(name)
    m = ChangepointRegression1D(z["x"], z["y"], float(z["xmin"]), float(z["xmax"]), float(z["lamb"]),
                                2 * int(z["k"]), float(z["alpha"]), float(z["beta"]), k=int(z["k"]))
    th = z["theta"]
    for fn, key in ((m.log_likelihood, "logL"), (m.log_prior, "logP"), (m.log_posterior, "logpost")):
        got, ref = fn(th), z[key]
        assert np.array_equal(np.isinf(got), np.isinf(ref)), key
        assert np.array_equal(np.sign(got[np.isinf(ref)]), np.sign(ref[np.isinf(ref)])), key
        fin = np.isfinite(ref)
        np.testing.assert_allclose(got[fin], ref[fin], rtol=RTOL)


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


# ------------------------------------------- fixtures recorded from the reference
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


def cp_model(z):
    from riemann_amd.models import ChangepointRegression1D
    return ChangepointRegression1D(z["x"], z["y"], float(z["xmin"]), float(z["xmax"]),
                                   float(z["lamb"]), 2 * int(z["k"]), float(z["alpha"]),
                                   float(z["beta"]), k=int(z["k"]))


def scaled_rw(z):
    from riemann_amd.proposals import MetropolisRandomWalk
    p = MetropolisRandomWalk(np.diag(z["Cdiag"]))
    p.scale = float(z["scale"])
    return p


def test_g5_changepoint_rw(ra):
    z = load("g5_changepoint_rw")
    s = ra.Sampler(cp_model(z), scaled_rw(z), z["theta0"], seed=int(z["seed"]))
    s.run(64)
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


# ------------------------------------------ larger runs against the CPU oracle
def oracle_run(orc, model, prop, th0, seed, T, chain0=0):
    th = np.ascontiguousarray(th0).copy()
    st = orc.rng_seed(seed, th.shape[0], chain0=chain0)
    lp = model.log_posterior(th)
    return orc.run(model, prop, th, lp, st, T)


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


def test_changepoint_rw_4096_chains(ra, orc):
    from riemann_amd.models import ChangepointRegression1D
    from riemann_amd.proposals import MetropolisRandomWalk
    from riemann_amd.workloads import changepoint_data
    x, y, truth, cdiag = changepoint_data()
    g = np.random.default_rng(5)
    th0 = truth + 0.01 * g.standard_normal((4096, 12))
    th0[:, :5] = np.sort(th0[:, :5], axis=1)
    p = MetropolisRandomWalk(np.diag(cdiag))
    p.scale = 0.1
    s = ra.Sampler(ChangepointRegression1D(x, y, 1.0, 3.0, 5.0, 10, 1, 1, k=5), p, th0, seed=3)
    s.run(60)
    ht, hl, acc = oracle_run(orc, orc.Changepoint(x, y, 5, 1.0, 3.0, 5.0, 1.0, 1.0),
                             orc.RandomWalk(np.diag(cdiag), 0.1), th0, 3, 60)
    np.testing.assert_array_equal(s._chain_accept, acc)
    np.testing.assert_allclose(s._chain_thetas, ht, rtol=RTOL, atol=1e-12)
    fin = np.isfinite(hl)
    assert np.array_equal(np.isfinite(s._chain_logpost), fin)
    np.testing.assert_allclose(s._chain_logpost[fin], hl[fin], rtol=RTOL)


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


# --------------------------------- size-independent properties at full chain count
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


# ----------------------------- scale adaptation on the device
def _adapt_gpu_check(ra, z, model, prop, T):
    s = ra.Sampler(model, prop, z["theta0"], seed=int(z["seed"]))
    s.run(T)
    check_run(z, s)
    np.testing.assert_array_equal(prop.Nsamples, z["ad_n"])
    np.testing.assert_array_equal(prop.Naccepts, z["ad_acc"])
    np.testing.assert_allclose(prop.scale, z["ad_scale"], rtol=1e-12)
    np.testing.assert_allclose(prop.accept_rate, z["ad_rate"], rtol=1e-15)
    return s


def test_g8_adapt_scale_random_walk(ra):
    from riemann_amd.models import MultiGaussianDist, benchmarks
    from riemann_amd.proposals import AdaptScaleRandomWalk
    z = load("g8_adapt_rw_gauss2d")
    s = _adapt_gpu_check(ra, z, MultiGaussianDist(z["mu"], z["C"]), AdaptScaleRandomWalk(z["Cprop"]), 200)
    s.run(10)                                  # adaptation state persists across run() calls
    assert np.all(s.proposal.Nsamples == 210)
    z = load("g8_adapt_rw_gauss100d")
    _adapt_gpu_check(ra, z, benchmarks.benchmark_gauss100d_corr,
                     AdaptScaleRandomWalk(float(z["prop_factor"]) * corr100()), 64)


def test_g8_adapt_scale_hmc(ra):
    from riemann_amd.models import benchmarks
    from riemann_amd.proposals import AdaptScaleHMC
    z = load("g8_adapt_hmc_gauss100d")
    m = benchmarks.benchmark_gauss100d_corr
    _adapt_gpu_check(ra, z, m, AdaptScaleHMC(float(z["eps0"]), int(z["nsteps"]), m.grad_log_likelihood), 40)


def test_adapt_scale_many_chains_vs_oracle(ra, orc):
    from riemann_amd.models import LogisticRegression
    from riemann_amd.proposals import AdaptScaleHMC
    from riemann_amd.workloads import logreg_data
    X, y = logreg_data(512, 24, seed=3)
    th0 = 0.1 * np.random.default_rng(7).standard_normal((300, 24))
    m = LogisticRegression(X, y, 100.0)
    p = AdaptScaleHMC(0.3, 3, m)
    s = ra.Sampler(m, p, th0, seed=21)
    s.run(30)
    om = orc.LogReg(X, y, 100.0)
    th, ad = th0.copy(), orc.new_adapt_state(300)
    ht, hl, acc = orc.run(om, orc.set_adapt(orc.HMC(0.3, 3, 24), 0.75), th, om.log_posterior(th),
                          orc.rng_seed(21, 300), 30, adapt=ad)
    np.testing.assert_array_equal(s._chain_accept, acc)
    np.testing.assert_allclose(s._chain_thetas, ht, rtol=RTOL, atol=1e-12)
    np.testing.assert_allclose(p.scale, ad[:, 2], rtol=1e-12)


# ---------------------------------------------- HMC with a mass matrix (a-7)
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


# ------------------------------------------------------- pCN
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


# ------------------------------------------------------------ edge cases
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


# ------------------------------------- C-ABI pooling over RCCL (world of one GPU)
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


# ------------------------------------------- mixture targets, parallel tempering
MIX2D = [(0.3, [0.0, 0.0], [[1.0, 0.9], [0.9, 1.0]]),
         (0.7, [0.0, 0.0], [[1.0, -0.9], [-0.9, 1.0]])]


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


def check_pt(z, pt):
    idx = z["slots"]
    th = np.transpose(pt._host_thetas, (0, 1, 2, 3))          # (n, Ne, K, d)
    np.testing.assert_allclose(th, z["thetas"][:, idx], rtol=RTOL, atol=1e-12)
    np.testing.assert_allclose(pt._host_logpost, z["logpost"][:, idx], rtol=RTOL)
    code = pt._host_codes
    sp, sa, wa = z["swap_prop"][:, idx], z["swap_acc"][:, idx], z["within_acc"][:, idx]
    np.testing.assert_array_equal(code >> 1, sp)
    sw = sp == 1
    known = sa >= 0
    np.testing.assert_array_equal((code & 1)[sw & known], sa[sw & known])
    np.testing.assert_array_equal((code & 1)[~sw], wa[~sw])


def test_g12_parallel_tempering_vs_reference_fixture(ra):
    from riemann_amd.models import MultiGaussianDist, benchmarks as b
    from riemann_amd.proposals import MetropolisRandomWalk
    from riemann_amd.samplers import PTSampler
    z = load("g12_pt_gauss2d")
    pt = PTSampler(MultiGaussianDist(z["mu"], z["C"]), MetropolisRandomWalk(z["Cprop"]), z["theta0"],
                   Pswap=float(z["pswap"]), seed=int(z["seed"]))
    np.testing.assert_array_equal(pt.betas, z["betas"])
    pt.run(40)
    pt.run(56)          # the histories continue across run() calls
    check_pt(z, pt)
    assert pt._chain_thetas.shape == (97, 4, 2)
    z = load("g12_pt_mix2d")
    pt = PTSampler(b.benchmark_gauss2d_mix2_cross, MetropolisRandomWalk(z["Cprop"]), z["theta0"],
                   Pswap=float(z["pswap"]), seed=int(z["seed"]))
    for _ in range(3):
        pt.sample()
    pt.run(93)
    check_pt(z, pt)
    # a single ensemble keeps the reference's shapes; rung views expose the tempered models
    pt1 = PTSampler(b.benchmark_gauss2d_mix2_cross, MetropolisRandomWalk(z["Cprop"]), z["theta0"][0],
                    Pswap=0.1, seed=int(z["seed"]))
    pt1.run(10)
    assert pt1._chain_thetas.shape == (11, 2) and pt1._chain_logpost.shape == (11,)
    np.testing.assert_allclose(pt1._chain_thetas, z["thetas"][:11, z["slots"][0, 0]], rtol=RTOL, atol=1e-12)
    th, lp = pt1.samplers[2].current_state()
    np.testing.assert_allclose(lp, pt1.samplers[2].model.log_posterior(th), rtol=RTOL)
    with pytest.raises(ra.ParameterError):
        PTSampler(b.benchmark_gauss2d_mix2_cross, MetropolisRandomWalk(z["Cprop"]), z["theta0"], Pswap=1.5)
    with pytest.raises(ra.ParameterError):
        PTSampler(b.benchmark_gauss2d_mix2_cross, MetropolisRandomWalk(z["Cprop"]), z["theta0"], betas=[1.0, 2.0])


@pytest.mark.parametrize("K,d", [(5, 100), (4, 100), (16, 3), (3, 40), (1, 2)])
def test_parallel_tempering_many_ensembles_vs_oracle(ra, orc, K, d):
    """Thousands of ladders against the oracle: states, tempered log posteriors, codes."""
    from riemann_amd.models import MultiGaussianDist
    from riemann_amd.proposals import MetropolisRandomWalk
    from riemann_amd.samplers import PTSampler
    g = np.random.default_rng(K * 100 + d)
    A = g.standard_normal((d, d))
    Cm = A @ A.T / d + 0.5 * np.eye(d)
    mu = g.standard_normal(d)
    Cp = (2.38 ** 2 / d) * Cm
    Ne, T = 700, 24
    betas = 0.6 ** np.arange(K)
    th0 = g.standard_normal((Ne, d))
    pt = PTSampler(MultiGaussianDist(mu, Cm), MetropolisRandomWalk(Cp), th0, betas=betas, Pswap=0.25,
                   seed=5, chain0=4096)
    pt.run(T)
    S, idx = orc.pt_slots(Ne, K)
    slots = np.zeros((S, d))
    slots[idx] = th0[:, None, :]
    st = orc.rng_seed(5, S, chain0=4096)
    ht, hl, code, _ = orc.run_pt(orc.Gaussian(mu, Cm), orc.RandomWalk(Cp), slots, st, T, betas, 0.25)
    np.testing.assert_array_equal(pt._host_codes, code[:, idx])
    np.testing.assert_allclose(pt._host_thetas, ht[:, idx], rtol=RTOL, atol=1e-12)
    np.testing.assert_allclose(pt._host_logpost, hl[:, idx], rtol=RTOL)
    if K > 1:
        assert (code[:, idx] == 3).sum() > 100 and (code[:, idx] == 2).sum() > 10
    # resumable stream: the slots' stream states agree with the oracle's afterwards
    rng = np.empty((S, 16), dtype=np.uint64)
    from riemann_amd import _lib
    _lib.call("rmh_state_get_rng", pt._state, rng)
    np.testing.assert_array_equal(rng[idx.ravel()], st[idx.ravel()])


def test_parallel_tempering_shards_by_tile(ra):
    """Ladders split over two 'ranks' at a tile boundary (chain0 = first global slot) give
    the same chains as one rank: streams are keyed by global slot id."""
    from riemann_amd.models import benchmarks as b
    from riemann_amd.proposals import MetropolisRandomWalk
    from riemann_amd.samplers import PTSampler
    m, Cp = b.benchmark_gauss2d_mix2_cross, 1.5 * np.eye(2)
    th0 = np.random.default_rng(3).standard_normal((96, 2))
    one = PTSampler(m, MetropolisRandomWalk(Cp), th0, seed=9)
    one.run(50)
    parts = []
    for r in range(2):          # 48 ladders of 5 = 16 tiles = 256 slots per rank
        p = PTSampler(m, MetropolisRandomWalk(Cp), th0[48 * r:48 * (r + 1)], seed=9, chain0=256 * r)
        p.run(50)
        parts.append(p)
    np.testing.assert_array_equal(one._chain_thetas, np.concatenate([p._chain_thetas for p in parts], axis=1))
    np.testing.assert_array_equal(one._host_codes, np.concatenate([p._host_codes for p in parts], axis=1))
    # the cold rung mixes between the two crossed components: both signs of x*y are visited
    xy = one._chain_thetas[10:, :, 0] * one._chain_thetas[10:, :, 1]
    assert (xy > 0.2).mean() > 0.1 and (xy < -0.2).mean() > 0.2


# ---------------------------------- a-11 (B): the reference's block-mixture proposal
def test_g13_changepoint_block_proposal_vs_reference_fixture(ra):
    from riemann_amd.proposals import ChangepointRegression1DProp
    z = load("g13_changepoint_block")
    m = cp_model(z)
    p = ChangepointRegression1DProp(m, float(z["hscale"]))
    assert (p.P_cumprop_cpx, p.P_cumprop_cpv, p.P_cumprop_sig, p.P_cumprop_dim) == (0.2, 0.4, 0.6, 1.0)
    np.testing.assert_allclose(np.diag(p.L) ** 2, z["Cdiag"], rtol=1e-14)
    p.P_cumprop_cpv = 1.0        # as the fixture: cpx with 0.2, else cpv
    s = ra.Sampler(m, p, z["theta0"], seed=int(z["seed"]))
    s.run(160)
    check_run(z, s)


def test_changepoint_block_proposal_many_chains_vs_oracle(ra, orc):
    """Default probabilities (0.2 / 0.32 / 0.288, else the null move) on 4096 chains."""
    from riemann_amd.proposals import ChangepointRegression1DProp
    from riemann_amd import workloads as W
    x, y, truth, cdiag = W.changepoint_data()
    k, d, N, T = 5, 12, 4096, 60
    from riemann_amd.models import ChangepointRegression1D
    m = ChangepointRegression1D(x, y, 1.0, 3.0, 5.0, 10, 1, 1, k=k)
    p = ChangepointRegression1DProp(m, 2.0)
    p.scale = 0.5
    g = np.random.default_rng(8)
    th0 = truth + 0.01 * g.standard_normal((N, d))
    th0[:, :k] = np.sort(th0[:, :k], axis=1)
    s = ra.Sampler(m, p, th0, seed=21, chain0=77)
    s.run(T)
    om = orc.Changepoint(x, y, k, 1.0, 3.0, 5.0, 1.0, 1.0)
    op = orc.BlockRandomWalk(np.diag(np.diag(p.L) ** 2), [k, k + 1, 1], [0.2, 0.4, 0.6], scale=0.5)
    th = th0.copy()
    st = orc.rng_seed(21, N, chain0=77)
    lp = om.log_posterior(th)
    ht, hl, acc = orc.run(om, op, th, lp, st, T)
    np.testing.assert_array_equal(s._chain_accept, acc)
    np.testing.assert_allclose(s._chain_thetas, ht, rtol=RTOL, atol=1e-12)
    np.testing.assert_allclose(s._chain_logpost, hl, rtol=RTOL)
    moved = np.diff(s._chain_thetas, axis=0) != 0
    nblocks = moved[:, :, :k].any(-1).astype(int) + moved[:, :, k:2 * k + 1].any(-1) + moved[:, :, -1]
    assert nblocks.max() == 1 and moved[:, :, -1].any() and moved[:, :, :k].any()
    # the plain full-vector walk is still available from the same object
    from riemann_amd import _lib
    with pytest.raises(ra.ParameterError):
        _lib.call("rmh_proposal_rw_set_blocks", p._h, 2, np.array([5, 6], dtype=np.int32), _lib.as_f64([0.5, 1.0]))


# ------------------------------------------- random walk on the logistic model
@pytest.mark.parametrize("n,d", [(2000, 100), (300, 16)])
def test_logreg_random_walk_plain_adaptive_and_tempered(ra, orc, n, d):
    from riemann_amd.models import LogisticRegression
    from riemann_amd.proposals import AdaptScaleRandomWalk, MetropolisRandomWalk
    from riemann_amd.samplers import PTSampler
    from riemann_amd.workloads import logreg_data
    X, y = logreg_data(n, d, seed=4)
    g = np.random.default_rng(n + d)
    th0 = 0.05 * g.standard_normal((200, d))
    A = g.standard_normal((d, d))
    Cp = (0.02 / d) * (A @ A.T / d + np.eye(d))
    m, om = LogisticRegression(X, y, 100.0), orc.LogReg(X, y, 100.0)
    T = 14
    s = ra.Sampler(m, MetropolisRandomWalk(Cp), th0, seed=3)
    s.run(T)
    ht, hl, acc = oracle_run(orc, om, orc.RandomWalk(Cp), th0, 3, T)
    np.testing.assert_array_equal(s._chain_accept, acc)
    np.testing.assert_allclose(s._chain_thetas, ht, rtol=RTOL, atol=1e-12)
    np.testing.assert_allclose(s._chain_logpost, hl, rtol=RTOL)
    assert 0.05 < acc.mean() < 0.95
    # AdaptScaleRandomWalk (randomwalk.py:29-38)
    s = ra.Sampler(m, AdaptScaleRandomWalk(Cp), th0, seed=4)
    s.run(T)
    th = np.ascontiguousarray(th0).copy()
    st = orc.rng_seed(4, th.shape[0])
    ad = orc.new_adapt_state(th.shape[0])
    ht, hl, acc = orc.run(om, orc.set_adapt(orc.RandomWalk(Cp), 0.25), th, om.log_posterior(th), st, T, adapt=ad)
    np.testing.assert_array_equal(s._chain_accept, acc)
    np.testing.assert_allclose(s._chain_thetas, ht, rtol=RTOL, atol=1e-12)
    # PTSampler over TemperedModel(logistic): the prior is not tempered (ptsampler.py:38-39)
    K, betas = 4, 0.5 ** np.arange(4)
    pt = PTSampler(m, MetropolisRandomWalk(Cp), th0[:60], betas=betas, Pswap=0.3, seed=5)
    pt.run(T)
    S, idx = orc.pt_slots(60, K)
    slots = np.zeros((S, d))
    slots[:] = th0[0]
    slots[idx] = th0[:60, None, :]
    st = orc.rng_seed(5, S)
    ht, hl, code, _ = orc.run_pt(om, orc.RandomWalk(Cp), slots, st, T, betas, 0.3)
    np.testing.assert_array_equal(pt._host_codes, code[:, idx])
    np.testing.assert_allclose(pt._host_thetas, ht[:, idx], rtol=RTOL, atol=1e-12)
    np.testing.assert_allclose(pt._host_logpost, hl[:, idx], rtol=RTOL)


def test_pcn_and_adaptive_scale_on_mixture_and_logistic(ra, orc):
    """The remaining model x proposal cells of the random-walk family, against the oracle."""
    from riemann_amd.models import LogisticRegression, benchmarks as b
    from riemann_amd.proposals import AdaptScalepCN, AdaptScaleRandomWalk, pCN
    from riemann_amd.workloads import logreg_data
    T = 40
    th0 = np.random.default_rng(2).standard_normal((300, 2))
    om = orc.Mixture(MIX2D)
    s = ra.Sampler(b.benchmark_gauss2d_mix2_cross, pCN(np.eye(2), 0.8), th0, seed=6)
    s.run(T)
    ht, hl, acc = oracle_run(orc, om, orc.PCN(np.eye(2), 0.8), th0, 6, T)
    np.testing.assert_array_equal(s._chain_accept, acc)
    np.testing.assert_allclose(s._chain_thetas, ht, rtol=RTOL, atol=1e-12)
    s = ra.Sampler(b.benchmark_gauss2d_mix2_cross, AdaptScaleRandomWalk(np.eye(2)), th0, seed=7)
    s.run(T)
    th, st, ad = th0.copy(), orc.rng_seed(7, 300), orc.new_adapt_state(300)
    ht, hl, acc = orc.run(om, orc.set_adapt(orc.RandomWalk(np.eye(2)), 0.25), th, om.log_posterior(th), st, T, adapt=ad)
    np.testing.assert_array_equal(s._chain_accept, acc)
    np.testing.assert_allclose(s._chain_thetas, ht, rtol=RTOL, atol=1e-12)
    X, y = logreg_data(500, 40, seed=9)
    m, ol = LogisticRegression(X, y, 100.0), orc.LogReg(X, y, 100.0)
    th0 = 0.05 * np.random.default_rng(3).standard_normal((200, 40))
    Cp = 0.05 * np.eye(40)
    s = ra.Sampler(m, pCN(Cp, 0.98), th0, seed=8)
    s.run(12)
    ht, hl, acc = oracle_run(orc, ol, orc.PCN(Cp, 0.98), th0, 8, 12)
    np.testing.assert_array_equal(s._chain_accept, acc)
    np.testing.assert_allclose(s._chain_thetas, ht, rtol=RTOL, atol=1e-12)
    np.testing.assert_allclose(s._chain_logpost, hl, rtol=RTOL)


# --------------------------------------------- Haario covariance adaptation (8f-1)
@pytest.mark.parametrize("tag", ["strict", "smooth", "marg"])
def test_g14_adaptive_covariance_vs_reference_fixture(ra, tag):
    """AdaptCovRandomWalk of the reference, one chain per Sampler (pooled == per-chain)."""
    from riemann_amd.models import MultiGaussianDist
    from riemann_amd.proposals import AdaptCovRandomWalk
    z = load("g14_adaptcov_" + tag)
    m = MultiGaussianDist(z["mu"], z["C"])
    T = z["accept"].shape[0]
    for c in range(3):
        p = AdaptCovRandomWalk(z["C0"], t_adapt=float(z["t_adapt"]), marginalize=bool(z["marginalize"]),
                               smooth_adapt=bool(z["smooth"]))
        s = ra.Sampler(m, p, z["theta0"][c], seed=int(z["seeds"][c]))
        s.run(60)
        s.run(T - 60)                       # the adaptation continues across run() calls
        np.testing.assert_array_equal(s._chain_accept, z["accept"][60:, c])
        # strict / marg start adapting from 4 samples (t_adapt = 1): the feedback covariance ->
        # step -> state -> covariance amplifies the rounding differences described below; the
        # well-conditioned smooth variant agrees to 1e-8, these to the contract's 1e-6
        rt = 1e-8 if tag == "smooth" else 1e-6
        np.testing.assert_allclose(s._chain_thetas, z["thetas"][60:, c], rtol=rt, atol=1e-12)
        np.testing.assert_allclose(s._chain_logpost, z["logpost"][60:, c], rtol=rt)
        # Cs = (SX2 - SX SX^T/n)/(n-1) (adaptive.py:83) cancels catastrophically when the chain
        # barely moves (strict/marg: steps of 1e-5 around 1), and the state differences above feed
        # back into it: there the adapted covariance is a sanity check, not a parity figure
        np.testing.assert_allclose(p.C, z["C_final"][c], rtol=1e-8 if tag == "smooth" else 2e-2, atol=1e-15)
        np.testing.assert_allclose(p.L @ p.L.T * 2 ** 0.4, p.C, rtol=1e-12, atol=1e-300)
        if tag == "smooth":
            np.testing.assert_allclose(p.L, z["L_final"][c], rtol=1e-8, atol=1e-15)
        assert p._S == T


def test_g14_adaptive_covariance_10d_fixture(ra):
    from riemann_amd.models import MultiGaussianDist
    from riemann_amd.proposals import HaarioRandomWalk
    z = load("g14_adaptcov_10d")
    p = HaarioRandomWalk(z["C0"], t_adapt=30, smooth_adapt=True)
    s = ra.Sampler(MultiGaussianDist(np.zeros(10), z["Ct"]), p, z["theta0"][0], seed=int(z["seed"]))
    s.run(120)
    np.testing.assert_array_equal(s._chain_accept, z["accept"][:, 0])
    np.testing.assert_allclose(s._chain_thetas, z["thetas"][:, 0], rtol=1e-8, atol=1e-12)
    np.testing.assert_allclose(p.L, z["L_final"][0], rtol=1e-8, atol=1e-14)


@pytest.mark.parametrize("d,N,squares", [(100, 1000, True), (12, 333, False), (111, 64, True)])
def test_adaptive_covariance_pooled_over_chains_vs_oracle(ra, orc, d, N, squares, tmp_path):
    """Pooled Haario adaptation on many chains: states, adapted covariance, burn/thin
    bookkeeping across update boundaries, checkpoint/restore."""
    from riemann_amd.models import MultiGaussianDist
    from riemann_amd.proposals import AdaptCovRandomWalk
    g = np.random.default_rng(d + N)
    A = g.standard_normal((d, d))
    Ct = A @ A.T / d + 0.3 * np.eye(d)
    C0 = (0.3 / d) * np.eye(d)
    th0 = g.standard_normal((N, d))
    T = 70
    m = MultiGaussianDist(np.zeros(d), Ct)
    p = AdaptCovRandomWalk(C0, t_adapt=50, smooth_adapt=True, exact_squares=squares)
    s = ra.Sampler(m, p, th0, seed=31)
    s.run(T)
    om = orc.Gaussian(np.zeros(d), Ct)
    th, st, stc = th0.copy(), orc.rng_seed(31, N), orc.new_adaptcov_state(C0)
    ht, hl, acc = orc.run_adaptcov(om, orc.RandomWalk(C0), th, om.log_posterior(th), st, T, stc, C0,
                                   t_adapt=50, smooth_adapt=True, exact_squares=squares)
    np.testing.assert_array_equal(s._chain_accept, acc)
    np.testing.assert_allclose(s._chain_thetas, ht, rtol=1e-7, atol=1e-10)
    Cf, Lf = orc.adaptcov_CL(stc, d)
    np.testing.assert_allclose(p.C, Cf, rtol=1e-8, atol=1e-14)
    np.testing.assert_allclose(p.L, Lf, rtol=1e-8, atol=1e-14)
    np.testing.assert_allclose(p._SX, stc[1:1 + d], rtol=1e-10)
    # thinning across update boundaries + checkpoint: continue two ways, same chains
    s.checkpoint(str(tmp_path / "acov.npz"))
    s.run(45, 5, 7)
    ref_th, ref_C = s._chain_thetas.copy(), p.C.copy()
    p2 = AdaptCovRandomWalk(C0, t_adapt=50, smooth_adapt=True, exact_squares=squares)
    s2 = ra.Sampler(m, p2, th0, seed=0)
    s2.restore(str(tmp_path / "acov.npz"))
    s2.run(45)
    np.testing.assert_array_equal(s2._chain_thetas[5::7], ref_th)
    np.testing.assert_array_equal(p2.C, ref_C)
    assert ref_th.shape[0] == 6


# ------------------------------------------------ d > 112: workgroup-per-16-chains kernels
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


@pytest.mark.parametrize("d,N", [(113, 50), (128, 33), (257, 16), (500, 40), (896, 17)])
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
        MultiGaussianDist(np.zeros(897), np.eye(897))


def test_g14_adaptive_scale_and_covariance_fixture_and_many_chains(ra, orc):
    """AdaptScaleCovRandomWalk (randomwalk.py:62-75): reference fixture (one chain), then
    per-chain scales + pooled covariance on many chains against the oracle."""
    from riemann_amd.models import MultiGaussianDist
    from riemann_amd.proposals import AdaptScaleCovRandomWalk
    z = load("g14_adaptscalecov_10d")
    m = MultiGaussianDist(np.zeros(10), z["Ct"])
    p = AdaptScaleCovRandomWalk(z["C0"], t_adapt=30, smooth_adapt=True)
    s = ra.Sampler(m, p, z["theta0"][0], seed=int(z["seed"]))
    s.run(120)
    np.testing.assert_array_equal(s._chain_accept, z["accept"][:, 0])
    np.testing.assert_allclose(s._chain_thetas, z["thetas"][:, 0], rtol=1e-8, atol=1e-12)
    np.testing.assert_allclose(p.scale, z["scale_final"][0], rtol=1e-12)
    assert p.Naccepts == z["Naccepts"][0]
    np.testing.assert_allclose(p.L, z["L_final"][0], rtol=1e-8, atol=1e-14)
    N, T = 500, 50
    g = np.random.default_rng(4)
    th0 = g.standard_normal((N, 10))
    p = AdaptScaleCovRandomWalk(z["C0"], t_adapt=200, smooth_adapt=True)
    s = ra.Sampler(m, p, th0, seed=12)
    s.run(T)
    om = orc.Gaussian(np.zeros(10), z["Ct"])
    th, st, stc, ad = th0.copy(), orc.rng_seed(12, N), orc.new_adaptcov_state(z["C0"]), orc.new_adapt_state(N)
    ht, hl, acc = orc.run_adaptcov(om, orc.set_adapt(orc.RandomWalk(z["C0"]), 0.25), th, om.log_posterior(th),
                                   st, T, stc, z["C0"], t_adapt=200, smooth_adapt=True, adapt=ad)
    np.testing.assert_array_equal(s._chain_accept, acc)
    np.testing.assert_allclose(s._chain_thetas, ht, rtol=1e-8, atol=1e-11)
    np.testing.assert_allclose(p.scale, ad[:, 2], rtol=1e-12)
    np.testing.assert_allclose(p.C, orc.adaptcov_CL(stc, 10)[0], rtol=1e-9, atol=1e-15)


def test_g14_adaptive_covariance_hmc_fixtures_and_many_chains(ra, orc):
    """AdaptCovHMC (adapted covariance = mass matrix, with the reference's M / chM mismatch)
    and AdaptScaleCovHMC (whose covariance part never runs in the reference)."""
    from riemann_amd.models import LogisticRegression, MultiGaussianDist
    from riemann_amd.proposals import AdaptCovHMC, AdaptScaleCovHMC
    from riemann_amd.workloads import logreg_data
    z = load("g14_adaptcovhmc_10d")
    m = MultiGaussianDist(np.zeros(10), z["Ct"])
    for c in range(2):
        p = AdaptCovHMC(float(z["eps"]), int(z["nsteps"]), m, z["M0"], t_adapt=float(z["t_adapt"]), smooth_adapt=True)
        s = ra.Sampler(m, p, z["theta0"][c], seed=int(z["seed"]), chain0=c)
        s.run(60)
        np.testing.assert_array_equal(s._chain_accept, z["accept"][:, c])
        np.testing.assert_allclose(s._chain_thetas, z["thetas"][:, c], rtol=1e-8, atol=1e-12)
        np.testing.assert_allclose(s._chain_logpost, z["logpost"][:, c], rtol=1e-8)
        np.testing.assert_allclose(p.chM, z["L_final"][c], rtol=1e-8, atol=1e-14)
        np.testing.assert_allclose(p.M, z["C_final"][c], rtol=1e-8, atol=1e-14)
    z = load("g14_adaptscalecovhmc_10d")
    for c in range(2):
        p = AdaptScaleCovHMC(float(z["eps"]), int(z["nsteps"]), m, z["M0"], t_adapt=40, smooth_adapt=True)
        s = ra.Sampler(m, p, z["theta0"][c], seed=int(z["seed"]), chain0=c)
        s.run(60)
        np.testing.assert_array_equal(s._chain_accept, z["accept"][:, c])
        np.testing.assert_allclose(s._chain_thetas, z["thetas"][:, c], rtol=1e-8, atol=1e-12)
        np.testing.assert_allclose(p.scale, z["scale_final"][c], rtol=1e-12)
        np.testing.assert_array_equal(p.C, z["C_final"][c])
    # pooled over chains, logistic target (two waves per SIMD variant is not used with a mass matrix)
    X, y = logreg_data(400, 24, seed=6)
    ml, ol = LogisticRegression(X, y, 100.0), orc.LogReg(X, y, 100.0)
    N, T, M0 = 300, 30, 40.0 * np.eye(24)
    th0 = 0.05 * np.random.default_rng(5).standard_normal((N, 24))
    p = AdaptCovHMC(0.9, 3, ml, M0, t_adapt=500, smooth_adapt=True)
    s = ra.Sampler(ml, p, th0, seed=14)
    s.run(T)
    th, st, stc = th0.copy(), orc.rng_seed(14, N), orc.new_adaptcov_state(M0)
    ht, hl, acc = orc.run_adaptcov(ol, orc.HMC(0.9, 3, 24, M=M0), th, ol.log_posterior(th), st, T, stc, M0,
                                   t_adapt=500, smooth_adapt=True)
    np.testing.assert_array_equal(s._chain_accept, acc)
    np.testing.assert_allclose(s._chain_thetas, ht, rtol=1e-7, atol=1e-10)
    np.testing.assert_allclose(p.C, orc.adaptcov_CL(stc, 24)[0], rtol=1e-8, atol=1e-14)
    assert 0.05 < acc.mean() < 0.99 and acc[:4].mean() > 0.5


def test_g10_adaptive_scale_pcn(ra, orc, tmp_path):
    """AdaptScalepCN: reference fixture, then many chains against the oracle, then a
    checkpoint across the per-chain rho."""
    from riemann_amd.models import MultiGaussianDist
    from riemann_amd.proposals import AdaptScalepCN
    z = load("g10_adaptscalepcn_gauss2d")
    m = MultiGaussianDist(z["mu"], z["C"])
    p = AdaptScalepCN(z["Cprop"], float(z["rho"]))
    s = ra.Sampler(m, p, z["theta0"], seed=int(z["seed"]))
    s.run(64)
    check_run(z, s)
    np.testing.assert_allclose(p.rho, z["rho_final"], rtol=1e-12)
    np.testing.assert_allclose(p.scale, z["scale_final"], rtol=1e-12)
    d, N, T = 40, 600, 30
    g = np.random.default_rng(2)
    A = g.standard_normal((d, d))
    Ct = A @ A.T / d + 0.3 * np.eye(d)
    th0 = g.standard_normal((N, d))
    m = MultiGaussianDist(np.zeros(d), Ct)
    p = AdaptScalepCN(0.1 * np.eye(d), 0.95)
    s = ra.Sampler(m, p, th0, seed=8)
    s.run(T)
    om = orc.Gaussian(np.zeros(d), Ct)
    th, st, ad, aux = th0.copy(), orc.rng_seed(8, N), orc.new_adapt_state(N), np.full(N, np.nan)
    ht, hl, acc = orc.run(om, orc.set_adapt(orc.PCN(0.1 * np.eye(d), 0.95), 0.25), th, om.log_posterior(th),
                          st, T, adapt=ad, aux=aux)
    np.testing.assert_array_equal(s._chain_accept, acc)
    np.testing.assert_allclose(s._chain_thetas, ht, rtol=RTOL, atol=1e-12)
    np.testing.assert_allclose(p.rho, aux, rtol=1e-12)
    s.checkpoint(str(tmp_path / "apcn.npz"))
    s.run(10)
    ref = s._chain_thetas.copy()
    p2 = AdaptScalepCN(0.1 * np.eye(d), 0.95)
    s2 = ra.Sampler(m, p2, th0, seed=0)
    s2.restore(str(tmp_path / "apcn.npz"))
    s2.run(10)
    np.testing.assert_array_equal(s2._chain_thetas, ref)


def test_adaptive_covariance_on_mixture_and_logistic_vs_oracle(ra, orc):
    from riemann_amd.models import LogisticRegression, benchmarks as b
    from riemann_amd.proposals import AdaptCovRandomWalk
    from riemann_amd.workloads import logreg_data
    g = np.random.default_rng(11)
    X, y = logreg_data(300, 16, seed=3)
    cases = [(b.benchmark_gauss2d_mix2_cross, orc.Mixture(MIX2D), 0.5 * np.eye(2), g.standard_normal((400, 2))),
             (LogisticRegression(X, y, 100.0), orc.LogReg(X, y, 100.0), 0.01 * np.eye(16),
              0.05 * g.standard_normal((300, 16)))]
    for m, om, C0, th0 in cases:
        d, N, T = th0.shape[1], th0.shape[0], 40
        p = AdaptCovRandomWalk(C0, t_adapt=300, smooth_adapt=True)
        s = ra.Sampler(m, p, th0, seed=17)
        s.run(T)
        th, st, stc = th0.copy(), orc.rng_seed(17, N), orc.new_adaptcov_state(C0)
        ht, hl, acc = orc.run_adaptcov(om, orc.RandomWalk(C0), th, om.log_posterior(th), st, T, stc, C0,
                                       t_adapt=300, smooth_adapt=True)
        np.testing.assert_array_equal(s._chain_accept, acc)
        np.testing.assert_allclose(s._chain_thetas, ht, rtol=1e-8, atol=1e-11)
        np.testing.assert_allclose(p.C, orc.adaptcov_CL(stc, d)[0], rtol=1e-9, atol=1e-15)


def test_changepoint_with_the_generic_kernels_vs_oracle(ra, orc):
    """Scale adaptation, pCN, AdaptScalepCN, covariance adaptation and parallel tempering on the
    changepoint model (the model as a policy of the generic fused kernels)."""
    from riemann_amd import workloads as W
    from riemann_amd.models import ChangepointRegression1D
    from riemann_amd.proposals import (AdaptCovRandomWalk, AdaptScalepCN, AdaptScaleRandomWalk,
                                       MetropolisRandomWalk, pCN)
    from riemann_amd.samplers import PTSampler
    x, y, truth, cdiag = W.changepoint_data()
    k, d, N, T = 5, 12, 700, 40
    m = ChangepointRegression1D(x, y, 1.0, 3.0, 5.0, 10, 1, 1, k=k)
    om = orc.Changepoint(x, y, k, 1.0, 3.0, 5.0, 1.0, 1.0)
    g = np.random.default_rng(9)
    th0 = truth + 0.01 * g.standard_normal((N, d))
    th0[:, :k] = np.sort(th0[:, :k], axis=1)
    th0[5, 0], th0[5, 1] = th0[5, 1], th0[5, 0]              # one chain starts outside the support
    Cp = 0.02 * np.diag(cdiag)

    def compare(s, ht, hl, acc, rtol=RTOL):
        np.testing.assert_array_equal(s._chain_accept, acc)
        np.testing.assert_allclose(s._chain_thetas, ht, rtol=RTOL, atol=1e-12)
        assert np.array_equal(np.isinf(s._chain_logpost), np.isinf(hl))
        fin = np.isfinite(hl)
        np.testing.assert_allclose(s._chain_logpost[fin], hl[fin], rtol=rtol)

    s = ra.Sampler(m, AdaptScaleRandomWalk(Cp), th0, seed=1)
    s.run(T)
    th, st, ad = th0.copy(), orc.rng_seed(1, N), orc.new_adapt_state(N)
    compare(s, *orc.run(om, orc.set_adapt(orc.RandomWalk(Cp), 0.25), th, om.log_posterior(th), st, T, adapt=ad))
    s = ra.Sampler(m, pCN(Cp, 0.999), th0, seed=2)
    s.run(T)
    compare(s, *oracle_run(orc, om, orc.PCN(Cp, 0.999), th0, 2, T))
    s = ra.Sampler(m, AdaptScalepCN(Cp, 0.999), th0, seed=3)
    s.run(T)
    th, st, ad, aux = th0.copy(), orc.rng_seed(3, N), orc.new_adapt_state(N), np.full(N, np.nan)
    compare(s, *orc.run(om, orc.set_adapt(orc.PCN(Cp, 0.999), 0.25), th, om.log_posterior(th), st, T, adapt=ad, aux=aux))
    good = np.delete(th0, 5, axis=0)                          # pooled moments need finite chains
    p = AdaptCovRandomWalk(Cp, t_adapt=5000, smooth_adapt=True)
    s = ra.Sampler(m, p, good, seed=4)
    s.run(T)
    th, st, stc = good.copy(), orc.rng_seed(4, N - 1), orc.new_adaptcov_state(Cp)
    # the pooled covariance (matrix-core sums vs the oracle's loops) feeds back into the states at
    # the 1e-10 level, and log L ~ sum r^2 / sig^2 with sig = 0.1 magnifies that: contract tolerance
    compare(s, *orc.run_adaptcov(om, orc.RandomWalk(Cp), th, om.log_posterior(th), st, T, stc, Cp,
                                 t_adapt=5000, smooth_adapt=True), rtol=1e-6)
    # variances of 1e-5 around values of 2: the one-pass covariance cancels ~5 digits
    np.testing.assert_allclose(p.C, orc.adaptcov_CL(stc, d)[0], rtol=1e-5, atol=1e-12)
    K, Ne, betas = 4, 150, 0.5 ** np.arange(4)
    pt = PTSampler(m, MetropolisRandomWalk(Cp), th0[:Ne], betas=betas, Pswap=0.3, seed=5)
    pt.run(T)
    S, idx = orc.pt_slots(Ne, K)
    slots = np.zeros((S, d))
    slots[:] = th0[0]
    slots[idx] = th0[:Ne, None, :]
    ht, hl, code, _ = orc.run_pt(om, orc.RandomWalk(Cp), slots, orc.rng_seed(5, S), T, betas, 0.3)
    np.testing.assert_array_equal(pt._host_codes, code[:, idx])
    np.testing.assert_allclose(pt._host_thetas, ht[:, idx], rtol=RTOL, atol=1e-12)


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


def test_adaptive_covariance_run_of_zero_steps(ra):
    """run(0) with a covariance-adaptive proposal keeps the state on entry (found by
    tools/fuzz_vs_oracle.py: the stretch loop never ran, so entry 0 was never written)."""
    from riemann_amd.models import MultiGaussianDist
    from riemann_amd.proposals import AdaptCovRandomWalk
    th0 = np.random.default_rng(1).standard_normal((17, 15))
    m = MultiGaussianDist(np.zeros(15), np.eye(15))
    s = ra.Sampler(m, AdaptCovRandomWalk(0.1 * np.eye(15), t_adapt=10, smooth_adapt=True), th0, seed=1)
    s.run(0, 0, 2)
    np.testing.assert_array_equal(s._chain_thetas, th0[None])
    np.testing.assert_allclose(s._chain_logpost[0], m.log_posterior(th0), rtol=RTOL)
    s.run(3)
    assert s._chain_thetas.shape == (4, 17, 15)
    np.testing.assert_array_equal(s._chain_thetas[0], th0)


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


# ----------------------------- trans-dimensional changepoint sampling (8f-4)
def rj_model(z):
    from riemann_amd.models import ChangepointRegression1D
    return ChangepointRegression1D(z["x"], z["y"], float(z["xmin"]), float(z["xmax"]), float(z["lamb"]),
                                   int(z["KM"]), float(z["alpha"]), float(z["beta"]), variable_k=True)


def test_g17_trans_dimensional_changepoint_sampler_vs_reference_fixture(ra):
    """The reference's ChangepointRegression1DProp with its birth/death branch: moves, accepts,
    the number of changepoints and the states of 4 chains x 300 iterations."""
    from riemann_amd.proposals import ChangepointRegression1DProp
    z = load("g17_changepoint_rj")
    m = rj_model(z)
    start = m.unpack_padded(z["theta0"])
    np.testing.assert_array_equal(m.pack_padded(start), z["theta0"])
    np.testing.assert_allclose(m.log_posterior(z["theta0"]), z["logpost"][0], rtol=RTOL)
    p = ChangepointRegression1DProp(m, float(z["hscale"]))
    assert p.P_cumprop_dim == 1.0
    p.P_cumprop_sig = 0.0                    # as the fixture (the reference's sig block raises under numpy 2)
    s = ra.Sampler(m, p, start, seed=int(z["seed"]))
    s.run(300)
    code = s._chain_accept
    np.testing.assert_array_equal(code >> 1, z["move"])
    np.testing.assert_array_equal(code & 1, z["accept"])
    th = s._chain_thetas
    np.testing.assert_array_equal(th[:, :, -1], z["thetas"][:, :, -1])
    fin = np.isfinite(z["thetas"])
    assert np.array_equal(np.isfinite(th), fin)
    np.testing.assert_allclose(th[fin], z["thetas"][fin], rtol=RTOL, atol=1e-12)
    np.testing.assert_allclose(s._chain_logpost, z["logpost"], rtol=RTOL)
    last = m.unpack_padded(th[-1])
    assert [len(t.cpx) for t in last] == list(z["thetas"][-1, :, -1].astype(int))


@pytest.mark.parametrize("KM,N", [(5, 3000), (12, 1000)])
def test_trans_dimensional_changepoint_many_chains_vs_oracle(ra, orc, KM, N):
    """Default probabilities (all five move types, births refused at capacity) on many chains."""
    from riemann_amd import workloads as W
    from riemann_amd.models import ChangepointRegression1D
    from riemann_amd.proposals import ChangepointRegression1DProp
    x, y, truth, cdiag = W.changepoint_data()
    m = ChangepointRegression1D(x, y, 1.0, 3.0, 5.0, KM, 1, 1, variable_k=True)
    om = orc.Changepoint(x, y, 1, 1.0, 3.0, 5.0, 1.0, 1.0)
    g = np.random.default_rng(KM)
    starts = []
    for c in range(N):
        k = int(g.integers(0, KM + 1))
        starts.append((np.sort(g.uniform(1.05, 2.95, size=k)), g.uniform(1.0, 3.0, size=k + 1), float(g.uniform(0.2, 1.0))))
    th0 = orc.cprj_pack(starts, KM)
    p = ChangepointRegression1DProp(m, 2.0)
    s = ra.Sampler(m, p, th0, seed=6, chain0=160)
    T = 80
    s.run(T)
    th = th0.copy()
    lp = orc.cprj_log_posterior(om, KM, th)
    ht, hl, acc, mv = orc.run_cprj(om, KM, 2.0, [0.2, 0.4, 0.6], th, lp, orc.rng_seed(6, N, chain0=160), T)
    code = s._chain_accept
    np.testing.assert_array_equal(code >> 1, mv)
    np.testing.assert_array_equal(code & 1, acc)
    got = s._chain_thetas
    fin = np.isfinite(ht)
    assert np.array_equal(np.isfinite(got), fin)
    np.testing.assert_allclose(got[fin], ht[fin], rtol=RTOL, atol=1e-12)
    np.testing.assert_allclose(s._chain_logpost, hl, rtol=RTOL)
    assert set(np.unique(mv)) >= {0, 1, 2, 3, 4} and (mv == 5).any()
    assert (acc[mv == 3] == 1).sum() > 20 and (acc[mv == 4] == 1).sum() > 20


def test_parallel_tempering_with_hmc_within_the_rungs(ra, orc):
    """PTSampler + VanillaHMC / MALA: reference fixture, then many ladders on a 100-d Gaussian
    and on the logistic model against the oracle.  The trajectories use the base model's
    gradient on every rung (the reference hands all rungs one proposal), the accept test the
    tempered posterior; an accepted swap exchanges states AND cached gradients."""
    from riemann_amd.models import LogisticRegression, MultiGaussianDist
    from riemann_amd.proposals import MALA, VanillaHMC
    from riemann_amd.samplers import PTSampler
    from riemann_amd.workloads import logreg_data
    z = load("g12_pt_hmc_gauss2d")
    m = MultiGaussianDist(z["mu"], z["C"])
    pt = PTSampler(m, VanillaHMC(float(z["eps"]), int(z["nsteps"]), m), z["theta0"], Pswap=float(z["pswap"]),
                   seed=int(z["seed"]))
    pt.run(50)
    pt.run(46)
    check_pt(z, pt)
    g = np.random.default_rng(8)
    d = 100
    A = g.standard_normal((d, d))
    Cm = A @ A.T / d + 0.5 * np.eye(d)
    X, y = logreg_data(300, 24, seed=5)
    cases = [(MultiGaussianDist(np.zeros(d), Cm), orc.Gaussian(np.zeros(d), Cm), d, 0.25, 3, 5),
             (LogisticRegression(X, y, 100.0), orc.LogReg(X, y, 100.0), 24, 0.4, 1, 4)]
    for m, om, dd, eps, L, K in cases:
        Ne, T = 200, 16
        betas = 0.7 ** np.arange(K)
        th0 = 0.3 * g.standard_normal((Ne, dd))
        prop = MALA(eps, m) if L == 1 else VanillaHMC(eps, L, m)
        pt = PTSampler(m, prop, th0, betas=betas, Pswap=0.3, seed=11)
        pt.run(T)
        S, idx = orc.pt_slots(Ne, K)
        slots = np.zeros((S, dd))
        slots[:] = th0[0]
        slots[idx] = th0[:, None, :]
        ht, hl, code, _ = orc.run_pt(om, orc.HMC(eps, L, dd), slots, orc.rng_seed(11, S), T, betas, 0.3)
        np.testing.assert_array_equal(pt._host_codes, code[:, idx])
        np.testing.assert_allclose(pt._host_thetas, ht[:, idx], rtol=1e-8, atol=1e-10)
        np.testing.assert_allclose(pt._host_logpost, hl[:, idx], rtol=1e-8)
        assert (code[:, idx] == 3).sum() > 20 and (code[:, idx] == 1).sum() > 100
