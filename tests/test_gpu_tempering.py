"""GPU parity tests of parallel tempering (PTSampler): reference fixtures G12 and oracle
comparisons, random-walk and HMC within-chain moves."""
import numpy as np
import pytest

from gpu_common import MIX2D, RTOL, check_pt, load  # noqa: F401

pytestmark = pytest.mark.gpu


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


def test_parallel_tempering_with_pcn_and_mass_matrix_hmc_within_the_rungs(ra, orc):
    """The other stateless proposals of the reference inside PTSampler: pCN (asymmetric logqratio,
    randomwalk.py:78-101) on the Gaussian, mixture, logistic and changepoint models, and
    VanillaHMC / MALA with a mass matrix (hamiltonian.py:71-87) on the Gaussian and logistic
    models; many ladders against the oracle.  Adaptive proposals stay refused: the reference
    hands one proposal object to all rungs (ptsampler.py:82)."""
    from riemann_amd.models import ChangepointRegression1D, LogisticRegression, MultiGaussianDist, benchmarks as b
    from riemann_amd.proposals import AdaptScaleRandomWalk, VanillaHMC, pCN
    from riemann_amd.samplers import PTSampler
    from riemann_amd.workloads import changepoint_data, logreg_data
    g = np.random.default_rng(21)
    d = 100
    A = g.standard_normal((d, d))
    Cm = A @ A.T / d + 0.5 * np.eye(d)
    X, y = logreg_data(300, 24, seed=5)
    x, yy, truth, cdiag = changepoint_data()
    mix = b.benchmark_gauss2d_mix2_cross
    omix = orc.Mixture(MIX2D)
    Mm = np.diag(g.uniform(0.5, 2.0, 24)) + 0.1 * np.ones((24, 24))
    Mg = np.linalg.inv(Cm)
    cases = [
        ("pcn gauss", MultiGaussianDist(np.zeros(d), Cm), orc.Gaussian(np.zeros(d), Cm), d,
         lambda m: pCN(Cm, 0.9), lambda: orc.PCN(Cm, 0.9), 5, 0.3 * g.standard_normal((150, d))),
        ("pcn logreg", LogisticRegression(X, y, 100.0), orc.LogReg(X, y, 100.0), 24,
         lambda m: pCN(0.01 * np.eye(24), 0.95), lambda: orc.PCN(0.01 * np.eye(24), 0.95), 4, 0.3 * g.standard_normal((150, 24))),
        ("pcn changepoint", ChangepointRegression1D(x, yy, 1.0, 3.0, 5.0, 10, 1, 1, k=5),
         orc.Changepoint(x, yy, 5, 1.0, 3.0, 5.0, 1.0, 1.0), 12,
         lambda m: pCN(np.diag(0.01 * cdiag), 0.999), lambda: orc.PCN(np.diag(0.01 * cdiag), 0.999), 3,
         None),
        ("hmc mass gauss", MultiGaussianDist(np.zeros(d), Cm), orc.Gaussian(np.zeros(d), Cm), d,
         lambda m: VanillaHMC(0.3, 3, m, M=Mg), lambda: orc.HMC(0.3, 3, d, M=Mg), 5, 0.3 * g.standard_normal((150, d))),
        ("mala mass logreg", LogisticRegression(X, y, 100.0), orc.LogReg(X, y, 100.0), 24,
         lambda m: VanillaHMC(0.3, 1, m, M=Mm), lambda: orc.HMC(0.3, 1, 24, M=Mm), 4, 0.3 * g.standard_normal((150, 24))),
    ]
    cases.append(("pcn mixture", mix, omix, 2, lambda m: pCN(1.5 * np.eye(2), 0.8), lambda: orc.PCN(1.5 * np.eye(2), 0.8),
                  5, g.standard_normal((150, 2))))
    cases.append(("hmc mixture", mix, omix, 2, lambda m: VanillaHMC(0.3, 4, m), lambda: orc.HMC(0.3, 4, 2),
                  5, g.standard_normal((150, 2))))
    M2 = np.array([[2.0, 0.3], [0.3, 0.5]])
    cases.append(("mala mass mixture", mix, omix, 2, lambda m: VanillaHMC(0.4, 1, m, M=M2), lambda: orc.HMC(0.4, 1, 2, M=M2),
                  4, g.standard_normal((150, 2))))
    for name, m, om, dd, mkp, mko, K, th0 in cases:
        if th0 is None:
            th0 = truth + 0.01 * g.standard_normal((150, 12))
            th0[:, :5] = np.sort(th0[:, :5], axis=1)
        Ne, T = th0.shape[0], 16
        betas = 0.7 ** np.arange(K)
        pt = PTSampler(m, mkp(m), th0, betas=betas, Pswap=0.3, seed=13)
        pt.run(T)
        S, idx = orc.pt_slots(Ne, K)
        slots = np.zeros((S, dd))
        slots[:] = th0[0]
        slots[idx] = th0[:, None, :]
        ht, hl, code, _ = orc.run_pt(om, mko(), slots, orc.rng_seed(13, S), T, betas, 0.3)
        np.testing.assert_array_equal(pt._host_codes, code[:, idx], err_msg=name)
        np.testing.assert_allclose(pt._host_thetas, ht[:, idx], rtol=1e-8, atol=1e-10, err_msg=name)
        np.testing.assert_allclose(pt._host_logpost, hl[:, idx], rtol=1e-8, err_msg=name)
        # (pCN pulls towards the origin, where the changepoint posterior has no mass: its within-chain
        # moves are all rejected there, which is the parity case that matters for that model)
        assert (code[:, idx] == 3).sum() > 10 and ((code[:, idx] == 1).sum() > 50 or name == "pcn changepoint"), name
    with pytest.raises(ra.ParameterError):
        PTSampler(MultiGaussianDist(np.zeros(2), np.eye(2)), AdaptScaleRandomWalk(np.eye(2)), np.zeros((4, 2))).run(3)
