"""Pins the CPU oracle (oracle/rmh_oracle.c) against the golden vectors that
tests/golden/make_golden.py recorded from the reference itself.

Bars: accept masks bit-exact; thetas / log-densities within 1e-9 relative of
the reference's LAPACK-based values (the contract in BASELINE.json is 1e-6;
the observed agreement is ~1e-13).  CPU only.
"""
import os

import numpy as np
import pytest

from oracle import oracle as orc

G = os.path.join(os.path.dirname(__file__), "golden")
RTOL = 1e-9


def load(name):
    return np.load(os.path.join(G, name + ".npz"))


def replay(z, model, prop, T=None):
    th = np.ascontiguousarray(z["theta0"]).copy()
    N = th.shape[0]
    st = orc.rng_seed(int(z["seed"]), N)
    lp = model.log_posterior(th)
    T = T or z["accept"].shape[0]
    ht, hl, acc = orc.run(model, prop, th, lp, st, T)
    return ht, hl, acc


def check_run(z, ht, hl, acc):
    np.testing.assert_array_equal(acc, z["accept"])
    np.testing.assert_allclose(ht, z["thetas"], rtol=RTOL, atol=1e-12)
    ref = z["logpost"]
    assert np.array_equal(np.isinf(hl), np.isinf(ref))
    fin = np.isfinite(ref)
    np.testing.assert_allclose(hl[fin], ref[fin], rtol=RTOL)


def corr100():
    return 0.1 * np.eye(100) + 0.9 * np.ones((100, 100))


def test_g0_rng_stream_is_reproduced():
    z = load("g0_rng")
    for d in (1, 2, 5, 12, 100):
        st = orc.rng_seed(int(z["seed"]), 3, chain0=int(z["chain0"]))
        np.testing.assert_array_equal(st, z["state0_d%d" % d])
        xi, u = orc.rng_stream(st, d, 3)
        np.testing.assert_array_equal(xi, z["xi_d%d" % d])
        np.testing.assert_array_equal(u, z["u_d%d" % d])
        np.testing.assert_array_equal(st, z["state3_d%d" % d])


def test_g1_gauss2d_rw():
    z = load("g1_gauss2d_rw")
    xi, u = orc.rng_stream(orc.rng_seed(int(z["seed"]), 4), 2, 64)
    np.testing.assert_array_equal(xi, z["xi"])
    np.testing.assert_array_equal(u, z["u"])
    check_run(z, *replay(z, orc.Gaussian(z["mu"], z["C"]), orc.RandomWalk(z["Cprop"])))


def test_g2_gauss100d_rw():
    z = load("g2_gauss100d_rw")
    Ct = corr100()
    check_run(z, *replay(z, orc.Gaussian(z["mu"], Ct), orc.RandomWalk(float(z["prop_factor"]) * Ct)))


@pytest.mark.parametrize("tag", ["mala", "hmc10"])
def test_g3_gauss100d_hmc(tag):
    z = load("g3_gauss100d_" + tag)
    check_run(z, *replay(z, orc.Gaussian(np.zeros(100), corr100()),
                         orc.HMC(float(z["eps"]), int(z["nsteps"]), 100)))


def test_g3_gauss2d_hmc5():
    z = load("g3_gauss2d_hmc5")
    check_run(z, *replay(z, orc.Gaussian(z["mu"], z["C"]),
                         orc.HMC(float(z["eps"]), int(z["nsteps"]), 2)))


@pytest.mark.parametrize("tag", ["mala", "hmc10"])
def test_g4_logreg_small(tag):
    z = load("g4_logreg256x16_" + tag)
    m = orc.LogReg(z["X"], z["y"], float(z["tau2"]))
    check_run(z, *replay(z, m, orc.HMC(float(z["eps"]), int(z["nsteps"]), 16)))


@pytest.mark.parametrize("tag", ["mala", "hmc10"])
def test_g4_logreg_full_size(tag):
    import hashlib
    from riemann_amd.workloads import logreg_data
    z = load("g4_logreg10000x100_" + tag)
    X, y = logreg_data(int(z["n"]), int(z["d"]), int(z["data_seed"]))
    assert hashlib.sha256(X.tobytes()).digest() == z["X_sha256"].tobytes()
    assert hashlib.sha256(y.tobytes()).digest() == z["y_sha256"].tobytes()
    m = orc.LogReg(X, y, float(z["tau2"]))
    check_run(z, *replay(z, m, orc.HMC(float(z["eps"]), int(z["nsteps"]), 100)))


def cp_model(z):
    return orc.Changepoint(z["x"], z["y"], int(z["k"]), float(z["xmin"]), float(z["xmax"]),
                           float(z["lamb"]), float(z["alpha"]), float(z["beta"]))


@pytest.mark.parametrize("name", ["g5_changepoint_eval", "g5_changepoint_eval_a25"])
def test_g5_changepoint_eval(name):
    z = load(name)
    m = cp_model(z)
    th = z["theta"]
    for fn, key in ((m.log_likelihood, "logL"), (m.log_prior, "logP"), (m.log_posterior, "logpost")):
        got, ref = fn(th), z[key]
        assert np.array_equal(np.isinf(got), np.isinf(ref)), key
        assert np.array_equal(np.sign(got[np.isinf(ref)]), np.sign(ref[np.isinf(ref)])), key
        fin = np.isfinite(ref)
        np.testing.assert_allclose(got[fin], ref[fin], rtol=RTOL)
    assert np.isinf(z["logpost"]).sum() >= 12  # the edge rows really are exercised
    if "idx" in z.files:  # integer path: bit-exact wherever the state is valid
        ok = np.isfinite(z["logpost"])
        np.testing.assert_array_equal(m.indices(th)[ok], z["idx"][ok])


def test_g5_changepoint_rw():
    z = load("g5_changepoint_rw")
    check_run(z, *replay(z, cp_model(z), orc.RandomWalk(np.diag(z["Cdiag"]), float(z["scale"]))))


def test_g6_neginf_start_accepts_nan_ratio():
    z = load("g6_neginf_start")
    ht, hl, acc = replay(z, cp_model(z), orc.RandomWalk(np.diag(z["Cdiag"]), float(z["scale"])))
    check_run(z, ht, hl, acc)
    # chains 0,1 start at -inf and accept their first move (NaN ratio, sampler.py:83-84)
    assert np.isinf(z["logpost"][0, :2]).all() and z["accept"][0, :2].all()


def test_g6_burn_thin_resume():
    z = load("g6_burn_thin_resume")
    m, p = orc.Gaussian(z["mu"], z["C"]), orc.RandomWalk(z["Cprop"])
    th = z["theta0"].copy()
    st = orc.rng_seed(int(z["seed"]), th.shape[0])
    lp = m.log_posterior(th)
    (n0, b0, t0), (n1, b1, t1) = z["runs"]
    ht, hl, _ = orc.run(m, p, th, lp, st, int(n0))
    assert z["thetas_run0"].shape[0] == 3  # run(10,4,3) keeps entries 4,7,10
    np.testing.assert_allclose(ht[b0::t0], z["thetas_run0"], rtol=RTOL)
    np.testing.assert_allclose(hl[b0::t0], z["logpost_run0"], rtol=RTOL)
    ht, hl, acc = orc.run(m, p, th, lp, st, int(n1))  # continues from the last state
    assert z["thetas_run1"].shape[0] == 6
    np.testing.assert_allclose(ht[b1::t1], z["thetas_run1"], rtol=RTOL)
    np.testing.assert_array_equal(acc, z["accept_run1"])


def test_oracle_math_close_to_libm():
    g = np.random.default_rng(0)
    xs = np.concatenate([g.random(2000), 10.0 ** g.uniform(-300, 300, 2000),
                         [1.0, 2.0 ** -53, 1 - 2.0 ** -53, 5e-324, 2.2e-308]])
    got = np.array([orc.olog(x) for x in xs])
    np.testing.assert_allclose(got, np.log(xs), rtol=3e-16, atol=1e-17)
    assert np.isnan(orc.olog(-1.0)) and orc.olog(0.0) == -np.inf and orc.olog(np.inf) == np.inf


def test_inverse_cdf_normals_are_accurate():
    """z(half word) against scipy's normal quantile at the tail probability the half word encodes."""
    from scipy.stats import norm
    g = np.random.default_rng(1)
    halves = [int(w) for w in g.integers(0, 2 ** 32, 3000, dtype=np.uint64)]
    halves += [1, 2, 3, 2 ** 32 - 1, 2 ** 31, 2 ** 31 + 1, 2 ** 20 + 5, 2 ** 12, 2 ** 7 + 1, 0]
    for h in halves:
        z = orc.normal_from_half(h)
        q = h | 1
        r = min(32 - q.bit_length(), 31)
        N = ((q << r) & (2 ** 32 - 1)) | 2 ** 31
        p = (N >> 1) / 2.0 ** 30 * 2.0 ** (-2 - r)          # = 2^(-2-r) (1 + j/8 + m27 2^-30)
        ref = norm.isf(p)
        # v4: degree-6 cell polynomials, 2.3e-12 as a map (tools/gen_icdf_table.py) -- the lattice the 32-bit
        # half word puts the probabilities on is three orders coarser (about 1e-9 in z)
        assert abs(abs(z) - ref) <= 4e-12 * max(1.0, ref), (h, z, ref)
        assert (z < 0) == bool(h & 1) or z == 0
    # symmetric by construction: the sign is bit 0, which the magnitude ignores
    assert orc.normal_from_half(12345 * 2) == -orc.normal_from_half(12345 * 2 + 1)
    # the largest magnitude a half word can give: p = 2^-33, 6.4 sigma
    assert 6.3 < abs(orc.normal_from_half(0)) < 6.5


def test_rng_matches_pure_python_restatement():
    """The stream spec, restated with Python integers and exact rational fma."""
    from tests.rng_spec import stream_normals_uniform
    for d in (1, 5, 12):
        st = orc.rng_seed(99, 2, chain0=3)
        xi, u = orc.rng_stream(st, d, 2)
        for c in range(2):
            exi, eu = stream_normals_uniform(99, 3 + c, d, 2)
            np.testing.assert_array_equal(xi[:, c], exi)
            np.testing.assert_array_equal(u[:, c], eu)


# ---- the 64-bit-resolution normal mode (Sampler(..., normals="f64")) -------------------------
def _p_of_word(w):
    """tail probability and sign a stream word encodes in the 64-bit normal mode (rmh_oracle.c section 1)"""
    q = w | 1
    r = min(64 - q.bit_length(), 55)
    N = ((q << r) & (2 ** 64 - 1)) | 2 ** 63
    j, mant = (N >> 60) & 7, (N >> 8) & (2 ** 52 - 1)
    return 2.0 ** (-2 - r) * (1 + (j + mant * 2.0 ** -52) / 8), w & 1


def test_f64_mode_normals_are_accurate_out_to_the_tails():
    """z(word) against scipy's normal quantile at the tail probability the word encodes: random words,
    every row of the table (leading zeros 0..63 -> p down to 2^-58, 8.5 sigma), cell edges."""
    from scipy.stats import norm
    g = np.random.default_rng(2)
    words = [int(w) for w in g.integers(0, 2 ** 64, 3000, dtype=np.uint64)]
    for r in range(64):                                  # a word with r leading zeros, random below
        words += [(1 << (63 - r)) | int(g.integers(0, 1 << (63 - r))) if r < 63 else 1 for _ in range(8)]
    words += [0, 1, 2, 3, 2 ** 64 - 1, 2 ** 63, 2 ** 63 + 1, 2 ** 60 - 1, 2 ** 60, (2 ** 61 - 1) << 3]
    zmax = 0.0
    for w in words:
        z = orc.normal_from_word(w)
        p, sign = _p_of_word(w)
        ref = norm.isf(p)
        assert abs(abs(z) - ref) <= 2e-15 * max(1.0, ref), (hex(w), z, ref)
        assert (z < 0) == bool(sign) or z == 0
        zmax = max(zmax, abs(z))
    assert orc.normal_from_word(0xABCDEF12345678 * 2) == -orc.normal_from_word(0xABCDEF12345678 * 2 + 1)
    assert 8.5 < zmax < 8.6 and abs(orc.normal_from_word(0)) == zmax      # p = 2^-57
    # finer than the default stream (which reads 32 bits): words that differ in bit 20 only give different normals
    assert orc.normal_from_word(0x4000000000100000) != orc.normal_from_word(0x4000000000000000)


def test_f64_mode_stream_matches_pure_python_restatement_and_fixture():
    from tests.rng_spec import stream_normals_uniform
    with orc.normal_mode("f64"):
        for d in (1, 5, 12):
            st = orc.rng_seed(99, 2, chain0=3)
            xi, u = orc.rng_stream(st, d, 2)
            for c in range(2):
                exi, eu = stream_normals_uniform(99, 3 + c, d, 2, mode="f64")
                np.testing.assert_array_equal(xi[:, c], exi)
                np.testing.assert_array_equal(u[:, c], eu)
        z = load("g18_rng_f64")
        for d in (1, 2, 5, 12, 100):
            st = orc.rng_seed(int(z["seed"]), 3, chain0=int(z["chain0"]))
            xi, u = orc.rng_stream(st, d, 3)
            np.testing.assert_array_equal(xi, z["xi_d%d" % d])
            np.testing.assert_array_equal(u, z["u_d%d" % d])
            np.testing.assert_array_equal(st, z["state3_d%d" % d])
        xi, u = orc.rng_stream(orc.rng_seed(7, 2048), 100, 4)
    assert orc.lib().orc_get_normal_mode() == 0           # the context manager restores the default
    assert abs(xi.mean()) < 4 / np.sqrt(xi.size) and abs(xi.var() - 1) < 0.01 and abs(np.mean(xi ** 4) - 3) < 0.06
    c = np.corrcoef(xi.reshape(-1, 100), rowvar=False)
    assert np.abs(c - np.eye(100)).max() < 0.08


def test_g18_reference_runs_on_the_f64_stream():
    """the oracle on the 64-bit-mode stream against the reference's Sampler on the same stream"""
    Ct = corr100()
    with orc.normal_mode("f64"):
        z = load("g18_gauss100d_rw_f64")
        m = orc.Gaussian(np.zeros(100), Ct)
        th = z["theta0"].copy()
        ht, hl, acc = orc.run(m, orc.RandomWalk(float(z["prop_factor"]) * Ct), th, m.log_posterior(th),
                              orc.rng_seed(int(z["seed"]), th.shape[0]), 32)
        check_run(z, ht, hl, acc)
        z = load("g18_gauss100d_hmc5_f64")
        th = z["theta0"].copy()
        ht, hl, acc = orc.run(m, orc.HMC(float(z["eps"]), int(z["nsteps"]), 100), th, m.log_posterior(th),
                              orc.rng_seed(int(z["seed"]), th.shape[0]), 12)
        check_run(z, ht, hl, acc)


def test_normals_are_standard_normal():
    xi, u = orc.rng_stream(orc.rng_seed(7, 4096), 100, 4)
    assert abs(xi.mean()) < 4 / np.sqrt(xi.size)
    assert abs(xi.var() - 1) < 0.01
    assert abs(np.mean(xi ** 4) - 3) < 0.05
    c = np.corrcoef(xi.reshape(-1, 100), rowvar=False)
    assert np.abs(c - np.eye(100)).max() < 0.06
    assert 0 <= u.min() and u.max() < 1 and abs(u.mean() - 0.5) < 0.01


def test_python_twin_matches_reference_fixtures():
    """oracle/py_twin.py (the single-chain numpy twin timed by bench.py as the
    'reference-equivalent' CPU rate) against the vectors recorded from the reference."""
    from oracle import py_twin as tw
    z = load("g2_gauss100d_rw")
    Ct = corr100()
    m, p = tw.TwinGaussian(z["mu"], Ct), tw.TwinRandomWalk(float(z["prop_factor"]) * Ct)
    for c in (0, 5):
        th, lp, acc = tw.run_chain(m, p, z["theta0"][c], z["xi"][:, c], z["u"][:, c])
        np.testing.assert_array_equal(acc, z["accept"][:, c])
        np.testing.assert_allclose(th, z["thetas"][:, c], rtol=1e-12, atol=1e-13)
        np.testing.assert_allclose(lp, z["logpost"][:, c], rtol=1e-12)
    z = load("g3_gauss100d_hmc10")
    m = tw.TwinGaussian(np.zeros(100), Ct)
    p = tw.TwinHMC(float(z["eps"]), int(z["nsteps"]), m.grad)
    th, lp, acc = tw.run_chain(m, p, z["theta0"][1], z["xi"][:, 1], z["u"][:, 1])
    np.testing.assert_array_equal(acc, z["accept"][:, 1])
    np.testing.assert_allclose(lp, z["logpost"][:, 1], rtol=1e-10)
    z = load("g4_logreg256x16_mala")
    m = tw.TwinLogReg(z["X"], z["y"], float(z["tau2"]))
    p = tw.TwinHMC(float(z["eps"]), 1, m.grad)
    th, lp, acc = tw.run_chain(m, p, z["theta0"][2], z["xi"][:, 2], z["u"][:, 2])
    np.testing.assert_array_equal(acc, z["accept"][:, 2])
    np.testing.assert_allclose(lp, z["logpost"][:, 2], rtol=1e-10)


def _adapt_check(z, model, prop, T):
    th = np.ascontiguousarray(z["theta0"]).copy()
    N = th.shape[0]
    ad = orc.new_adapt_state(N)
    ht, hl, acc = orc.run(model, orc.set_adapt(prop, float(z["target"])), th, model.log_posterior(th),
                          orc.rng_seed(int(z["seed"]), N), T, adapt=ad)
    check_run(z, ht, hl, acc)
    np.testing.assert_array_equal(ad[:, 0], z["ad_n"])
    np.testing.assert_array_equal(ad[:, 1], z["ad_acc"])
    np.testing.assert_allclose(ad[:, 2], z["ad_scale"], rtol=1e-12)
    np.testing.assert_allclose(ad[:, 1] / ad[:, 0], z["ad_rate"], rtol=1e-15)


def test_g8_adaptive_scale_random_walk_and_hmc():
    """AdaptScaleRandomWalk / AdaptScaleHMC (proposals/adaptive.py:11-35) incl. the
    first-call quirk (last_theta is None => counted as a move)."""
    z = load("g8_adapt_rw_gauss2d")
    _adapt_check(z, orc.Gaussian(z["mu"], z["C"]), orc.RandomWalk(z["Cprop"]), 200)
    z = load("g8_adapt_rw_gauss100d")
    Ct = corr100()
    _adapt_check(z, orc.Gaussian(np.zeros(100), Ct), orc.RandomWalk(float(z["prop_factor"]) * Ct), 64)
    z = load("g8_adapt_hmc_gauss100d")
    _adapt_check(z, orc.Gaussian(np.zeros(100), Ct), orc.HMC(float(z["eps0"]), int(z["nsteps"]), 100), 40)


def test_g9_hmc_with_mass_matrix():
    """VanillaHMC(eps, Nsteps, grad, M) (hamiltonian.py:71-91 with the M branches)."""
    z = load("g9_gauss100d_hmc_mass")
    check_run(z, *replay(z, orc.Gaussian(np.zeros(100), corr100()),
                         orc.HMC(float(z["eps"]), int(z["nsteps"]), 100, M=z["M"])))
    z = load("g9_logreg256x16_hmc_mass")
    check_run(z, *replay(z, orc.LogReg(z["X"], z["y"], float(z["tau2"])),
                         orc.HMC(float(z["eps"]), int(z["nsteps"]), 16, M=z["M"])))


def test_g10_pcn():
    """pCN(C, rho) (proposals/randomwalk.py:78-101)."""
    z = load("g10_pcn_gauss100d")
    Ct = corr100()
    check_run(z, *replay(z, orc.Gaussian(np.zeros(100), Ct),
                         orc.PCN(0.7 * Ct + 0.3 * np.eye(100), float(z["rho"]))))
    z = load("g10_pcn_gauss2d")
    check_run(z, *replay(z, orc.Gaussian(z["mu"], z["C"]), orc.PCN(z["Cprop"], float(z["rho"]))))


MIX1D = [(0.3, [-5.0], [[1.0]]), (0.7, [5.0], [[1.0]])]                       # benchmarks.py:75-78
MIX2D = [(0.3, [0.0, 0.0], [[1.0, 0.9], [0.9, 1.0]]),                         # benchmarks.py:81-84
         (0.7, [0.0, 0.0], [[1.0, -0.9], [-0.9, 1.0]])]


def test_g11_mixture_values_and_random_walk():
    z = load("g11_mixture_values")
    for name, comps in (("mix1d", MIX1D), ("mix2d", MIX2D)):
        m = orc.Mixture(comps)
        np.testing.assert_array_equal(m.weights, z[name + "_weights"])
        th = z[name + "_theta"]
        for fn, key in ((m.log_likelihood, "_logL"), (m.log_prior, "_logP"), (m.log_posterior, "_logpost")):
            got, ref = fn(th), z[name + key]
            assert np.array_equal(np.isinf(got), np.isinf(ref))
            fin = np.isfinite(ref)
            np.testing.assert_allclose(got[fin], ref[fin], rtol=RTOL, atol=1e-15)
        assert np.isinf(z[name + "_logpost"]).any()    # the underflow -> -inf case is in the fixture
    z = load("g11_mixture_rw")
    check_run(z, *replay(z, orc.Mixture(MIX2D), orc.RandomWalk(z["Cprop"])))
    z = load("g11_mixture1d_rw")
    check_run(z, *replay(z, orc.Mixture(MIX1D), orc.RandomWalk(z["Cprop"])))
    for tag in ("mala", "hmc4"):          # the build's mixture gradient inside the reference's VanillaHMC
        z = load("g11_mixture_" + tag)
        check_run(z, *replay(z, orc.Mixture(MIX2D), orc.HMC(float(z["eps"]), int(z["nsteps"]), 2)))


def pt_replay(z, model, prop=None):
    """The oracle's PT run on the fixture's slot layout: (hist_theta, hist_logpost, code)."""
    idx = z["slots"]
    Ne, K = idx.shape
    S, idx2 = orc.pt_slots(Ne, K)
    np.testing.assert_array_equal(idx, idx2)
    th = np.zeros((S, z["theta0"].shape[1]))
    th[idx] = z["theta0"][:, None, :]            # PTSampler gives every rung the same theta0 (:82)
    st = orc.rng_seed(int(z["seed"]), S)
    T = z["swap_prop"].shape[0]
    prop = prop or orc.RandomWalk(z["Cprop"])
    ht, hl, code, _ = orc.run_pt(model, prop, th, st, T, z["betas"], float(z["pswap"]))
    return ht, hl, code, idx.ravel()


def check_pt(z, ht, hl, code, valid):
    np.testing.assert_array_equal(code[:, valid] >> 1, z["swap_prop"][:, valid])
    known = z["swap_acc"][:, valid] >= 0
    sw = z["swap_prop"][:, valid] == 1
    np.testing.assert_array_equal((code[:, valid] & 1)[sw & known], z["swap_acc"][:, valid][sw & known])
    np.testing.assert_array_equal((code[:, valid] & 1)[~sw], z["within_acc"][:, valid][~sw])
    np.testing.assert_allclose(ht[:, valid], z["thetas"][:, valid], rtol=RTOL, atol=1e-12)
    np.testing.assert_allclose(hl[:, valid], z["logpost"][:, valid], rtol=RTOL)
    assert sw.sum() > 20 and (z["swap_acc"][:, valid] == 1).sum() > 5


def test_g12_parallel_tempering():
    z = load("g12_pt_gauss2d")
    check_pt(z, *pt_replay(z, orc.Gaussian(z["mu"], z["C"])))
    z = load("g12_pt_mix2d")
    check_pt(z, *pt_replay(z, orc.Mixture(MIX2D)))
    z = load("g12_pt_hmc_gauss2d")       # HMC within the rungs, base-model gradient
    check_pt(z, *pt_replay(z, orc.Gaussian(z["mu"], z["C"]), orc.HMC(float(z["eps"]), int(z["nsteps"]), 2)))


def test_g13_changepoint_block_proposal():
    """a-11 variant (B): the reference's ChangepointRegression1DProp at fixed k."""
    z = load("g13_changepoint_block")
    model = cp_model(z)
    prop = orc.BlockRandomWalk(np.diag(z["Cdiag"]), z["sizes"], z["pcum"])
    check_run(z, *replay(z, model, prop))
    # which block moved: only the picked block's coordinates differ between proposal and state
    k = int(z["k"])
    moved = np.diff(z["thetas"], axis=0) != 0
    acc = z["accept"].astype(bool)
    assert (moved[:, :, :k].any(-1) == (acc & (z["picked"] == 0))).all()
    assert (moved[:, :, k:2 * k + 1].any(-1) == (acc & (z["picked"] == 1))).all()
    assert not moved[:, :, -1].any()


def test_block_proposal_null_move_and_sig_block():
    """Blocks the reference cannot record here: sig (P = 1 after two failures) and the
    no-block case (the proposal is theta itself, always accepted, nothing changes)."""
    z = load("g13_changepoint_block")
    model = cp_model(z)
    k = int(z["k"])
    th = np.ascontiguousarray(z["theta0"]).copy()
    st = orc.rng_seed(7, th.shape[0])
    lp = model.log_posterior(th)
    prop = orc.BlockRandomWalk(np.diag(z["Cdiag"]), z["sizes"], [0.0, 0.0, 1.0], scale=0.3)
    ht, _, acc = orc.run(model, prop, th, lp, st, 50)
    moved = np.diff(ht, axis=0) != 0
    assert not moved[:, :, :2 * k + 1].any() and moved[:, :, -1].any()
    prop0 = orc.BlockRandomWalk(np.diag(z["Cdiag"]), z["sizes"], [0.0, 0.0, 0.0])
    ht, _, acc = orc.run(model, prop0, th, lp, st, 20)
    assert acc.all() and not (np.diff(ht, axis=0) != 0).any()


def _adaptcov_replay(z, model, C0, seed, th0, T):
    th = np.ascontiguousarray(np.atleast_2d(th0)).copy()
    st = orc.rng_seed(int(seed), 1)
    stc = orc.new_adaptcov_state(C0)
    lp = model.log_posterior(th)
    ht, hl, acc = orc.run_adaptcov(model, orc.RandomWalk(C0), th, lp, st, T, stc, C0,
                                   t_adapt=float(z["t_adapt"]),
                                   marginalize=bool(z["marginalize"]) if "marginalize" in z else False,
                                   smooth_adapt=bool(z["smooth"]) if "smooth" in z else True)
    return ht, hl, acc, orc.adaptcov_CL(stc, th.shape[1])


@pytest.mark.parametrize("tag", ["strict", "smooth", "marg"])
def test_g14_adaptive_covariance_random_walk(tag):
    """AdaptCovRandomWalk of the reference, one chain per run (pooled == per-chain there)."""
    z = load("g14_adaptcov_" + tag)
    model = orc.Gaussian(z["mu"], z["C"])
    T = z["accept"].shape[0]
    for c in range(3):
        ht, hl, acc, (Cf, Lf) = _adaptcov_replay(z, model, z["C0"], z["seeds"][c], z["theta0"][c], T)
        np.testing.assert_array_equal(acc[:, 0], z["accept"][:, c])
        # strict / marg adapt from four samples (t_adapt = 1): Cs = (SX2 - SX SX^T/n)/(n-1)
        # (adaptive.py:83) cancels catastrophically while the chain barely moves, the covariances
        # are nearly singular, and the feedback covariance -> step -> state amplifies the rounding
        # differences between numpy/LAPACK and the loops here: those fixtures are held to the
        # contract's 1e-6, the well-conditioned smooth variant to 1e-8 / 1e-9
        rt = 1e-8 if tag == "smooth" else 1e-6
        np.testing.assert_allclose(ht[:, 0], z["thetas"][:, c], rtol=rt, atol=1e-12)
        np.testing.assert_allclose(hl[:, 0], z["logpost"][:, c], rtol=rt)
        if tag == "smooth":
            np.testing.assert_allclose(Cf, z["C_final"][c], rtol=1e-9, atol=1e-15)
            np.testing.assert_allclose(Lf, z["L_final"][c], rtol=1e-9, atol=1e-15)
        else:
            np.testing.assert_allclose(Cf, z["C_final"][c], rtol=2e-2, atol=1e-15)
    assert not np.allclose(z["C_final"][0], z["C0"] / 2 ** 0.4)      # it did adapt


def test_g14_adaptive_covariance_10d():
    z = load("g14_adaptcov_10d")
    model = orc.Gaussian(np.zeros(10), z["Ct"])
    ht, hl, acc, (Cf, Lf) = _adaptcov_replay(z, model, z["C0"], z["seed"], z["theta0"], 120)
    np.testing.assert_array_equal(acc, z["accept"])
    np.testing.assert_allclose(ht, z["thetas"], rtol=1e-8, atol=1e-12)
    np.testing.assert_allclose(Lf, z["L_final"][0], rtol=1e-8, atol=1e-14)


def big_gauss(d=200, seed=15):
    """The target of fixture G15 (rebuilt from its seed: 320 KB of covariance are not stored)."""
    g = np.random.default_rng(seed)
    A = g.standard_normal((d, d))
    Ct = A @ A.T / d + 0.2 * np.eye(d)
    mu = g.standard_normal(d)
    return mu, Ct, (2.38 ** 2 / d) * Ct


def test_g15_gauss200d_random_walk():
    z = load("g15_gauss200d_rw")
    mu, Ct, Cp = big_gauss()
    check_run(z, *replay(z, orc.Gaussian(mu, Ct), orc.RandomWalk(Cp)))


def test_g14_adaptive_scale_and_covariance_10d():
    """AdaptScaleCovRandomWalk (randomwalk.py:62-75)."""
    z = load("g14_adaptscalecov_10d")
    model = orc.Gaussian(np.zeros(10), z["Ct"])
    th = np.ascontiguousarray(z["theta0"]).copy()
    st, stc, ad = orc.rng_seed(int(z["seed"]), 1), orc.new_adaptcov_state(z["C0"]), orc.new_adapt_state(1)
    prop = orc.set_adapt(orc.RandomWalk(z["C0"]), 0.25)
    ht, hl, acc = orc.run_adaptcov(model, prop, th, model.log_posterior(th), st, 120, stc, z["C0"],
                                   t_adapt=30, smooth_adapt=True, adapt=ad)
    np.testing.assert_array_equal(acc, z["accept"])
    np.testing.assert_allclose(ht, z["thetas"], rtol=1e-8, atol=1e-12)
    np.testing.assert_allclose(ad[0, 2], z["scale_final"][0], rtol=1e-12)
    assert ad[0, 1] == z["Naccepts"][0]
    np.testing.assert_allclose(orc.adaptcov_CL(stc, 10)[1], z["L_final"][0], rtol=1e-8, atol=1e-14)


@pytest.mark.parametrize("name,adaptive", [("g14_adaptcovhmc_10d", False), ("g14_adaptscalecovhmc_10d", True)])
def test_g14_adaptive_covariance_hmc(name, adaptive):
    """AdaptCovHMC (hamiltonian.py:107-120): the adapted covariance is the mass matrix.
    AdaptScaleCovHMC (:123-134) defines no adapt() of its own, so the method resolution order
    finds AdaptScaleProposal.adapt only: the scale adapts, the covariance never does, and the
    class is AdaptScaleHMC with the constant mass matrix M0 -- restated as that.
    Each chain of a fixture is an independent single-chain reference run."""
    z = load(name)
    model = orc.Gaussian(np.zeros(10), z["Ct"])
    T = z["accept"].shape[0]
    for c in range(z["theta0"].shape[0]):
        th = np.ascontiguousarray(z["theta0"][c:c + 1]).copy()
        st = orc.rng_seed(int(z["seed"]), c + 1)[c:c + 1].copy()
        prop = orc.HMC(float(z["eps"]), int(z["nsteps"]), 10, M=z["M0"])
        lp = model.log_posterior(th)
        if adaptive:
            ad = orc.new_adapt_state(1)
            ht, hl, acc = orc.run(model, orc.set_adapt(prop, 0.75), th, lp, st, T, adapt=ad)
            np.testing.assert_allclose(ad[0, 2], z["scale_final"][c], rtol=1e-12)
            np.testing.assert_array_equal(z["C_final"][c], z["M0"])        # never adapted
        else:
            stc = orc.new_adaptcov_state(z["M0"])
            ht, hl, acc = orc.run_adaptcov(model, prop, th, lp, st, T, stc, z["M0"],
                                           t_adapt=float(z["t_adapt"]), smooth_adapt=True)
            np.testing.assert_allclose(orc.adaptcov_CL(stc, 10)[1], z["L_final"][c], rtol=1e-8, atol=1e-14)
        np.testing.assert_array_equal(acc[:, 0], z["accept"][:, c])
        np.testing.assert_allclose(ht[:, 0], z["thetas"][:, c], rtol=1e-8, atol=1e-12)
        np.testing.assert_allclose(hl[:, 0], z["logpost"][:, c], rtol=1e-8)


def test_g10_adaptive_scale_pcn():
    """AdaptScalepCN (randomwalk.py:104-119)."""
    z = load("g10_adaptscalepcn_gauss2d")
    model = orc.Gaussian(z["mu"], z["C"])
    th = np.ascontiguousarray(z["theta0"]).copy()
    N = th.shape[0]
    st, ad, aux = orc.rng_seed(int(z["seed"]), N), orc.new_adapt_state(N), np.full(N, np.nan)
    prop = orc.set_adapt(orc.PCN(z["Cprop"], float(z["rho"])), 0.25)
    ht, hl, acc = orc.run(model, prop, th, model.log_posterior(th), st, 64, adapt=ad, aux=aux)
    check_run(z, ht, hl, acc)
    np.testing.assert_allclose(aux, z["rho_final"], rtol=1e-12)
    np.testing.assert_allclose(ad[:, 2], z["scale_final"], rtol=1e-12)


def test_g16_the_reference_example_run():
    """examples/test_randomwalk.py:34-40 end to end: 10 000 adaptive-HMC steps, burn-in 1000."""
    z = load("g16_example_test_randomwalk")
    model = orc.Gaussian(z["mu"], z["C"])
    th = np.ascontiguousarray(z["theta0"]).copy()
    st, ad = orc.rng_seed(int(z["seed"]), 1), orc.new_adapt_state(1)
    prop = orc.set_adapt(orc.HMC(0.1, 5, 2, M=np.eye(2)), 0.75)
    ht, hl, acc = orc.run(model, prop, th, model.log_posterior(th), st, 10000, adapt=ad)
    np.testing.assert_allclose(ht[1000:], z["thetas"], rtol=1e-7, atol=1e-10)
    np.testing.assert_allclose(hl[1000:], z["logpost"], rtol=1e-7, atol=1e-10)
    assert ad[0, 1] == z["Naccepts"][0]
    np.testing.assert_allclose(ad[0, 2], z["scale_final"][0], rtol=1e-10)


def test_g17_trans_dimensional_changepoint_sampler():
    """The reference's ChangepointRegression1DProp with its birth/death branch
    (riemann/test_changepoint.py:43-73, models/changepoint.py:193-240)."""
    z = load("g17_changepoint_rj")
    KM = int(z["KM"])
    model = orc.Changepoint(z["x"], z["y"], 1, float(z["xmin"]), float(z["xmax"]), float(z["lamb"]),
                            float(z["alpha"]), float(z["beta"]))
    th = np.ascontiguousarray(z["theta0"]).copy()
    N = th.shape[0]
    lp = orc.cprj_log_posterior(model, KM, th)
    np.testing.assert_allclose(lp, z["logpost"][0], rtol=RTOL)
    st = orc.rng_seed(int(z["seed"]), N)
    T = z["accept"].shape[0]
    ht, hl, acc, mv = orc.run_cprj(model, KM, float(z["hscale"]), z["pcum"], th, lp, st, T)
    # a birth proposed at k = KM is "refused" here (code 5); the reference proposes it like any
    # other birth (code 3) -- and, in this fixture, rejects it
    np.testing.assert_array_equal(np.where(mv == 5, 3, mv), z["move"])
    np.testing.assert_array_equal(acc, z["accept"])
    np.testing.assert_array_equal(ht[:, :, -1], z["thetas"][:, :, -1])          # k, exactly
    fin = np.isfinite(z["thetas"])
    assert np.array_equal(np.isfinite(ht), fin)
    np.testing.assert_allclose(ht[fin], z["thetas"][fin], rtol=RTOL, atol=1e-12)
    np.testing.assert_allclose(hl, z["logpost"], rtol=RTOL)
    k = z["thetas"][:, :, -1]
    assert k.min() == 0 and k.max() >= 7          # the fixture walks through many dimensions
