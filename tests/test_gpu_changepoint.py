"""GPU parity tests of the changepoint model: fixed k (fixtures G5, G13), the generic fused
kernels on it, and trans-dimensional sampling (fixture G17)."""
import zlib

import numpy as np
import pytest

from gpu_common import RTOL, check_run, cp_model, load, oracle_run, rj_model, scaled_rw  # noqa: F401

pytestmark = pytest.mark.gpu


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


def test_g5_changepoint_rw(ra):
    z = load("g5_changepoint_rw")
    s = ra.Sampler(cp_model(z), scaled_rw(z), z["theta0"], seed=int(z["seed"]))
    s.run(64)
    check_run(z, s)


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


@pytest.mark.parametrize("policy", [(), ("CP_SS64",)])
def test_g13_changepoint_block_proposal_vs_reference_fixture(ra, policy):
    """policy CP_SS64: the one-lane-per-chain kernel (k_cp_rw_ss64<.., BLK>), which takes over by itself
    from 65 536 chains."""
    from riemann_amd._lib import kernel_policy
    from riemann_amd.proposals import ChangepointRegression1DProp
    z = load("g13_changepoint_block")
    m = cp_model(z)
    p = ChangepointRegression1DProp(m, float(z["hscale"]))
    assert (p.P_cumprop_cpx, p.P_cumprop_cpv, p.P_cumprop_sig, p.P_cumprop_dim) == (0.2, 0.4, 0.6, 1.0)
    np.testing.assert_allclose(np.diag(p.L) ** 2, z["Cdiag"], rtol=1e-14)
    p.P_cumprop_cpv = 1.0        # as the fixture: cpx with 0.2, else cpv
    with kernel_policy(*policy):
        s = ra.Sampler(m, p, z["theta0"], seed=int(z["seed"]))
        s.run(160)
    check_run(z, s)


@pytest.mark.parametrize("policy", [(), ("CP_SS64",), ("CP_SS64", "CP_NO_BUCKETS")])
def test_changepoint_block_proposal_many_chains_vs_oracle(ra, orc, policy):
    """Default probabilities (0.2 / 0.32 / 0.288, else the null move) on 4096 chains (4100 on the
    one-lane-per-chain kernel: a last wavefront of four chains)."""
    from riemann_amd._lib import kernel_policy
    from riemann_amd.proposals import ChangepointRegression1DProp
    from riemann_amd import workloads as W
    x, y, truth, cdiag = W.changepoint_data()
    k, d, N, T = 5, 12, 4096 + (4 if policy else 0), 60
    from riemann_amd.models import ChangepointRegression1D
    m = ChangepointRegression1D(x, y, 1.0, 3.0, 5.0, 10, 1, 1, k=k)
    p = ChangepointRegression1DProp(m, 2.0)
    p.scale = 0.5
    g = np.random.default_rng(8)
    th0 = truth + 0.01 * g.standard_normal((N, d))
    th0[:, :k] = np.sort(th0[:, :k], axis=1)
    with kernel_policy(*policy):
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


@pytest.mark.parametrize("policy", [(), ("CP_SS64",)])
def test_g17_trans_dimensional_changepoint_sampler_vs_reference_fixture(ra, policy):
    """The reference's ChangepointRegression1DProp with its birth/death branch: moves, accepts,
    the number of changepoints and the states of 4 chains x 300 iterations.  policy CP_SS64: the
    one-lane-per-chain kernel k_cprj_run64, which takes over by itself from 65 536 chains."""
    from riemann_amd._lib import kernel_policy
    from riemann_amd.proposals import ChangepointRegression1DProp
    z = load("g17_changepoint_rj")
    m = rj_model(z)
    start = m.unpack_padded(z["theta0"])
    np.testing.assert_array_equal(m.pack_padded(start), z["theta0"])
    np.testing.assert_allclose(m.log_posterior(z["theta0"]), z["logpost"][0], rtol=RTOL)
    p = ChangepointRegression1DProp(m, float(z["hscale"]))
    assert p.P_cumprop_dim == 1.0
    p.P_cumprop_sig = 0.0                    # as the fixture (the reference's sig block raises under numpy 2)
    with kernel_policy(*policy):
        s = ra.Sampler(m, p, start, seed=int(z["seed"]))
        s.run(300)
    code = s._chain_accept
    mv = code >> 1          # a birth at k = kmax is "refused" (5) where the reference proposes (3) and rejects
    np.testing.assert_array_equal(np.where(mv == 5, 3, mv), z["move"])
    np.testing.assert_array_equal(code & 1, z["accept"])
    th = s._chain_thetas
    np.testing.assert_array_equal(th[:, :, -1], z["thetas"][:, :, -1])
    fin = np.isfinite(z["thetas"])
    assert np.array_equal(np.isfinite(th), fin)
    np.testing.assert_allclose(th[fin], z["thetas"][fin], rtol=RTOL, atol=1e-12)
    np.testing.assert_allclose(s._chain_logpost, z["logpost"], rtol=RTOL)
    last = m.unpack_padded(th[-1])
    assert [len(t.cpx) for t in last] == list(z["thetas"][-1, :, -1].astype(int))


@pytest.mark.parametrize("KM,N,policy", [(5, 3000, ()), (12, 1000, ()), (5, 3001, ("CP_SS64",)), (12, 1000, ("CP_SS64",)),
                                         (10, 2100, ("CP_SS64",)), (10, 700, ("CP_SS64", "CP_NO_BUCKETS"))])
def test_trans_dimensional_changepoint_many_chains_vs_oracle(ra, orc, KM, N, policy):
    """Default probabilities (all five move types, births refused at capacity) on many chains."""
    from riemann_amd._lib import kernel_policy
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
    T = 80
    with kernel_policy(*policy):
        s = ra.Sampler(m, p, th0, seed=6, chain0=160)
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


@pytest.mark.parametrize("case", ["n33", "n127_no_padding", "n500_dense", "clustered", "duplicates", "unsorted_k3"])
def test_one_lane_per_chain_kernel_search_variants_vs_oracle(ra, orc, case):
    """k_cp_rw_ss64 finds the piece boundaries through a bucket table (two or three search steps
    inside a bucket) or, where the data do not allow one (no +inf padding left, clustered x), through
    the full binary search; data sizes on both sides of the compile-time search depths, duplicate and
    unsorted x.  Forced here (by itself the kernel takes over at 65 536 chains); states, log
    posteriors and accept masks against the oracle's O(n) likelihood."""
    from riemann_amd._lib import kernel_policy
    from riemann_amd import workloads as W
    from riemann_amd.models import ChangepointRegression1D
    from riemann_amd.proposals import MetropolisRandomWalk
    g = np.random.default_rng(zlib.crc32(case.encode()) % 1000)     # (not hash(): salted per process)
    k = 3 if case == "unsorted_k3" else 5
    n = {"n33": 33, "n127_no_padding": 127, "n500_dense": 500}.get(case, 100)
    x, y, truth, cdiag = W.changepoint_data(seed=11, k=k, n=n)
    if case == "clustered":        # 90 points inside 1 % of the range: no bucket table can split them
        x = np.sort(np.concatenate([1.5 + 0.02 * g.uniform(size=90), g.uniform(1.0, 3.0, size=10)]))
        y = truth[k + np.searchsorted(truth[:k], x)] + 0.1 * g.standard_normal(n)
    if case == "duplicates":
        x = np.sort(np.round(x, 1))
        y = truth[k + np.searchsorted(truth[:k], x)] + 0.1 * g.standard_normal(n)
    if case == "unsorted_k3":
        perm = g.permutation(n)
        x, y = x[perm], y[perm]
    d, N, T = 2 * k + 2, 700, 50
    th0 = truth + 0.01 * g.standard_normal((N, d))
    th0[:, :k] = np.sort(th0[:, :k], axis=1)
    with kernel_policy("CP_SS64"):
        s = ra.Sampler(ChangepointRegression1D(x, y, 1.0, 3.0, 5.0, 10, 1, 1, k=k),
                       MetropolisRandomWalk(np.diag(cdiag)), th0, seed=4, chain0=9)
        s.run(T)
    om = orc.Changepoint(x, y, k, 1.0, 3.0, 5.0, 1.0, 1.0)
    th, st = th0.copy(), orc.rng_seed(4, N, chain0=9)
    ht, hl, acc = orc.run(om, orc.RandomWalk(np.diag(cdiag)), th, om.log_posterior(th), st, T)
    np.testing.assert_array_equal(s._chain_accept, acc)
    np.testing.assert_allclose(s._chain_thetas, ht, rtol=RTOL, atol=1e-12)
    fin = np.isfinite(hl)
    # the log posterior of these chains passes through zero (terms of size ~10 cancel): a relative
    # bound alone fails by chance on a value of 1e-4 (seen: 5e-10 relative = 5e-14 absolute)
    np.testing.assert_allclose(s._chain_logpost[fin], hl[fin], rtol=RTOL, atol=1e-11)
    assert 0.01 < acc.mean() < 0.95


@pytest.mark.parametrize("k,n", [(5, 100), (3, 60), (8, 100)])
def test_adapted_scale_walk_on_the_one_lane_per_chain_kernel_vs_oracle_and_generic_kernel(ra, orc, k, n):
    """AdaptScaleRandomWalk on the changepoint model: k_cp_rw_ss64<..., ADAPT> (forced here by the kernel policy; by
    itself it takes over at 65 536 chains) against the oracle and against the generic kernel over the changepoint
    policy -- accept masks, states, log posteriors and the chains' adapted scales."""
    from riemann_amd._lib import kernel_policy
    from riemann_amd import workloads as W
    from riemann_amd.models import ChangepointRegression1D
    from riemann_amd.proposals import AdaptScaleRandomWalk
    g = np.random.default_rng(100 * k + n)
    x, y, truth, cdiag = W.changepoint_data(seed=11, k=k, n=n)
    d, N, T = 2 * k + 2, 700, 60
    th0 = truth + 0.01 * g.standard_normal((N, d))
    th0[:, :k] = np.sort(th0[:, :k], axis=1)
    Cp = 0.1 * np.diag(cdiag)
    out = {}
    for pol in (("CP_SS64",), ()):
        with kernel_policy(*pol):
            p = AdaptScaleRandomWalk(Cp)
            s = ra.Sampler(ChangepointRegression1D(x, y, 1.0, 3.0, 5.0, 2 * k, 1, 1, k=k), p, th0, seed=4, chain0=9)
            s.run(T // 2)
            a1 = s._chain_accept.copy()
            s.run(T - T // 2)                      # the adaptation state persists across run() calls
            out[pol] = (np.concatenate([a1, s._chain_accept]), s._chain_thetas.copy(), s._chain_logpost.copy(),
                        np.array(p.scale).copy(), np.array(p.Nsamples).copy())
    om = orc.Changepoint(x, y, k, 1.0, 3.0, 5.0, 1.0, 1.0)
    th, st, ad = th0.copy(), orc.rng_seed(4, N, chain0=9), orc.new_adapt_state(N)
    ht, hl, acc = orc.run(om, orc.set_adapt(orc.RandomWalk(Cp), 0.25), th, om.log_posterior(th), st, T, adapt=ad)
    for pol in out:
        a, th_, lp_, sc, ns = out[pol]
        np.testing.assert_array_equal(a, acc)
        np.testing.assert_allclose(th_, ht[T // 2:], rtol=RTOL, atol=1e-12)
        fin = np.isfinite(hl[T // 2:])
        np.testing.assert_allclose(lp_[fin], hl[T // 2:][fin], rtol=RTOL, atol=1e-11)
        np.testing.assert_allclose(sc, ad[:, 2], rtol=1e-12)
        np.testing.assert_array_equal(ns, ad[:, 0])
    assert 0.05 < acc.mean() < 0.9
