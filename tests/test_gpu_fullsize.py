"""Full-size property tests of BASELINE.json configs 3, 4 and 5: 65 536 chains per GPU (the size
bench.py measures) on the benchmark's own data, through size-independent properties --
  (i)   a shard started with chain0 = a equals chains [a, b) of the whole job, bit for bit
        (the multi-GPU partition is exact);
  (ii)  run(a) + run(b) == run(a + b), with and without a recorded history;
  (iii) thinning keeps every Nthin-th state --
plus oracle comparisons with every wave of a workgroup live at n = 10 000 (512 chains: the
8-waves-per-workgroup staging of X streams 313 stages past eight waves).  cfg2's twin is
test_gpu_parity.py::test_full_size_sharding_and_resume_invariance."""
import numpy as np
import pytest

from gpu_common import RTOL, oracle_run

pytestmark = pytest.mark.gpu

N = 65536


def _properties(ra, make, T1, T2, thin, shard=(16384, 32768)):
    """make() -> (model, proposal, theta0 (N, d)); the proposal object is shared by the samplers."""
    m, p, th0 = make()
    whole = ra.Sampler(m, p, th0, seed=1)
    whole.run(T1 + T2)
    th_w, lp_w = whole.current_state()
    acc_w = whole._chain_accept.copy()
    hist_w = whole._chain_thetas[T1::thin].copy() if T2 % thin == 0 else None
    del whole
    a, b = shard
    part = ra.Sampler(m, p, th0[a:b], seed=1, chain0=a)
    part.run(T1 + T2)
    th_s, lp_s = part.current_state()
    np.testing.assert_array_equal(th_s, th_w[a:b])
    np.testing.assert_array_equal(lp_s, lp_w[a:b])
    np.testing.assert_array_equal(part._chain_accept, acc_w[:, a:b])
    del part
    two = ra.Sampler(m, p, th0, seed=1)
    two.run(T1, record=False)
    two.run(T2, 0, thin)
    th_t, lp_t = two.current_state()
    np.testing.assert_array_equal(th_t, th_w)
    np.testing.assert_array_equal(lp_t, lp_w)
    if hist_w is not None:
        np.testing.assert_array_equal(two._chain_thetas, hist_w)
    return acc_w


def _logreg(eps, L):
    from riemann_amd.models import LogisticRegression
    from riemann_amd.proposals import MALA, VanillaHMC
    from riemann_amd.workloads import logreg_data
    X, y = logreg_data(10000, 100, seed=2)
    m = LogisticRegression(X, y, 100.0)
    th0 = 0.02 * np.random.default_rng(3).standard_normal((N, 100))
    return m, (MALA(eps, m) if L == 1 else VanillaHMC(eps, L, m)), th0


def test_cfg3_logistic_mala_full_size_properties(ra):
    acc = _properties(ra, lambda: _logreg(0.175, 1), 6, 6, 3)
    assert 0.3 < acc.mean() < 0.9


def test_cfg4_logistic_hmc_full_size_properties(ra):
    acc = _properties(ra, lambda: _logreg(0.10, 10), 1, 2, 1)
    assert 0.3 < acc.mean() <= 1.0


def test_cfg5_changepoint_full_size_properties(ra):
    from riemann_amd.models import ChangepointRegression1D
    from riemann_amd.proposals import MetropolisRandomWalk
    from riemann_amd.workloads import changepoint_data

    def make():
        x, y, truth, cdiag = changepoint_data()
        th0 = truth + 0.01 * np.random.default_rng(5).standard_normal((N, 12))
        th0[:, :5] = np.sort(th0[:, :5], axis=1)
        p = MetropolisRandomWalk(np.diag(cdiag))
        p.scale = 0.1
        return ChangepointRegression1D(x, y, 1.0, 3.0, 5.0, 10, 1, 1, k=5), p, th0

    acc = _properties(ra, make, 120, 90, 3)
    assert 0.05 < acc.mean() < 0.9


def test_cfg5_block_mixture_and_trans_dimensional_proposals_full_size_properties(ra):
    """The reference's own changepoint proposals at the bench size, on the sufficient-statistics
    kernels: the block mixture at fixed k and the sampler over the number of changepoints."""
    from riemann_amd.models import ChangepointParams, ChangepointRegression1D
    from riemann_amd.proposals import ChangepointRegression1DProp
    from riemann_amd.workloads import changepoint_data
    x, y, truth, _ = changepoint_data()

    def block():
        th0 = truth + 0.01 * np.random.default_rng(6).standard_normal((N, 12))
        th0[:, :5] = np.sort(th0[:, :5], axis=1)
        m = ChangepointRegression1D(x, y, 1.0, 3.0, 5.0, 10, 1, 1, k=5)
        p = ChangepointRegression1DProp(m, 2.0)
        p.scale = 0.5
        return m, p, th0

    acc = _properties(ra, block, 60, 45, 3)
    assert 0.02 < acc.mean() < 0.9

    def rj():
        m = ChangepointRegression1D(x, y, 1.0, 3.0, 5.0, 10, 1, 1, variable_k=True)
        p = ChangepointRegression1DProp(m, 2.0)
        th0 = m.pack_padded([ChangepointParams([2.0], [1.0, 3.0], 0.1)] * N)
        return m, p, th0

    acc = _properties(ra, rj, 60, 45, 3)
    assert 0.01 < (acc & 1).mean() < 0.9


@pytest.mark.parametrize("L,eps,T", [(1, 0.175, 5), (4, 0.12, 2), (10, 0.10, 1)])
def test_logistic_10000x100_vs_oracle_with_eight_live_waves(ra, orc, L, eps, T):
    """512 chains = four full workgroups of eight waves on the cfg3/cfg4 data against the oracle;
    L = 10 is cfg4's own trajectory: nine gradient-only evaluations with all eight waves live."""
    from riemann_amd.models import LogisticRegression
    from riemann_amd.proposals import MALA, VanillaHMC
    from riemann_amd.workloads import logreg_data
    X, y = logreg_data(10000, 100, seed=2)
    m = LogisticRegression(X, y, 100.0)
    th0 = 0.02 * np.random.default_rng(9).standard_normal((512, 100))
    s = ra.Sampler(m, MALA(eps, m) if L == 1 else VanillaHMC(eps, L, m), th0, seed=4)
    s.run(T)
    ht, hl, acc = oracle_run(orc, orc.LogReg(X, y, 100.0), orc.HMC(eps, L, 100), th0, 4, T)
    np.testing.assert_array_equal(s._chain_accept, acc)
    np.testing.assert_allclose(s._chain_thetas, ht, rtol=RTOL, atol=1e-12)
    np.testing.assert_allclose(s._chain_logpost, hl, rtol=RTOL)


def test_cfg5_changepoint_16384_chains_vs_oracle(ra, orc):
    """The sufficient-statistics kernel on the cfg5 data against the oracle's O(n) likelihood."""
    from riemann_amd.models import ChangepointRegression1D
    from riemann_amd.proposals import MetropolisRandomWalk
    from riemann_amd.workloads import changepoint_data
    x, y, truth, cdiag = changepoint_data()
    n = 16384
    th0 = truth + 0.01 * np.random.default_rng(5).standard_normal((n, 12))
    th0[:, :5] = np.sort(th0[:, :5], axis=1)
    p = MetropolisRandomWalk(np.diag(cdiag))
    p.scale = 0.1
    s = ra.Sampler(ChangepointRegression1D(x, y, 1.0, 3.0, 5.0, 10, 1, 1, k=5), p, th0, seed=6)
    s.run(100)
    ht, hl, acc = oracle_run(orc, orc.Changepoint(x, y, 5, 1.0, 3.0, 5.0, 1.0, 1.0),
                             orc.RandomWalk(np.diag(cdiag), 0.1), th0, 6, 100)
    np.testing.assert_array_equal(s._chain_accept, acc)
    np.testing.assert_allclose(s._chain_thetas, ht, rtol=RTOL, atol=1e-12)
    fin = np.isfinite(hl)
    assert np.array_equal(np.isfinite(s._chain_logpost), fin)
    np.testing.assert_allclose(s._chain_logpost[fin], hl[fin], rtol=RTOL)


def test_cfg5_three_kernels_agree_at_full_size(ra):
    """cfg5 at 65 536 chains through the three kernels that can run it -- k_cp_rw (O(n) likelihood,
    round 1), k_cp_rw_ss (sufficient statistics, four lanes per chain) and k_cp_rw_ss64 (one lane
    per chain): same stream, same accept decisions, states equal to rounding."""
    from riemann_amd._lib import kernel_policy
    from riemann_amd.models import ChangepointRegression1D
    from riemann_amd.proposals import MetropolisRandomWalk
    from riemann_amd.workloads import changepoint_data
    x, y, truth, cdiag = changepoint_data()
    th0 = truth + 0.01 * np.random.default_rng(5).standard_normal((N, 12))
    th0[:, :5] = np.sort(th0[:, :5], axis=1)
    out = {}
    for name, pol in (("ss64", ()), ("ss16", ("CP_SS16",)), ("on", ("CP_NO_SS",))):
        with kernel_policy(*pol):
            p = MetropolisRandomWalk(np.diag(cdiag))
            p.scale = 0.1
            s = ra.Sampler(ChangepointRegression1D(x, y, 1.0, 3.0, 5.0, 10, 1, 1, k=5), p, th0, seed=8)
            s.run(150, 0, 50)
            out[name] = (s._chain_accept.copy(), s._chain_thetas.copy(), s._chain_logpost.copy())
    for name in ("ss16", "on"):
        np.testing.assert_array_equal(out["ss64"][0], out[name][0])
        np.testing.assert_allclose(out["ss64"][1], out[name][1], rtol=RTOL, atol=1e-12)
        np.testing.assert_allclose(out["ss64"][2], out[name][2], rtol=RTOL)


@pytest.mark.parametrize("n", [1000, 17, 64 * 9 + 16])
def test_cfg5_one_lane_per_chain_kernel_ragged_chain_counts(ra, orc, n):
    """k_cp_rw_ss64 owns 64 chains per wavefront: chain counts that leave its last wavefront one,
    two or three 16-chain blocks short (forced here; by itself it takes over at 65 536 chains)."""
    from riemann_amd._lib import kernel_policy
    from riemann_amd.models import ChangepointRegression1D
    from riemann_amd.proposals import MetropolisRandomWalk
    from riemann_amd.workloads import changepoint_data
    x, y, truth, cdiag = changepoint_data()
    th0 = truth + 0.01 * np.random.default_rng(n).standard_normal((n, 12))
    th0[:, :5] = np.sort(th0[:, :5], axis=1)
    p = MetropolisRandomWalk(np.diag(cdiag))
    p.scale = 0.1
    with kernel_policy("CP_SS64"):
        s = ra.Sampler(ChangepointRegression1D(x, y, 1.0, 3.0, 5.0, 10, 1, 1, k=5), p, th0, seed=6, chain0=5)
        s.run(60, 7, 2)
        s.run(41)
    om = orc.Changepoint(x, y, 5, 1.0, 3.0, 5.0, 1.0, 1.0)
    th, st = th0.copy(), orc.rng_seed(6, n, chain0=5)
    lp = om.log_posterior(th)
    ht, hl, _ = orc.run(om, orc.RandomWalk(np.diag(cdiag), 0.1), th, lp, st, 60)
    th, lp = ht[7::2][-1].copy(), hl[7::2][-1].copy()           # the last KEPT state (59)
    ht, hl, acc = orc.run(om, orc.RandomWalk(np.diag(cdiag), 0.1), th, lp, st, 41)
    np.testing.assert_array_equal(s._chain_accept, acc)
    np.testing.assert_allclose(s._chain_thetas, ht, rtol=RTOL, atol=1e-12)


def test_chain_count_limit_of_the_32_bit_buffer_offsets(ra):
    """The kernels address a (rows, chains) array through 32-bit byte offsets: (16 NT + 4) (pad16(N) + 128) 8
    must stay below 2^32 (rmh_state_create; the + 4 rows and + 128 columns are the HMC kernels' parking
    array).  Just below the limit a run addresses its last chains correctly (they move as the same global
    chain ids move in a small Sampler); just above, construction fails with the library's message."""
    from riemann_amd.models import MultiGaussianDist
    from riemann_amd.proposals import MetropolisRandomWalk
    d = 2                                            # one row tile: 20 rows
    lim = (1 << 32) // (20 * 8) - 128                # chains (padded to 16) at which the product reaches 2^32
    n_ok = (lim - 16) // 16 * 16
    m, p = MultiGaussianDist(np.zeros(d), np.eye(d)), MetropolisRandomWalk(0.5 * np.eye(d))
    th0 = np.zeros((n_ok, d))
    th0[-40:] = np.arange(80, dtype=np.float64).reshape(40, 2) / 40.0
    big = ra.Sampler(m, p, th0, seed=3)
    big.run(5, record=False)
    tail, _ = big.current_state()
    tail = tail[-40:].copy()
    del big
    small = ra.Sampler(m, p, th0[-40:], seed=3, chain0=n_ok - 40)
    small.run(5, record=False)
    np.testing.assert_array_equal(small.current_state()[0], tail)
    # the HMC kernels park momenta in a (rows + 4, chains + 128) array behind the same offsets
    from riemann_amd.proposals import MALA
    big = ra.Sampler(m, MALA(0.4, m), th0, seed=4)
    big.run(3, record=False)
    tail = big.current_state()[0][-40:].copy()
    del big
    small = ra.Sampler(m, MALA(0.4, m), th0[-40:], seed=4, chain0=n_ok - 40)
    small.run(3, record=False)
    np.testing.assert_array_equal(small.current_state()[0], tail)
    with pytest.raises(ra.ParameterError, match="too many chains"):
        ra.Sampler(m, p, np.zeros((lim + 32, d)), seed=3)


def test_late_rank_shards_of_the_multi_gpu_jobs(ra):
    """The shards the LAST ranks of BASELINE cfg4 (8 x 65 536 chains: rank 7 starts at chain
    458 752) and cfg5 (4 x 65 536: rank 3 at 196 608) run, against those chains of the whole job
    run in one process: bit for bit (32-bit slot arithmetic and stream ids at the far end)."""
    from riemann_amd.models import ChangepointRegression1D
    from riemann_amd.proposals import MetropolisRandomWalk
    from riemann_amd.workloads import changepoint_data
    # cfg4: logistic 10 000 x 100, HMC L = 10
    m, p, _ = _logreg(0.10, 10)
    n4, a4 = 8 * N, 7 * N
    th0 = (0.02 * np.random.default_rng(3).standard_normal((n4, 100))).astype(np.float64)
    whole = ra.Sampler(m, p, th0, seed=1)
    whole.run(1)
    th_w, lp_w = whole.current_state()
    acc_w = whole._chain_accept[:, a4:].copy()
    th_w, lp_w = th_w[a4:].copy(), lp_w[a4:].copy()
    del whole
    part = ra.Sampler(m, p, th0[a4:], seed=1, chain0=a4)
    part.run(1)
    np.testing.assert_array_equal(part.current_state()[0], th_w)
    np.testing.assert_array_equal(part.current_state()[1], lp_w)
    np.testing.assert_array_equal(part._chain_accept, acc_w)
    assert 0.3 < acc_w.mean() <= 1.0
    del part, th0
    # cfg5: changepoint model, random walk
    x, y, truth, cdiag = changepoint_data()
    n5, a5 = 4 * N, 3 * N
    th0 = truth + 0.01 * np.random.default_rng(5).standard_normal((n5, 12))
    th0[:, :5] = np.sort(th0[:, :5], axis=1)
    pr = MetropolisRandomWalk(np.diag(cdiag))
    pr.scale = 0.1
    mc = ChangepointRegression1D(x, y, 1.0, 3.0, 5.0, 10, 1, 1, k=5)
    whole = ra.Sampler(mc, pr, th0, seed=2)
    whole.run(60, 0, 20)
    ref = (whole._chain_thetas[:, a5:].copy(), whole._chain_logpost[:, a5:].copy(), whole._chain_accept[:, a5:].copy())
    del whole
    part = ra.Sampler(mc, pr, th0[a5:], seed=2, chain0=a5)
    part.run(60, 0, 20)
    np.testing.assert_array_equal(part._chain_thetas, ref[0])
    np.testing.assert_array_equal(part._chain_logpost, ref[1])
    np.testing.assert_array_equal(part._chain_accept, ref[2])
    assert 0.05 < ref[2].mean() < 0.9
