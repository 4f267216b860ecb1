"""GPU parity tests of the 64-bit-resolution normal mode, Sampler(..., normals="f64"): the stream (fixture
G18 and the oracle, bit for bit), the reference's runs recorded on that stream, and every kernel family
that serves the mode against the oracle run in the same mode."""
import numpy as np
import pytest

from gpu_common import RTOL, check_run, corr100, load, oracle_run, rj_model  # noqa: F401

pytestmark = pytest.mark.gpu


def test_f64_stream_bit_exact_vs_fixture_and_oracle(ra, orc):
    from riemann_amd import _lib
    z = load("g18_rng_f64")
    for d in (1, 2, 5, 12, 100):
        xi, u = np.empty((3, 3, d)), np.empty((3, 3))
        _lib.call("rmh_debug_rng_stream_mode", int(z["seed"]), int(z["chain0"]), 3, d, 3, 1, xi, u)
        np.testing.assert_array_equal(xi, z["xi_d%d" % d])
        np.testing.assert_array_equal(u, z["u_d%d" % d])
    N, T = 1000, 3
    for d in (100, 7):
        xi, u = np.empty((T, N, d)), np.empty((T, N))
        _lib.call("rmh_debug_rng_stream_mode", 77, 123456789, N, d, T, 1, xi, u)
        with orc.normal_mode("f64"):
            exi, eu = orc.rng_stream(orc.rng_seed(77, N, chain0=123456789), d, T)
        np.testing.assert_array_equal(xi, exi)
        np.testing.assert_array_equal(u, eu)
    # the host-copy entry point (mixed plugins draw through it), stream state advanced alike
    st, st2 = orc.rng_seed(3, 50), orc.rng_seed(3, 50)
    xi, u = np.empty((2, 50, 9)), np.empty((2, 50))
    _lib.call("rmh_rng_draw_mode", st, 50, 9, 2, 1, xi.ctypes.data, u.ctypes.data)
    with orc.normal_mode("f64"):
        exi, eu = orc.rng_stream(st2, 9, 2)
    np.testing.assert_array_equal(xi, exi)
    np.testing.assert_array_equal(st, st2)


def test_g18_reference_runs_recorded_on_the_f64_stream(ra):
    """cfg2's shape (d = 100: the tail kernel is routed around) and a 5-step HMC trajectory, as the
    reference's Sampler ran them on the 64-bit-mode stream."""
    from riemann_amd.models import benchmarks
    from riemann_amd.proposals import MetropolisRandomWalk, VanillaHMC
    z = load("g18_gauss100d_rw_f64")
    m = benchmarks.benchmark_gauss100d_corr
    s = ra.Sampler(m, MetropolisRandomWalk(float(z["prop_factor"]) * corr100()), z["theta0"], seed=int(z["seed"]),
                   normals="f64")
    s.run(32)
    check_run(z, s)
    z = load("g18_gauss100d_hmc5_f64")
    s = ra.Sampler(m, VanillaHMC(float(z["eps"]), int(z["nsteps"]), m.grad_log_likelihood), z["theta0"],
                   seed=int(z["seed"]), normals="f64")
    s.run(12)
    check_run(z, s)
    # and the default stream gives a different run from the same seed
    s2 = ra.Sampler(m, VanillaHMC(float(z["eps"]), int(z["nsteps"]), m.grad_log_likelihood), z["theta0"],
                    seed=int(z["seed"]))
    s2.run(12)
    assert not np.array_equal(s2._chain_thetas, s._chain_thetas)


def _cases(orc):
    from riemann_amd.models import ChangepointRegression1D, LogisticRegression, MixtureDist, MultiGaussianDist
    from riemann_amd import proposals as P
    from riemann_amd.workloads import changepoint_data, logreg_data
    g = np.random.default_rng(64)

    def spd(d, ridge=0.5):
        A = g.standard_normal((d, d))
        return A @ A.T / d + ridge * np.eye(d)

    out = []
    for d in (3, 36, 100, 112):                      # d = 36, 100: tail dimensions on the generic kernel
        Cm = spd(d)
        out.append(("rw gauss d=%d" % d, MultiGaussianDist(np.zeros(d), Cm), orc.Gaussian(np.zeros(d), Cm),
                    P.MetropolisRandomWalk((2.38 ** 2 / d) * Cm), orc.RandomWalk((2.38 ** 2 / d) * Cm), d, 12))
    d = 40
    Cm, Mm = spd(d), spd(d, 1.0)
    gm, om = MultiGaussianDist(np.zeros(d), Cm), orc.Gaussian(np.zeros(d), Cm)
    Cq = 2.0 * Cm + 0.1 * np.eye(d)                  # (pCN with the target's own covariance accepts every move)
    out.append(("pcn gauss", gm, om, P.pCN(Cq, 0.95), orc.PCN(Cq, 0.95), d, 12))
    out.append(("hmc mass gauss", gm, om, P.VanillaHMC(0.4, 3, gm, M=Mm), orc.HMC(0.4, 3, d, M=Mm), d, 8))
    X, y = logreg_data(256, 100, seed=4)              # d = 100: the gradient-tail kernel is routed around
    out.append(("mala logreg d=100", LogisticRegression(X, y, 100.0), orc.LogReg(X, y, 100.0),
                None, orc.HMC(0.15, 1, 100), 100, 8))
    X2, y2 = logreg_data(200, 20, seed=5)
    lm = LogisticRegression(X2, y2, 100.0)
    out.append(("rw logreg", lm, orc.LogReg(X2, y2, 100.0), P.MetropolisRandomWalk(0.05 * np.eye(20)),
                orc.RandomWalk(0.05 * np.eye(20)), 20, 10))
    comps = [(0.3, np.full(6, -0.5), spd(6)), (0.7, np.full(6, 0.5), spd(6))]
    mm = MixtureDist([[w, MultiGaussianDist(mu, C)] for w, mu, C in comps])
    omix = orc.Mixture(comps)
    out.append(("hmc mixture", mm, omix, P.VanillaHMC(0.3, 3, mm), orc.HMC(0.3, 3, 6), 6, 10))
    x, yy, truth, cdiag = changepoint_data(seed=11, k=3, n=60)
    cm = ChangepointRegression1D(x, yy, 1.0, 3.0, 5.0, 6, 1, 1, k=3)
    out.append(("rw changepoint", cm, orc.Changepoint(x, yy, 3, 1.0, 3.0, 5.0, 1.0, 1.0),
                P.MetropolisRandomWalk(0.1 * np.diag(cdiag)), orc.RandomWalk(0.1 * np.diag(cdiag)), 8, 20))
    return out, truth


def test_every_kernel_family_in_f64_mode_vs_oracle(ra, orc):
    from riemann_amd import proposals as P
    cases, truth = _cases(orc)
    g = np.random.default_rng(1)
    for name, m, om, prop, oprop, d, T in cases:
        N = 300
        th0 = 0.3 * g.standard_normal((N, d))
        if "changepoint" in name:
            th0 = truth + 0.01 * g.standard_normal((N, d))
            th0[:, :3] = np.sort(th0[:, :3], axis=1)
        if prop is None:
            prop = P.MALA(0.15, m)
        s = ra.Sampler(m, prop, th0, seed=21, chain0=32, normals="f64")
        s.run(T)
        with orc.normal_mode("f64"):
            ht, hl, acc = oracle_run(orc, om, oprop, th0, 21, T, chain0=32)
        np.testing.assert_array_equal(s._chain_accept, acc, err_msg=name)
        np.testing.assert_allclose(s._chain_thetas, ht, rtol=1e-8, atol=1e-10, err_msg=name)
        fin = np.isfinite(hl)
        np.testing.assert_allclose(s._chain_logpost[fin], hl[fin], rtol=1e-8, atol=1e-10, err_msg=name)
        assert 0.02 < acc.mean() < 1.0, name
        # the stream continues where the oracle's does: ceil(d/4) words per sub-stream and vector
        from riemann_amd import _lib
        st = np.empty((N, 16), dtype=np.uint64)
        _lib.call("rmh_state_get_rng", s._state, st)
        st2 = orc.rng_seed(21, N, chain0=32)
        with orc.normal_mode("f64"):
            orc.rng_stream(st2, d, T)
        np.testing.assert_array_equal(st, st2, err_msg=name)


def test_adaptive_and_tempered_runs_in_f64_mode_vs_oracle(ra, orc):
    from riemann_amd.models import MultiGaussianDist
    from riemann_amd.proposals import AdaptCovRandomWalk, AdaptScaleRandomWalk, MetropolisRandomWalk
    from riemann_amd.samplers import PTSampler
    g = np.random.default_rng(5)
    d = 100
    A = g.standard_normal((d, d))
    Cm = A @ A.T / d + 0.5 * np.eye(d)
    Cp = (2.38 ** 2 / d) * Cm
    m, om = MultiGaussianDist(np.zeros(d), Cm), orc.Gaussian(np.zeros(d), Cm)
    # per-chain scale adaptation
    N, T = 200, 30
    th0 = g.standard_normal((N, d))
    pr = AdaptScaleRandomWalk(Cp)
    s = ra.Sampler(m, pr, th0, seed=8, normals="f64")
    s.run(T)
    with orc.normal_mode("f64"):
        th = th0.copy()
        ad = orc.new_adapt_state(N)
        ht, hl, acc = orc.run(om, orc.set_adapt(orc.RandomWalk(Cp), 0.25), th, om.log_posterior(th),
                              orc.rng_seed(8, N), T, adapt=ad)
    np.testing.assert_array_equal(s._chain_accept, acc)
    np.testing.assert_allclose(s._chain_thetas, ht, rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(pr.scale, ad[:, 2], rtol=1e-12)
    # pooled covariance adaptation (the fused-moment kernel is a 32-bit tail kernel: routed to the scratch history)
    pc = AdaptCovRandomWalk(Cp, t_adapt=20, smooth_adapt=True)
    s = ra.Sampler(m, pc, th0, seed=9, normals="f64")
    s.run(40)
    with orc.normal_mode("f64"):
        th = th0.copy()
        st = orc.new_adaptcov_state(Cp)
        ht, hl, acc = orc.run_adaptcov(om, orc.RandomWalk(Cp), th, om.log_posterior(th), orc.rng_seed(9, N), 40, st,
                                       Cp, t_adapt=20, smooth_adapt=True)
    np.testing.assert_array_equal(s._chain_accept, acc)
    np.testing.assert_allclose(s._chain_thetas, ht, rtol=1e-7, atol=1e-9)
    # parallel tempering, K = 5
    Ne, K, T = 150, 5, 16
    betas = 0.6 ** np.arange(K)
    th0 = g.standard_normal((Ne, d))
    pt = PTSampler(m, MetropolisRandomWalk(Cp), th0, betas=betas, Pswap=0.25, seed=5, chain0=4096, normals="f64")
    pt.run(T)
    S, idx = orc.pt_slots(Ne, K)
    slots = np.zeros((S, d))
    slots[idx] = th0[:, None, :]
    with orc.normal_mode("f64"):
        ht, hl, code, _ = orc.run_pt(om, orc.RandomWalk(Cp), slots, orc.rng_seed(5, S, chain0=4096), T, betas, 0.25)
    np.testing.assert_array_equal(pt._host_codes, code[:, idx])
    np.testing.assert_allclose(pt._host_thetas, ht[:, idx], rtol=RTOL, atol=1e-12)
    np.testing.assert_allclose(pt._host_logpost, hl[:, idx], rtol=RTOL)


def test_f64_mode_refusals_and_checkpoints(ra, tmp_path):
    from riemann_amd.models import MultiGaussianDist
    from riemann_amd.proposals import ChangepointRegression1DProp, MetropolisRandomWalk
    from riemann_amd.sampling_errors import ParameterError
    m = MultiGaussianDist(np.zeros(4), np.eye(4))
    with pytest.raises(ParameterError, match="normals must be"):
        ra.Sampler(m, MetropolisRandomWalk(np.eye(4)), np.zeros((3, 4)), normals="f16")
    big = MultiGaussianDist(np.zeros(130), np.eye(130))
    with pytest.raises(ParameterError, match="d > 112"):
        ra.Sampler(big, MetropolisRandomWalk(np.eye(130)), np.zeros((3, 130)), normals="f64")
    z = load("g17_changepoint_rj")
    rj = rj_model(z)
    with pytest.raises(ParameterError, match="variable-k"):
        ra.Sampler(rj, ChangepointRegression1DProp(rj, float(z["hscale"])), z["theta0"], normals="f64")

    class HostModel(object):                     # a Python plugin next to the device proposal: the mixed path
        def log_posterior(self, th):
            return -0.5 * float(np.sum(th * th))

    class HostProposal(object):
        def propose(self, th):
            return th, 0.0
    with pytest.raises(ParameterError, match="draws its own"):
        ra.Sampler(HostModel(), HostProposal(), np.zeros(4), normals="f64")
    # mixed plugins in the 64-bit mode == the fused device run of the same target; stand-alone propose() likewise
    th0 = np.random.default_rng(3).standard_normal((20, 4))
    mixed = ra.Sampler(HostModel(), MetropolisRandomWalk(0.5 * np.eye(4)), th0, seed=6, normals="f64")
    mixed.run(15)
    fused = ra.Sampler(m, MetropolisRandomWalk(0.5 * np.eye(4)), th0, seed=6, normals="f64")
    fused.run(15)
    np.testing.assert_array_equal(mixed._chain_accept, fused._chain_accept)
    np.testing.assert_allclose(mixed._chain_thetas, fused._chain_thetas, rtol=1e-12, atol=1e-14)
    fused32 = ra.Sampler(m, MetropolisRandomWalk(0.5 * np.eye(4)), th0, seed=6)
    fused32.run(15)
    assert not np.array_equal(fused32._chain_thetas, fused._chain_thetas)
    pr = MetropolisRandomWalk(0.5 * np.eye(4))
    pr.normals = "f64"
    pr.seed(6)
    thp, lq = pr.propose(th0)
    np.testing.assert_allclose(thp[fused._chain_accept[0] == 1], fused._chain_thetas[1][fused._chain_accept[0] == 1],
                               rtol=1e-12, atol=1e-14)
    # a checkpoint remembers its mode; resuming continues the same chains
    th0 = np.random.default_rng(0).standard_normal((40, 4))
    a = ra.Sampler(m, MetropolisRandomWalk(np.eye(4)), th0, seed=2, normals="f64")
    a.run(10)
    a.checkpoint(str(tmp_path / "c.npz"))
    a.run(10)
    b = ra.Sampler(m, MetropolisRandomWalk(np.eye(4)), th0, seed=99, normals="f64")
    b.restore(str(tmp_path / "c.npz"))
    b.run(10)
    np.testing.assert_array_equal(b._chain_thetas[-10:], a._chain_thetas[-10:])
    np.testing.assert_array_equal(b._chain_accept[-10:], a._chain_accept[-10:])
    c = ra.Sampler(m, MetropolisRandomWalk(np.eye(4)), th0, seed=2)
    with pytest.raises(ParameterError, match="normals="):
        c.restore(str(tmp_path / "c.npz"))
