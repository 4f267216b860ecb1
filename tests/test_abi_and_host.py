"""CPU-only checks of the product side: the C-ABI library loads and exports
every symbol include/riemann_hip.h declares, the ctypes table covers them, the
host logic (stream seeding, error types, plugin API surface) behaves, and the
package refuses to sample without the device.  No compute call is made."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "riemann_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(rmh_[a-z0-9_]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    from riemann_amd import _lib, build
    so = build.build()
    lib = ctypes.CDLL(so)
    names = declared_symbols()
    assert len(names) >= 35
    for n in names:
        assert hasattr(lib, n), n
    bound = set(_lib.SIGNATURES) | {"rmh_last_error"}
    assert bound == set(names), (bound ^ set(names))
    assert _lib.load() is _lib.load()


def test_no_gpu_means_loud_failure_not_fallback():
    import riemann_amd
    from riemann_amd import _lib
    from riemann_amd.models import MultiGaussianDist
    n = ctypes.c_int(0)
    rc = _lib.load().rmh_device_count(ctypes.byref(n))
    if rc == 0 and n.value > 0:
        pytest.skip("a GPU is visible here")
    with pytest.raises(RuntimeError):
        MultiGaussianDist(np.zeros(2), np.eye(2))
    # the abstract bases behave as the reference's (model.py:27-33, proposal.py:10-17)
    with pytest.raises(NotImplementedError):
        riemann_amd.Sampler(riemann_amd.Model(), riemann_amd.Proposal(), np.zeros(2))

    # a user's own Python plugins are the plugin API, not a CPU sampler: the chain's random
    # stream and accept uniforms still come from the device, so without one run() fails loudly
    class Flat(riemann_amd.Model):
        def log_prior(self, theta):
            return 0.0

        def log_likelihood(self, theta):
            return -0.5 * float(theta @ theta)

    class Walk(riemann_amd.Proposal):
        def propose(self, theta):
            return theta + 0.1, 0.0

    s = riemann_amd.Sampler(Flat(), Walk(), np.zeros(2))
    with pytest.raises(RuntimeError):
        s.run(1)


def test_product_does_not_import_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "riemann_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f
                assert "rmh_oracle" not in src and "librmh_oracle" not in src, f


def test_host_seeding_matches_oracle_stream():
    from oracle import oracle as orc
    from riemann_amd.rng import rng_seed_host
    for seed, N, c0 in ((0, 1, 0), (12345, 37, 7), (2 ** 63 + 5, 200, 2 ** 40)):
        np.testing.assert_array_equal(rng_seed_host(seed, N, c0), orc.rng_seed(seed, N, chain0=c0))


def test_error_types_and_api_surface():
    import riemann_amd as ra
    from riemann_amd.models import ChangepointParams, MultiGaussianDist
    e = ra.ParameterError("boom")
    assert str(e) == "ParameterError: boom" and isinstance(e, ra.RiemannBaseError)
    z = np.load(os.path.join(ROOT, "tests", "golden", "g7_errors.npz"))
    with pytest.raises(ra.ParameterError) as ei:     # raised on the host, before any launch
        MultiGaussianDist(np.zeros(2), np.ones((2, 3)))
    assert str(ei.value) == str(z["gauss_nonsquare"])
    with pytest.raises(ra.ParameterError) as ei:
        MultiGaussianDist(np.zeros(3), np.eye(2))
    assert str(ei.value) == str(z["gauss_mismatch"])
    for cls, names in ((ra.Model, ("pack", "unpack", "log_likelihood", "log_prior", "log_posterior",
                                   "logL", "logP", "__call__")),
                       (ra.Proposal, ("propose", "adapt")),
                       (ra.Sampler, ("run", "current_state", "_add_state", "sample"))):
        for n in names:
            assert callable(getattr(cls, n)), (cls, n)
    with pytest.raises(NotImplementedError):
        ra.Model().log_prior(np.zeros(2))
    with pytest.raises(NotImplementedError):
        ra.Proposal().propose(np.zeros(2))
    p = ChangepointParams([1.5, 2.0], [1.0, 2.0, 3.0], 0.1)
    v = p.pack()
    assert v.shape == (6,)
    q = ChangepointParams.unpack(v)
    np.testing.assert_array_equal(q.cpx, p.cpx)
    np.testing.assert_array_equal(q.cpv, p.cpv)
    with pytest.raises(ValueError):
        ChangepointParams([1.0], [1.0], 0.1)


def test_model_base_log_posterior_guard():
    """models/model.py:49-55: nan or +-inf in either term -> -inf, batched too."""
    import riemann_amd as ra

    class M(ra.Model):
        def log_prior(self, t):
            return np.asarray(t)[..., 0]

        def log_likelihood(self, t):
            return np.asarray(t)[..., 1]

    m = M()
    assert m.log_posterior(np.array([1.0, 2.0])) == 3.0
    for bad in (np.nan, np.inf, -np.inf):
        assert m.log_posterior(np.array([bad, 1.0])) == -np.inf
        assert m(np.array([1.0, bad])) == -np.inf
    out = m.log_posterior(np.array([[1.0, 2.0], [np.inf, 0.0], [0.0, np.nan]]))
    np.testing.assert_array_equal(out, [3.0, -np.inf, -np.inf])


def test_dpp_hazard_scanner_and_built_kernels():
    """The hand-written v_fmac_f64_dpp (tail rows of rmh_run.hpp, gradient tail of rmh_models.hpp) must
    not read a VGPR through DPP within two wait states of a VALU write, nor follow a VALU write of
    EXEC within five; the compiler does not check inside inline asm.  The build scans the
    disassembly of the objects it links (riemann_amd/build.py); here the scanner itself is
    checked and the built objects are scanned again (no GPU needed)."""
    import os
    from riemann_amd import _dpp_hazards as chk
    fake = """
	s_nop 0
	s_nop 0
	v_mov_b64_e32 v[10:11], v[2:3]
	v_fmac_f64_dpp v[4:5], v[10:11], v[6:7] row_newbcast:3 row_mask:0xf bank_mask:0xf
	ds_read_b64 v[20:21], v9
	s_nop 1
	v_fmac_f64_dpp v[4:5], v[20:21], v[6:7] row_newbcast:1 row_mask:0xf bank_mask:0xf
	v_add_f64 v[30:31], v[0:1], v[2:3]
	s_nop 1
	v_fmac_f64_dpp v[4:5], v[30:31], v[6:7] row_newbcast:1 row_mask:0xf bank_mask:0xf
	v_permlane16_swap_b32_e32 v40, v41
	v_fmac_f64_dpp v[4:5], v[40:41], v[6:7] row_newbcast:1 row_mask:0xf bank_mask:0xf
	v_cmpx_lt_f64_e32 v[0:1], v[2:3]
	s_nop 3
	v_fmac_f64_dpp v[4:5], v[50:51], v[6:7] row_newbcast:1 row_mask:0xf bank_mask:0xf
.LBB0_1:
	s_nop 0
	v_fmac_f64_dpp v[4:5], v[50:51], v[6:7] row_newbcast:1 row_mask:0xf bank_mask:0xf
"""
    total, bad = chk.scan(fake)
    assert total == 6 and len(bad) == 4, bad
    assert "v_mov_b64" in bad[0] and "v_permlane16_swap" in bad[1] and "EXEC" in bad[2] and "block entry" in bad[3]
    # branch targets of a disassembly count as block entries
    dis = """
0000000000001000 <k>:
	s_cbranch_scc1 3                                           // 000000001000: BF850003
	v_add_f64 v[30:31], v[0:1], v[2:3]                         // 000000001004: D2800000 00000000
	s_nop 1                                                    // 00000000100C: BF800001
	v_fmac_f64_dpp v[4:5], v[30:31], v[6:7] row_newbcast:1 row_mask:0xf bank_mask:0xf// 000000001010: 0808A4FA FF01511E
"""
    total, bad = chk.scan(dis)
    assert total == 1 and len(bad) == 1 and "block entry" in bad[0], bad
    total, bad = chk.main()
    assert total >= 600 and not bad, bad


def test_host_adaptation_rules_replay_the_reference_fixtures():
    """The ONE host statement of AdaptScaleProposal.adapt and of the Haario update
    (riemann_amd/proposals/adaptive.py: adapt_scale_step, adapt_cov_due, adapt_cov_update -- what the
    mixed-plugin Sampler calls) replayed on the runs recorded from the reference's classes: fed the
    fixture's own accept flags / states they end in the fixture's adaptation state (g8, g14)."""
    import glob
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # (the module imports the ctypes binding; loading the library needs no GPU)
    sys.path.insert(0, root)
    from riemann_amd.proposals.adaptive import adapt_cov_due, adapt_cov_update, adapt_scale_step
    gold = os.path.join(root, "tests", "golden")
    for f in sorted(glob.glob(os.path.join(gold, "g8_adapt_*.npz"))):
        z = np.load(f)
        acc = z["accept"]
        ada = np.zeros((acc.shape[1], 3))
        ada[:, 2] = 1.0
        for t in range(acc.shape[0]):
            adapt_scale_step(ada, acc[t] != 0, float(z["target"]))
        np.testing.assert_array_equal(ada[:, 0], z["ad_n"])
        np.testing.assert_array_equal(ada[:, 1], z["ad_acc"])
        np.testing.assert_allclose(ada[:, 2], z["ad_scale"], rtol=1e-12)
    for tag in ("smooth", "strict", "marg"):
        z = np.load(os.path.join(gold, "g14_adaptcov_%s.npz" % tag))
        th = z["thetas"]
        for c in range(th.shape[1]):                       # three independent one-chain runs
            d = th.shape[2]
            SX, SX2, C, L = np.zeros(d), np.zeros((d, d)), z["C0"].copy(), None
            for n in range(1, th.shape[0]):                # adapt() after every step, sampler.py:88
                x = th[n, c]
                SX += x
                SX2 += np.outer(x, x)
                if adapt_cov_due(n):
                    C, L = adapt_cov_update(float(n), SX, SX2, z["C0"], float(z["t_adapt"]), bool(z["smooth"]),
                                            bool(z["marginalize"]))
            np.testing.assert_allclose(C, z["C_final"][c], rtol=1e-7, atol=1e-13)
            np.testing.assert_allclose(L, z["L_final"][c], rtol=1e-7, atol=1e-13)
    assert [n for n in range(1, 30) if adapt_cov_due(n, True)] == [4, 9, 16, 25]
    assert adapt_cov_due(11) and adapt_cov_due(14) and not adapt_cov_due(2)   # the float test's extra hits
