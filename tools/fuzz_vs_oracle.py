#!/usr/bin/env python3
"""Randomised differential test: random (model, proposal, d, N, T, Nburn, Nthin) configurations on
the device against the CPU oracle.  Test infrastructure (uses oracle/).   python tools/fuzz_vs_oracle.py [n] [seed]

Every configuration runs under a random kernel policy (riemann_amd._lib.kernel_policy: half of them the
library's own choice, the others a random subset of the A/B switches), so that the specialised kernels --
one-lane-per-chain changepoint kernels at small chain counts, the VALU tail rows against the padded tiles,
the in-kernel moments of the covariance adaptation against the scratch history -- all meet the oracle.
A third of the configurations that can (not d > 112, block-mixture or variable-k proposals) draw their normals
in the 64-bit-resolution mode, Sampler(..., normals="f64"), against the oracle in the same mode."""
import os
import sys
import traceback

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import riemann_amd as ra                                                     # noqa: E402
from oracle import oracle as orc                                              # noqa: E402
from riemann_amd.models import ChangepointRegression1D, LogisticRegression, MultiGaussianDist  # noqa: E402
from riemann_amd.models.benchmarks import MixtureDist                        # noqa: E402
from riemann_amd import proposals as P                                        # noqa: E402
from riemann_amd.samplers import PTSampler                                    # noqa: E402


def spd(g, d, lo=0.3):
    A = g.standard_normal((d, d))
    return A @ A.T / d + lo * np.eye(d)


def one_cprj(g):
    """trans-dimensional changepoint sampling: random capacity, data, start dimensions, move mix"""
    KM = int(g.integers(1, 15))
    N = int(g.choice([1, 15, 16, 17, 100, int(g.integers(1, 400))]))
    T = int(g.integers(0, 40))
    n = int(g.choice([3, 24, 100, 101]))
    x = np.sort(g.uniform(1.0, 3.0, size=n))
    y = g.uniform(1.0, 3.0, size=n)
    al, be = (1.0, 1.0) if g.uniform() < 0.5 else (2.5, 1.5)
    lamb = float(g.uniform(1.0, 8.0))
    m = ChangepointRegression1D(x, y, 1.0, 3.0, lamb, KM, al, be, variable_k=True)
    om = orc.Changepoint(x, y, 1, 1.0, 3.0, lamb, al, be)
    starts = []
    for c in range(N):
        k = int(g.integers(0, KM + 1))
        starts.append((np.sort(g.uniform(1.05, 2.95, size=k)), g.uniform(0.5, 3.0, size=k + 1), float(g.uniform(0.2, 1.0))))
    th0 = orc.cprj_pack(starts, KM)
    pc = sorted(float(v) for v in g.uniform(0, 1, size=3)) if g.uniform() < 0.5 else [0.2, 0.4, 0.6]
    hs = float(g.uniform(0.5, 3.0))
    p = P.ChangepointRegression1DProp(m, hs)
    p.P_cumprop_cpx, p.P_cumprop_cpv, p.P_cumprop_sig = pc
    seed = int(g.integers(0, 2 ** 31))
    desc = "cprj KM=%d N=%d T=%d n=%d" % (KM, N, T, n)
    s = ra.Sampler(m, p, th0, seed=seed)
    T1 = int(g.integers(0, T + 1))
    s.run(T1)
    first = (s._chain_thetas.copy(), s._chain_logpost.copy(), s._chain_accept.copy())
    s.run(T - T1)
    th = th0.copy()
    lp = orc.cprj_log_posterior(om, KM, th)
    np.testing.assert_allclose(m.log_posterior(th0), lp, rtol=1e-8, atol=1e-10, err_msg=desc + " eval")
    ht, hl, acc, mv = orc.run_cprj(om, KM, hs, pc, th, lp, orc.rng_seed(seed, N), T)
    code = np.concatenate([first[2], s._chain_accept])
    got = np.concatenate([first[0], s._chain_thetas[1:]])
    glp = np.concatenate([first[1], s._chain_logpost[1:]])
    assert np.array_equal(code >> 1, mv) and np.array_equal(code & 1, acc), desc + " moves/accepts"
    fin = np.isfinite(ht)
    assert np.array_equal(np.isfinite(got), fin), desc + " finite"
    np.testing.assert_allclose(got[fin], ht[fin], rtol=1e-8, atol=1e-11, err_msg=desc)
    fl = np.isfinite(hl)
    assert np.array_equal(np.isfinite(glp), fl), desc + " finite lp"
    np.testing.assert_allclose(glp[fl], hl[fl], rtol=1e-8, atol=1e-10, err_msg=desc)
    return desc + " cprj"


def one(g, it, want64=False):
    kind = g.choice(["gauss", "gauss", "logreg", "mix", "cp", "bigd", "cprj"])
    if kind == "cprj":
        return one_cprj(g)
    dmax = {"gauss": 112, "logreg": 112, "mix": 100, "cp": 18, "bigd": 1024}[kind]
    d = int(g.choice([1, 2, 3, 4, 5, 15, 16, 17, 20, 31, 32, 33, 36, 47, 48, 49, 52, 63, 64, 65, 68, 79, 80, 81, 84, 95, 96, 97, 100,
                      111, 112, int(g.integers(1, 113)), int(g.integers(33, 97))]))
    d = min(d, dmax)
    if kind == "bigd":
        d = int(g.choice([113, 127, 128, 129, 256, 257, 300, 512, 513, int(g.integers(113, 700))]))
    if kind == "cp":
        kc = int(g.integers(1, 9))
        d = 2 * kc + 2
    N = int(g.choice([1, 2, 15, 16, 17, 63, 64, 65, 100, int(g.integers(1, 300))]))
    T = int(g.integers(0, 14))
    nb = int(g.integers(0, T + 2))
    nt = int(g.integers(1, 4))
    th0 = 0.3 * g.standard_normal((N, d))
    if kind == "cp":
        n = int(g.choice([4, 25, 100, 101]))
        x = np.sort(g.uniform(1.0, 3.0, size=n))
        y = g.uniform(1.0, 3.0, size=n)
        al, be = (1.0, 1.0) if g.uniform() < 0.5 else (2.5, 1.5)
        m = ChangepointRegression1D(x, y, 1.0, 3.0, float(kc), 2 * kc, al, be, k=kc)
        om = orc.Changepoint(x, y, kc, 1.0, 3.0, float(kc), al, be)
        th0 = np.concatenate([np.sort(g.uniform(1.05, 2.95, size=(N, kc)), axis=1),
                              g.uniform(1.0, 3.0, size=(N, kc + 1)), g.uniform(0.3, 1.0, size=(N, 1))], axis=1)
        if N > 2:
            th0[1, 0] = 3.5          # a chain outside the support
    elif kind in ("gauss", "bigd"):
        mu, Ct = g.standard_normal(d), spd(g, d)
        m, om = MultiGaussianDist(mu, Ct), orc.Gaussian(mu, Ct)
        th0 = th0 + mu
    elif kind == "logreg":
        n = int(g.choice([1, 31, 32, 33, 100]))
        X, y = g.standard_normal((n, d)) / np.sqrt(d), (g.uniform(size=n) < 0.5).astype(float)
        m, om = LogisticRegression(X, y, 10.0), orc.LogReg(X, y, 10.0)
    else:
        comps = [(float(g.uniform(0.2, 1)), g.standard_normal(d), spd(g, d)) for _ in range(int(g.integers(1, 4)))]
        m = MixtureDist([(w, MultiGaussianDist(mu, C)) for w, mu, C in comps])
        om = orc.Mixture(comps)
    # the Model entry points on the starting states
    for fn in ("log_prior", "log_likelihood", "log_posterior"):
        a, b = np.atleast_1d(getattr(m, fn)(th0)), getattr(om, fn)(th0)
        assert np.array_equal(np.isfinite(a), np.isfinite(b)), "%s %s d=%d finite" % (kind, fn, d)
        fin = np.isfinite(b)
        np.testing.assert_allclose(a[fin], b[fin], rtol=1e-8, atol=1e-10, err_msg="%s %s d=%d" % (kind, fn, d))
    if kind in ("gauss", "logreg", "mix"):
        np.testing.assert_allclose(m.grad_log_posterior(th0), om.grad_log_posterior(th0), rtol=1e-7, atol=1e-9,
                                   err_msg="%s grad d=%d" % (kind, d))
    Cp = (0.5 / d) * spd(g, d)
    props = ["rw", "arw", "pcn", "apcn", "acov", "pt"]
    if kind in ("gauss", "logreg", "mix"):
        props += ["hmc", "mala", "hmcm", "ahmc", "acovhmc"]
    if kind != "bigd":
        props += ["asc"]
    if kind == "cp":
        Cp = 0.002 * np.eye(d)
        props += ["block"]
    if kind == "bigd":
        props = ["rw"]
    pk = str(g.choice(props))
    nm = "f64" if (want64 and kind != "bigd" and pk != "block") else "f32x2"
    orc.lib().orc_set_normal_mode(1 if nm == "f64" else 0)     # (main() resets it after the configuration)
    desc = "%s d=%d N=%d T=%d nb=%d nt=%d %s%s" % (kind, d, N, T, nb, nt, pk, " f64" if nm == "f64" else "")
    seed = int(g.integers(0, 2 ** 31))
    ad = aux = None
    th, st = th0.copy(), orc.rng_seed(seed, N)
    if pk == "pt":
        K = int(g.choice([1, 2, 3, 4, 5, 7, 8, 15, 16, int(g.integers(1, 17))]))
        betas = np.sort(g.uniform(0.0, 1.0, size=K))[::-1].copy()
        pswap = float(g.uniform(0.05, 0.9))
        chain0 = 16 * int(g.integers(0, 1000))
        # within-chain moves: random walk, pCN (any model), HMC / MALA with or without a mass matrix
        # (models with a gradient)
        hmc = kind in ("gauss", "logreg", "mix") and g.uniform() < 0.5
        Lh = int(g.integers(1, 4))
        if hmc:
            Mm = spd(g, d) if g.uniform() < 0.5 else None
            wprop, oprop = P.VanillaHMC(0.2, Lh, m, M=Mm), orc.HMC(0.2, Lh, d, M=Mm)
            desc += " mass" if Mm is not None else ""
        elif g.uniform() < 0.4:
            hmc = False
            wprop, oprop = P.pCN(Cp, 0.9), orc.PCN(Cp, 0.9)
            desc += " pcn"
        else:
            wprop, oprop = P.MetropolisRandomWalk(Cp), orc.RandomWalk(Cp)
        pt = PTSampler(m, wprop, th0, betas=betas, Pswap=pswap, seed=seed, chain0=chain0, normals=nm)
        T1 = int(g.integers(0, T + 1))
        pt.run(T1)                       # histories append across run() calls
        pt.run(T - T1)
        S, idx = orc.pt_slots(N, K)
        slots = np.zeros((S, d))
        slots[:] = th0[0]
        slots[idx] = th0[:, None, :]
        ht, hl, code, _ = orc.run_pt(om, oprop, slots, orc.rng_seed(seed, S, chain0=chain0), T, betas, pswap)
        desc += " K=%d T1=%d%s" % (K, T1, " hmc" if hmc else "")
        assert np.array_equal(pt._host_codes, code[:, idx]), desc + " codes"
        np.testing.assert_allclose(pt._host_thetas, ht[:, idx], rtol=1e-8, atol=1e-11, err_msg=desc)
        ref = hl[:, idx]
        fin = np.isfinite(ref)
        assert np.array_equal(np.isfinite(pt._host_logpost), fin), desc + " finite"
        np.testing.assert_allclose(pt._host_logpost[fin], ref[fin], rtol=1e-8, atol=1e-10, err_msg=desc)
        return desc
    if pk == "block":
        pc = [float(g.uniform(0, 1)), float(g.uniform(0, 1)), float(g.uniform(0, 1))]
        p = P.ChangepointRegression1DProp(m, 2.0)
        p.P_cumprop_cpx, p.P_cumprop_cpv, p.P_cumprop_sig = pc
        p.scale = 0.3
        op = orc.BlockRandomWalk(np.diag(np.diag(p.L) ** 2), [kc, kc + 1, 1], pc, scale=0.3)
    elif pk == "rw":
        p, op = P.MetropolisRandomWalk(Cp), orc.RandomWalk(Cp)
    elif pk == "arw":
        p, op, ad = P.AdaptScaleRandomWalk(Cp), orc.set_adapt(orc.RandomWalk(Cp), 0.25), orc.new_adapt_state(N)
    elif pk == "pcn":
        p, op = P.pCN(Cp, 0.9), orc.PCN(Cp, 0.9)
    elif pk == "apcn":
        p, op = P.AdaptScalepCN(Cp, 0.9), orc.set_adapt(orc.PCN(Cp, 0.9), 0.25)
        ad, aux = orc.new_adapt_state(N), np.full(N, np.nan)
    elif pk == "hmc":
        L = int(g.integers(1, 5))
        p, op = P.VanillaHMC(0.2, L, m), orc.HMC(0.2, L, d)
    elif pk == "mala":
        p, op = P.MALA(0.2, m), orc.HMC(0.2, 1, d)
    elif pk == "hmcm":
        Mm = spd(g, d)
        p, op = P.VanillaHMC(0.2, 3, m, M=Mm), orc.HMC(0.2, 3, d, M=Mm)
    elif pk == "ahmc":
        p, op, ad = P.AdaptScaleHMC(0.2, 2, m), orc.set_adapt(orc.HMC(0.2, 2, d), 0.75), orc.new_adapt_state(N)
    elif pk in ("acov", "asc", "acovhmc"):
        if d > 111:
            return desc + " (skipped)"
        sq = bool(g.integers(0, 2))
        if pk == "acov":
            p, op = P.AdaptCovRandomWalk(Cp, t_adapt=50, smooth_adapt=True, exact_squares=sq), orc.RandomWalk(Cp)
        elif pk == "asc":
            p = P.AdaptScaleCovRandomWalk(Cp, t_adapt=50, smooth_adapt=True, exact_squares=sq)
            op, ad = orc.set_adapt(orc.RandomWalk(Cp), 0.25), orc.new_adapt_state(N)
        else:
            Cp = spd(g, d)
            p, op = P.AdaptCovHMC(0.2, 2, m, Cp, t_adapt=50, smooth_adapt=True, exact_squares=sq), orc.HMC(0.2, 2, d, M=Cp)
        s = ra.Sampler(m, p, th0, seed=seed, normals=nm)
        s.run(T, nb, nt)
        stc = orc.new_adaptcov_state(Cp)
        ht, hl, acc = orc.run_adaptcov(om, op, th, om.log_posterior(th), st, T, stc, Cp,
                                       t_adapt=50, smooth_adapt=True, exact_squares=sq, adapt=ad)
        assert np.array_equal(s._chain_accept, acc), desc + " accept"
        np.testing.assert_allclose(s._chain_thetas, ht[nb::nt], rtol=1e-7, atol=1e-10, err_msg=desc)
        return desc
    s = ra.Sampler(m, p, th0, seed=seed, normals=nm)
    s.run(T, nb, nt)
    lp0 = om.log_posterior(th)
    ht, hl, acc = orc.run(om, op, th, lp0, st, T, adapt=ad, aux=aux)

    def check(tag, ht, hl, acc, nb, nt):
        assert np.array_equal(s._chain_accept, acc), desc + tag + " accept"
        np.testing.assert_allclose(s._chain_thetas, ht[nb::nt], rtol=1e-8, atol=1e-11, err_msg=desc + tag)
        ref = hl[nb::nt]
        assert np.array_equal(np.isfinite(s._chain_logpost), np.isfinite(ref)), desc + tag + " finite"
        fin = np.isfinite(ref)
        np.testing.assert_allclose(s._chain_logpost[fin], ref[fin], rtol=1e-8, atol=1e-10, err_msg=desc + tag)
    check("", ht, hl, acc, nb, nt)
    kept = (T - nb) // nt + 1 if T >= nb else 0
    if kept:    # a second run() continues from the LAST KEPT entry (sampler.py:49-50, 53-54)
        last = nb + (kept - 1) * nt
        th2, lp2 = np.ascontiguousarray(ht[last]).copy(), np.ascontiguousarray(hl[last]).copy()
        T2, nb2, nt2 = int(g.integers(0, 8)), int(g.integers(0, 4)), int(g.integers(1, 3))
        desc += " | T=%d nb=%d nt=%d" % (T2, nb2, nt2)
        s.run(T2, nb2, nt2)
        ht, hl, acc = orc.run(om, op, th2, lp2, st, T2, adapt=ad, aux=aux)
        check(" second run", ht, hl, acc, nb2, nt2)
    return desc


def random_policy(gp):
    """() half of the time, else a random subset of the kernel switches"""
    if gp.uniform() < 0.5:
        return ()
    names = ["NO_TAIL", "CP_NO_SS", "CP_NO_BUCKETS", "CPRJ_NO_SS", "ACOV_HOST", "ACOV_UNFUSED"]
    pol = [n for n in names if gp.uniform() < 0.3]
    r = gp.uniform()
    if r < 0.45:
        pol.append("CP_SS64")
    elif r < 0.6:
        pol.append("CP_SS16")
    return tuple(pol)


def main():
    from riemann_amd._lib import kernel_policy
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    g, gp = np.random.default_rng(seed), np.random.default_rng(seed + 977)
    bad = 0
    for it in range(n):
        pol = random_policy(gp)
        want64 = gp.uniform() < 0.34
        try:
            with kernel_policy(*pol):
                d = one(g, it, want64)
            print("ok  ", d, ("[" + ",".join(pol) + "]") if pol else "", flush=True)
        except Exception as e:           # keep going: collect every failing configuration
            bad += 1
            print("FAIL", "[" + ",".join(pol) + "]", repr(e)[:300], flush=True)
            traceback.print_exc(limit=2)
        finally:
            orc.lib().orc_set_normal_mode(0)
    print("%d configurations, %d failed" % (n, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
