"""Helpers shared by the GPU parity test modules (fixtures ``ra`` / ``orc`` live in conftest.py).

Bars (BASELINE.json north_star): accept masks bit-exact; log-densities and states within 1e-6
relative -- asserted at RTOL = 1e-9 unless a test says why not.  The random stream is asserted
BIT-exact."""
import os

import numpy as np

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

RTOL = 1e-9

MIX2D = [(0.3, [0.0, 0.0], [[1.0, 0.9], [0.9, 1.0]]),
         (0.7, [0.0, 0.0], [[1.0, -0.9], [-0.9, 1.0]])]


def load(name):
    return np.load(os.path.join(G, name + ".npz"))


def corr100():
    return 0.1 * np.eye(100) + 0.9 * np.ones((100, 100))


def check_run(z, s, sfx=""):
    np.testing.assert_array_equal(s._chain_accept, z["accept" + sfx])
    np.testing.assert_allclose(s._chain_thetas, z["thetas" + sfx], rtol=RTOL, atol=1e-12)
    ref, got = z["logpost" + sfx], s._chain_logpost
    assert np.array_equal(np.isinf(got), np.isinf(ref))
    fin = np.isfinite(ref)
    np.testing.assert_allclose(got[fin], ref[fin], rtol=RTOL)


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


def oracle_run(orc, model, prop, th0, seed, T, chain0=0):
    th = np.ascontiguousarray(th0).copy()
    st = orc.rng_seed(seed, th.shape[0], chain0=chain0)
    lp = model.log_posterior(th)
    return orc.run(model, prop, th, lp, st, T)


def _adapt_gpu_check(ra, z, model, prop, T):
    s = ra.Sampler(model, prop, z["theta0"], seed=int(z["seed"]))
    s.run(T)
    check_run(z, s)
    np.testing.assert_array_equal(prop.Nsamples, z["ad_n"])
    np.testing.assert_array_equal(prop.Naccepts, z["ad_acc"])
    np.testing.assert_allclose(prop.scale, z["ad_scale"], rtol=1e-12)
    np.testing.assert_allclose(prop.accept_rate, z["ad_rate"], rtol=1e-15)
    return s


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


def rj_model(z):
    from riemann_amd.models import ChangepointRegression1D
    return ChangepointRegression1D(z["x"], z["y"], float(z["xmin"]), float(z["xmax"]), float(z["lamb"]),
                                   int(z["KM"]), float(z["alpha"]), float(z["beta"]), variable_k=True)
