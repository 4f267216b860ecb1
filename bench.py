#!/usr/bin/env python3
"""Benchmark of the many-chain MH hot path on MI355X (BASELINE.json metric:
MH chain-iterations/sec aggregate over N chains; ESS/sec).

    python bench.py --gpus 1 --steps K --warmup W [--workload logreg_mala]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

    cfg4 (BASELINE.json configs[3]):  ... --nproc-per-node 8 bench.py --gpus 8 --workload logreg_hmc
    cfg5 (configs[4]):                ... --nproc-per-node 4 bench.py --gpus 4 --workload changepoint_rw

A "step" is one fused ``rmh_run`` launch: ``iters`` MH iterations for every
chain of this rank, every state recorded into an HBM-resident history
(the reference records every state, riemann/samplers/sampler.py:69-70).
Inputs are synthetic (riemann_amd/workloads.py) and resident in HBM before the
timed region.  Chains shard across ranks by global chain id with no collective
in the timed region (weak scaling: 65 536 chains per GPU).  The launcher only
provides RANK / WORLD_SIZE / MASTER_*: nothing of PyTorch is imported -- the
barrier, the max-over-ranks of the timing and the pooling of the final states
(one all-gather, one all-reduce of moments) go through the library's own RCCL
communicator (rmh_comm_*, csrc/rmh_comm.hip), the 128-byte id over a socket
(riemann_amd.distributed.HostGroup).

Prints ONE JSON line (rank 0).  Default workload = north_star's target, BASELINE.json
configs[2] (cfg3: logistic regression 10 000 x 100, MALA, 65 536 chains, f64); at one GPU
the line carries an "also" list with short runs of cfg2, cfg4 (per-GPU share) and cfg5.
"""
import argparse
import ctypes as C
import gc
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP64_MFMA_PEAK = 78.6e12   # MI355X FP64 matrix = FP64 vector peak, AMD public spec
HBM_PEAK = 8.0e12          # /opt/skills/guides/MI355X_MICROARCH.md, chip-level parameters
WORKLOADS = ["logreg_mala", "gauss_rw", "logreg_hmc", "changepoint_rw", "pt_gauss"]


def workload(name, chains):
    """-> dict(model, proposal, theta0, flops, bytes, bound, iters, oracle builders, label)"""
    from riemann_amd import workloads as W
    from riemann_amd.models import ChangepointRegression1D, LogisticRegression, MultiGaussianDist
    from riemann_amd.proposals import MALA, MetropolisRandomWalk, VanillaHMC
    if name == "gauss_rw":      # BASELINE.json configs[1] / SURVEY.md 8d cfg2
        d = 100
        Ct = W.corr_gauss_cov(d)
        Cp = (2.38 ** 2 / d) * Ct
        th0 = np.random.default_rng(1).standard_normal((chains, d)) @ np.linalg.cholesky(Ct).T
        return dict(model=MultiGaussianDist(np.zeros(d), Ct), proposal=MetropolisRandomWalk(Cp),
                    theta0=th0, d=d, iters=1000, flops=2 * d * (d + 1) + 4 * d, bytes=8 * (d + 1),
                    bound="mfma", label="cfg2: 100-d correlated Gaussian, random-walk MH",
                    omodel=lambda o: o.Gaussian(np.zeros(d), Ct), oprop=lambda o: o.RandomWalk(Cp),
                    twin=lambda t: (t.TwinGaussian(np.zeros(d), Ct), t.TwinRandomWalk(Cp), 20000),
                    batched=lambda b: (b.BatchGaussian(np.zeros(d), Ct), dict(L_prop=np.linalg.cholesky(Cp)), 16384, 150),
                    cpu_sample=(65536, 200))
    if name in ("logreg_mala", "logreg_hmc"):   # configs[2] / configs[3]
        n, d = 10000, 100
        X, y = W.logreg_data(n, d, seed=2)
        m = LogisticRegression(X, y, 100.0)
        L, eps = (1, 0.175) if name == "logreg_mala" else (10, 0.10)
        th0 = np.zeros((chains, d))
        return dict(model=m, proposal=MALA(eps, m) if L == 1 else VanillaHMC(eps, L, m),
                    theta0=th0, d=d, iters=200 if L == 1 else 100, flops=4 * n * d * L,
                    bytes=2 * 8 * (2 * d + 1) + 8 * (d + 1), bound="mfma",
                    label=("cfg3: logistic regression 10000x100, MALA, 65536 chains/GPU" if L == 1 else
                           "cfg4: logistic regression 10000x100, HMC L=10, 65536 chains/GPU"),
                    omodel=lambda o: o.LogReg(X, y, 100.0), oprop=lambda o: o.HMC(eps, L, d),
                    twin=lambda t: ((lambda tm: (tm, t.TwinHMC(eps, L, tm.grad), 300 if L == 1 else 40))
                                    (t.TwinLogReg(X, y, 100.0))),
                    batched=lambda b: (b.BatchLogReg(X, y, 100.0), dict(eps=eps, nsteps=L),
                                       2048 if L == 1 else 1024, 6 if L == 1 else 2),
                    cpu_sample=(8192, 16) if L == 1 else (2048, 4))
    if name == "changepoint_rw":                # configs[4]
        x, y, truth, cdiag = W.changepoint_data()
        k, d = 5, 12
        g = np.random.default_rng(5)
        th0 = truth + 0.01 * g.standard_normal((chains, d))
        th0[:, :k] = np.sort(th0[:, :k], axis=1)
        p = MetropolisRandomWalk(np.diag(cdiag))
        p.scale = 0.1
        return dict(model=ChangepointRegression1D(x, y, 1.0, 3.0, 5.0, 10, 1, 1, k=k), proposal=p,
                    theta0=th0, d=d, iters=2000, flops=900, bytes=8 * (d + 1), bound="hbm",
                    label="cfg5: changepoint k=5 (n=100), random-walk MH",
                    omodel=lambda o: o.Changepoint(x, y, k, 1.0, 3.0, 5.0, 1.0, 1.0),
                    oprop=lambda o: o.RandomWalk(np.diag(cdiag), 0.1),
                    batched=lambda b: (b.BatchChangepoint(x, y, k, 1.0, 3.0, 5.0, 1.0, 1.0),
                                       dict(L_prop=0.1 * np.diag(np.sqrt(cdiag))), 16384, 60),
                    # the slowest changepoint position does not decorrelate at these lengths: tau_int 1.2e4 from 2e5 iterations,
                    # 7.4e4 from 2e6 (window not closed either time, reported as such): the longer, more conservative trace
                    ess_run=(2000000, 2000),
                    cpu_sample=(65536, 1000))
    if name == "pt_gauss":      # PTSampler, default ladder, on the cfg2 target
        d, K = 100, 5
        Ct = W.corr_gauss_cov(d)
        Cp = (2.38 ** 2 / d) * Ct
        ne = (chains // 16) * (16 // K)          # `chains` slots hold 16//K ladders per 16
        th0 = np.random.default_rng(1).standard_normal((ne, d)) @ np.linalg.cholesky(Ct).T
        return dict(model=MultiGaussianDist(np.zeros(d), Ct), proposal=MetropolisRandomWalk(Cp),
                    theta0=th0, d=d, iters=1000, flops=2 * d * (d + 1) + 4 * d, bytes=8 * (d + 1),
                    bound="mfma", pt=dict(betas=0.5 ** np.arange(K), pswap=0.1),
                    label="8f-2: parallel tempering, 5 temperatures, 100-d correlated Gaussian, random-walk MH",
                    omodel=lambda o: o.Gaussian(np.zeros(d), Ct), oprop=lambda o: o.RandomWalk(Cp),
                    cpu_sample=(240, 2000))
    raise SystemExit("unknown workload " + name)


def measured_profile(name):
    """Committed PMC measurements of the workload's run kernel (profiles/pmc_traffic.json:
    FETCH_SIZE and WRITE_SIZE collected in separate rocprofv3 --pmc runs of this same command,
    FETCH doubled as the gfx950 note of MI355X_MICROARCH.md prescribes), or {}."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            return json.load(f)[name]
    except Exception:
        return {}


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(w, seed):
    """The oracle (C port of the reference path, OpenMP over chains) on a bounded
    sample of the same workload, on this box's host cores."""
    from oracle import oracle as orc
    N, T = w["cpu_sample"]
    N = min(N, w["theta0"].shape[0])
    m, p = w["omodel"](orc), w["oprop"](orc)
    if "pt" in w:   # N ladders, serial restatement of PTSampler.run
        K = len(w["pt"]["betas"])
        S, idx = orc.pt_slots(N, K)
        th = np.zeros((S, w["d"]))
        th[idx] = w["theta0"][:N, None, :]
        st = orc.rng_seed(seed, S)
        t0 = time.perf_counter()
        orc.run_pt(m, p, th, st, T, w["pt"]["betas"], w["pt"]["pswap"], history=False)
        dt = time.perf_counter() - t0
        return {"value": N * K * T / dt, "unit": "chain-iterations/s", "cores": 1, "kind": "port",
                "cpu": cpu_model(),
                "sample": "%d ladders x %d temperatures x %d sweeps, oracle/rmh_oracle.c orc_run_pt "
                          "(one thread), %.1f s" % (N, K, T, dt)}
    th = np.ascontiguousarray(w["theta0"][:N]).copy()
    st = orc.rng_seed(seed, N)
    lp = m.log_posterior(th)
    # the one-GPU box share is 16 host cores (task environment); never oversubscribe
    threads = max(1, min(orc.max_threads(), len(os.sched_getaffinity(0)), 16))
    orc.run(m, p, th, lp, st, 1, history=False, nthreads=threads)   # warm
    t0 = time.perf_counter()
    orc.run(m, p, th, lp, st, T, history=False, nthreads=threads)
    dt = time.perf_counter() - t0
    out = {"value": N * T / dt, "unit": "chain-iterations/s", "cores": threads, "kind": "port",
           "cpu": cpu_model(),
           "sample": "%d chains x %d iterations of the same workload, oracle/rmh_oracle.c "
                     "(OpenMP over chains), %.1f s" % (N, T, dt)}
    bb = w.get("batched")
    if bb is not None:   # the CPU number a practitioner would recognise: chains as a matrix, level-3 BLAS
        from oracle import batched_twin
        bm, kw, Nb, Tb = bb(batched_twin)
        Nb = min(Nb, w["theta0"].shape[0])
        try:
            from threadpoolctl import threadpool_limits
            lim = threadpool_limits(limits=threads)
        except Exception:
            lim = None
        try:
            batched_twin.run(bm, w["theta0"][:Nb], 1, seed, **kw)            # warm
            dtb, accb = batched_twin.run(bm, w["theta0"][:Nb], Tb, seed, **kw)
        finally:
            if lim is not None:
                lim.restore_original_limits() if hasattr(lim, "restore_original_limits") else lim.unregister()
        out["batched_blas"] = {"value": Nb * Tb / dtb, "unit": "chain-iterations/s", "cores": threads, "kind": "port",
                               "accept_rate": accb,
                               "sample": "%d chains x %d iterations as one matrix: numpy + threaded BLAS "
                                         "(theta @ X^T / triangular solves with N right-hand sides), "
                                         "oracle/batched_twin.py, %.1f s" % (Nb, Tb, dtb)}
    tw = w.get("twin")
    if tw is not None:   # the reference-equivalent rate: numpy twin of the reference, 1 chain, 1 core
        from oracle import py_twin
        tm, tp, Tt = tw(py_twin)
        xi, u = orc.rng_stream(orc.rng_seed(seed, 1), w["d"], Tt)
        t0 = time.perf_counter()
        py_twin.run_chain(tm, tp, w["theta0"][0], xi[:, 0], u[:, 0])
        out["python_single_chain"] = {"value": Tt / (time.perf_counter() - t0), "cores": 1,
                                      "sample": "oracle/py_twin.py (numpy twin of the reference's "
                                                "Sampler/Proposal/Model), 1 chain x %d iterations" % Tt}
    return out


def tau_estimate(w, seed):
    """tau_int (the slowest parameter) and accept rate from an untimed diagnostic run: 256 chains, a
    thinned trace, reduced ON THE DEVICE (Sampler.history_stats: chain-averaged autocorrelation
    functions -> Sokal window; the history is not downloaded)."""
    import riemann_amd as ra
    T, thin = w.get("ess_run") or ((20000, 20) if w["flops"] < 1e6 else ((600, 1) if w["flops"] < 2e7 else (150, 1)))
    s = ra.Sampler(w["model"], w["proposal"], w["theta0"][:256], seed=seed + 1000)
    s.run(T, T // 5, thin)
    st = s.history_stats(nlags=128)
    acc = _lib_accept_rate(s)
    return float(np.max(st["tau"])) * thin, acc, bool(np.all(st["closed"])), s._kept


def _lib_accept_rate(s):
    from riemann_amd import _lib
    a = np.empty((s._iters, s.Nchains), dtype=np.uint8)
    _lib.call("rmh_history_accept", s._hist, 0, s._iters, a)
    return float((a & 1).mean())


class Ranks(object):
    """Barrier / max over the ranks: the library's RCCL communicator when it formed, the host
    socket group otherwise (e.g. several ranks rehearsing on ONE GPU, where RCCL refuses)."""

    def __init__(self):
        from riemann_amd import _lib
        from riemann_amd.distributed import HostGroup, comm_init
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.group = HostGroup.from_env() if self.world > 1 else HostGroup(0, 1)
        self.rccl, self.rccl_error = False, None
        if os.environ.get("RMH_BENCH_NO_RCCL") != "1":
            # RCCL prints a version banner on stdout when a communicator forms: keep stdout for
            # the ONE JSON line (the banner goes to stderr)
            sys.stdout.flush()
            keep = os.dup(1)
            os.dup2(2, 1)
            try:
                comm_init(self.group)
                _lib.call("rmh_comm_barrier")          # builds the communicator's channels untimed
                self.rccl = True
            except Exception as e:     # keep the measurement: the timed region has no collective
                self.rccl_error = repr(e)[:300]
            finally:
                os.dup2(keep, 1)
                os.close(keep)
        ok = self.group.allgather(self.rccl)
        if not all(ok):                # all or none: a half-formed communicator would hang
            if self.rccl:              # formed here but not everywhere: give it back
                try:
                    _lib.call("rmh_comm_destroy")
                except Exception:
                    pass
            self.rccl = False

    def barrier(self):
        from riemann_amd import _lib
        _lib.call("rmh_synchronize")
        if self.rccl:
            _lib.call("rmh_comm_barrier")
        else:
            self.group.barrier()

    def max(self, x):
        from riemann_amd import _lib
        if self.rccl:
            b = np.array([float(x)])
            _lib.call("rmh_comm_allreduce_host", b, 1, 1)
            return float(b[0])
        return self.group.max(x)


def measure(name, a, ranks, steps, warmup, iters=0, with_cpu=True, with_ess=True, with_pool=True):
    """Time `steps` fused launches of one workload on this rank's shard; rank 0 gets the result dict."""
    import riemann_amd as ra
    from riemann_amd import _lib
    rank, world = ranks.rank, ranks.world
    w = workload(name, a.chains)
    iters = iters or w["iters"]
    N, d = w["theta0"].shape

    # chains of this rank: global ids [rank*N, (rank+1)*N) -- RNG keyed by global id
    pt = w.get("pt")
    if pt:   # N ladders of K replicas in S chain slots; streams keyed by global slot id
        from riemann_amd.samplers import PTSampler
        s = PTSampler(w["model"], w["proposal"], w["theta0"], betas=pt["betas"], Pswap=pt["pswap"],
                      seed=a.seed, chain0=rank * a.chains, normals=a.normals)
        K = s.Ntemps
        bet = _lib.as_f64(s.betas)
    else:
        s = ra.Sampler(w["model"], w["proposal"], w["theta0"], seed=a.seed, chain0=rank * N, normals=a.normals)
    hist = C.c_void_p()
    if not a.no_record:
        _lib.call("rmh_history_create", s._state, iters + 1, iters, C.byref(hist))
    w["proposal"]._sync()

    def step():
        if pt:
            _lib.call("rmh_run_pt", w["model"]._h, w["proposal"]._h, s._state, bet, K, pt["pswap"],
                      iters, 0, 1, hist if hist else None)
        else:
            _lib.call("rmh_run", w["model"]._h, w["proposal"]._h, s._state, iters, 0, 1,
                      hist if hist else None)

    for _ in range(warmup):
        step()
    ranks.barrier()
    kernel_ms, step_end = [], []
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
        # HIP events on the library's stream around each launch; reading them
        # waits for the launch, which the next launch on the same stream would do anyway
        ms = C.c_double()
        _lib.call("rmh_last_run_ms", C.byref(ms), None)
        kernel_ms.append(ms.value)
        step_end.append(time.perf_counter())
    ranks.barrier()
    dt = ranks.max(time.perf_counter() - t0)
    # the dominant kernel's launch time: mean over this rank's launches; over the ranks the slowest one
    # prices the roofline (and the fastest is printed beside it)
    kms = float(np.mean(kernel_ms))
    kms_lo, kms_hi = (-ranks.max(-kms), ranks.max(kms)) if world > 1 else (kms, kms)

    # pool the final states of all ranks over RCCL (diagnostics only, untimed): the library's own
    # communicator, device-resident state blocks -> one ncclAllGather + one ncclAllReduce
    pooled = None
    if with_pool and not pt:
        if ranks.rccl:
            try:
                from riemann_amd.distributed import pool
                allth, mean, cov, info = pool(s)
                pooled = dict(info, pooled_mean_abs_max=float(np.abs(mean).max()),
                              pooled_var_mean=float(np.diag(cov).mean()))
            except Exception as e:  # never lose the measurement to the diagnostics
                pooled = {"rccl_ranks": 0, "allgather_error": repr(e)[:300]}
        else:
            pooled = {"rccl_ranks": 0, "rccl_error": ranks.rccl_error}
    if hist:
        _lib.call("rmh_history_free", hist)
    if rank != 0:
        del s
        gc.collect()
        return None
    if pt:
        N = N * K                # replicas: every one makes a within-chain or a swap move per sweep
    total_iters = world * N * iters * steps
    value = total_iters / dt
    kms = kms_hi
    flops = w["flops"] * N * iters
    bytes_ = (w["bytes"] * (0 if a.no_record else 1)) * N * iters + 16 * (d + 1) * N
    if w["bound"] == "mfma":
        ach, peak, unit = flops / (kms * 1e-3) / 1e12, FP64_MFMA_PEAK / 1e12, "TFLOP/s"
    else:
        ach, peak, unit = bytes_ / (kms * 1e-3) / 1e9, HBM_PEAK / 1e9, "GB/s"
    prof = measured_profile(name)
    tr = prof.get("hbm_bytes_per_chain_iteration")
    out = {
        "value": value, "unit": "chain-iterations/s", "steps": steps, "warmup": warmup,
        "ms_per_step": dt / steps * 1e3,
        "config": {"workload": w["label"], "chains_per_gpu": N, "dim": d,
                   "iterations_per_step": iters, "recorded": (not a.no_record),
                   "parallelism": "chains sharded by global id, %d rank(s)" % world},
        "roofline": {"bound": w["bound"], "achieved": ach, "peak": peak, "unit": unit,
                     "frac": ach / peak, "traffic": (tr * N * iters if tr else None),
                     "traffic_source": ("profiles/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of "
                                        "this kernel, bytes per chain-iteration) x chains x iterations of this "
                                        "launch; not measured in this run") if tr else None,
                     "kernel_ms": kms,
                     "kernel_ms_over_ranks": [kms_lo, kms_hi],
                     "step_wall_ms": [round((b - a_) * 1e3, 3) for a_, b in zip([t0] + step_end, step_end)],
                     "algorithmic_bytes": bytes_,
                     "hbm_algorithmic_GBps": bytes_ / (kms * 1e-3) / 1e9,
                     "flops_per_chain_iteration": w["flops"],
                     "bytes_per_chain_iteration": w["bytes"]},
    }
    if name == "changepoint_rw" and prof.get("valu_insts_per_chain_iteration"):
        # the kernel's real bound: vector-instruction issue, one per 4 cycles per SIMD on 1024 SIMDs at
        # 2.4 GHz, at the committed instruction count per chain-iteration (profiles/r02_sq_counters.json)
        vi = prof["valu_insts_per_chain_iteration"]
        roof = 1024 * 2.4e9 / 4.0 / vi
        out["roofline"]["valu_issue"] = {"insts_per_chain_iteration": vi, "roof_chain_iterations_per_s": roof,
                                         "frac": (N * iters / (kms * 1e-3)) / roof}
    if pooled:
        out["pool"] = pooled
    if with_cpu and world == 1:   # the CPU leg is timed at N=1 only
        out["cpu_baseline"] = cpu_baseline(w, a.seed)
    if a.normals != "f32x2":
        out["config"]["normals"] = a.normals
    if pt:
        out["config"]["ladders_per_gpu"] = N // K
        out["config"]["temperatures"] = K
        out["config"]["chain_slots_per_gpu"] = a.chains
    if with_ess and world == 1 and not pt:
        try:
            tau, acc, closed, nkept = tau_estimate(w, a.seed)
            out["ess"] = {"tau_int_max": tau, "accept_rate": acc, "ess_per_sec": value / max(tau, 1.0),
                          "window_closed": closed, "kept_states": nkept,
                          "estimator": "chain-averaged autocorrelation functions reduced on the device "
                                       "(rmh_history_reduce), Sokal window c=5, 256 chains "
                                       "(riemann_amd/diagnostics.py); parity unpinned: the reference's "
                                       "emcee.autocorr.integrated_time is not installed here"}
        except Exception as e:
            out["ess"] = {"error": repr(e)[:200]}
    del s, w
    gc.collect()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default=None, choices=WORKLOADS,
                    help="default: logreg_mala (cfg3) with cfg2/cfg4/cfg5 alongside in \"also\"")
    ap.add_argument("--chains", type=int, default=65536, help="chains per GPU")
    ap.add_argument("--iters", type=int, default=0, help="MH iterations per step (0 = workload default)")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ess", action="store_true")
    ap.add_argument("--no-also", action="store_true", help="skip the short runs of the other configs")
    ap.add_argument("--no-record", action="store_true", help="do not record the history (diagnostic)")
    ap.add_argument("--normals", default="f32x2", choices=["f32x2", "f64"],
                    help="resolution of the stream's normals (Sampler(..., normals=...)); f64: one normal per 64-bit word "
                         "(not for changepoint_rw's specialised kernel: that workload then runs the generic one)")
    a = ap.parse_args()

    from riemann_amd import _lib
    local = int(os.environ.get("LOCAL_RANK", "0"))
    ndev = C.c_int()
    _lib.call("rmh_device_count", C.byref(ndev))
    _lib.call("rmh_init", local % max(ndev.value, 1))   # ranks that see one device each land on it
    ranks = Ranks()
    main_wl = a.workload or "logreg_mala"
    out = measure(main_wl, a, ranks, a.steps, a.warmup, iters=a.iters, with_cpu=not a.no_cpu_baseline,
                  with_ess=not a.no_ess)
    also = []
    if a.workload is None and not a.no_also and ranks.world == 1:
        # short runs of the other BASELINE configs in the same process: cfg4 and cfg5 are
        # multi-GPU configurations, measured here as their per-GPU share (65 536 chains)
        # (two warm-up launches: the first timed cfg2 step of round 2's line was 10 % slow after one)
        for name, st, wu, it in (("gauss_rw", 3, 2, 0), ("logreg_hmc", 2, 2, 20), ("changepoint_rw", 3, 2, 0)):
            try:
                r = measure(name, a, ranks, st, wu, iters=it, with_cpu=not a.no_cpu_baseline,
                            with_ess=not a.no_ess, with_pool=False)
                r["workload"] = name
                also.append(r)
            except Exception as e:
                also.append({"workload": name, "error": repr(e)[:300]})
    if ranks.rank == 0:
        line = {"metric": "MH chain-iterations/sec", "value": out["value"], "unit": out["unit"],
                "n_gpus": ranks.world, "steps": out["steps"], "warmup": out["warmup"],
                "ms_per_step": out["ms_per_step"], "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": "f64", "data": "synthetic"}
        for k in ("config", "roofline", "pool", "cpu_baseline", "ess"):
            if k in out:
                line[k] = out[k]
        if also:
            line["also"] = also
        print(json.dumps(line))
    sys.stdout.flush()
    ranks.group.barrier()
    if ranks.rccl:
        _lib.call("rmh_comm_destroy")
    ranks.group.close()


if __name__ == "__main__":
    main()
