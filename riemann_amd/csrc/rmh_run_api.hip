// rmh_run_api.hip -- Sampler.run (rmh_run), PTSampler.run (rmh_run_pt) and the pooled
// covariance adaptation (rmh_proposal_set_adapt_cov ...): validation, dispatch to the fused
// kernels, the launch schedule between covariance updates.  See rmh_api.hip for the rest of
// the C ABI and rmh_host.hpp for the shared handle types.
#include "rmh_host.hpp"


using namespace rmh;
using namespace rmh::host;

namespace rmh {
int comm_active();                                  // rmh_comm.hip
int comm_allreduce_sum(double *dev, size_t count);
} // namespace rmh

// ---- AdaptCovProposal, pooled over the chains (proposals/adaptive.py:38-103) -------------
constexpr int kAcovDeviceMaxDim = 111; // covariance updates on the device (the moment kernel's limit)
constexpr int kAcovChunk = 64;   // iterations per launch at most (scratch history size)
constexpr int kAcovWGs = 512;    // workgroups of the moment kernel: two per CU (59 KB of LDS each)

static int acov_ntri(const rmh_proposal *p) { return p->acov_NT * (p->acov_NT + 1) / 2; }

// is an update due after the n-th adapt() call?  adaptive.py:81-82: `np.sqrt(n)**2 == n and
// n > 2` -- meant as "n is a perfect square", but the float test also passes for about half
// of the other n (11, 14, 17, ...); the literal test is the reference's behaviour,
// acov_squares selects the documented intent.
static bool acov_due(const rmh_proposal *p, double n)
{
    if (!(n > 2)) return false;
    const double r = std::sqrt(n);
    if (p->acov_squares) {
        const double q = std::floor(r + 0.5);
        return q * q == n;
    }
    volatile double rr = r * r;
    return rr == n;
}

// (d+1) x (d+1) moment matrix [[SX2, SX], [SX^T, count]] <- tile-pair layout of the device sums
static void acov_unpack(const rmh_proposal *p, const std::vector<double> &acc, std::vector<double> &Mm)
{
    const int D1 = p->d + 1, NT = p->acov_NT;
    Mm.assign((size_t)D1 * D1, 0.0);
    for (int i = 0, t = 0; i < NT; ++i)
        for (int j = 0; j <= i; ++j, ++t)
            for (int r = 0; r < 4; ++r)
                for (int l = 0; l < 64; ++l) {
                    const int row = 16 * i + (l >> 4) + 4 * r, col = 16 * j + (l & 15);
                    if (row < D1 && col < D1) {
                        const double v = acc[(size_t)t * 256 + r * 64 + l];
                        Mm[(size_t)row * D1 + col] = v;
                        Mm[(size_t)col * D1 + row] = v;
                    }
                }
}

static void acov_pack(const rmh_proposal *p, const std::vector<double> &Mm, std::vector<double> &acc)
{
    const int D1 = p->d + 1, NT = p->acov_NT;
    acc.assign((size_t)acov_ntri(p) * 256, 0.0);
    for (int i = 0, t = 0; i < NT; ++i)
        for (int j = 0; j <= i; ++j, ++t)
            for (int r = 0; r < 4; ++r)
                for (int l = 0; l < 64; ++l) {
                    const int row = 16 * i + (l >> 4) + 4 * r, col = 16 * j + (l & 15);
                    if (row < D1 && col < D1) acc[(size_t)t * 256 + r * 64 + l] = Mm[(size_t)row * D1 + col];
                }
}

namespace rmh {
namespace host {
bool cholesky_lower(const std::vector<double> &Cm, int d, std::vector<double> &L)
{
    L.assign((size_t)d * d, 0.0);
    for (int j = 0; j < d; ++j) {
        double sj = Cm[(size_t)j * d + j];
        for (int k = 0; k < j; ++k) sj -= L[(size_t)j * d + k] * L[(size_t)j * d + k];
        if (!(sj > 0.0)) return false;
        const double ljj = std::sqrt(sj);
        L[(size_t)j * d + j] = ljj;
        for (int i = j + 1; i < d; ++i) {
            double t = Cm[(size_t)i * d + j];
            for (int k = 0; k < j; ++k) t -= L[(size_t)i * d + k] * L[(size_t)j * d + k];
            L[(size_t)i * d + j] = t / ljj;
        }
    }
    return true;
}
} // namespace host
} // namespace rmh

// the covariance update of adaptive.py:83-102 from the pooled moments (local sums are
// all-reduced over the ranks first when a communicator exists)
// refresh the host copies of C and L after device-side updates
static int acov_sync_host(rmh_proposal *p)
{
    if (!p->acov_host_stale) return RMH_OK;
    const size_t n = (size_t)p->d * p->d;
    int st = 0;
    p->acov_C.resize(n);
    p->L.resize(n);
    RMH_HIP(hipMemcpyAsync(p->acov_C.data(), p->d_acov_C, n * sizeof(double), hipMemcpyDeviceToHost, g_stream));
    RMH_HIP(hipMemcpyAsync(p->L.data(), p->d_acov_L, n * sizeof(double), hipMemcpyDeviceToHost, g_stream));
    RMH_HIP(hipMemcpyAsync(&st, p->d_acov_status, sizeof(int), hipMemcpyDeviceToHost, g_stream));
    RMH_HIP(hipStreamSynchronize(g_stream));
    p->acov_host_stale = false;
    if (st) // numpy raises LinAlgError at adaptive.py:102
        return fail(RMH_ERR_HIP, "AdaptCov: the adapted covariance is not positive definite");
    return RMH_OK;
}

// random-walk variant: the whole update stays on the device (no host round trip)
static int acov_update_device(rmh_proposal *p)
{
    const int d = p->d, n = acov_ntri(p) * 256;
    const double *acc = p->d_acov_acc;
    if (rmh::comm_active()) { // pooled over the ranks: all-reduce a copy of this rank's sums
        RMH_HIP(hipMemcpyAsync(p->d_acov_tmp, p->d_acov_acc, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, g_stream));
        int rc = rmh::comm_allreduce_sum(p->d_acov_tmp, (size_t)n);
        if (rc) return rc;
        acc = p->d_acov_tmp;
    }
    AcovUpdateArgs a{};
    a.acc = acc;
    a.C0 = p->d_acov_C0;
    a.Cout = p->d_acov_C;
    a.Lout = p->d_acov_L;
    a.Limg = p->d_Ltiles;
    a.Ltail = has_tail(d) ? p->d_Ltail : nullptr;
    a.status = p->d_acov_status;
    a.d = d;
    a.NT = p->NT;
    a.t_adapt = p->acov_t_adapt;
    a.f4 = std::pow((double)d, 0.4);
    a.f2 = std::pow((double)d, 0.2);
    a.img_scale = p->adapt_target > 0 ? 1.0 : p->scale;   // as sync_proposal packs it
    double tr = 0.0;
    for (int i = 0; i < d; ++i) tr += p->acov_C0[(size_t)i * d + i];
    a.creg = 1e-12 * (tr / d);
    a.marginalize = p->acov_marg;
    a.smooth = p->acov_smooth;
    RMH_HIP(launch_acov_update(a, g_stream));
    p->acov_host_stale = true;
    return RMH_OK;
}

static int acov_update(rmh_proposal *p)
{
    // random walk: the one-workgroup device update (blocked Cholesky in LDS, k_acov_update): no
    // host round trip between the stretches of an adaptive run (tools/adaptcov_timing.py)
    if (p->kind == kRW && p->d <= kAcovDeviceMaxDim && !policy(RMH_POLICY_ACOV_HOST)) return acov_update_device(p);
    const int d = p->d, D1 = d + 1, n = acov_ntri(p) * 256;
    std::vector<double> acc((size_t)n), Mm;
    if (rmh::comm_active()) {
        double *tmp = nullptr;
        int rc = dev_alloc(&tmp, (size_t)n);
        if (rc) return rc;
        hipError_t e = hipMemcpyAsync(tmp, p->d_acov_acc, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, g_stream);
        if (e == hipSuccess) rc = rmh::comm_allreduce_sum(tmp, (size_t)n);
        if (e == hipSuccess && !rc) e = hipMemcpyAsync(acc.data(), tmp, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, g_stream);
        if (e == hipSuccess && !rc) e = hipStreamSynchronize(g_stream);
        (void)hipFree(tmp);
        if (rc) return rc;
        RMH_HIP(e);
    } else {
        RMH_HIP(hipMemcpyAsync(acc.data(), p->d_acov_acc, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, g_stream));
        RMH_HIP(hipStreamSynchronize(g_stream));
    }
    acov_unpack(p, acc, Mm);
    const double ntot = Mm[(size_t)d * D1 + d];   // pooled sample count: N (all ranks) x n
    double tr = 0.0;
    for (int i = 0; i < d; ++i) tr += p->acov_C0[(size_t)i * d + i];
    const double creg = 1e-12 * (tr / d), f4 = std::pow((double)d, 0.4), f2 = std::pow((double)d, 0.2);
    std::vector<double> Cn((size_t)d * d), L;
    for (int i = 0; i < d; ++i)
        for (int j = 0; j < d; ++j) {
            const double cs = (Mm[(size_t)i * D1 + j] - Mm[(size_t)i * D1 + d] * Mm[(size_t)j * D1 + d] / ntot) / (ntot - 1);
            double cv;
            if (p->acov_smooth) cv = (ntot * cs + p->acov_t_adapt * p->acov_C0[(size_t)i * d + j]) / (ntot + p->acov_t_adapt);
            else if (ntot < p->acov_t_adapt) cv = p->acov_C0[(size_t)i * d + j];
            else cv = cs + (i == j ? creg : 0.0);
            if (p->acov_marg && i != j) cv = 0.0;
            Cn[(size_t)i * d + j] = cv / f4;
        }
    if (!cholesky_lower(Cn, d, L))   // numpy raises LinAlgError at adaptive.py:102
        return fail(RMH_ERR_HIP, "AdaptCov: the adapted covariance is not positive definite");
    for (double &v : L) v /= f2;
    p->acov_C = Cn;
    p->L = L;
    p->dirty = true;
    if (p->kind == kHMC) { // M = C = d^0.4 L L^T from now on, chM = L
        p->minv = 1.0 / f4;
        return install_mass(p, L.data());
    }
    return sync_proposal(p);
}

// (the VALU-tail kernels serve the 64-bit normal mode for the plain and adapted-scale walks: not within tempering,
// not with the moments fused in)
static bool tail_ok(const ChainArgs &) { return !policy(RMH_POLICY_NO_TAIL); }

// adapt: the AdaptScale variant (per-chain scale in c.adapt, unscaled tile image)
static hipError_t launch_rw_plain(rmh_model *m, rmh_proposal *p, const ChainArgs &c, const RunArgs &r, bool adapt)
{
    RwArgs ra{p->d_Ltiles, nullptr, 0.0};
    switch (m->kind) {
    case kGauss:
        if (m->ga.Wtail && p->d_Ltail && tail_ok(c)) { // d = 97..100: rmh_run.hpp, k_run_rw_tail
            ra.Ltail = p->d_Ltail;
            return launch_gauss_rw_tail(m->ga, ra, c, r, adapt, g_stream);
        }
        return launch_gauss_rw(m->NT, m->ga, ra, c, r, adapt, g_stream);
    case kMixture: return launch_mix_rw(m->NT, m->ma, ra, c, r, adapt, g_stream);
    case kLogReg: return launch_logreg_rw(m->NT, m->la, ra, c, r, adapt, g_stream);
    case kChangepoint: return launch_cp_generic(m->k, m->ca, ra, c, r, adapt ? 1 : 0, g_stream);
    default: return hipErrorInvalidValue;
    }
}

// first state number >= lo that [Nburn::Nthin] keeps (sampler.py:53-54), and its history entry
static void first_kept(int64_t lo, int64_t nburn, int64_t nthin, int64_t *e, int64_t *k)
{
    const int64_t q = lo <= nburn ? 0 : (lo - nburn + nthin - 1) / nthin;
    *e = nburn + q * nthin;
    *k = q;
}

// Sampler.run with an AdaptCov proposal: one fused launch per stretch between covariance
// updates.  Round 3, d = 97..100 random walk on the Gaussian model (BASELINE cfg 2's shape):
// the sampling kernel accumulates the pooled moments of adapt() itself (MomFused, rmh_run.hpp)
// and records straight into the caller's history -- no scratch history, no copies, two launches
// per stretch (run, reduce) plus the update.  Everything else: every state of a stretch goes to
// a scratch history that k_moments reads, and the entries the caller keeps are copied on.
static int run_adaptcov(rmh_model *m, rmh_proposal *p, rmh_state *s, const ChainArgs &c, int64_t T,
                        int64_t nburn, int64_t nthin, rmh_history *h, int *launches)
{
    int rc;
    const int64_t rows = 4 * s->ks, ld = s->ld;
    const bool fused = p->kind == kRW && m->kind == kGauss && m->ga.Wtail && p->d_Ltail &&
                       p->acov_NT == 7 && tail_ok(c) && !c.nmode && !policy(RMH_POLICY_ACOV_UNFUSED);
    int mom_passes = 1;
    const int64_t nwg = fused ? gauss_rw_tail_mom_grid(c.nblocks, &mom_passes) : kAcovWGs;
    const size_t need = (size_t)nwg * acov_ntri(p) * 256;
    if (p->acov_partial_cap < need) {
        (void)hipFree(p->d_acov_partial);
        p->d_acov_partial = nullptr;
        p->acov_partial_cap = 0;
        if ((rc = dev_alloc(&p->d_acov_partial, need))) return rc;
        p->acov_partial_cap = need;
    }
    if (!fused && !s->sc_theta) {
        if ((rc = dev_alloc(&s->sc_theta, (size_t)(kAcovChunk + 1) * rows * ld)) ||
            (rc = dev_alloc(&s->sc_logpost, (size_t)(kAcovChunk + 1) * ld)) ||
            (rc = dev_alloc(&s->sc_accept, (size_t)kAcovChunk * ld)))
            return rc;
        RMH_HIP(hipMemsetAsync(s->sc_theta, 0, (size_t)(kAcovChunk + 1) * rows * ld * sizeof(double), g_stream));
    }
    if (T == 0 && h && nburn == 0) { // run(0): the history is the state on entry (sampler.py:49-50)
        RMH_HIP(hipMemcpyAsync(h->thetas, s->theta, (size_t)rows * ld * sizeof(double), hipMemcpyDeviceToDevice, g_stream));
        RMH_HIP(hipMemcpyAsync(h->logpost, s->logpost, (size_t)ld * sizeof(double), hipMemcpyDeviceToDevice, g_stream));
    }
    int64_t done = 0;
    while (done < T) {
        // the proposal is fixed until the next update falls due (and a scratch history holds
        // kAcovChunk iterations; the fused path has no such limit but keeps launches short
        // enough for the partial sums to stay a few thousand terms deep)
        int64_t seg = std::min<int64_t>(T - done, fused ? 1024 : kAcovChunk);
        for (int64_t i = 1; i <= seg; ++i)
            if (acov_due(p, p->acov_n + (double)i)) {
                seg = i;
                break;
            }
        RunArgs r{};
        r.T = seg;
        if (fused) {
            // states done .. done+seg of the run; number `done` was recorded by the previous stretch
            int64_t e0, k0;
            first_kept(done == 0 ? 0 : done + 1, nburn, nthin, &e0, &k0);
            r.nburn = e0 - done;
            r.nthin = nthin;
            if (h) {
                r.h_theta = h->thetas + k0 * rows * ld;
                r.h_logpost = h->logpost + k0 * ld;
                r.h_accept = h->accept ? h->accept + done * ld : nullptr;
            }
            r.mom_partial = p->d_acov_partial;
            r.mom_N = s->N;
            r.mom_passes = mom_passes;
            RwArgs ra{p->d_Ltiles, nullptr, 0.0};
            ra.Ltail = p->d_Ltail;
            RMH_HIP(launch_gauss_rw_tail(m->ga, ra, c, r, c.adapt != nullptr, g_stream));   // (AdaptScaleCov: the chains' own scales)
            ++*launches;
            // adapt() after every step: the states 1..seg of the stretch, summed over the workgroups
            RMH_HIP(launch_moments_reduce(p->d_acov_partial, nwg, acov_ntri(p) * 256, p->d_acov_acc, g_stream));
        } else {
            r.nburn = 0;
            r.nthin = 1;
            r.h_theta = s->sc_theta;
            r.h_logpost = s->sc_logpost;
            r.h_accept = s->sc_accept;
            if (p->kind == kHMC) {
                HmcArgs ha{p->eps, p->nsteps, p->d_chM, p->d_V, p->d_VT, p->minv};
                RMH_HIP(m->kind == kGauss     ? launch_gauss_hmc(m->NT, m->ga, ha, c, r, false, g_stream)
                        : m->kind == kMixture ? launch_mix_hmc(m->NT, m->ma, ha, c, r, false, g_stream)
                                              : launch_logreg_hmc(m->NT, m->la, ha, c, r, false, g_stream));
            } else {
                RMH_HIP(launch_rw_plain(m, p, c, r, c.adapt != nullptr));
            }
            ++*launches;
            if (h) {
                for (int64_t i = (done == 0 ? 0 : 1); i <= seg; ++i) {
                    const int64_t e = done + i;
                    if (e < nburn || (e - nburn) % nthin) continue;
                    const int64_t k = (e - nburn) / nthin;
                    RMH_HIP(hipMemcpyAsync(h->thetas + k * rows * ld, s->sc_theta + i * rows * ld,
                                           (size_t)rows * ld * sizeof(double), hipMemcpyDeviceToDevice, g_stream));
                    RMH_HIP(hipMemcpyAsync(h->logpost + k * ld, s->sc_logpost + i * ld, (size_t)ld * sizeof(double),
                                           hipMemcpyDeviceToDevice, g_stream));
                }
                if (h->accept)
                    RMH_HIP(hipMemcpyAsync(h->accept + done * ld, s->sc_accept, (size_t)seg * ld, hipMemcpyDeviceToDevice, g_stream));
            }
            // adapt() after every step: accumulate the states 1..seg of the segment
            RMH_HIP(launch_moments(p->acov_NT, s->sc_theta, 1, seg + 1, (int)rows, ld, s->N, s->d,
                                   p->d_acov_partial, kAcovWGs, p->d_acov_acc, g_stream));
        }
        p->acov_n += (double)seg;
        done += seg;
        if (acov_due(p, p->acov_n) && (rc = acov_update(p))) return rc;
    }
    if (p->acov_host_stale) { // one look at the not-positive-definite flag per run
        int st = 0;
        RMH_HIP(hipMemcpyAsync(&st, p->d_acov_status, sizeof(int), hipMemcpyDeviceToHost, g_stream));
        RMH_HIP(hipStreamSynchronize(g_stream));
        if (st) return fail(RMH_ERR_HIP, "AdaptCov: the adapted covariance is not positive definite");
    }
    return RMH_OK;
}

static int run_impl(rmh_model_t m, rmh_proposal_t p, rmh_state_t s, int64_t T, int64_t nburn,
                    int64_t nthin, rmh_history_t h, const double *betas, int K, double pswap)
{
    RMH_NEED_INIT();
    if (!m || !p || !s) return fail(RMH_ERR_PARAM, "run: null handle");
    if (s->m != m) return fail(RMH_ERR_PARAM, "run: state belongs to another model");
    if (p->d != m->d) return fail(RMH_ERR_PARAM, "theta and L have incompatible shapes"); // randomwalk.py:24
    if (T < 0 || nburn < 0 || nthin < 1) return fail(RMH_ERR_PARAM, "run: need Nsamples >= 0, Nburn >= 0, Nthin >= 1");
    if (!s->valid) return fail(RMH_ERR_STATE, "run: no current state (set theta0 first, or the previous run kept nothing)");
    const int64_t kept = (T >= nburn) ? (T - nburn) / nthin + 1 : 0;
    if (h) {
        if (h->s != s) return fail(RMH_ERR_PARAM, "run: history belongs to another state");
        if (h->entries < kept || h->iters < T) return fail(RMH_ERR_PARAM, "run: history too small");
    }
    int rc;
    if (p->kind == kHMC) {
        if (m->kind == kChangepoint)
            return fail(RMH_ERR_PARAM, "hmc: the changepoint model has no gradient");
        if ((rc = ensure_grad(s))) return rc;
    } else {
        // a re-pack of the tile image (changed .scale) reads the host copy of L
        if (p->acov && p->dirty && (rc = acov_sync_host(p))) return rc;
        if ((rc = sync_proposal(p))) return rc;
    }
    const bool adapt = p->adapt_target > 0;
    if (adapt && (rc = ensure_adapt(s))) return rc;
    ChainArgs c = chain_args(s);
    c.adapt = adapt ? s->adapt : nullptr;
    c.adapt_target = p->adapt_target;
    RunArgs r{};
    r.T = T;
    r.nburn = nburn;
    r.nthin = nthin;
    r.h_theta = h ? h->thetas : nullptr;
    r.h_logpost = h ? h->logpost : nullptr;
    r.h_accept = h ? h->accept : nullptr;
    // The chain's current state after run() is the LAST KEPT entry (sampler.py:49-50, 53-54),
    // recorded or not.  Without a history and with (T - Nburn) % Nthin != 0 the kernels record
    // exactly that one state (number `last`) into a one-entry sink of the chain state.
    const int64_t last = kept > 0 ? nburn + (kept - 1) * nthin : 0;
    const bool need_last = !h && kept > 0 && last != T;
    rmh_history sink;
    if (need_last) {
        const size_t rows = (size_t)4 * s->ks;
        if (!s->last_theta) {
            if ((rc = dev_alloc(&s->last_theta, rows * s->ld)) || (rc = dev_alloc(&s->last_logpost, (size_t)s->ld)))
                return rc;
            RMH_HIP(hipMemsetAsync(s->last_theta, 0, rows * s->ld * sizeof(double), g_stream));
        }
        sink.s = s;
        sink.entries = 1;
        sink.thetas = s->last_theta;
        sink.logpost = s->last_logpost;
        r.nburn = last;
        r.nthin = T + 1;
        r.h_theta = sink.thetas;
        r.h_logpost = sink.logpost;
    }
    if (K > 0) { // parallel tempering: the slots' inverse temperatures
        // adaptive proposals: the reference hands ONE proposal object to every rung (ptsampler.py:82), so
        // its adaptation state would couple the rungs of a sweep sequentially -- not built
        if ((p->kind != kRW && p->kind != kHMC && p->kind != kPCN) || adapt || p->acov || p->nblk > 0)
            return fail(RMH_ERR_PARAM, "run_pt: built for MetropolisRandomWalk, pCN and VanillaHMC / MALA "
                                       "within-chain moves (no adaptation)");
        if (p->kind == kHMC && m->kind != kGauss && m->kind != kLogReg && m->kind != kMixture)
            return fail(RMH_ERR_PARAM, "run_pt: this model has no gradient");
        std::vector<double> b((size_t)s->ld, 1.0);
        const int used = (16 / K) * K;
        for (int64_t c0 = 0; c0 < s->ld; ++c0)
            if ((c0 & 15) < used) b[c0] = betas[(c0 & 15) % K];
        if (!s->pt_beta && (rc = dev_alloc(&s->pt_beta, (size_t)s->ld))) return rc;
        RMH_HIP(hipMemcpy(s->pt_beta, b.data(), b.size() * sizeof(double), hipMemcpyHostToDevice));
        r.pt_beta = s->pt_beta;
        r.pt_K = K;
        r.pt_pswap = pswap;
    }
    RMH_HIP(hipEventRecord(g_ev0, g_stream));
    hipError_t e = hipSuccess;
    int launches = 1;
    if (p->kind == kCpRJ || m->kind == kChangepointRJ) { // trans-dimensional changepoint sampling
        if (p->kind != kCpRJ || m->kind != kChangepointRJ || adapt || K > 0)
            return fail(RMH_ERR_PARAM, "run: the variable-k changepoint model goes with "
                                       "ChangepointRegression1DProp (and nothing else)");
        if (c.nmode) return fail(RMH_ERR_PARAM, "run: normals=\"f64\" is not built for the variable-k changepoint kernels");
        CpRjArgs ra = m->rj;
        ra.sv = std::sqrt(0.01 * p->hscale * p->hscale / ra.base.n);   // test_changepoint.py:37
        ra.ss = std::sqrt(0.01 * p->hscale);                            // :38
        for (int b = 0; b < 3; ++b) ra.pcum[b] = p->blk_p[b];
        e = launch_cprj_run(ra, c, r, g_stream);
    } else if (m->big) { // 112 < d <= 896
        if (p->kind != kRW || adapt || K > 0 || p->acov || p->nblk > 0)
            return fail(RMH_ERR_PARAM, "run: for d > 112 only the plain random walk on the Gaussian model is built");
        if (c.nmode) return fail(RMH_ERR_PARAM, "run: normals=\"f64\" is not built for d > 112");
        RwArgs ra{p->d_Ltiles, nullptr, 0.0};
        e = launch_big_gauss_rw(m->NT, m->ga, ra, c, r, g_stream);
    } else if (p->acov) {
        if (K > 0 || p->kind == kPCN || p->nblk > 0 ||
            (p->kind == kHMC && (adapt || m->kind == kChangepoint)))
            return fail(RMH_ERR_PARAM, "run: AdaptCov is built for the random walk (plain or with scale "
                                       "adaptation) and for HMC on the Gaussian, mixture (random walk) "
                                       "and logistic models");
        launches = 0;
        if ((rc = run_adaptcov(m, p, s, c, T, r.nburn, r.nthin, need_last ? &sink : h, &launches))) return rc;
    } else if (K > 0 && p->kind == kHMC) {
        HmcArgs ha{p->eps, p->nsteps, p->d_chM, p->d_V, p->d_VT, p->minv};
        e = m->kind == kGauss     ? launch_gauss_hmc(m->NT, m->ga, ha, c, r, false, g_stream)
            : m->kind == kMixture ? launch_mix_hmc(m->NT, m->ma, ha, c, r, false, g_stream)
                                  : launch_logreg_hmc(m->NT, m->la, ha, c, r, false, g_stream);
    } else if (K > 0 && p->kind == kPCN) {
        RwArgs ra{p->d_Ltiles, p->d_V, p->rho};
        e = m->kind == kGauss         ? launch_gauss_pcn(m->NT, m->ga, ra, c, r, false, g_stream)
            : m->kind == kLogReg      ? launch_logreg_pcn(m->NT, m->la, ra, c, r, false, g_stream)
            : m->kind == kChangepoint ? launch_cp_generic(m->k, m->ca, ra, c, r, 5, g_stream)
                                      : launch_mix_pcn(m->NT, m->ma, ra, c, r, false, g_stream);
    } else if (K > 0) {
        RwArgs ra{p->d_Ltiles, nullptr, 0.0};
        const bool tail = m->kind == kGauss && m->ga.Wtail && p->d_Ltail && tail_ok(c) && !c.nmode;
        if (tail) ra.Ltail = p->d_Ltail;
        e = tail                      ? launch_gauss_rw_tail(m->ga, ra, c, r, false, g_stream)
            : m->kind == kGauss       ? launch_gauss_pt(m->NT, m->ga, ra, c, r, g_stream)
            : m->kind == kLogReg      ? launch_logreg_rw(m->NT, m->la, ra, c, r, false, g_stream)
            : m->kind == kChangepoint ? launch_cp_generic(m->k, m->ca, ra, c, r, 4, g_stream)
                                      : launch_mix_rw(m->NT, m->ma, ra, c, r, false, g_stream);
    } else if (p->kind == kPCN) {
        RwArgs ra{p->d_Ltiles, p->d_V, p->rho};
        e = m->kind == kGauss         ? launch_gauss_pcn(m->NT, m->ga, ra, c, r, adapt, g_stream)
            : m->kind == kLogReg      ? launch_logreg_pcn(m->NT, m->la, ra, c, r, adapt, g_stream)
            : m->kind == kChangepoint ? launch_cp_generic(m->k, m->ca, ra, c, r, adapt ? 3 : 2, g_stream)
                                      : launch_mix_pcn(m->NT, m->ma, ra, c, r, adapt, g_stream);
    } else if (p->kind == kRW) {
        RwArgs ra{p->d_Ltiles, nullptr, 0.0};
        if (p->nblk > 0) {
            if (m->kind != kChangepoint || adapt)
                return fail(RMH_ERR_PARAM, "run: block-mixture proposals are built for the changepoint model, "
                                           "without scale adaptation");
            if (c.nmode) return fail(RMH_ERR_PARAM, "run: normals=\"f64\" is not built for the block-mixture proposal");
            ra.nblk = p->nblk;
            for (int b = 0; b < 3; ++b) {
                ra.blk_end[b] = p->blk_end[b];
                ra.blk_p[b] = p->blk_p[b];
            }
        }
        if (m->kind == kGauss && m->ga.Wtail && p->d_Ltail && tail_ok(c)) {
            ra.Ltail = p->d_Ltail;   // d = 97..100: rows 96..99 on the VALU instead of a 7th row tile
            e = launch_gauss_rw_tail(m->ga, ra, c, r, adapt, g_stream);
        } else if (m->kind == kGauss) e = launch_gauss_rw(m->NT, m->ga, ra, c, r, adapt, g_stream);
        else if (m->kind == kChangepoint && (adapt || c.nmode)) { // (64-bit normals: the generic kernel over the changepoint policy)
            e = hipErrorNotReady;
            if (adapt && !c.nmode && p->nblk == 0 && p->L_is_diag && m->ca.ss_xp && m->ca.alpha_m1 == 0.0 &&
                !policy(RMH_POLICY_CP_NO_SS)) { // cfg 5's conditions: the one-lane-per-chain kernel with the chains' own scales
                ra.Ldiag = p->d_Ldiag;          // (unscaled in this mode, as the tile image)
                e = launch_cp_rw_ss(m->k, m->ca, ra, c, r, g_stream, true);
            }
            if (e == hipErrorNotReady) e = launch_cp_generic(m->k, m->ca, ra, c, r, adapt ? 1 : 0, g_stream);
        }
        else if (m->kind == kChangepoint && p->L_is_diag && m->ca.ss_xp && m->ca.alpha_m1 == 0.0 &&
                 !policy(RMH_POLICY_CP_NO_SS)) {
            // BASELINE cfg 5 (and the block-mixture proposal at fixed k): diagonal proposal factor,
            // alpha = 1: likelihood from sufficient statistics
            ra.Ldiag = p->d_Ldiag;
            e = launch_cp_rw_ss(m->k, m->ca, ra, c, r, g_stream);
        } else if (m->kind == kChangepoint) e = launch_cp_rw(m->k, m->ca, ra, c, r, g_stream);
        else if (m->kind == kMixture) e = launch_mix_rw(m->NT, m->ma, ra, c, r, adapt, g_stream);
        else e = launch_logreg_rw(m->NT, m->la, ra, c, r, adapt, g_stream);
    } else {
        HmcArgs ha{p->eps, p->nsteps, p->d_chM, p->d_V, p->d_VT, p->minv};
        if (m->kind == kMixture) e = launch_mix_hmc(m->NT, m->ma, ha, c, r, adapt, g_stream);
        else
        if (m->kind == kGauss) e = launch_gauss_hmc(m->NT, m->ga, ha, c, r, adapt, g_stream);
        else e = launch_logreg_hmc(m->NT, m->la, ha, c, r, adapt, g_stream);
    }
    RMH_HIP(e);
    RMH_HIP(hipEventRecord(g_ev1, g_stream));
    g_timed = true;
    g_last_launches = launches;
    // every kernel but the HMC ones moves theta and leaves the cached gradient behind
    if (p->kind != kHMC) s->grad_valid = false;
    // the chain's current state is the LAST KEPT entry (sampler.py:49-50, 53-54)
    if (kept == 0) {
        s->valid = false;
    } else if (last != T) {
        const int64_t rows = 4 * s->ks;
        const double *src_t = h ? h->thetas + (kept - 1) * rows * s->ld : sink.thetas;
        const double *src_l = h ? h->logpost + (kept - 1) * s->ld : sink.logpost;
        RMH_HIP(hipMemcpyAsync(s->theta, src_t, (size_t)rows * s->ld * sizeof(double), hipMemcpyDeviceToDevice, g_stream));
        RMH_HIP(hipMemcpyAsync(s->logpost, src_l, (size_t)s->ld * sizeof(double), hipMemcpyDeviceToDevice, g_stream));
        s->grad_valid = false;
    }
    return RMH_OK;
}

extern "C" {

int rmh_run(rmh_model_t m, rmh_proposal_t p, rmh_state_t s, int64_t T, int64_t nburn,
            int64_t nthin, rmh_history_t h)
{
    return run_impl(m, p, s, T, nburn, nthin, h, nullptr, 0, 0.0);
}

int rmh_run_pt(rmh_model_t m, rmh_proposal_t p, rmh_state_t s, const double *betas, int K,
               double pswap, int64_t T, int64_t nburn, int64_t nthin, rmh_history_t h)
{
    if (!betas || K < 1 || K > 16) return fail(RMH_ERR_PARAM, "run_pt: need 1 <= K <= 16 temperatures");
    for (int i = 0; i < K; ++i)
        if (!(betas[i] >= 0.0 && betas[i] <= 1.0)) // TemperedModel.__init__, ptsampler.py:26-27
            return fail(RMH_ERR_PARAM, "beta must be a number between 0 and 1");
    if (!(pswap > 0.0 && pswap < 1.0))             // PTSampler.__init__, ptsampler.py:72-73
        return fail(RMH_ERR_PARAM, "Pswap must be a number between 0 and 1");
    return run_impl(m, p, s, T, nburn, nthin, h, betas, K, pswap);
}

} // extern "C"

extern "C" {

// ---- AdaptCovProposal entry points (proposals/adaptive.py:38-103) ----
int rmh_proposal_set_adapt_cov(rmh_proposal_t p, const double *C0, double t_adapt, int marginalize,
                               int smooth_adapt, int exact_squares)
{
    RMH_NEED_INIT();
    if (!p || (p->kind != kRW && p->kind != kHMC))
        return fail(RMH_ERR_PARAM, "set_adapt_cov: not a random-walk or HMC proposal");
    if (!C0) {
        p->acov = false;
        return RMH_OK;
    }
    const int d = p->d, NTm = nt_for_dim(d + 1);
    if (NTm < 0) return fail(RMH_ERR_PARAM, "set_adapt_cov: d > 111 is not supported by the moment kernel");
    std::vector<double> C(C0, C0 + (size_t)d * d), L;
    if (!cholesky_lower(C, d, L)) return fail(RMH_ERR_PARAM, "set_adapt_cov: C0 is not positive definite");
    p->acov = true;
    p->acov_marg = marginalize != 0;
    p->acov_smooth = smooth_adapt != 0;
    p->acov_squares = exact_squares != 0;
    p->acov_t_adapt = t_adapt;
    p->acov_n = 0.0;
    p->acov_NT = NTm;
    p->acov_C0 = C;
    p->acov_C = C;                 // adaptive.py:60-62: C = C0, L = cholesky(C0)
    p->L = L;
    p->dirty = true;
    if (p->kind == kHMC) {         // AdaptCovHMC: M, chM = C, L (hamiltonian.py:117-119)
        p->minv = 1.0;
        int rcm = install_mass(p, L.data());
        if (rcm) return rcm;
    }
    const size_t n = (size_t)acov_ntri(p) * 256;
    (void)hipFree(p->d_acov_acc);
    (void)hipFree(p->d_acov_partial);
    p->d_acov_acc = p->d_acov_partial = nullptr;
    p->acov_partial_cap = 0;
    int rc = dev_alloc(&p->d_acov_acc, n);
    if (rc) return rc;
    RMH_HIP(hipMemset(p->d_acov_acc, 0, n * sizeof(double)));
    (void)hipFree(p->d_acov_C0); (void)hipFree(p->d_acov_C); (void)hipFree(p->d_acov_L);
    (void)hipFree(p->d_acov_tmp); (void)hipFree(p->d_acov_status);
    p->d_acov_C0 = p->d_acov_C = p->d_acov_L = p->d_acov_tmp = nullptr;
    p->d_acov_status = nullptr;
    p->acov_host_stale = false;
    if ((rc = upload(&p->d_acov_C0, C.data(), C.size())) || (rc = upload(&p->d_acov_C, C.data(), C.size())) ||
        (rc = upload(&p->d_acov_L, L.data(), L.size())) || (rc = dev_alloc(&p->d_acov_tmp, n)) ||
        (rc = dev_alloc(&p->d_acov_status, (size_t)1)))
        return rc;
    RMH_HIP(hipMemset(p->d_acov_status, 0, sizeof(int)));
    return RMH_OK;
}

// st: [n, SX (d), SX2 (d,d), C (d,d), L (d,d), minv] -- this rank's sums; minv: HMC only
int rmh_proposal_adapt_cov_get(rmh_proposal_t p, double *st)
{
    if (!p || !p->acov || !st) return fail(RMH_ERR_PARAM, "adapt_cov_get: not an AdaptCov proposal");
    const int d = p->d, D1 = d + 1;
    int rcs = acov_sync_host(p);
    if (rcs) return rcs;
    std::vector<double> acc((size_t)acov_ntri(p) * 256), Mm;
    RMH_HIP(hipMemcpy(acc.data(), p->d_acov_acc, acc.size() * sizeof(double), hipMemcpyDeviceToHost));
    acov_unpack(p, acc, Mm);
    st[0] = p->acov_n;
    for (int i = 0; i < d; ++i) st[1 + i] = Mm[(size_t)i * D1 + d];
    for (int i = 0; i < d; ++i)
        for (int j = 0; j < d; ++j) st[1 + d + (size_t)i * d + j] = Mm[(size_t)i * D1 + j];
    double *C = st + 1 + d + (size_t)d * d, *L = C + (size_t)d * d;
    for (size_t i = 0; i < (size_t)d * d; ++i) {
        C[i] = p->acov_C[i];
        L[i] = p->L[i];
    }
    L[(size_t)d * d] = p->kind == kHMC ? p->minv : 0.0;
    return RMH_OK;
}

// `count` = samples pooled in the sums (N chains x n adapt() calls on this rank)
int rmh_proposal_adapt_cov_set(rmh_proposal_t p, const double *st, double count)
{
    if (!p || !p->acov || !st) return fail(RMH_ERR_PARAM, "adapt_cov_set: not an AdaptCov proposal");
    const int d = p->d, D1 = d + 1;
    std::vector<double> Mm((size_t)D1 * D1, 0.0), acc;
    for (int i = 0; i < d; ++i) Mm[(size_t)i * D1 + d] = Mm[(size_t)d * D1 + i] = st[1 + i];
    for (int i = 0; i < d; ++i)
        for (int j = 0; j < d; ++j) Mm[(size_t)i * D1 + j] = st[1 + d + (size_t)i * d + j];
    Mm[(size_t)d * D1 + d] = count;
    acov_pack(p, Mm, acc);
    RMH_HIP(hipMemcpy(p->d_acov_acc, acc.data(), acc.size() * sizeof(double), hipMemcpyHostToDevice));
    p->acov_n = st[0];
    const double *C = st + 1 + d + (size_t)d * d, *L = C + (size_t)d * d;
    p->acov_C.assign(C, C + (size_t)d * d);
    p->L.assign(L, L + (size_t)d * d);
    p->dirty = true;
    p->acov_host_stale = false;
    RMH_HIP(hipMemcpy(p->d_acov_C, C, (size_t)d * d * sizeof(double), hipMemcpyHostToDevice));
    RMH_HIP(hipMemcpy(p->d_acov_L, L, (size_t)d * d * sizeof(double), hipMemcpyHostToDevice));
    if (p->kind == kHMC) {
        p->minv = L[(size_t)d * d] != 0.0 ? L[(size_t)d * d] : 1.0;
        return install_mass(p, p->L.data());
    }
    return RMH_OK;
}

} // extern "C"
