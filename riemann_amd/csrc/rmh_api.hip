// rmh_api.hip -- the C ABI of libriemann_hip.so (include/riemann_hip.h): host
// side of the MI355X back end.  Owns the device buffers behind the opaque
// handles, packs host operands into the kernels' tile images and launches the
// kernels of rmh_kernels_*.hip on the library's stream.  No computation of the
// hot path happens on the host: a missing GPU is an error, never a fallback.
// (Sampler.run / PTSampler.run and the covariance adaptation: rmh_run_api.hip.)
#include "rmh_host.hpp"
#include "rmh_icdf_table.h"
#include "rmh_icdf64_table.h"
#include "rmh_logistic_table.h"

using namespace rmh;
using namespace rmh::host;

namespace rmh {
int g_kernel_policy = 0;   // rmh_debug_set_kernel_policy
namespace host {

// ------------------------------------------------------------------ context
thread_local std::string g_err;
bool g_init = false;
int g_device = -1;
hipStream_t g_stream = nullptr;
hipEvent_t g_ev0 = nullptr, g_ev1 = nullptr;
hipStream_t g_copy = nullptr;                         // device -> host copies overlapping the library's stream
hipEvent_t g_cp_ready[2] = {nullptr, nullptr}, g_cp_done[2] = {nullptr, nullptr};
int g_last_launches = 0;
bool g_timed = false;
double *g_icdf = nullptr;   // device copy of the inverse-CDF table in its LDS layout
double *g_icdf64 = nullptr; // the 64-bit normal mode's table
double *g_logi = nullptr; // device copy of the logistic link-function table

int fail(int code, const std::string &msg)
{
    g_err = msg;
    return code;
}

int64_t pad16(int64_t n) { return (n + 15) / 16 * 16; }

// tile (i, s) of a (d x d) row-major matrix in A-operand lane order, scaled
void pack_tile(const double *M, int d, int i, int s, double scale, bool transpose, double *out)
{
    for (int l = 0; l < 64; ++l) {
        const int row = 16 * i + (l & 15), col = 4 * s + (l >> 4);
        double v = 0.0;
        if (row < d && col < d) v = transpose ? M[col * d + row] : M[row * d + col];
        out[l] = scale * v;
    }
}

// lower-triangular image: tiles (i, s), s <= 4i+3, index 2i(i+1)+s
std::vector<double> pack_lower(const double *M, int d, int NT, double scale)
{
    std::vector<double> img((size_t)tri_tiles(NT) * 64, 0.0);
    for (int i = 0; i < NT; ++i)
        for (int s = 0; s < 4 * (i + 1); ++s)
            pack_tile(M, d, i, s, scale, false, &img[(size_t)(2 * i * (i + 1) + s) * 64]);
    return img;
}

// upper-triangular image of M^T (M lower): tiles (i, s), s >= 4i
std::vector<double> pack_upper_T(const double *M, int d, int NT)
{
    const int S = 4 * NT;
    std::vector<double> img((size_t)tri_tiles(NT) * 64, 0.0);
    for (int i = 0; i < NT; ++i)
        for (int s = 4 * i; s < S; ++s)
            pack_tile(M, d, i, s, 1.0, true, &img[(size_t)(i * S - 2 * i * (i - 1) + (s - 4 * i)) * 64]);
    return img;
}

// rows 96..99 of a lower-triangular matrix in the layout k_run_rw_tail reads (rmh_run.hpp,
// TailAcc): [group S of four k-steps][lane group g][4 q + r] = M[96 + r][4 (4S + q) + g]
std::vector<double> pack_tail(const double *M, int d, double scale)
{
    // d = 16 NF + 1 .. 16 NF + 4: rows 16 NF .. of a lower-triangular matrix, [S][g][4 q + r] =
    // scale * M[16 NF + r][4 (4 S + q) + g] for the k-steps s = 4 S + q <= 4 NF (TailAcc, rmh_run.hpp)
    const int NF = (d - 1) / 16, r0 = 16 * NF;
    std::vector<double> t((size_t)(NF + 1) * 64, 0.0);
    for (int s = 0; s <= 4 * NF; ++s)
        for (int g = 0; g < 4; ++g)
            for (int r = 0; r < 4; ++r) {
                const int row = r0 + r, col = 4 * s + g;
                if (row < d && col < d && col <= row)
                    t[(size_t)((s / 4) * 4 + g) * 16 + 4 * (s % 4) + r] = scale * M[(size_t)row * d + col];
            }
    return t;
}

bool lower_ok(const double *L, int d)
{
    for (int i = 0; i < d; ++i) {
        if (!(L[i * d + i] > 0.0) || !std::isfinite(L[i * d + i])) return false;
        for (int j = i + 1; j < d; ++j)
            if (L[i * d + j] != 0.0) return false;
    }
    return true;
}


ChainArgs chain_args(const rmh_state *s)
{
    ChainArgs c{};
    c.theta = s->theta;
    c.logpost = s->logpost;
    c.grad = s->grad;
    c.mom = s->mom;
    c.rng = s->rng;
    c.ld = s->ld;
    c.nslots = 4 * s->ld;
    c.nblocks = s->ld / 16;
    c.d = s->d;
    c.dp = s->dp;
    c.ks = s->ks;
    c.M = (s->d + 7) / 8;   // stream words per sub-stream per iteration: two normals each
    c.nmode = s->nmode;
    c.icdf = s->nmode ? g_icdf64 : g_icdf;
    return c;
}

int sync_proposal(rmh_proposal *p)
{
    if ((p->kind != kRW && p->kind != kPCN) || !p->dirty) return RMH_OK;
    // adaptive variant: the chains carry their own scale, the image stays unscaled;
    // pCN: the image is rho_c * L (randomwalk.py:86, 93)
    const double sc = p->kind == kPCN ? std::sqrt(1 - p->rho * p->rho) : p->adapt_target > 0 ? 1.0 : p->scale;
    std::vector<double> img = pack_lower(p->L.data(), p->d, p->NT, sc);
    if (p->kind == kPCN && !p->d_V) { // (rho_c L)^-1 for the two solves of randomwalk.py:97-98
        const int d = p->d;
        std::vector<double> V((size_t)d * d, 0.0);
        for (int col = 0; col < d; ++col)
            for (int i = col; i < d; ++i) {
                double acc = (i == col) ? 1.0 : 0.0;
                for (int j = col; j < i; ++j) acc -= sc * p->L[i * d + j] * V[j * d + col];
                V[i * d + col] = acc / (sc * p->L[i * d + i]);
            }
        std::vector<double> vimg = pack_lower(V.data(), d, p->NT, 1.0);
        int rc = upload(&p->d_V, vimg.data(), vimg.size());
        if (rc) return rc;
    }
    if (!p->d_Ltiles) {
        int rc = dev_alloc(&p->d_Ltiles, img.size());
        if (rc) return rc;
    }
    RMH_HIP(hipMemcpy(p->d_Ltiles, img.data(), img.size() * sizeof(double), hipMemcpyHostToDevice));
    if (p->kind == kRW && has_tail(p->d)) { // rows 96.. for the VALU tail of k_run_rw_tail
        std::vector<double> lt = pack_tail(p->L.data(), p->d, sc);
        if (!p->d_Ltail) {
            int rc = dev_alloc(&p->d_Ltail, lt.size());
            if (rc) return rc;
        }
        RMH_HIP(hipMemcpy(p->d_Ltail, lt.data(), lt.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    // diagonal L (e.g. the block proposal of riemann/test_changepoint.py:36-38 as one matrix):
    // the proposal is an axpy per coordinate, no matrix product (k_cp_rw_ss)
    p->L_is_diag = p->kind == kRW;
    for (int i = 0; i < p->d && p->L_is_diag; ++i)
        for (int j = 0; j < i; ++j)
            if (p->L[(size_t)i * p->d + j] != 0.0) {
                p->L_is_diag = false;
                break;
            }
    if (p->L_is_diag) {
        std::vector<double> dg((size_t)p->d);
        for (int i = 0; i < p->d; ++i) dg[i] = sc * p->L[(size_t)i * p->d + i];
        if (!p->d_Ldiag) {
            int rc = dev_alloc(&p->d_Ldiag, dg.size());
            if (rc) return rc;
        }
        RMH_HIP(hipMemcpy(p->d_Ldiag, dg.data(), dg.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    p->dirty = false;
    return RMH_OK;
}

// log posterior (and optionally gradient / parts) of the states in `c`
int eval_states(rmh_model *m, const ChainArgs &c, bool want_grad, double *lp, double *grad,
                double *lprior, double *llik)
{
    hipError_t e = hipSuccess;
    switch (m->kind) {
    case kGauss:
        if (m->big) {
            if (want_grad) return fail(RMH_ERR_PARAM, "gaussian: the gradient is built for d <= 112 only");
            e = launch_big_gauss_eval(m->NT, m->ga, c, lp, lprior, llik, g_stream);
            break;
        }
        e = launch_gauss_eval(m->NT, m->ga, c, want_grad, lp, grad, lprior, llik, g_stream);
        break;
    case kLogReg:
        e = launch_logreg_eval(m->NT, m->la, c, want_grad, lp, grad, lprior, llik, g_stream);
        break;
    case kChangepointRJ:
        if (want_grad)
            return fail(RMH_ERR_PARAM, "the changepoint model has no gradient "
                                       "(riemann/models/changepoint.py:108-113)");
        e = launch_cprj_eval(m->rj, c, lprior, llik, lp, g_stream);
        break;
    case kMixture:
        e = launch_mix_eval(m->NT, m->ma, c, want_grad, lp, grad, lprior, llik, g_stream);
        break;
    case kChangepoint:
        if (want_grad)
            return fail(RMH_ERR_PARAM, "the changepoint model has no gradient "
                                       "(riemann/models/changepoint.py:108-113)");
        e = launch_cp_eval(m->k, m->ca, c, lprior, llik, lp, g_stream);
        break;
    default:
        return fail(RMH_ERR_PARAM, "unknown model kind");
    }
    RMH_HIP(e);
    return RMH_OK;
}

int ensure_grad(rmh_state *s)
{
    if (s->grad_valid) return RMH_OK;
    if (!s->grad) {
        int rc = dev_alloc(&s->grad, (size_t)s->dp * s->ld);
        if (rc) return rc;
        RMH_HIP(hipMemsetAsync(s->grad, 0, (size_t)s->dp * s->ld * sizeof(double), g_stream));
    }
    if (!s->mom) {
        // (dp + 4, ld + 128): the last four rows park per-lane scalars (k_run_hmc)
        int rc = dev_alloc(&s->mom, (size_t)(s->dp + 4) * (s->ld + 128));
        if (rc) return rc;
    }
    ChainArgs c = chain_args(s);
    int rc = eval_states(s->m, c, true, nullptr, s->grad, nullptr, nullptr);
    if (rc) return rc;
    s->grad_valid = true;
    return RMH_OK;
}

} // namespace host
} // namespace rmh

namespace {

// temporary device state for the host-in/host-out entry points

struct TmpState {
    rmh_state s;
    double *staging = nullptr;
    ~TmpState()
    {
        (void)hipFree(s.theta);
        (void)hipFree(s.grad);
        (void)hipFree(s.mom);
        (void)hipFree(s.adapt);
        (void)hipFree(s.logpost);
        (void)hipFree(s.rng);
        (void)hipFree(staging);
    }
};

int tmp_state(TmpState &t, rmh_model *m, const double *theta, int64_t N)
{
    rmh_state &s = t.s;
    s.m = m;
    s.N = N;
    s.ld = pad16(N);
    s.d = m->d;
    s.NT = m->NT;
    s.dp = 16 * m->NT;
    s.ks = (m->d + 3) / 4;
    int rc;
    if ((rc = dev_alloc(&s.theta, (size_t)s.dp * s.ld))) return rc;
    if ((rc = dev_alloc(&s.logpost, (size_t)s.ld))) return rc;
    if ((rc = upload(&t.staging, theta, (size_t)N * s.d))) return rc;
    RMH_HIP(hipMemsetAsync(s.theta, 0, (size_t)s.dp * s.ld * sizeof(double), g_stream));
    RMH_HIP(launch_transpose_nd_to_dn(t.staging, s.theta, N, s.d, s.ld, g_stream));
    return RMH_OK;
}

void rng_to_device_layout(const uint64_t *host, int64_t N, int64_t ld, std::vector<uint64_t> &dev)
{
    const int64_t nslots = 4 * ld;
    dev.assign((size_t)4 * nslots, 0);
    for (int64_t c = 0; c < N; ++c)
        for (int g = 0; g < 4; ++g) {
            const int64_t slot = (c >> 4) * 64 + g * 16 + (c & 15);
            for (int w = 0; w < 4; ++w) dev[(size_t)w * nslots + slot] = host[c * 16 + g * 4 + w];
        }
}

void rng_from_device_layout(const std::vector<uint64_t> &dev, int64_t N, int64_t ld, uint64_t *host)
{
    const int64_t nslots = 4 * ld;
    for (int64_t c = 0; c < N; ++c)
        for (int g = 0; g < 4; ++g) {
            const int64_t slot = (c >> 4) * 64 + g * 16 + (c & 15);
            for (int w = 0; w < 4; ++w) host[c * 16 + g * 4 + w] = dev[(size_t)w * nslots + slot];
        }
}

} // namespace

// hooks for rmh_comm.hip
namespace rmh {
int comm_fail(int code, const std::string &msg) { return fail(code, msg); }
hipStream_t comm_stream() { return g_stream; }
int comm_hist_info(rmh_history_t h, rmh_state_t *s, double **thetas, int64_t *entries, int64_t *rows)
{
    if (!h) return fail(RMH_ERR_PARAM, "null history");
    *s = h->s;
    *thetas = h->thetas;
    *entries = h->entries;
    *rows = 4 * (int64_t)h->s->ks;
    return RMH_OK;
}

int comm_state_info(rmh_state_t s, double **theta, int64_t *N, int64_t *ld, int *d, int *dp)
{
    if (!s) return fail(RMH_ERR_PARAM, "null state");
    if (!s->valid) return fail(RMH_ERR_STATE, "state: no current state (empty history)");
    *theta = s->theta;
    *N = s->N;
    *ld = s->ld;
    *d = s->d;
    *dp = s->dp;
    return RMH_OK;
}
} // namespace rmh

// ------------------------------------------------------------------ context
extern "C" {

const char *rmh_last_error(void) { return g_err.c_str(); }

int rmh_device_count(int *count)
{
    RMH_HIP(hipGetDeviceCount(count));
    return RMH_OK;
}

int rmh_init(int device)
{
    if (g_init && (device == g_device || device < 0)) return RMH_OK; // device < 0: "whatever is bound"
    if (device < 0) return fail(RMH_ERR_STATE, "rmh_init: no device bound yet");
    if (g_init) return fail(RMH_ERR_STATE, "rmh_init: already bound to another device");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(RMH_ERR_HIP, "rmh_init: no HIP device visible (this library has no CPU path)");
    if (device < 0 || device >= n) return fail(RMH_ERR_PARAM, "rmh_init: bad device index");
    RMH_HIP(hipSetDevice(device));
    RMH_HIP(hipStreamCreate(&g_stream));
    RMH_HIP(hipEventCreate(&g_ev0));
    RMH_HIP(hipEventCreate(&g_ev1));
    RMH_HIP(hipStreamCreate(&g_copy));
    for (int k = 0; k < 2; ++k) {
        RMH_HIP(hipEventCreateWithFlags(&g_cp_ready[k], hipEventDisableTiming));
        RMH_HIP(hipEventCreateWithFlags(&g_cp_done[k], hipEventDisableTiming));
    }
    {   // inverse-normal-CDF table of the random stream, padded to an 80-byte cell stride
        static_assert(RMH_ICDF_ROWS == 56 && RMH_ICDF_SUB == 8 && RMH_ICDF_NC == 7, "table shape");
        // kIcdfStride, kIcdfRowStride of rmh_device.hpp
        const int cell_stride = 10, row_stride = RMH_ICDF_SUB * cell_stride;
        std::vector<double> pad((size_t)RMH_ICDF_ROWS * row_stride, 0.0);
        for (int r = 0; r < RMH_ICDF_ROWS; ++r)
            for (int j = 0; j < RMH_ICDF_SUB; ++j)
                for (int k = 0; k < RMH_ICDF_NC; ++k)
                    pad[(size_t)r * row_stride + j * cell_stride + k] =
                        rmh_icdf_table[(r * RMH_ICDF_SUB + j) * RMH_ICDF_NC + k];
        int rc = upload(&g_icdf, pad.data(), pad.size());
        if (rc) return rc;
        // the 64-bit normal mode's table: kIcdf64Stride, kIcdf64RowStride of rmh_device.hpp
        static_assert(RMH_ICDF64_ROWS == 56 && RMH_ICDF64_SUB == 8 && RMH_ICDF64_NC == 11, "table shape");
        const int cs64 = 12, rs64 = RMH_ICDF64_SUB * cs64 + 2;
        std::vector<double> pad64((size_t)RMH_ICDF64_ROWS * rs64, 0.0);
        for (int r = 0; r < RMH_ICDF64_ROWS; ++r)
            for (int j = 0; j < RMH_ICDF64_SUB; ++j)
                for (int k = 0; k < RMH_ICDF64_NC; ++k)
                    pad64[(size_t)r * rs64 + j * cs64 + k] = rmh_icdf64_table[(r * RMH_ICDF64_SUB + j) * RMH_ICDF64_NC + k];
        if ((rc = upload(&g_icdf64, pad64.data(), pad64.size()))) return rc;
        static_assert(RMH_LOGI_NI == 600 && RMH_LOGI_W == 16 && RMH_LOGI_NC == 6, "table shape");
        if ((rc = upload(&g_logi, rmh_logistic_table, sizeof(rmh_logistic_table) / sizeof(double)))) return rc;
    }
    g_device = device;
    g_init = true;
    return RMH_OK;
}

int rmh_device_name(char *buf, int buflen)
{
    RMH_NEED_INIT();
    hipDeviceProp_t prop;
    RMH_HIP(hipGetDeviceProperties(&prop, g_device));
    snprintf(buf, (size_t)buflen, "%s %s", prop.name, prop.gcnArchName);
    return RMH_OK;
}

int rmh_synchronize(void)
{
    RMH_NEED_INIT();
    RMH_HIP(hipStreamSynchronize(g_stream));
    return RMH_OK;
}

int rmh_last_run_ms(double *ms, int *launches)
{
    RMH_NEED_INIT();
    if (!g_timed) return fail(RMH_ERR_STATE, "rmh_last_run_ms: no run recorded");
    RMH_HIP(hipEventSynchronize(g_ev1));
    float f = 0.f;
    RMH_HIP(hipEventElapsedTime(&f, g_ev0, g_ev1));
    if (ms) *ms = f;
    if (launches) *launches = g_last_launches;
    return RMH_OK;
}

// ------------------------------------------------------------------- models
int rmh_model_gaussian_create(const double *mu, const double *L, int d, rmh_model_t *out)
{
    RMH_NEED_INIT();
    if (d < 1) return fail(RMH_ERR_PARAM, "gaussian: d must be >= 1");
    const bool big = nt_for_dim(d) < 0;
    const int NT = big ? nt_big(d) : nt_for_dim(d);
    if (NT < 0) return fail(RMH_ERR_PARAM, "gaussian: d > 1024 is not supported");
    if (!lower_ok(L, d))
        return fail(RMH_ERR_PARAM, "gaussian: L must be a lower-triangular Cholesky factor");
    // W = L^-1 by forward substitution (columns of the identity)
    std::vector<double> W((size_t)d * d, 0.0);
    for (int col = 0; col < d; ++col)
        for (int i = col; i < d; ++i) {
            double acc = (i == col) ? 1.0 : 0.0;
            for (int j = col; j < i; ++j) acc -= L[i * d + j] * W[j * d + col];
            W[i * d + col] = acc / L[i * d + i];
        }
    double slog = 0.0;
    for (int i = 0; i < d; ++i) slog += std::log(L[i * d + i]);
    rmh_model *m = new rmh_model;
    m->kind = kGauss;
    m->d = d;
    m->NT = NT;
    m->big = big;
    std::vector<double> lo = pack_lower(W.data(), d, NT, 1.0), up;
    if (!big) up = pack_upper_T(W.data(), d, NT);   // gradient image: register-tile kernels only
    else up.assign(64, 0.0);
    std::vector<double> mup((size_t)16 * NT, 0.0);
    for (int i = 0; i < d; ++i) mup[i] = mu[i];
    int rc;
    if ((rc = upload(&m->d_W, lo.data(), lo.size())) || (rc = upload(&m->d_WT, up.data(), up.size())) ||
        (rc = upload(&m->d_mu, mup.data(), mup.size()))) {
        rmh_model_free(m);
        return rc;
    }
    if (has_tail(d)) {
        std::vector<double> wt = pack_tail(W.data(), d, 1.0);
        if ((rc = upload(&m->d_Wtail, wt.data(), wt.size()))) {
            rmh_model_free(m);
            return rc;
        }
    }
    m->ga.Wtail = m->d_Wtail;
    m->ga.Wtiles = m->d_W;
    m->ga.WTtiles = m->d_WT;
    m->ga.mu_pad = m->d_mu;
    m->ga.dlog2pi = d * std::log(2 * M_PI);   // gaussian.py:52
    m->ga.logdetC = 2 * slog;                 // gaussian.py:43
    m->ga.ks = (d + 3) / 4;
    *out = m;
    return RMH_OK;
}

int rmh_model_mixture_create(const double *weights, const double *mu, const double *L, int ncomp,
                             int d, double log_prior, rmh_model_t *out)
{
    RMH_NEED_INIT();
    if (!weights || !mu || !L || !out || ncomp < 1 || d < 1) return fail(RMH_ERR_PARAM, "mixture: bad argument");
    const int NT = nt_for_dim(d);
    if (NT < 0) return fail(RMH_ERR_PARAM, "mixture: d > 112 is not supported by the MFMA tile yet");
    std::vector<double> imgs, imgsT, mus((size_t)ncomp * 16 * NT, 0.0), cst(ncomp);
    for (int k = 0; k < ncomp; ++k) {
        const double *Lk = L + (size_t)k * d * d;
        if (!lower_ok(Lk, d))
            return fail(RMH_ERR_PARAM, "mixture: every L must be a lower-triangular Cholesky factor");
        std::vector<double> W((size_t)d * d, 0.0);
        for (int col = 0; col < d; ++col)
            for (int i = col; i < d; ++i) {
                double acc = (i == col) ? 1.0 : 0.0;
                for (int j = col; j < i; ++j) acc -= Lk[i * d + j] * W[j * d + col];
                W[i * d + col] = acc / Lk[i * d + i];
            }
        std::vector<double> lo = pack_lower(W.data(), d, NT, 1.0), up = pack_upper_T(W.data(), d, NT);
        imgs.insert(imgs.end(), lo.begin(), lo.end());
        imgsT.insert(imgsT.end(), up.begin(), up.end());
        for (int i = 0; i < d; ++i) mus[(size_t)k * 16 * NT + i] = mu[(size_t)k * d + i];
        double slog = 0.0;
        for (int i = 0; i < d; ++i) slog += std::log(Lk[i * d + i]);
        cst[k] = 2 * slog;                     // gaussian.py:43
    }
    rmh_model *m = new rmh_model;
    m->kind = kMixture;
    m->d = d;
    m->NT = NT;
    int rc;
    if ((rc = upload(&m->d_mixW, imgs.data(), imgs.size())) ||
        (rc = upload(&m->d_mixWT, imgsT.data(), imgsT.size())) ||
        (rc = upload(&m->d_mixmu, mus.data(), mus.size())) ||
        (rc = upload(&m->d_mixc, cst.data(), cst.size())) ||
        (rc = upload(&m->d_mixw, weights, (size_t)ncomp))) {
        rmh_model_free(m);
        return rc;
    }
    m->ma.Wtiles = m->d_mixW;
    m->ma.WTtiles = m->d_mixWT;
    m->ma.mu_pad = m->d_mixmu;
    m->ma.cst = m->d_mixc;
    m->ma.wts = m->d_mixw;
    m->ma.ncomp = ncomp;
    m->ma.ks = (d + 3) / 4;
    m->ma.dlog2pi = d * std::log(2 * M_PI);    // gaussian.py:52
    m->ma.logP0 = log_prior;
    *out = m;
    return RMH_OK;
}

int rmh_model_logreg_create(const double *X, const double *y, int64_t n, int d, double tau2,
                            rmh_model_t *out)
{
    RMH_NEED_INIT();
    if (d < 1 || n < 1) return fail(RMH_ERR_PARAM, "logreg: need n >= 1 rows and d >= 1 columns");
    if (!(tau2 > 0)) return fail(RMH_ERR_PARAM, "logreg: tau2 must be positive");
    const int NT = nt_for_dim(d);
    if (NT < 0) return fail(RMH_ERR_PARAM, "logreg: d > 112 is not supported by the MFMA tile yet");
    // stage images of kLogRegCH row tiles each (layout: LogRegModel, rmh_models.hpp)
    constexpr int CH = kLogRegCH, RPS = 16 * CH;   // data rows per stage
    const int64_t nchunks = (n + RPS - 1) / RPS;
    const int XT = 256 * NT, CHD = kChunkDoubles(NT);
    std::vector<double> img((size_t)nchunks * CHD, 0.0);
    for (int64_t row = 0; row < n; ++row) {
        double *ch = &img[(size_t)(row / RPS) * CHD];
        const int k2 = (int)((row % RPS) / 16), i = (int)(row & 15);
        double *t = ch + k2 * XT;
        for (int k = 0; k < d; ++k) {
            const int s = k >> 2, gg = k & 3;
            const int F = ((s & 3) << 2) | gg;
            t[s * 64 + gg * 16 + (i ^ F)] = 16.0 * X[row * d + k];   // 16 X: rmh_device.hpp, logistic_lockstep
        }
        ch[CH * XT + k2 * 32 + 2 * i] = y[row] - 0.5;      // (y - 1/2, w / 16) pairs: w = 1 marks a real row
        ch[CH * XT + k2 * 32 + 2 * i + 1] = 0.0625;
    }
    rmh_model *m = new rmh_model;
    m->kind = kLogReg;
    m->d = d;
    m->NT = NT;
    int rc = upload(&m->d_X, img.data(), img.size());
    if (rc) {
        rmh_model_free(m);
        return rc;
    }
    m->la.Ximg = m->d_X;
    m->la.nchunks = nchunks;
    m->la.ks = (d + 3) / 4;
    m->la.logi = g_logi;
    m->la.inv_tau2 = 1.0 / tau2;
    m->la.two_tau2 = 2.0 * tau2;
    *out = m;
    return RMH_OK;
}

// Sufficient statistics of the changepoint likelihood (k_cp_rw_ss, k_cp_rw_ss64, k_cprj_run): with x
// sorted, the data of piece p are a contiguous run, so sum (ypred - y)^2 needs only the run
// boundaries (searches) and prefix sums of y.  Leaves a.ss_xp null when n is too large for the LDS.
static int build_cp_ss(rmh_model *m, CpArgs &a, const double *x, const double *y, int n)
{
    a.ss_xp = nullptr;
    a.ss_steps = 0;
    a.ss_ym = a.ss_p2 = 0.0;
    a.ss_nb = a.ss_s2 = 0;
    a.ss_x0 = a.ss_invh = 0.0;
    if (n > 2047) return RMH_OK;
    int steps = 1;
    while ((2 << (steps - 1)) <= n) ++steps;          // floor(log2 n) + 1
    const int np = (1 << steps) - 1;                  // >= n
    std::vector<int> ord((size_t)n);
    for (int i = 0; i < n; ++i) ord[i] = i;
    std::stable_sort(ord.begin(), ord.end(), [&](int u, int v) { return x[u] < x[v]; });
    double ym = 0.0;
    for (int i = 0; i < n; ++i) ym += y[i];
    ym /= n;
    std::vector<double> xp((size_t)np + (np + 1), 0.0);
    double p1 = 0.0, p2 = 0.0;
    for (int i = 0; i < np; ++i) xp[i] = i < n ? x[ord[i]] : HUGE_VAL;
    for (int i = 0; i <= np; ++i) {
        xp[(size_t)np + i] = p1;                      // P1[i] = sum of the first i centred y
        if (i < n) {
            const double yc = y[ord[i]] - ym;
            p1 += yc;
            p2 += yc * yc;
        }
    }
    // bucket table (rmh_internal.hpp, CpArgs): the smallest power of two of buckets that leaves at
    // most 3 data points in any bucket (2 search steps), else 7 (3 steps) at 1024 buckets, else none;
    // the probes reach 2^s2 - 2 entries beyond a bucket's first point, which the +inf padding must cover
    const double x0 = xp[0], x1 = xp[(size_t)n - 1];
    if (x1 > x0 && std::isfinite(x0) && std::isfinite(x1)) {
        for (int nb = 64; nb <= 1024 && !a.ss_nb; nb *= 2) {
            const double invh = nb / (x1 - x0);
            std::vector<int> L((size_t)nb + 1, 0);
            int maxw = 0;
            for (int i = 0; i < n; ++i) {
                const int b = (int)std::fmin(std::fmax((xp[i] - x0) * invh, 0.0), (double)(nb - 1));
                ++L[(size_t)b + 1];
            }
            for (int b = 0; b < nb; ++b) {
                maxw = std::max(maxw, L[(size_t)b + 1]);
                L[(size_t)b + 1] += L[b];
            }
            const int s2 = maxw <= 3 ? 2 : (maxw <= 7 && nb == 1024) ? 3 : 0;
            if (!s2 || np - n < (1 << s2) - 1) continue;
            a.ss_nb = nb;
            a.ss_s2 = s2;
            a.ss_x0 = x0;
            a.ss_invh = invh;
            xp.resize((size_t)2 * np + 2 + (nb + 2) / 2 + 1, 0.0);
            std::vector<int32_t> L8((size_t)nb + 2, 0);
            for (int b = 0; b <= nb; ++b) L8[b] = 8 * L[b];
            std::memcpy(xp.data() + 2 * np + 2, L8.data(), L8.size() * sizeof(int32_t));
        }
    }
    int rc;
    if ((rc = upload(&m->d_ssxp, xp.data(), xp.size()))) return rc;
    a.ss_xp = m->d_ssxp;
    a.ss_steps = steps;
    a.ss_ym = ym;
    a.ss_p2 = p2;
    return RMH_OK;
}

int rmh_model_changepoint_create(const double *x, const double *y, int n, int k, double xmin,
                                 double xmax, double lamb, double alpha, double beta,
                                 rmh_model_t *out)
{
    RMH_NEED_INIT();
    if (k < 1 || k > 8) return fail(RMH_ERR_PARAM, "changepoint: 1 <= k <= 8 changepoints supported");
    if (n < 1 || n > 8192) return fail(RMH_ERR_PARAM, "changepoint: 1 <= n <= 8192 data points supported");
    rmh_model *m = new rmh_model;
    m->kind = kChangepoint;
    m->k = k;
    m->d = 2 * k + 2;
    m->NT = nt_for_dim(m->d);
    std::vector<double> xy((size_t)2 * n);
    for (int i = 0; i < n; ++i) {
        xy[i] = x[i];
        xy[n + i] = y[i];
    }
    int rc = upload(&m->d_xy, xy.data(), xy.size());
    if (rc) {
        rmh_model_free(m);
        return rc;
    }
    const double kp = k + 1; // "k = len(theta.cpv)", changepoint.py:133
    CpArgs &a = m->ca;
    a.xy = m->d_xy;
    a.n = n;
    a.xmin = xmin;
    a.xmax = xmax;
    a.alpha_m1 = alpha - 1;
    a.beta = beta;
    a.c_k = kp * std::log(lamb) - std::lgamma(kp) - lamb; // :134
    a.c_v = alpha * std::log(beta);                       // :19
    a.lgam_alpha = std::lgamma(alpha);
    a.c_s = std::lgamma(2 * kp + 1);                      // :143
    a.c_sL = kp * std::log(xmax - xmin);
    a.nlog2pi = n * std::log(2 * M_PI);                   // :119
    if ((rc = build_cp_ss(m, a, x, y, n))) {
        rmh_model_free(m);
        return rc;
    }
    *out = m;
    return RMH_OK;
}

int rmh_model_changepoint_rj_create(const double *x, const double *y, int n, int kmax, double xmin,
                                    double xmax, double lamb, double alpha, double beta,
                                    rmh_model_t *out)
{
    RMH_NEED_INIT();
    if (kmax < 1 || kmax > 14) return fail(RMH_ERR_PARAM, "changepoint: 1 <= kmax <= 14 supported");
    if (n < 1 || n > 8192) return fail(RMH_ERR_PARAM, "changepoint: 1 <= n <= 8192 data points supported");
    rmh_model *m = new rmh_model;
    m->kind = kChangepointRJ;
    m->k = kmax;
    m->d = 2 * kmax + 3;
    m->NT = nt_for_dim(m->d);
    std::vector<double> xy((size_t)2 * n), kt((size_t)4 * (kmax + 2), 0.0);
    for (int i = 0; i < n; ++i) {
        xy[i] = x[i];
        xy[n + i] = y[i];
    }
    for (int kp = 1; kp <= kmax + 1; ++kp) { // kp = pieces = len(theta.cpv), changepoint.py:133
        kt[4 * kp + 0] = kp * std::log(lamb) - std::lgamma((double)kp) - lamb;   // :134
        kt[4 * kp + 1] = std::lgamma(2.0 * kp + 1);                               // :143
        kt[4 * kp + 2] = kp * std::log(xmax - xmin);
        kt[4 * kp + 3] = std::sqrt(0.01 * (xmax - xmin) / kp);                    // test_changepoint.py:36
    }
    int rc;
    if ((rc = upload(&m->d_xy, xy.data(), xy.size())) || (rc = upload(&m->d_ktab, kt.data(), kt.size()))) {
        rmh_model_free(m);
        return rc;
    }
    CpArgs &a = m->rj.base;
    a.xy = m->d_xy;
    a.n = n;
    a.xmin = xmin;
    a.xmax = xmax;
    a.alpha_m1 = alpha - 1;
    a.beta = beta;
    a.c_v = alpha * std::log(beta);
    a.lgam_alpha = std::lgamma(alpha);
    a.nlog2pi = n * std::log(2 * M_PI);
    m->rj.ktab = m->d_ktab;
    m->rj.KM = kmax;
    if ((rc = build_cp_ss(m, a, x, y, n))) {
        rmh_model_free(m);
        return rc;
    }
    *out = m;
    return RMH_OK;
}

int rmh_model_dim(rmh_model_t m, int *d)
{
    if (!m) return fail(RMH_ERR_PARAM, "null model");
    *d = m->d;
    return RMH_OK;
}

int rmh_model_free(rmh_model_t m)
{
    if (!m) return RMH_OK;
    (void)hipFree(m->d_W);
    (void)hipFree(m->d_WT);
    (void)hipFree(m->d_mu);
    (void)hipFree(m->d_Wtail);
    (void)hipFree(m->d_X);
    (void)hipFree(m->d_xy);
    (void)hipFree(m->d_ssxp);
    (void)hipFree(m->d_ktab);
    (void)hipFree(m->d_mixW);
    (void)hipFree(m->d_mixWT);
    (void)hipFree(m->d_mixmu);
    (void)hipFree(m->d_mixc);
    (void)hipFree(m->d_mixw);
    delete m;
    return RMH_OK;
}

static int eval_host(rmh_model_t m, const double *theta, int64_t N, int which, double *out)
{
    RMH_NEED_INIT();
    if (!m || !theta || !out || N < 1) return fail(RMH_ERR_PARAM, "bad argument");
    TmpState t;
    int rc = tmp_state(t, m, theta, N);
    if (rc) return rc;
    ChainArgs c = chain_args(&t.s);
    if (which == 3) { // gradient
        if ((rc = dev_alloc(&t.s.grad, (size_t)t.s.dp * t.s.ld))) return rc;
        RMH_HIP(hipMemsetAsync(t.s.grad, 0, (size_t)t.s.dp * t.s.ld * sizeof(double), g_stream));
        if ((rc = eval_states(m, c, true, nullptr, t.s.grad, nullptr, nullptr))) return rc;
        double *nd = nullptr;
        if ((rc = dev_alloc(&nd, (size_t)N * m->d))) return rc;
        hipError_t e = launch_transpose_dn_to_nd(t.s.grad, nd, N, m->d, t.s.ld, 1, g_stream);
        if (e == hipSuccess) e = hipMemcpyAsync(out, nd, (size_t)N * m->d * sizeof(double), hipMemcpyDeviceToHost, g_stream);
        if (e == hipSuccess) e = hipStreamSynchronize(g_stream);
        (void)hipFree(nd);
        RMH_HIP(e);
        return RMH_OK;
    }
    double *lp = t.s.logpost;
    rc = eval_states(m, c, false, which == 2 ? lp : nullptr, nullptr, which == 0 ? lp : nullptr,
                     which == 1 ? lp : nullptr);
    if (rc) return rc;
    RMH_HIP(hipMemcpyAsync(out, lp, (size_t)N * sizeof(double), hipMemcpyDeviceToHost, g_stream));
    RMH_HIP(hipStreamSynchronize(g_stream));
    return RMH_OK;
}

int rmh_log_prior(rmh_model_t m, const double *theta, int64_t N, double *out) { return eval_host(m, theta, N, 0, out); }
int rmh_log_likelihood(rmh_model_t m, const double *theta, int64_t N, double *out) { return eval_host(m, theta, N, 1, out); }
int rmh_log_posterior(rmh_model_t m, const double *theta, int64_t N, double *out) { return eval_host(m, theta, N, 2, out); }
int rmh_grad_log_posterior(rmh_model_t m, const double *theta, int64_t N, double *out) { return eval_host(m, theta, N, 3, out); }

// ---------------------------------------------------------------- proposals
int rmh_proposal_rw_create(const double *L, int d, double scale, rmh_proposal_t *out)
{
    RMH_NEED_INIT();
    if (d < 1) return fail(RMH_ERR_PARAM, "random walk: d must be >= 1");
    const int NT = nt_for_dim(d) < 0 ? nt_big(d) : nt_for_dim(d);
    if (NT < 0) return fail(RMH_ERR_PARAM, "random walk: d > 1024 is not supported");
    if (!lower_ok(L, d))
        return fail(RMH_ERR_PARAM, "random walk: L must be a lower-triangular Cholesky factor");
    rmh_proposal *p = new rmh_proposal;
    p->kind = kRW;
    p->d = d;
    p->NT = NT;
    p->scale = scale;
    p->L.assign(L, L + (size_t)d * d);
    p->dirty = true;
    *out = p;
    return RMH_OK;
}

int rmh_proposal_rw_set_blocks(rmh_proposal_t p, int nblk, const int *sizes, const double *pcum)
{
    if (!p || p->kind != kRW) return fail(RMH_ERR_PARAM, "set_blocks: not a random-walk proposal");
    if (nblk == 0) {
        p->nblk = 0;
        return RMH_OK;
    }
    if (nblk < 1 || nblk > 3 || !sizes || !pcum) return fail(RMH_ERR_PARAM, "set_blocks: 1 to 3 blocks");
    int end = 0;
    for (int b = 0; b < nblk; ++b) {
        if (sizes[b] < 1) return fail(RMH_ERR_PARAM, "set_blocks: empty block");
        end += sizes[b];
    }
    if (end != p->d) return fail(RMH_ERR_PARAM, "set_blocks: the blocks must cover the d coordinates");
    // every block has its own MetropolisRandomWalk (test_changepoint.py:36-41): L block diagonal
    end = 0;
    for (int b = 0; b < nblk; ++b) {
        for (int i = end; i < end + sizes[b]; ++i)
            for (int j = 0; j < end; ++j)
                if (p->L[(size_t)i * p->d + j] != 0.0)
                    return fail(RMH_ERR_PARAM, "set_blocks: L is not block diagonal over the blocks");
        end += sizes[b];
    }
    end = 0;
    for (int b = 0; b < 3; ++b) {
        if (b < nblk) end += sizes[b];
        p->blk_end[b] = end;
        p->blk_p[b] = b < nblk ? pcum[b] : 0.0;
    }
    p->nblk = nblk;
    return RMH_OK;
}

int rmh_proposal_cprj_create(double hscale, const double *pcum, int d, rmh_proposal_t *out)
{
    RMH_NEED_INIT();
    if (!pcum || !out || d < 5 || !(hscale > 0.0)) return fail(RMH_ERR_PARAM, "changepoint proposal: bad argument");
    rmh_proposal *p = new rmh_proposal;
    p->kind = kCpRJ;
    p->d = d;
    p->NT = nt_for_dim(d);
    p->hscale = hscale;
    for (int b = 0; b < 3; ++b) p->blk_p[b] = pcum[b];
    p->dirty = false;
    *out = p;
    return RMH_OK;
}

int rmh_proposal_cprj_set_probs(rmh_proposal_t p, const double *pcum)
{
    if (!p || p->kind != kCpRJ || !pcum) return fail(RMH_ERR_PARAM, "not a trans-dimensional changepoint proposal");
    for (int b = 0; b < 3; ++b) p->blk_p[b] = pcum[b];
    return RMH_OK;
}

int rmh_proposal_pcn_create(const double *L, int d, double rho, rmh_proposal_t *out)
{
    RMH_NEED_INIT();
    if (!(rho > -1.0 && rho < 1.0)) return fail(RMH_ERR_PARAM, "pCN: need |rho| < 1");
    int rc = rmh_proposal_rw_create(L, d, 1.0, out);
    if (rc) return rc;
    (*out)->kind = kPCN;
    (*out)->rho = rho;
    return RMH_OK;
}

int rmh_proposal_hmc_create(double eps, int nsteps, int d, rmh_proposal_t *out)
{
    RMH_NEED_INIT();
    if (d < 1) return fail(RMH_ERR_PARAM, "hmc: d must be >= 1");
    if (nsteps < 1) return fail(RMH_ERR_PARAM, "hmc: Nsteps must be an integer >= 1 (hamiltonian.py:64)");
    rmh_proposal *p = new rmh_proposal;
    p->kind = kHMC;
    p->d = d;
    p->NT = nt_for_dim(d);
    p->eps = eps;
    p->nsteps = nsteps;
    *out = p;
    return RMH_OK;
}

// tile images of chM, V = chM^-1 and V^T for the mass-matrix kernels
} // extern "C"

namespace rmh {
namespace host {
int install_mass(rmh_proposal *p, const double *chM)
{
    if (!chM) {
        (void)hipFree(p->d_chM); (void)hipFree(p->d_V); (void)hipFree(p->d_VT);
        p->d_chM = p->d_V = p->d_VT = nullptr;
        return RMH_OK;
    }
    const int d = p->d, NT = p->NT;
    if (NT < 0) return fail(RMH_ERR_PARAM, "hmc: d > 112 is not supported by the MFMA tile yet");
    if (!lower_ok(chM, d)) return fail(RMH_ERR_PARAM, "mass matrix: chM must be a lower-triangular Cholesky factor");
    std::vector<double> V((size_t)d * d, 0.0); // V = chM^-1 by forward substitution
    for (int col = 0; col < d; ++col)
        for (int i = col; i < d; ++i) {
            double acc = (i == col) ? 1.0 : 0.0;
            for (int j = col; j < i; ++j) acc -= chM[i * d + j] * V[j * d + col];
            V[i * d + col] = acc / chM[i * d + i];
        }
    std::vector<double> a = pack_lower(chM, d, NT, 1.0), b = pack_lower(V.data(), d, NT, 1.0),
                        c = pack_upper_T(V.data(), d, NT);
    // AdaptCovHMC installs a new mass matrix at every covariance update (545 times per 1000 iterations on the
    // reference's schedule): the three images keep their device buffers (no hipFree / hipMalloc per update; measured:
    // no difference -- the 400 us of such an update are the host's Cholesky, inverse and packing)
    int rc;
    if (!p->d_chM || !p->d_V || !p->d_VT) {
        (void)hipFree(p->d_chM); (void)hipFree(p->d_V); (void)hipFree(p->d_VT);
        p->d_chM = p->d_V = p->d_VT = nullptr;
        if ((rc = dev_alloc(&p->d_chM, a.size())) || (rc = dev_alloc(&p->d_V, b.size())) ||
            (rc = dev_alloc(&p->d_VT, c.size())))
            return rc;
    }
    RMH_HIP(hipMemcpyAsync(p->d_chM, a.data(), a.size() * sizeof(double), hipMemcpyHostToDevice, g_stream));
    RMH_HIP(hipMemcpyAsync(p->d_V, b.data(), b.size() * sizeof(double), hipMemcpyHostToDevice, g_stream));
    RMH_HIP(hipMemcpyAsync(p->d_VT, c.data(), c.size() * sizeof(double), hipMemcpyHostToDevice, g_stream));
    RMH_HIP(hipStreamSynchronize(g_stream));   // a, b, c are pageable and go out of scope
    return RMH_OK;
}
} // namespace host
} // namespace rmh

extern "C" {

int rmh_proposal_hmc_set_mass(rmh_proposal_t p, const double *chM)
{
    RMH_NEED_INIT();
    if (!p || p->kind != kHMC) return fail(RMH_ERR_PARAM, "mass matrix: not an HMC proposal");
    if (p->acov) return fail(RMH_ERR_PARAM, "mass matrix: owned by the covariance adaptation");
    p->minv = 1.0;
    return install_mass(p, chM);
}

int rmh_proposal_set_scale(rmh_proposal_t p, double scale)
{
    if (!p) return fail(RMH_ERR_PARAM, "null proposal");
    if (p->scale != scale) {
        p->scale = scale;
        p->dirty = true;
    }
    return RMH_OK;
}

int rmh_proposal_set_eps(rmh_proposal_t p, double eps)
{
    if (!p) return fail(RMH_ERR_PARAM, "null proposal");
    p->eps = eps;
    return RMH_OK;
}

int rmh_proposal_set_adapt(rmh_proposal_t p, double target)
{
    if (!p) return fail(RMH_ERR_PARAM, "null proposal");
    if (!(target >= 0.0 && target < 1.0)) return fail(RMH_ERR_PARAM, "adapt: target accept rate must be in [0, 1)");
    if ((p->adapt_target > 0) != (target > 0)) p->dirty = true;
    p->adapt_target = target;
    return RMH_OK;
}

int rmh_proposal_free(rmh_proposal_t p)
{
    if (!p) return RMH_OK;
    (void)hipFree(p->d_Ltiles);
    (void)hipFree(p->d_Ltail);
    (void)hipFree(p->d_Ldiag);
    (void)hipFree(p->d_chM);
    (void)hipFree(p->d_V);
    (void)hipFree(p->d_VT);
    (void)hipFree(p->d_acov_acc);
    (void)hipFree(p->d_acov_partial);
    (void)hipFree(p->d_acov_C0);
    (void)hipFree(p->d_acov_C);
    (void)hipFree(p->d_acov_L);
    (void)hipFree(p->d_acov_tmp);
    (void)hipFree(p->d_acov_status);
    delete p;
    return RMH_OK;
}

int rmh_propose(rmh_model_t m, rmh_proposal_t p, const double *theta, int64_t N, uint64_t *rng,
                double *theta_p, double *logq)
{
    return rmh_propose_mode(m, p, theta, N, rng, 0, theta_p, logq);
}

int rmh_propose_mode(rmh_model_t m, rmh_proposal_t p, const double *theta, int64_t N, uint64_t *rng, int normal_mode,
                     double *theta_p, double *logq)
{
    RMH_NEED_INIT();
    if (!p || !theta || !rng || !theta_p || !logq || N < 1 || (normal_mode != 0 && normal_mode != 1))
        return fail(RMH_ERR_PARAM, "bad argument");
    if (p->kind == kCpRJ) return fail(RMH_ERR_PARAM, "propose: stand-alone trans-dimensional proposals are not built");
    if (p->kind == kHMC && !m) return fail(RMH_ERR_PARAM, "hmc: needs a device Model for the gradient");
    if (m && m->d != p->d)
        return fail(RMH_ERR_PARAM, "theta and L have incompatible shapes"); // randomwalk.py:24
    // a model-less random walk borrows a dimension-only stand-in
    rmh_model shape;
    shape.d = p->d;
    shape.NT = p->NT;
    TmpState t;
    int rc = tmp_state(t, m ? m : &shape, theta, N);
    if (rc) return rc;
    rmh_state &s = t.s;
    s.nmode = normal_mode;
    std::vector<uint64_t> dev;
    rng_to_device_layout(rng, N, s.ld, dev);
    if ((rc = upload(&s.rng, dev.data(), dev.size()))) return rc;
    double *outp = nullptr, *lq = nullptr, *nd = nullptr;
    if ((rc = dev_alloc(&outp, (size_t)s.dp * s.ld)) || (rc = dev_alloc(&lq, (size_t)s.ld)) ||
        (rc = dev_alloc(&nd, (size_t)N * s.d))) {
        (void)hipFree(outp); (void)hipFree(lq); (void)hipFree(nd);
        return rc;
    }
    ChainArgs c = chain_args(&s);
    hipError_t e = hipMemsetAsync(lq, 0, (size_t)s.ld * sizeof(double), g_stream);
    if (e == hipSuccess) e = hipMemsetAsync(outp, 0, (size_t)s.dp * s.ld * sizeof(double), g_stream);
    if (e == hipSuccess) {
        if (p->kind == kPCN) {
            rc = sync_proposal(p);
            if (!rc) {
                RwArgs ra{p->d_Ltiles, p->d_V, p->rho};
                e = launch_propose_rw(p->NT, ra, c, outp, lq, g_stream); // randomwalk.py:89-101
            }
        } else if (p->kind == kRW && p->nblk > 0) {
            rc = fail(RMH_ERR_PARAM, "propose: stand-alone block-mixture proposals are not built");
        } else if (p->kind == kRW && p->NT > 7) {
            rc = fail(RMH_ERR_PARAM, "propose: stand-alone proposals are built for d <= 112");
        } else if (p->kind == kRW) {
            rc = sync_proposal(p);
            if (!rc) {
                RwArgs ra{p->d_Ltiles, nullptr, 0.0};
                e = launch_propose_rw(p->NT, ra, c, outp, lq, g_stream); // logqratio = 0.0, randomwalk.py:26
            }
        } else {
            if (p->acov) rc = fail(RMH_ERR_PARAM, "propose: stand-alone proposals of an adaptive proposal are not built");
            HmcArgs ha{p->eps, p->nsteps, p->d_chM, p->d_V, p->d_VT, p->minv};
            if (rc) {} else if (m->kind == kGauss) e = launch_gauss_propose_hmc(m->NT, m->ga, ha, c, outp, lq, g_stream);
            else if (m->kind == kLogReg) e = launch_logreg_propose_hmc(m->NT, m->la, ha, c, outp, lq, g_stream);
            else if (m->kind == kMixture) e = launch_mix_propose_hmc(m->NT, m->ma, ha, c, outp, lq, g_stream);
            else rc = fail(RMH_ERR_PARAM, "hmc: this model has no gradient");
        }
    }
    if (!rc && e == hipSuccess) e = launch_transpose_dn_to_nd(outp, nd, N, s.d, s.ld, 1, g_stream);
    if (!rc && e == hipSuccess) e = hipMemcpyAsync(theta_p, nd, (size_t)N * s.d * sizeof(double), hipMemcpyDeviceToHost, g_stream);
    if (!rc && e == hipSuccess) e = hipMemcpyAsync(logq, lq, (size_t)N * sizeof(double), hipMemcpyDeviceToHost, g_stream);
    if (!rc && e == hipSuccess) e = hipMemcpyAsync(dev.data(), s.rng, dev.size() * sizeof(uint64_t), hipMemcpyDeviceToHost, g_stream);
    if (!rc && e == hipSuccess) e = hipStreamSynchronize(g_stream);
    (void)hipFree(outp); (void)hipFree(lq); (void)hipFree(nd);
    if (rc) return rc;
    RMH_HIP(e);
    rng_from_device_layout(dev, N, s.ld, rng);
    return RMH_OK;
}

// -------------------------------------------------------------------- state
int rmh_state_create(rmh_model_t m, int64_t N, uint64_t seed, uint64_t chain0, rmh_state_t *out)
{
    RMH_NEED_INIT();
    if (!m || N < 1) return fail(RMH_ERR_PARAM, "state: need a model and N >= 1 chains");
    // the kernels address a (rows, ld) array through 32-bit byte offsets (raw buffer instructions)
    // (the HMC momentum parking array is (rows, ld + 128) through the same offsets)
    if ((int64_t)(16 * m->NT + 4) * (pad16(N) + 128) * 8 >= ((int64_t)1 << 32))
        return fail(RMH_ERR_PARAM, "state: too many chains for one state at this dimension "
                                   "(rows x chains x 8 bytes must stay below 4 GiB); split the chains "
                                   "over several Samplers");
    rmh_state *s = new rmh_state;
    s->m = m;
    s->N = N;
    s->ld = pad16(N);
    s->d = m->d;
    s->NT = m->NT;
    s->dp = 16 * m->NT;
    s->ks = (m->d + 3) / 4;
    int rc;
    if ((rc = dev_alloc(&s->theta, (size_t)s->dp * s->ld)) || (rc = dev_alloc(&s->logpost, (size_t)s->ld)) ||
        (rc = dev_alloc(&s->rng, (size_t)16 * s->ld))) {
        rmh_state_free(s);
        return rc;
    }
    hipError_t e = hipMemsetAsync(s->theta, 0, (size_t)s->dp * s->ld * sizeof(double), g_stream);
    if (e == hipSuccess) e = hipMemsetAsync(s->logpost, 0, (size_t)s->ld * sizeof(double), g_stream);
    if (e == hipSuccess) e = launch_rng_seed(s->rng, s->ld, seed, chain0, g_stream);
    if (e != hipSuccess) {
        rmh_state_free(s);
        RMH_HIP(e);
    }
    *out = s;
    return RMH_OK;
}

int rmh_state_set_theta(rmh_state_t s, const double *theta)
{
    RMH_NEED_INIT();
    if (!s || !theta) return fail(RMH_ERR_PARAM, "bad argument");
    double *staging = nullptr;
    int rc = upload(&staging, theta, (size_t)s->N * s->d);
    if (rc) return rc;
    hipError_t e = hipMemsetAsync(s->theta, 0, (size_t)s->dp * s->ld * sizeof(double), g_stream);
    if (e == hipSuccess) e = launch_transpose_nd_to_dn(staging, s->theta, s->N, s->d, s->ld, g_stream);
    if (e == hipSuccess) {
        ChainArgs c = chain_args(s);
        rc = eval_states(s->m, c, false, s->logpost, nullptr, nullptr, nullptr); // sampler.py:42
    }
    hipError_t e2 = hipStreamSynchronize(g_stream);
    (void)hipFree(staging);
    if (rc) return rc;
    RMH_HIP(e);
    RMH_HIP(e2);
    s->valid = true;
    s->grad_valid = false;
    return RMH_OK;
}

int rmh_state_get(rmh_state_t s, double *theta, double *logpost)
{
    RMH_NEED_INIT();
    if (!s) return fail(RMH_ERR_PARAM, "null state");
    if (!s->valid) return fail(RMH_ERR_STATE, "state: no current state (empty history)");
    if (theta) {
        double *nd = nullptr;
        int rc = dev_alloc(&nd, (size_t)s->N * s->d);
        if (rc) return rc;
        hipError_t e = launch_transpose_dn_to_nd(s->theta, nd, s->N, s->d, s->ld, 1, g_stream);
        if (e == hipSuccess) e = hipMemcpyAsync(theta, nd, (size_t)s->N * s->d * sizeof(double), hipMemcpyDeviceToHost, g_stream);
        if (e == hipSuccess) e = hipStreamSynchronize(g_stream);
        (void)hipFree(nd);
        RMH_HIP(e);
    }
    if (logpost) {
        RMH_HIP(hipMemcpyAsync(logpost, s->logpost, (size_t)s->N * sizeof(double), hipMemcpyDeviceToHost, g_stream));
        RMH_HIP(hipStreamSynchronize(g_stream));
    }
    return RMH_OK;
}

int rmh_state_get_rng(rmh_state_t s, uint64_t *rng)
{
    RMH_NEED_INIT();
    if (!s || !rng) return fail(RMH_ERR_PARAM, "bad argument");
    std::vector<uint64_t> dev((size_t)16 * s->ld);
    RMH_HIP(hipMemcpyAsync(dev.data(), s->rng, dev.size() * sizeof(uint64_t), hipMemcpyDeviceToHost, g_stream));
    RMH_HIP(hipStreamSynchronize(g_stream));
    rng_from_device_layout(dev, s->N, s->ld, rng);
    return RMH_OK;
}

int rmh_state_set_rng(rmh_state_t s, const uint64_t *rng)
{
    RMH_NEED_INIT();
    if (!s || !rng) return fail(RMH_ERR_PARAM, "bad argument");
    std::vector<uint64_t> dev;
    rng_to_device_layout(rng, s->N, s->ld, dev);
    RMH_HIP(hipMemcpyAsync(s->rng, dev.data(), dev.size() * sizeof(uint64_t), hipMemcpyHostToDevice, g_stream));
    RMH_HIP(hipStreamSynchronize(g_stream));
    return RMH_OK;
}

} // extern "C"

namespace rmh {
namespace host {
int ensure_adapt(rmh_state *s)
{
    if (s->adapt) return RMH_OK;
    // rows: Nsamples, Naccepts, scale (adaptive.py:18-23) and one auxiliary per-chain value
    // (AdaptScalepCN's recursive rho; NaN = not started)
    std::vector<double> init((size_t)4 * s->ld, 0.0);
    for (int64_t c = 0; c < s->ld; ++c) {
        init[2 * s->ld + c] = 1.0; // adaptive.py:23
        init[3 * s->ld + c] = std::nan("");
    }
    return upload(&s->adapt, init.data(), init.size());
}
} // namespace host
} // namespace rmh

extern "C" {

// Sampler(..., normals="f64"): the chains' normals come one per 64-bit stream word (rmh_device.hpp); 0 = the
// default, two per word.  A property of the state: every run / evaluation of these chains uses it.
int rmh_state_set_normal_mode(rmh_state_t s, int mode)
{
    RMH_NEED_INIT();
    if (!s || (mode != 0 && mode != 1)) return fail(RMH_ERR_PARAM, "state_set_normal_mode: mode is 0 or 1");
    if (mode && (s->m->big || s->m->kind == kChangepointRJ))
        return fail(RMH_ERR_PARAM, "normals=\"f64\" is not built for d > 112 or the variable-k changepoint model");
    s->nmode = mode;
    return RMH_OK;
}

int rmh_state_get_normal_mode(rmh_state_t s, int *mode)
{
    if (!s || !mode) return fail(RMH_ERR_PARAM, "state_get_normal_mode: null argument");
    *mode = s->nmode;
    return RMH_OK;
}

int rmh_state_get_adapt(rmh_state_t s, double *adapt)
{
    RMH_NEED_INIT();
    if (!s || !adapt) return fail(RMH_ERR_PARAM, "bad argument");
    int rc = ensure_adapt(s);
    if (rc) return rc;
    std::vector<double> h((size_t)3 * s->ld);
    RMH_HIP(hipMemcpyAsync(h.data(), s->adapt, h.size() * sizeof(double), hipMemcpyDeviceToHost, g_stream));
    RMH_HIP(hipStreamSynchronize(g_stream));
    for (int64_t c = 0; c < s->N; ++c)
        for (int k = 0; k < 3; ++k) adapt[c * 3 + k] = h[(size_t)k * s->ld + c];
    return RMH_OK;
}

int rmh_state_set_adapt(rmh_state_t s, const double *adapt)
{
    RMH_NEED_INIT();
    if (!s || !adapt) return fail(RMH_ERR_PARAM, "bad argument");
    int rc = ensure_adapt(s);
    if (rc) return rc;
    std::vector<double> h((size_t)3 * s->ld, 0.0);
    for (int64_t c = 0; c < s->ld; ++c) h[2 * s->ld + c] = 1.0;
    for (int64_t c = 0; c < s->N; ++c)
        for (int k = 0; k < 3; ++k) h[(size_t)k * s->ld + c] = adapt[c * 3 + k];
    RMH_HIP(hipMemcpyAsync(s->adapt, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice, g_stream));
    RMH_HIP(hipStreamSynchronize(g_stream));
    return RMH_OK;
}

int rmh_state_get_adapt_aux(rmh_state_t s, double *aux)
{
    RMH_NEED_INIT();
    if (!s || !aux) return fail(RMH_ERR_PARAM, "bad argument");
    int rc = ensure_adapt(s);
    if (rc) return rc;
    RMH_HIP(hipMemcpyAsync(aux, s->adapt + 3 * s->ld, (size_t)s->N * sizeof(double), hipMemcpyDeviceToHost, g_stream));
    RMH_HIP(hipStreamSynchronize(g_stream));
    return RMH_OK;
}

int rmh_state_set_adapt_aux(rmh_state_t s, const double *aux)
{
    RMH_NEED_INIT();
    if (!s || !aux) return fail(RMH_ERR_PARAM, "bad argument");
    int rc = ensure_adapt(s);
    if (rc) return rc;
    RMH_HIP(hipMemcpyAsync(s->adapt + 3 * s->ld, aux, (size_t)s->N * sizeof(double), hipMemcpyHostToDevice, g_stream));
    RMH_HIP(hipStreamSynchronize(g_stream));
    return RMH_OK;
}

int rmh_state_device_ptrs(rmh_state_t s, void **theta_dn, void **logpost, int64_t *ld)
{
    if (!s) return fail(RMH_ERR_PARAM, "null state");
    if (theta_dn) *theta_dn = s->theta;
    if (logpost) *logpost = s->logpost;
    if (ld) *ld = s->ld;
    return RMH_OK;
}

int rmh_state_free(rmh_state_t s)
{
    if (!s) return RMH_OK;
    (void)hipFree(s->theta);
    (void)hipFree(s->grad);
    (void)hipFree(s->mom);
    (void)hipFree(s->adapt);
    (void)hipFree(s->pt_beta);
    (void)hipFree(s->sc_theta);
    (void)hipFree(s->sc_logpost);
    (void)hipFree(s->sc_accept);
    (void)hipFree(s->last_theta);
    (void)hipFree(s->last_logpost);
    (void)hipFree(s->logpost);
    (void)hipFree(s->rng);
    delete s;
    return RMH_OK;
}

// ------------------------------------------------------------------ history
int rmh_history_create(rmh_state_t s, int64_t entries, int64_t iters, rmh_history_t *out)
{
    RMH_NEED_INIT();
    if (!s || entries < 0 || iters < 0) return fail(RMH_ERR_PARAM, "bad argument");
    rmh_history *h = new rmh_history;
    h->s = s;
    h->entries = entries;
    h->iters = iters;
    int rc;
    if ((rc = dev_alloc(&h->thetas, (size_t)entries * 4 * s->ks * s->ld)) ||
        (rc = dev_alloc(&h->logpost, (size_t)entries * s->ld)) ||
        (rc = dev_alloc(&h->accept, (size_t)iters * s->ld))) {
        rmh_history_free(h);
        return rc;
    }
    if (entries > 0) // rows d..4*ks-1 of an entry are never written by the lane kernels
        RMH_HIP(hipMemsetAsync(h->thetas, 0, (size_t)entries * 4 * s->ks * s->ld * sizeof(double), g_stream));
    *out = h;
    return RMH_OK;
}

int rmh_history_entries(rmh_history_t h, int64_t *entries, int64_t *iters)
{
    if (!h) return fail(RMH_ERR_PARAM, "null history");
    if (entries) *entries = h->entries;
    if (iters) *iters = h->iters;
    return RMH_OK;
}

// Pinned host memory for history downloads: into pinned memory the chunks below move by DMA at
// the PCIe rate while the next chunk is transposed; into pageable memory every chunk is staged
// by the runtime (about 20 GB/s measured).
int rmh_host_alloc(void **ptr, uint64_t bytes)
{
    RMH_NEED_INIT();
    if (!ptr || bytes == 0) return fail(RMH_ERR_PARAM, "host_alloc: bad argument");
    RMH_HIP(hipHostMalloc(ptr, (size_t)bytes, hipHostMallocDefault));
    return RMH_OK;
}

int rmh_host_free(void *ptr)
{
    if (ptr) RMH_HIP(hipHostFree(ptr));
    return RMH_OK;
}

// Entries [e0, e1) as (n, N, d) row-major: transposed on the device chunk by chunk (<= 256 MiB)
// into one of TWO staging buffers on the library's stream while the previous chunk travels to the
// host on a copy stream (events both ways), so transposes and transfers overlap.
int rmh_history_download(rmh_history_t h, int64_t e0, int64_t e1, double *thetas, double *logpost)
{
    RMH_NEED_INIT();
    if (!h || e0 < 0 || e1 > h->entries || e0 > e1) return fail(RMH_ERR_PARAM, "history: bad entry range");
    const rmh_state *s = h->s;
    const int64_t rows = 4 * s->ks, per = s->N * s->d;
    if (thetas && e1 > e0) {
        int64_t chunk = (256LL << 20) / (per * 8);
        if (chunk < 1) chunk = 1;
        if (chunk > e1 - e0) chunk = e1 - e0;
        // the two staging buffers are kept by the library (hipMalloc / hipFree of 2 x 256 MiB per
        // call cost as much as moving several GB)
        static double *stage[2] = {nullptr, nullptr};
        static size_t stage_n = 0;
        if (stage_n < (size_t)chunk * per) {
            (void)hipFree(stage[0]);
            (void)hipFree(stage[1]);
            stage[0] = stage[1] = nullptr;
            stage_n = 0;
            int rc = dev_alloc(&stage[0], (size_t)chunk * per);
            if (!rc) rc = dev_alloc(&stage[1], (size_t)chunk * per);
            if (rc) {
                (void)hipFree(stage[0]);
                stage[0] = nullptr;
                return rc;
            }
            stage_n = (size_t)chunk * per;
        }
        double *nd[2] = {stage[0], stage[1]};
        hipError_t e = hipSuccess;
        int k = 0;
        for (int64_t a = e0; a < e1 && e == hipSuccess; a += chunk, ++k) {
            const int64_t n = (e1 - a < chunk) ? e1 - a : chunk;
            const int b = k & 1;
            if (k >= 2) e = hipStreamWaitEvent(g_stream, g_cp_done[b], 0);   // the buffer's previous copy has left
            // entries are 4*ks rows apart; the transpose kernel strides by d rows, so go one
            // entry at a time when the two differ
            if (e != hipSuccess) break;
            if (rows == s->d) {
                e = launch_transpose_dn_to_nd(h->thetas + a * rows * s->ld, nd[b], s->N, s->d, s->ld, n, g_stream);
            } else {
                for (int64_t q = 0; q < n && e == hipSuccess; ++q)
                    e = launch_transpose_dn_to_nd(h->thetas + (a + q) * rows * s->ld, nd[b] + q * per,
                                                  s->N, s->d, s->ld, 1, g_stream);
            }
            if (e == hipSuccess) e = hipEventRecord(g_cp_ready[b], g_stream);
            if (e == hipSuccess) e = hipStreamWaitEvent(g_copy, g_cp_ready[b], 0);
            if (e == hipSuccess)
                e = hipMemcpyAsync(thetas + (a - e0) * per, nd[b], (size_t)n * per * sizeof(double),
                                   hipMemcpyDeviceToHost, g_copy);
            if (e == hipSuccess) e = hipEventRecord(g_cp_done[b], g_copy);
        }
        hipError_t e2 = hipStreamSynchronize(g_copy);
        hipError_t e3 = hipStreamSynchronize(g_stream);
        RMH_HIP(e);
        RMH_HIP(e2);
        RMH_HIP(e3);
    }
    if (logpost && e1 > e0) {
        RMH_HIP(hipMemcpy2DAsync(logpost, (size_t)s->N * sizeof(double), h->logpost + e0 * s->ld,
                                 (size_t)s->ld * sizeof(double), (size_t)s->N * sizeof(double),
                                 (size_t)(e1 - e0), hipMemcpyDeviceToHost, g_stream));
        RMH_HIP(hipStreamSynchronize(g_stream));
    }
    return RMH_OK;
}

// Diagnostics of a recorded history without bringing it to the host (at the cfg2 size the history
// is 53 GB, its reductions are 105 MB): per chain and parameter the mean and the (biased)
// variance over entries [e0, e1), and per parameter the autocorrelation function averaged over
// the chains, lags 0 .. nlags-1 (nlags a multiple of 32, at most 320) -- the inputs of
// integrated_time / ESS (riemann_amd/diagnostics.py; the reference prints
// emcee.autocorr.integrated_time of one chain, examples/test_randomwalk.py:42).
// mean_dn, var_dn: (d, N) parameter-major; acf: (nlags, d); any of them may be null.
static int history_reduce_impl(rmh_history_t h, int64_t e0, int64_t e1, int nlags, double *mean_dn, double *var_dn,
                               double *acf, int64_t *acf_sums)
{
    RMH_NEED_INIT();
    if (!h || e0 < 0 || e1 > h->entries || e1 - e0 < 2) return fail(RMH_ERR_PARAM, "history_reduce: bad entry range");
    if (nlags < 32 || nlags % 32 || nlags > 320) return fail(RMH_ERR_PARAM, "history_reduce: nlags must be 32, 64, ... 320");
    const rmh_state *s = h->s;
    const int64_t N = s->N, nblk = (N + 63) / 64;
    const int d = s->d, ng = nlags / 32;
    double *dm = nullptr, *dv = nullptr;
    long long *dp = nullptr, *da = nullptr;
    int rc;
    if ((rc = dev_alloc(&dm, (size_t)d * N)) || (rc = dev_alloc(&dv, (size_t)d * N)) ||
        (rc = dev_alloc(&dp, (size_t)nblk * d * nlags)) || (rc = dev_alloc(&da, (size_t)nlags * d))) {
        (void)hipFree(dm); (void)hipFree(dv); (void)hipFree(dp); (void)hipFree(da);
        return rc;
    }
    std::vector<long long> sums((size_t)nlags * d);
    hipError_t e = launch_hist_reduce(h->thetas, e0, e1, 4 * s->ks, s->ld, N, d, ng, dm, dv, dp, da, g_stream);
    if (e == hipSuccess && mean_dn) e = hipMemcpyAsync(mean_dn, dm, (size_t)d * N * sizeof(double), hipMemcpyDeviceToHost, g_stream);
    if (e == hipSuccess && var_dn) e = hipMemcpyAsync(var_dn, dv, (size_t)d * N * sizeof(double), hipMemcpyDeviceToHost, g_stream);
    if (e == hipSuccess && (acf || acf_sums))
        e = hipMemcpyAsync(sums.data(), da, sums.size() * sizeof(long long), hipMemcpyDeviceToHost, g_stream);
    if (e == hipSuccess) e = hipStreamSynchronize(g_stream);
    (void)hipFree(dm); (void)hipFree(dv); (void)hipFree(dp); (void)hipFree(da);
    RMH_HIP(e);
    // the chain average of the integer sums (RMH_ACF_SCALE): one rounding, whatever the chain count
    const double den = (double)N * RMH_ACF_SCALE;
    for (size_t i = 0; i < sums.size(); ++i) {
        if (acf) acf[i] = (double)sums[i] / den;
        if (acf_sums) acf_sums[i] = (int64_t)sums[i];
    }
    return RMH_OK;
}

int rmh_history_reduce(rmh_history_t h, int64_t e0, int64_t e1, int nlags, double *mean_dn, double *var_dn,
                       double *acf)
{
    return history_reduce_impl(h, e0, e1, nlags, mean_dn, var_dn, acf, nullptr);
}

int rmh_history_reduce_sums(rmh_history_t h, int64_t e0, int64_t e1, int nlags, double *mean_dn, double *var_dn,
                            int64_t *acf_sums)
{
    return history_reduce_impl(h, e0, e1, nlags, mean_dn, var_dn, nullptr, acf_sums);
}

// [count, sum theta (d), sum theta theta^T (d, d)] over the chains and entries [e0, e1) of a recorded
// history, on the matrix cores (k_moments: the pooled moment kernel of the covariance adaptation)
int rmh_history_moments(rmh_history_t h, int64_t e0, int64_t e1, double *moments)
{
    RMH_NEED_INIT();
    if (!h || !moments || e0 < 0 || e1 > h->entries || e1 <= e0)
        return fail(RMH_ERR_PARAM, "history_moments: bad entry range");
    const rmh_state *s = h->s;
    const int d = s->d, D1 = d + 1, NT = nt_for_dim(D1);
    if (NT < 0) return fail(RMH_ERR_PARAM, "history_moments: d > 111 is not supported by the moment kernel");
    const int nwg = 512, ntri = NT * (NT + 1) / 2;
    double *part = nullptr, *acc = nullptr;
    int rc;
    if ((rc = dev_alloc(&part, (size_t)nwg * ntri * 256)) || (rc = dev_alloc(&acc, (size_t)ntri * 256))) {
        (void)hipFree(part);
        return rc;
    }
    std::vector<double> hacc((size_t)ntri * 256);
    hipError_t e = hipMemsetAsync(acc, 0, hacc.size() * sizeof(double), g_stream);
    if (e == hipSuccess)
        e = launch_moments(NT, h->thetas, e0, e1, 4 * s->ks, s->ld, s->N, d, part, nwg, acc, g_stream);
    if (e == hipSuccess) e = hipMemcpyAsync(hacc.data(), acc, hacc.size() * sizeof(double), hipMemcpyDeviceToHost, g_stream);
    if (e == hipSuccess) e = hipStreamSynchronize(g_stream);
    (void)hipFree(part); (void)hipFree(acc);
    RMH_HIP(e);
    auto at = [&](int row, int col) {   // tile-pair layout of k_moments, lower triangle
        if (row < col) std::swap(row, col);
        const int i = row >> 4, j = col >> 4, t = i * (i + 1) / 2 + j, rr = row & 15;
        return hacc[(size_t)t * 256 + (rr >> 2) * 64 + (rr & 3) * 16 + (col & 15)];
    };
    moments[0] = at(d, d);
    for (int p = 0; p < d; ++p) moments[1 + p] = at(p, d);
    for (int p = 0; p < d; ++p)
        for (int q = 0; q < d; ++q) moments[1 + d + (size_t)p * d + q] = at(p, q);
    return RMH_OK;
}

int rmh_history_accept(rmh_history_t h, int64_t t0, int64_t t1, uint8_t *accept)
{
    RMH_NEED_INIT();
    if (!h || !accept || t0 < 0 || t1 > h->iters || t0 > t1) return fail(RMH_ERR_PARAM, "history: bad iteration range");
    if (t1 == t0) return RMH_OK;
    const rmh_state *s = h->s;
    RMH_HIP(hipMemcpy2DAsync(accept, (size_t)s->N, h->accept + t0 * s->ld, (size_t)s->ld, (size_t)s->N,
                             (size_t)(t1 - t0), hipMemcpyDeviceToHost, g_stream));
    RMH_HIP(hipStreamSynchronize(g_stream));
    return RMH_OK;
}

int rmh_history_device_ptrs(rmh_history_t h, void **thetas_edn, void **logpost_en, void **accept_tn, int64_t *ld)
{
    if (!h) return fail(RMH_ERR_PARAM, "null history");
    if (thetas_edn) *thetas_edn = h->thetas;
    if (logpost_en) *logpost_en = h->logpost;
    if (accept_tn) *accept_tn = h->accept;
    if (ld) *ld = h->s->ld;
    return RMH_OK;
}

int rmh_history_free(rmh_history_t h)
{
    if (!h) return RMH_OK;
    (void)hipFree(h->thetas);
    (void)hipFree(h->logpost);
    (void)hipFree(h->accept);
    delete h;
    return RMH_OK;
}

// ---------------------------------------------------------------------- run
} // extern "C"

extern "C" {

// The stream of a chain for callers that run part of Sampler.sample on the host (a Python Model
// or Proposal plugin next to device ones, samplers/sampler.py of this package): T iterations'
// worth of draws -- d normals (randomwalk.py:25 / hamiltonian.py:79), then the accept uniform
// (sampler.py:84) -- exactly as the fused kernels consume them.  d = 0: the uniforms only.
int rmh_rng_draw(uint64_t *rng, int64_t N, int d, int T, double *xi, double *u)
{
    return rmh_rng_draw_mode(rng, N, d, T, 0, xi, u);
}

int rmh_rng_draw_mode(uint64_t *rng, int64_t N, int d, int T, int normal_mode, double *xi, double *u)
{
    RMH_NEED_INIT();
    if (!rng || N < 1 || d < 0 || T < 1 || (d > 0 && !xi) || (d == 0 && !u) || (normal_mode != 0 && normal_mode != 1))
        return fail(RMH_ERR_PARAM, "rng_draw: bad argument");
    const int64_t ld = pad16(N);
    std::vector<uint64_t> dev;
    rng_to_device_layout(rng, N, ld, dev);
    uint64_t *drng = nullptr;
    double *dxi = nullptr, *du = nullptr;
    int rc;
    if ((rc = upload(&drng, dev.data(), dev.size())) || (rc = dev_alloc(&dxi, (size_t)T * ld * (d > 0 ? d : 1))) ||
        (u && (rc = dev_alloc(&du, (size_t)T * ld)))) {
        (void)hipFree(drng); (void)hipFree(dxi); (void)hipFree(du);
        return rc;
    }
    hipError_t e = launch_rng_stream(drng, ld, d, T, normal_mode ? g_icdf64 : g_icdf, normal_mode, dxi, du, g_stream);
    if (e == hipSuccess && d > 0)
        e = hipMemcpy2DAsync(xi, (size_t)N * d * 8, dxi, (size_t)ld * d * 8, (size_t)N * d * 8, (size_t)T,
                             hipMemcpyDeviceToHost, g_stream);
    if (e == hipSuccess && u)
        e = hipMemcpy2DAsync(u, (size_t)N * 8, du, (size_t)ld * 8, (size_t)N * 8, (size_t)T,
                             hipMemcpyDeviceToHost, g_stream);
    if (e == hipSuccess) e = hipMemcpyAsync(dev.data(), drng, dev.size() * sizeof(uint64_t), hipMemcpyDeviceToHost, g_stream);
    if (e == hipSuccess) e = hipStreamSynchronize(g_stream);
    (void)hipFree(drng); (void)hipFree(dxi); (void)hipFree(du);
    RMH_HIP(e);
    rng_from_device_layout(dev, N, ld, rng);
    return RMH_OK;
}

// --------------------------------------------------------------- test hooks
int rmh_debug_rng_stream(uint64_t seed, uint64_t chain0, int64_t N, int d, int T, double *xi, double *u)
{
    return rmh_debug_rng_stream_mode(seed, chain0, N, d, T, 0, xi, u);
}

int rmh_debug_rng_stream_mode(uint64_t seed, uint64_t chain0, int64_t N, int d, int T, int normal_mode, double *xi,
                              double *u)
{
    RMH_NEED_INIT();
    if (N < 1 || d < 1 || T < 1 || (normal_mode != 0 && normal_mode != 1)) return fail(RMH_ERR_PARAM, "bad argument");
    const int64_t ld = pad16(N);
    uint64_t *rng = nullptr;
    double *dxi = nullptr, *du = nullptr;
    int rc;
    if ((rc = dev_alloc(&rng, (size_t)16 * ld)) || (rc = dev_alloc(&dxi, (size_t)T * ld * d)) ||
        (rc = dev_alloc(&du, (size_t)T * ld))) {
        (void)hipFree(rng); (void)hipFree(dxi); (void)hipFree(du);
        return rc;
    }
    hipError_t e = launch_rng_seed(rng, ld, seed, chain0, g_stream);
    if (e == hipSuccess)
        e = launch_rng_stream(rng, ld, d, T, normal_mode ? g_icdf64 : g_icdf, normal_mode, dxi, du, g_stream);
    if (e == hipSuccess)
        e = hipMemcpy2DAsync(xi, (size_t)N * d * 8, dxi, (size_t)ld * d * 8, (size_t)N * d * 8, (size_t)T,
                             hipMemcpyDeviceToHost, g_stream);
    if (e == hipSuccess)
        e = hipMemcpy2DAsync(u, (size_t)N * 8, du, (size_t)ld * 8, (size_t)N * 8, (size_t)T,
                             hipMemcpyDeviceToHost, g_stream);
    if (e == hipSuccess) e = hipStreamSynchronize(g_stream);
    (void)hipFree(rng); (void)hipFree(dxi); (void)hipFree(du);
    RMH_HIP(e);
    return RMH_OK;
}

int rmh_debug_set_kernel_policy(int mask, int *old)
{
    if (old) *old = rmh::g_kernel_policy;
    rmh::g_kernel_policy = mask;
    return RMH_OK;
}

int rmh_debug_mfma(const double *A, const double *B, const double *C, double *D)
{
    RMH_NEED_INIT();
    double *dA = nullptr, *dB = nullptr, *dC = nullptr, *dD = nullptr;
    int rc;
    if ((rc = upload(&dA, A, 64)) || (rc = upload(&dB, B, 64)) || (rc = upload(&dC, C, 256)) ||
        (rc = dev_alloc(&dD, 256))) {
        (void)hipFree(dA); (void)hipFree(dB); (void)hipFree(dC); (void)hipFree(dD);
        return rc;
    }
    hipError_t e = launch_debug_mfma(dA, dB, dC, dD, g_stream);
    if (e == hipSuccess) e = hipMemcpyAsync(D, dD, 256 * sizeof(double), hipMemcpyDeviceToHost, g_stream);
    if (e == hipSuccess) e = hipStreamSynchronize(g_stream);
    (void)hipFree(dA); (void)hipFree(dB); (void)hipFree(dC); (void)hipFree(dD);
    RMH_HIP(e);
    return RMH_OK;
}

} // extern "C"
