// rmh_host.hpp -- host-side declarations shared by the C-ABI translation units
// (rmh_api.hip: context, models, proposals, chain state, history, propose;
//  rmh_run_api.hip: Sampler.run / PTSampler.run and the covariance adaptation).
#pragma once
#include "../../include/riemann_hip.h"
#include "rmh_internal.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

// ------------------------------------------------------------------ handles
// (global-scope types of the C ABI; their members are kernel argument blocks of namespace rmh)
struct rmh_model {
    int kind = 0, d = 0, NT = 0;
    bool big = false; // 112 < d <= 1024: rmh_kernels_big.hip (Gaussian, value only)
    // gaussian
    double *d_W = nullptr, *d_WT = nullptr, *d_mu = nullptr, *d_Wtail = nullptr;
    rmh::GaussArgs ga{};
    // logreg
    double *d_X = nullptr;
    rmh::LogRegArgs la{};
    // changepoint
    int k = 0;
    double *d_xy = nullptr, *d_ssxp = nullptr;
    rmh::CpArgs ca{};
    // changepoint with a variable number of changepoints (capacity k): table by pieces
    double *d_ktab = nullptr;
    rmh::CpRjArgs rj{};
    // mixture of gaussians
    double *d_mixWT = nullptr;
    double *d_mixW = nullptr, *d_mixmu = nullptr, *d_mixc = nullptr, *d_mixw = nullptr;
    rmh::MixArgs ma{};
};

struct rmh_proposal {
    int kind = 0, d = 0, NT = 0;
    double scale = 1.0, eps = 0.0, adapt_target = 0.0, rho = 0.0;
    int nsteps = 0;
    std::vector<double> L;     // host copy of the Cholesky factor
    double *d_Ltiles = nullptr;
    double *d_Ltail = nullptr;  // rows 96.. of scale*L for k_run_rw_tail (d = 97..100), or null
    double *d_Ldiag = nullptr;  // diagonal of scale*L when L is diagonal (k_cp_rw_ss)
    bool L_is_diag = false;
    bool dirty = true;
    double *d_chM = nullptr, *d_V = nullptr, *d_VT = nullptr; // mass-matrix tile images (HMC)
    double minv = 1.0;                                        // M^-1 = minv (chM chM^T)^-1
    double hscale = 0.0;                                      // ChangepointRegression1DProp (kCpRJ)
    int nblk = 0, blk_end[3] = {0, 0, 0};                     // block-mixture random walk
    double blk_p[3] = {0.0, 0.0, 0.0};
    // AdaptCovProposal (proposals/adaptive.py:38-103), pooled over the chains
    bool acov = false, acov_marg = false, acov_smooth = false, acov_squares = false;
    double acov_t_adapt = 1.0, acov_n = 0.0;   // n: adapt() calls per chain so far
    int acov_NT = 0;                            // tiles of the (d+1)-row moment matrix
    std::vector<double> acov_C0, acov_C;        // initial and current covariance (host)
    double *d_acov_acc = nullptr, *d_acov_partial = nullptr; // moment sums, per-workgroup partials
    size_t acov_partial_cap = 0;                             // doubles allocated behind d_acov_partial
    // device-side update (random-walk variant): C0, the current C and L, a scratch copy of the
    // sums for the all-reduce, the not-positive-definite flag; the host copies acov_C / L are
    // refreshed on demand (acov_host_stale)
    double *d_acov_C0 = nullptr, *d_acov_C = nullptr, *d_acov_L = nullptr, *d_acov_tmp = nullptr;
    int *d_acov_status = nullptr;
    bool acov_host_stale = false;
};

struct rmh_state {
    rmh_model *m = nullptr;
    int64_t N = 0, ld = 0;
    int d = 0, dp = 0, ks = 0, NT = 0;
    double *theta = nullptr, *grad = nullptr, *mom = nullptr, *logpost = nullptr;
    double *adapt = nullptr; // 3 x ld: Nsamples, Naccepts, scale (AdaptScaleProposal)
    double *pt_beta = nullptr; // ld: inverse temperature of every slot (rmh_run_pt)
    // scratch history of one AdaptCov segment (every state of <= kAcovChunk iterations)
    double *sc_theta = nullptr, *sc_logpost = nullptr;
    uint8_t *sc_accept = nullptr;
    // one-entry sink for the last kept state of a run without a history (rmh_run_api.hip)
    double *last_theta = nullptr, *last_logpost = nullptr;
    uint64_t *rng = nullptr;
    bool valid = false, grad_valid = false;
    int nmode = 0; // 1: the 64-bit-resolution normal mode (rmh_state_set_normal_mode)
};

struct rmh_history {
    rmh_state *s = nullptr;
    int64_t entries = 0, iters = 0;
    double *thetas = nullptr, *logpost = nullptr;
    uint8_t *accept = nullptr;
};

namespace rmh {
namespace host {

// ------------------------------------------------------------------ context (rmh_api.hip)
extern thread_local std::string g_err;
extern bool g_init;
extern int g_device;
extern hipStream_t g_stream;
extern hipEvent_t g_ev0, g_ev1;
extern hipStream_t g_copy;
extern hipEvent_t g_cp_ready[2], g_cp_done[2];
extern int g_last_launches;
extern bool g_timed;
extern double *g_icdf;   // device copy of the inverse-CDF table in its LDS layout (rmh_device.hpp)
extern double *g_icdf64; // the 64-bit normal mode's table, likewise
extern double *g_logi; // device copy of the logistic link-function table

int fail(int code, const std::string &msg);

#define RMH_HIP(expr)                                                                    \
    do {                                                                                 \
        hipError_t e_ = (expr);                                                          \
        if (e_ != hipSuccess)                                                            \
            return fail(RMH_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

#define RMH_NEED_INIT()                                                                  \
    do {                                                                                 \
        if (!g_init) {                                                                   \
            int rc_ = ::rmh_init(0);                                                       \
            if (rc_) return rc_;                                                         \
        }                                                                                \
    } while (0)

template <class T>
int dev_alloc(T **p, size_t n)
{
    *p = nullptr;
    if (n == 0) return RMH_OK;
    RMH_HIP(hipMalloc(reinterpret_cast<void **>(p), n * sizeof(T)));
    return RMH_OK;
}

template <class T>
int upload(T **p, const T *h, size_t n)
{
    int rc = dev_alloc(p, n);
    if (rc) return rc;
    RMH_HIP(hipMemcpy(*p, h, n * sizeof(T), hipMemcpyHostToDevice));
    return RMH_OK;
}


int64_t pad16(int64_t n);
std::vector<double> pack_lower(const double *M, int d, int NT, double scale);
std::vector<double> pack_upper_T(const double *M, int d, int NT);
bool lower_ok(const double *L, int d);
// d = 16 NF + 1 .. 16 NF + 4: rows 16 NF .. of a lower-triangular (d, d) matrix in the layout of TailAcc (rmh_run.hpp)
std::vector<double> pack_tail(const double *M, int d, double scale);
// the last row tile holds at most four rows: they run on the VALU (k_run_rw_tail), d = 17..20, 33..36, ... 97..100
inline bool has_tail(int d) { return d > 16 && d <= 100 && (d - 1) % 16 < 4; }
bool cholesky_lower(const std::vector<double> &Cm, int d, std::vector<double> &L);

ChainArgs chain_args(const rmh_state *s);
int sync_proposal(rmh_proposal *p);
int eval_states(rmh_model *m, const ChainArgs &c, bool want_grad, double *lp, double *grad, double *lprior,
                double *llik);
int ensure_grad(rmh_state *s);
int ensure_adapt(rmh_state *s);
int install_mass(rmh_proposal *p, const double *chM); // tile images of chM, chM^-1, chM^-T

} // namespace host
} // namespace rmh
