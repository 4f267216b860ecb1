// rmh_internal.hpp -- structures shared by the kernel files and the C-ABI layer.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/riemann_hip.h"   // RMH_POLICY_*
#include <stdint.h>
#include <vector>

namespace rmh {

enum ModelKind { kGauss = 1, kLogReg = 2, kChangepoint = 3, kMixture = 4, kChangepointRJ = 5 };
enum ProposalKind { kRW = 1, kHMC = 2, kPCN = 3, kCpRJ = 4 };

// rmh_debug_set_kernel_policy (include/riemann_hip.h, RMH_POLICY_*): tests and tools switch
// specialised kernels off or on through it; the launch path reads no environment variable
extern int g_kernel_policy;
inline bool policy(int bit) { return (g_kernel_policy & bit) != 0; }

// ---- device-side argument blocks (passed by value to the kernels) ----------
struct GaussArgs {
    const double *Wtiles;  // lower-triangular tile image of W = L^-1
    const double *WTtiles; // upper-triangular tile image of W^T (gradient)
    const double *mu_pad;  // 16*NT doubles, zero padded
    double dlog2pi;        // d*log(2 pi)
    double logdetC;
    int ks;                // ceil(d/4): k-steps that carry data
    // rows 96.. of W when the last 16-row tile holds at most 4 rows (d = 97..100), laid out
    // [s][g][r] = W[96 + r][4s + g]: see k_run_rw_tail (rmh_run.hpp); null otherwise
    const double *Wtail;
};

// benchmarks.MixtureDist of MultiGaussianDist components (models/benchmarks.py:31-62)
struct MixArgs {
    const double *Wtiles;  // ncomp lower-triangular tile images of W_k = L_k^-1 (global memory, L2)
    const double *WTtiles; // ncomp upper-triangular tile images of W_k^T (gradient)
    const double *mu_pad;  // ncomp x 16*NT doubles
    const double *cst;     // ncomp: logdetC_k
    const double *wts;     // ncomp normalised weights
    int ncomp;
    int ks;
    double dlog2pi;
    double logP0;          // the constant log_prior, log(sum(w exp(0)))
};

struct LogRegArgs {
    const double *Ximg;    // stage images, see LogRegModel (rmh_models.hpp)
    int64_t nchunks;       // stages of 2 row tiles (32 data rows)
    int ks;                // ceil(d/4): k-steps of the eta contraction
    const double *logi;    // link-function table (rmh_device.hpp, logistic_lockstep)
    double inv_tau2;
    double two_tau2;
};

struct CpArgs {
    const double *xy;      // x[n], y[n]
    int n;
    double xmin, xmax, alpha_m1, beta;
    double c_k;            // k' log(lamb) - lgamma(k') - lamb          (k' = pieces)
    double c_v;            // per piece: alpha log(beta)                 (added per element)
    double lgam_alpha;     // lgamma(alpha)                              (subtracted per element)
    double c_s;            // lgamma(2k'+1)
    double c_sL;           // k' log(xmax-xmin)
    double nlog2pi;        // n log(2 pi)
    // sufficient statistics of the likelihood (k_cp_rw_ss, rmh_kernels_cp.hip): x sorted and
    // padded with +inf to 2^ss_steps - 1 entries, then the prefix sums P1[j] = sum_{i<j} (y_i - ym)
    // of the y in that order (2^ss_steps entries); null when n is too large for the LDS
    const double *ss_xp;
    int ss_steps;          // steps of the branch-free upper-bound search: floor(log2 n) + 1
    double ss_ym, ss_p2;   // mean of y, sum (y_i - ym)^2
    // bucket table in front of the searches (k_cp_rw_ss64): ss_nb uniform buckets over the data range,
    // bucket(c) = (int)min(max((c - ss_x0) ss_invh, 0), ss_nb - 1) -- a monotone map evaluated with the
    // same IEEE operations on the host (for the data) and on the device (for the key), so that
    // #{x <= c} lies in [L[b], L[b+1]], L[b] = #{i : bucket(x_i) < b}.  The ss_nb + 1 entries of L
    // (times 8: byte offsets into xs) follow the prefix sums in ss_xp as int32, from double
    // 2 np + 2 on; ss_s2 = search steps left inside a bucket (2 or 3); ss_nb = 0: no table.
    int ss_nb, ss_s2;
    double ss_x0, ss_invh;
};

// Trans-dimensional changepoint sampling: model with a variable number
// k <= KM of changepoints, state = padded vector [cpx (KM, +inf beyond k), cpv (KM+1, 1.0
// beyond k+1), sig, k], and the proposal of riemann/test_changepoint.py:43-73.
struct CpRjArgs {
    CpArgs base;           // data and the k-independent constants (c_k, c_s, c_sL unused)
    const double *ktab;    // (KM+2) x 4 by pieces k' = k+1: c_k, lgamma(2k'+1), k' log(xmax-xmin), sqrt(C_cpx(k))
    int KM;                // capacity = the model's kmax
    double sv, ss;         // sqrt(C_cpv), sqrt(C_sig) (test_changepoint.py:36-41), set by the proposal
    double pcum[3];        // P_cumprop_cpx, _cpv, _sig
};

struct ChainArgs {
    double *theta;      // (dp, ld), rows >= d are zero
    double *logpost;    // (ld)
    double *grad;       // (dp, ld) cached gradient of the log posterior (HMC), or null
    double *mom;        // (dp, ld + 128) momentum parking space (HMC), or null
    uint64_t *rng;      // 4 words x nslots
    int64_t ld;         // padded chain count (multiple of 16)
    int64_t nslots;     // 4*ld
    int64_t nblocks;    // ld/16
    int d;
    int dp;             // rows of theta/grad: 16*NT (MFMA kernels)
    int ks;             // ceil(d/4): history entries hold 4*ks rows
    int M;              // 64-bit words per sub-stream per iteration: ceil(d/8), two normals per word
    const double *icdf; // inverse-normal-CDF table in its LDS layout (rmh_device.hpp): the 32-bit one, or the 64-bit mode's
    int nmode;          // 0: two normals per stream word; 1: one normal per word (Sampler(..., normals="f64"))
    double *adapt;      // AdaptScaleProposal state, 3 rows of ld: Nsamples, Naccepts, scale; or null
    double adapt_target;
};

struct RunArgs {
    int64_t T, nburn, nthin;
    double *h_theta;    // (entries, 4*ks, ld) or null
    double *h_logpost;  // (entries, ld)
    uint8_t *h_accept;  // (T, ld) or null
    // parallel tempering (PTSampler, samplers/ptsampler.py:42-127); pt_K = 0: off
    const double *pt_beta; // (ld) inverse temperature of every slot
    int pt_K;              // ladder size: a 16-slot tile holds 16/K ensembles, rung-minor
    double pt_pswap;
    // pooled second moments accumulated inside the sampling kernel (AdaptCov, MomFused in
    // rmh_run.hpp): per-workgroup partial sums of z z^T, z = (theta, 1), over the states 1..T of
    // the launch, NTRI x 256 doubles per workgroup in the tile-pair layout of k_moments; chains
    // >= mom_N (padding) do not count.  Null: off.
    double *mom_partial;
    int64_t mom_N;
    int mom_passes;     // chain sets per workgroup (the grid is sized to one workgroup per CU)
};

struct RwArgs {
    const double *Ltiles; // lower-triangular tile image of scale*L (rho_c*L for pCN), staged in LDS
    const double *Vtiles; // pCN: tile image of (rho_c*L)^-1, read from global memory (L2)
    double rho;           // pCN (proposals/randomwalk.py:83-86)
    const double *Ltail;  // rows 96.. of scale*L in the layout of GaussArgs::Wtail, or null
    // block-mixture random walk (ChangepointRegression1DProp, riemann/test_changepoint.py:18-56):
    // nblk <= 3 consecutive coordinate blocks ending at blk_end[b], picked by chained
    // uniforms against blk_p[b]; nblk = 0: plain full-vector random walk
    int nblk;
    int blk_end[3];
    double blk_p[3];
    const double *Ldiag;  // the diagonal of scale*L when L is diagonal (d entries), else null
};

struct HmcArgs {
    double eps;
    int nsteps;
    // mass matrix M (hamiltonian.py:71-74), or null: tile images in GLOBAL memory
    // (L2 resident; LDS is full) of chM = cholesky(M), V = chM^-1 and V^T
    const double *chM, *V, *VT;
    // M^-1 = minv (chM chM^T)^-1.  1 for VanillaHMC(M); AdaptCovHMC sets M = C and chM = L
    // with C = d^0.4 L L^T (hamiltonian.py:117-119 with adaptive.py:101-102): minv = d^-0.4
    double minv;
};

// logistic model: row tiles (of 16 data rows) per LDS stage and doubles per stage image
// (LogRegModel, rmh_models.hpp); the y / weight block of a stage is CH x 32 <= 128 doubles
constexpr int kLogRegCH = 2; // 3 was measured slower in round 1 (MALA 0.63 -> 0.41: spills)
constexpr int kChunkDoubles(int NT) { return kLogRegCH * 256 * NT + 128; }

// tile counts of the triangular images
constexpr int tri_tiles(int NT) { return 2 * NT * (NT + 1); }
inline int nt_for_dim(int d) { return d <= 16 ? 1 : d <= 112 ? (d + 15) / 16 : -1; }
// 112 < d <= 1024: the workgroup-per-16-chains kernels of rmh_kernels_big.hip (Gaussian model,
// random walk): ceil(d/16) tiles, -1 beyond
constexpr int kBigMaxDim = 1024;
inline int nt_big(int d) { return (d > 112 && d <= kBigMaxDim) ? (d + 15) / 16 : -1; }

// ---- host-side launchers implemented in the kernel files --------------------
// all return hipError_t of the launch
hipError_t launch_gauss_eval(int NT, const GaussArgs &m, const ChainArgs &c, int want_grad,
                             double *lp_out, double *grad_out, double *lprior_out,
                             double *llik_out, hipStream_t st);
// `adapt`: run the AdaptScale* variant (c.adapt / c.adapt_target set, tile image unscaled)
hipError_t launch_gauss_rw(int NT, const GaussArgs &m, const RwArgs &p, const ChainArgs &c,
                           const RunArgs &r, bool adapt, hipStream_t st);
hipError_t launch_gauss_rw_tail(const GaussArgs &m, const RwArgs &p, const ChainArgs &c,
                                const RunArgs &r, bool adapt, hipStream_t st);
// grid of launch_gauss_rw_tail with r.mom_partial set, for `nblocks` 16-chain blocks: workgroups (= partial
// sums written) and the chain sets each of them takes (RunArgs::mom_passes)
int64_t gauss_rw_tail_mom_grid(int64_t nblocks, int *passes);
hipError_t launch_gauss_pt(int NT, const GaussArgs &m, const RwArgs &p, const ChainArgs &c,
                           const RunArgs &r, hipStream_t st);
hipError_t launch_mix_eval(int NT, const MixArgs &m, const ChainArgs &c, int want_grad, double *lp_out,
                           double *grad_out, double *lprior_out, double *llik_out, hipStream_t st);
hipError_t launch_mix_hmc(int NT, const MixArgs &m, const HmcArgs &p, const ChainArgs &c,
                          const RunArgs &r, bool adapt, hipStream_t st);
hipError_t launch_mix_propose_hmc(int NT, const MixArgs &m, const HmcArgs &p, const ChainArgs &c,
                                  double *theta_p, double *logq, hipStream_t st);
// r.pt_K > 0 selects the parallel-tempering variant
hipError_t launch_mix_rw(int NT, const MixArgs &m, const RwArgs &p, const ChainArgs &c,
                         const RunArgs &r, bool adapt, hipStream_t st);
hipError_t launch_mix_pcn(int NT, const MixArgs &m, const RwArgs &p, const ChainArgs &c,
                          const RunArgs &r, bool adapt, hipStream_t st);
hipError_t launch_logreg_pcn(int NT, const LogRegArgs &m, const RwArgs &p, const ChainArgs &c,
                             const RunArgs &r, bool adapt, hipStream_t st);
hipError_t launch_gauss_pcn(int NT, const GaussArgs &m, const RwArgs &p, const ChainArgs &c,
                            const RunArgs &r, bool adapt, hipStream_t st);
// p.chM != null selects the mass-matrix variant
hipError_t launch_gauss_hmc(int NT, const GaussArgs &m, const HmcArgs &p, const ChainArgs &c,
                            const RunArgs &r, bool adapt, hipStream_t st);
hipError_t launch_propose_rw(int NT, const RwArgs &p, const ChainArgs &c, double *theta_p,
                             double *logq, hipStream_t st);
hipError_t launch_gauss_propose_hmc(int NT, const GaussArgs &m, const HmcArgs &p,
                                    const ChainArgs &c, double *theta_p, double *logq,
                                    hipStream_t st);

hipError_t launch_logreg_eval(int NT, const LogRegArgs &m, const ChainArgs &c, int want_grad,
                              double *lp_out, double *grad_out, double *lprior_out,
                              double *llik_out, hipStream_t st);
hipError_t launch_logreg_rw(int NT, const LogRegArgs &m, const RwArgs &p, const ChainArgs &c,
                            const RunArgs &r, bool adapt, hipStream_t st);
hipError_t launch_logreg_hmc(int NT, const LogRegArgs &m, const HmcArgs &p, const ChainArgs &c,
                             const RunArgs &r, bool adapt, hipStream_t st);
hipError_t launch_logreg_propose_hmc(int NT, const LogRegArgs &m, const HmcArgs &p,
                                     const ChainArgs &c, double *theta_p, double *logq,
                                     hipStream_t st);

hipError_t launch_cp_eval(int k, const CpArgs &m, const ChainArgs &c, double *lprior,
                          double *llik, double *lpost, hipStream_t st);
hipError_t launch_cp_rw(int k, const CpArgs &m, const RwArgs &p, const ChainArgs &c,
                        const RunArgs &r, hipStream_t st);
// sufficient-statistics kernels (diagonal proposal, alpha = 1, plain random walk; adapt: the chains' own scales -- on the
// one-lane-per-chain kernel only, hipErrorNotReady where that one does not apply)
hipError_t launch_cp_rw_ss(int k, const CpArgs &m, const RwArgs &p, const ChainArgs &c,
                           const RunArgs &r, hipStream_t st, bool adapt = false);

// device-side covariance update of the pooled AdaptCov (k_acov_update, rmh_kernels_mfma.hip)
struct AcovUpdateArgs {
    const double *acc;   // moment sums, tile-pair layout of k_moments, (d+1) x (d+1)
    const double *C0;    // (d, d)
    double *Cout, *Lout; // (d, d) adapted covariance (C / d^0.4) and factor (chol / d^0.2)
    double *Limg;        // lower-triangular tile image of img_scale * L, NT tiles
    double *Ltail;       // rows 96.. of img_scale * L for k_run_rw_tail (d = 97..100), or null
    int *status;
    int d, NT;
    double t_adapt, f4, f2, img_scale, creg;
    int marginalize, smooth;
};
hipError_t launch_acov_update(const AcovUpdateArgs &a, hipStream_t st);
// pooled second moments of history entries [e0, e1): see rmh_kernels_mfma.hip
hipError_t launch_moments(int NT, const double *hist, int64_t e0, int64_t e1, int rows, int64_t ld,
                          int64_t N, int d, double *partial, int nwg, double *acc, hipStream_t st);
// acc[x] += sum over the nwg per-workgroup partials (n doubles each), in a fixed order
hipError_t launch_moments_reduce(const double *partial, int64_t nwg, int n, double *acc, hipStream_t st);
hipError_t launch_big_gauss_rw(int NT, const GaussArgs &m, const RwArgs &p, const ChainArgs &c,
                               const RunArgs &r, hipStream_t st);
hipError_t launch_big_gauss_eval(int NT, const GaussArgs &m, const ChainArgs &c, double *lp_out,
                                 double *lprior_out, double *llik_out, hipStream_t st);
// generic fused kernels over the changepoint policy (rmh_kernels_cp.hip): variant 0 plain,
// 1 AdaptScale random walk, 2 pCN, 3 AdaptScalepCN, 4 parallel tempering
hipError_t launch_cp_generic(int k, const CpArgs &m, const RwArgs &p, const ChainArgs &c,
                             const RunArgs &r, int variant, hipStream_t st);
hipError_t launch_cprj_eval(const CpRjArgs &m, const ChainArgs &c, double *lprior, double *llik,
                            double *lpost, hipStream_t st);
hipError_t launch_cprj_run(const CpRjArgs &m, const ChainArgs &c, const RunArgs &r, hipStream_t st);
// per-chain means / variances and chain-averaged autocorrelations of history entries [e0, e1)
hipError_t launch_hist_reduce(const double *hist, int64_t e0, int64_t e1, int rows, int64_t ld, int64_t N, int d,
                              int ngroups, double *mean_dn, double *var_dn, long long *partial, long long *sums,
                              hipStream_t st);
hipError_t launch_rng_seed(uint64_t *rng, int64_t ld, uint64_t seed, uint64_t chain0,
                           hipStream_t st);
hipError_t launch_rng_stream(uint64_t *rng, int64_t ld, int d, int T, const double *icdf, int nmode,
                             double *xi, double *u, hipStream_t st);
hipError_t launch_transpose_nd_to_dn(const double *src, double *dst, int64_t N, int d,
                                     int64_t ld, hipStream_t st);
hipError_t launch_transpose_dn_to_nd(const double *src, double *dst, int64_t N, int d,
                                     int64_t ld, int64_t entries, hipStream_t st);
hipError_t launch_debug_mfma(const double *A, const double *B, const double *C, double *D,
                             hipStream_t st);

} // namespace rmh
