// rmh_kernels_mfma.hip -- instantiations and launchers of the MFMA kernels of the Gaussian
// model (random-walk, pCN, HMC/MALA, parallel tempering), the AdaptCov moment / update kernels
// and the small utility kernels (stream seeding, layout transposes, test hooks).
// The logistic and mixture models' instantiations: rmh_kernels_logreg.hip, rmh_kernels_mix.hip
// (separate translation units so that the three compile in parallel).
#include "rmh_run.hpp"
#include "rmh_launch.hpp"

namespace rmh {

// ------------------------------------------------------------------ Gaussian
hipError_t launch_gauss_eval(int NT, const GaussArgs &m, const ChainArgs &c, int want_grad,
                             double *lp_out, double *grad_out, double *lprior_out,
                             double *llik_out, hipStream_t st)
{
    RMH_NT_DISPATCH(NT, {
        using M = GaussModel<kNT>;
        const size_t lds = sizeof(double) * M::lds_doubles(want_grad);
        if (want_grad)
            return launch(k_eval<M, true, 4>, c.nblocks, 4, lds, st, m, c, lp_out, grad_out,
                          lprior_out, llik_out);
        return launch(k_eval<M, false, 4>, c.nblocks, 4, lds, st, m, c, lp_out, grad_out,
                      lprior_out, llik_out);
    })
}

hipError_t launch_gauss_rw(int NT, const GaussArgs &m, const RwArgs &p, const ChainArgs &c,
                           const RunArgs &r, bool adapt, hipStream_t st)
{
    RMH_NT_DISPATCH(NT, {
        using M = GaussModel<kNT>;
        const size_t lds = sizeof(double) * (tri_tiles(kNT) * 64 + M::lds_doubles(false) + icdf_doubles(c.nmode));
        // d > 64: the two tile images fill the LDS of a CU, so one workgroup per CU;
        // eight waves give every SIMD a second wavefront to cover the s_waitcnt
        // stalls (LDS table reads of the normals, state/record traffic): +16 %
        constexpr int NW = (kNT >= 5) ? 8 : 4;
        if (c.d == 16 * kNT && !policy(RMH_POLICY_NO_TAIL)) { // every k-step and stream word is real: compile-time counts
            if (adapt) return launch(k_run_rw<M, NW, true, false, false, false, true>, c.nblocks, NW, lds, st, m, p, c, r);
            return launch(k_run_rw<M, NW, false, false, false, false, true>, c.nblocks, NW, lds, st, m, p, c, r);
        }
        if (adapt) return launch(k_run_rw<M, NW, true, false>, c.nblocks, NW, lds, st, m, p, c, r);
        return launch(k_run_rw<M, NW, false, false>, c.nblocks, NW, lds, st, m, p, c, r);
    })
}

// d = 16 NF + 1 .. 16 NF + 4, random walk (plain, per-chain adapted scale, or within parallel tempering
// when r.pt_K > 0): the NF full row tiles on the matrix core, the four tail rows on the VALU
// (k_run_rw_tail); r.mom_partial: the AdaptCov variant with the pooled moments in the kernel (NF = 6)
template <int NF>
static hipError_t gauss_rw_tail_launch(const GaussArgs &m, const RwArgs &p, const ChainArgs &c, const RunArgs &r,
                                       bool adapt, hipStream_t st)
{
    // as the generic kernels: eight waves once the tile images fill most of a CU's LDS
    constexpr int NW = (NF >= 4) ? 8 : 4;
    const size_t lds = sizeof(double) * (2 * tri_tiles(NF) * 64 + 2 * tail_doubles(NF) + 4 * 4 * (NF + 1) + icdf_doubles(c.nmode));
    if (c.nmode) { // the 64-bit normal mode: plain and adapted-scale walks (the callers route tempering and the fused moments around)
        if (r.mom_partial || r.pt_K > 0) return hipErrorInvalidValue;
        if (adapt) return launch(k_run_rw_tail<NF, NW, true, false, false, true>, c.nblocks, NW, lds, st, m, p, c, r);
        return launch(k_run_rw_tail<NF, NW, false, false, false, true>, c.nblocks, NW, lds, st, m, p, c, r);
    }
    if (r.mom_partial) { // AdaptCov: the pooled moments of the states 1..T accumulated in the kernel
        if constexpr (NF == 6) {
            if (r.pt_K > 0) return hipErrorInvalidValue;
            int passes = 1;
            const int64_t grid = gauss_rw_tail_mom_grid(c.nblocks, &passes);
            if (passes != r.mom_passes) return hipErrorInvalidValue;
            if (adapt) // AdaptScaleCovRandomWalk: the chains' own scales between the pooled covariance updates
                return launch(k_run_rw_tail<6, 8, true, false, true>, grid * 8, 8, lds + sizeof(double) * 2 * kMomPanel,
                              st, m, p, c, r);
            return launch(k_run_rw_tail<6, 8, false, false, true>, grid * 8, 8, lds + sizeof(double) * 2 * kMomPanel,
                          st, m, p, c, r);
        }
        return hipErrorInvalidValue;
    }
    if (r.pt_K > 0) return launch(k_run_rw_tail<NF, NW, false, true>, c.nblocks, NW, lds, st, m, p, c, r);
    if (adapt) return launch(k_run_rw_tail<NF, NW, true, false>, c.nblocks, NW, lds, st, m, p, c, r);
    return launch(k_run_rw_tail<NF, NW, false, false>, c.nblocks, NW, lds, st, m, p, c, r);
}

hipError_t launch_gauss_rw_tail(const GaussArgs &m, const RwArgs &p, const ChainArgs &c,
                                const RunArgs &r, bool adapt, hipStream_t st)
{
    switch ((c.d - 1) / 16) {   // NF full row tiles
    case 1: return gauss_rw_tail_launch<1>(m, p, c, r, adapt, st);
    case 2: return gauss_rw_tail_launch<2>(m, p, c, r, adapt, st);
    case 3: return gauss_rw_tail_launch<3>(m, p, c, r, adapt, st);
    case 4: return gauss_rw_tail_launch<4>(m, p, c, r, adapt, st);
    case 5: return gauss_rw_tail_launch<5>(m, p, c, r, adapt, st);
    case 6: return gauss_rw_tail_launch<6>(m, p, c, r, adapt, st);
    default: return hipErrorInvalidValue;
    }
}

int64_t gauss_rw_tail_mom_grid(int64_t nblocks, int *passes)
{
    const int64_t nwg = (nblocks + 7) / 8;
    int cus = 256;
    hipDeviceProp_t prop;
    int dev = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
        cus = prop.multiProcessorCount;
    int64_t np = (nwg + cus - 1) / cus;           // one workgroup per CU, as many chain sets as that takes
    np = np < 1 ? 1 : np > 16 ? 16 : np;
    *passes = (int)np;
    return (nwg + np - 1) / np;
}

// parallel tempering over a Gaussian target: one wave per SIMD (register budget)
hipError_t launch_gauss_pt(int NT, const GaussArgs &m, const RwArgs &p, const ChainArgs &c,
                           const RunArgs &r, hipStream_t st)
{
    RMH_NT_DISPATCH(NT, {
        using M = GaussModel<kNT>;
        const size_t lds = sizeof(double) * (tri_tiles(kNT) * 64 + M::lds_doubles(false) + icdf_doubles(c.nmode));
        constexpr int NW = (kNT >= 5) ? 8 : 4;
        if (c.d == 16 * kNT && !policy(RMH_POLICY_NO_TAIL))   // compile-time k-step and word counts (k_run_rw FULL)
            return launch(k_run_rw<M, NW, false, false, true, false, true>, c.nblocks, NW, lds, st, m, p, c, r);
        return launch(k_run_rw<M, NW, false, false, true>, c.nblocks, NW, lds, st, m, p, c, r);
    })
}

hipError_t launch_gauss_pcn(int NT, const GaussArgs &m, const RwArgs &p, const ChainArgs &c,
                            const RunArgs &r, bool adapt, hipStream_t st)
{
    RMH_NT_DISPATCH(NT, {
        using M = GaussModel<kNT>;
        const size_t lds = sizeof(double) * (tri_tiles(kNT) * 64 + M::lds_doubles(false) + icdf_doubles(c.nmode));
        if (r.pt_K > 0) {
            if (c.d == 16 * kNT && !policy(RMH_POLICY_NO_TAIL))   // compile-time k-step and word counts (k_run_rw FULL)
                return launch(k_run_rw<M, 4, false, true, true, false, true>, c.nblocks, 4, lds, st, m, p, c, r);
            return launch(k_run_rw<M, 4, false, true, true>, c.nblocks, 4, lds, st, m, p, c, r);
        }
        // (compile-time k-step counts at d = 16 NT and d = 16 (NT - 1) + 1 .. 4: launch_rw_ks, rmh_launch.hpp)
        if (adapt) return launch_rw_ks<M, 4, true, true, false>(m, p, c, r, lds, st);
        return launch_rw_ks<M, 4, false, true, false>(m, p, c, r, lds, st);
    })
}

hipError_t launch_gauss_hmc(int NT, const GaussArgs &m, const HmcArgs &p, const ChainArgs &c,
                            const RunArgs &r, bool adapt, hipStream_t st)
{
    RMH_NT_DISPATCH(NT, {
        using M = GaussModel<kNT>;
        const size_t lds = sizeof(double) * (M::lds_doubles(true) + icdf_doubles(c.nmode));
        if (r.pt_K > 0) { // parallel tempering: HMC / MALA (with or without a mass matrix) within the rungs
            if (p.chM) return launch(k_run_hmc<M, 4, false, true, true>, c.nblocks, 4, lds, st, m, p, c, r);
            return launch(k_run_hmc<M, 4, false, false, true>, c.nblocks, 4, lds, st, m, p, c, r);
        }
        // (compile-time k-step counts at d = 16 NT and d = 16 (NT - 1) + 1 .. 4: launch_hmc_ks, rmh_launch.hpp)
        if (p.chM) {
            if (adapt) return launch_hmc_ks<M, 4, true, true>(m, p, c, r, lds, st);
            return launch_hmc_ks<M, 4, false, true>(m, p, c, r, lds, st);
        }
        if (adapt) return launch_hmc_ks<M, 4, true, false>(m, p, c, r, lds, st);
        return launch_hmc_ks<M, 4, false, false>(m, p, c, r, lds, st);
    })
}

// p.Vtiles != null: the pCN proposal (logq written), else the random walk
hipError_t launch_propose_rw(int NT, const RwArgs &p, const ChainArgs &c, double *theta_p,
                             double *logq, hipStream_t st)
{
    RMH_NT_DISPATCH(NT, {
        const size_t lds = sizeof(double) * (tri_tiles(kNT) * 64 + icdf_doubles(c.nmode));
        if (p.Vtiles) return launch(k_propose_rw<kNT, 4, true>, c.nblocks, 4, lds, st, p, c, theta_p, logq);
        return launch(k_propose_rw<kNT, 4, false>, c.nblocks, 4, lds, st, p, c, theta_p, logq);
    })
}

hipError_t launch_gauss_propose_hmc(int NT, const GaussArgs &m, const HmcArgs &p,
                                    const ChainArgs &c, double *theta_p, double *logq,
                                    hipStream_t st)
{
    RMH_NT_DISPATCH(NT, {
        using M = GaussModel<kNT>;
        const size_t lds = sizeof(double) * (M::lds_doubles(true) + icdf_doubles(c.nmode));
        if (p.chM) return launch(k_propose_hmc<M, 4, true>, c.nblocks, 4, lds, st, m, p, c, theta_p, logq);
        return launch(k_propose_hmc<M, 4, false>, c.nblocks, 4, lds, st, m, p, c, theta_p, logq);
    })
}

// ------------------------------------------------ pooled second moments (AdaptCov)
// Sum over the states (entries e0..e1-1 of a history block, chains < N) of z z^T with
// z = (theta, 1): the accumulators S, SX, SX2 of AdaptCovProposal.adapt
// (riemann/proposals/adaptive.py:73-76) pooled over chains, as ONE symmetric
// (d+1) x (d+1) product on the matrix cores.  A workgroup stages a panel of 64 chains of
// one entry in LDS (rows = parameters, padded row stride 66: the (row, chain) gathers of
// the A/B operands hit 32 distinct even banks per half wave), then every k-block of 4
// chains feeds v_mfma_f64_16x16x4_f64 with A = Z[tile i], B = Z[tile j]^T -- the two
// operands have the same lane map, (row l&15, chain l>>4).  The NT(NT+1)/2 lower tile
// pairs are dealt round-robin to the 4 waves.  Per-workgroup partial sums are written
// out and added up in a fixed order by k_moments_reduce (reproducible; no atomics).
constexpr int kMomStride = 66;

template <int NT>
__global__ void __launch_bounds__(256) k_moments(const double *hist, int64_t e0, int64_t e1, int rows,
                                                 int64_t ld, int64_t N, int d, double *partial)
{
    constexpr int DP = 16 * NT, NTRI = NT * (NT + 1) / 2, TPW = (NTRI + 3) / 4;
    extern __shared__ __attribute__((aligned(16))) double X[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int m = lane & 15, k = lane >> 4;
    int ti[TPW], tj[TPW];
#pragma unroll
    for (int tt = 0; tt < TPW; ++tt) {
        const int t = 4 * tt + w;
        int i = 0;
        while ((i + 1) * (i + 2) / 2 <= t) ++i;
        ti[tt] = i;
        tj[tt] = t - i * (i + 1) / 2;
    }
    double4_t acc[TPW];
#pragma unroll
    for (int tt = 0; tt < TPW; ++tt) acc[tt] = double4_t{0.0, 0.0, 0.0, 0.0};
    const int64_t ncp = (ld + 63) / 64, npanels = (e1 - e0) * ncp;
    for (int64_t pid = blockIdx.x; pid < npanels; pid += gridDim.x) {
        const int64_t e = e0 + pid / ncp, c0 = (pid % ncp) * 64;
        const double *src = hist + e * (int64_t)rows * ld;
        for (int idx = tid; idx < DP * 64; idx += 256) {
            const int row = idx >> 6, col = idx & 63;
            const int64_t c = c0 + col;
            double v = 0.0;
            if (c < N) v = row < d ? src[(int64_t)row * ld + c] : (row == d ? 1.0 : 0.0);
            X[row * kMomStride + col] = v;
        }
        __syncthreads();
#pragma unroll 4
        for (int q = 0; q < 16; ++q) {
#pragma unroll
            for (int tt = 0; tt < TPW; ++tt) {
                if (4 * tt + w < NTRI) {
                    const double a = X[(16 * ti[tt] + m) * kMomStride + 4 * q + k];
                    const double b = X[(16 * tj[tt] + m) * kMomStride + 4 * q + k];
                    acc[tt] = mfma(a, b, acc[tt]);
                }
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int tt = 0; tt < TPW; ++tt) {
        const int t = 4 * tt + w;
        if (t < NTRI) {
            double *dst = partial + ((int64_t)blockIdx.x * NTRI + t) * 256 + lane;
            dst[0] = acc[tt].x;
            dst[64] = acc[tt].y;
            dst[128] = acc[tt].z;
            dst[192] = acc[tt].w;
        }
    }
}

// acc[x] += sum_g partial[g][x], g in a fixed order: 16 threads per output each add a
// strided sixteenth of the partials, then thread 0 of the group adds the 16 in order
__global__ void __launch_bounds__(256) k_moments_reduce(const double *partial, int64_t nwg, int n, double *acc)
{
    __shared__ double part[16][17];
    const int o = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const int x = blockIdx.x * 16 + o;
    double s = 0.0;
    if (x < n)
        for (int64_t g = grp; g < nwg; g += 16) s += partial[g * n + x];
    part[o][grp] = s;
    __syncthreads();
    if (grp == 0 && x < n) {
        double t = acc[x];
        for (int g = 0; g < 16; ++g) t += part[o][g];
        acc[x] = t;
    }
}

hipError_t launch_moments_reduce(const double *partial, int64_t nwg, int n, double *acc, hipStream_t st)
{
    hipLaunchKernelGGL(k_moments_reduce, dim3((n + 15) / 16), dim3(256), 0, st, partial, nwg, n, acc);
    return hipGetLastError();
}

// acc (NTRI*256 doubles, tile-pair order) += moments of hist entries [e0, e1)
hipError_t launch_moments(int NT, const double *hist, int64_t e0, int64_t e1, int rows, int64_t ld,
                          int64_t N, int d, double *partial, int nwg, double *acc, hipStream_t st)
{
    if (e1 <= e0) return hipSuccess;
    RMH_NT_DISPATCH(NT, {
        constexpr int NTRI = kNT * (kNT + 1) / 2;
        const size_t lds = sizeof(double) * 16 * kNT * kMomStride;
        hipError_t e = launch(k_moments<kNT>, (int64_t)nwg * 4, 4, lds, st, hist, e0, e1, rows, ld, N, d, partial);
        if (e != hipSuccess) return e;
        const int n = NTRI * 256;
        return launch_moments_reduce(partial, nwg, n, acc, st);
    })
}

// ---- the covariance update of AdaptCovProposal.adapt (adaptive.py:83-102) on the device ----
// One workgroup: C from the pooled moments (same expressions, same order as the host version
// in rmh_run_api.hip), Cholesky in LDS (right-looking; every element receives its
// subtractions in the same order as the host's left-looking loop, so the factor is bit
// identical), L /= d^0.2, and the proposal's tile image scale * L written in place -- the
// launch schedule of an adaptive run then needs no host round trip between updates.
// status: set to 1 when a pivot is not positive (numpy raises LinAlgError at adaptive.py:102).

// the value lane `l` (a compile-time constant) holds, as a wave-uniform scalar
RMH_DEV double readlane_f64(double x, int l)
{
    const uint64_t u = (uint64_t)__double_as_longlong(x);
    const uint32_t lo = __builtin_amdgcn_readlane((uint32_t)u, l), hi = __builtin_amdgcn_readlane((uint32_t)(u >> 32), l);
    return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
}

RMH_DEV double mom_at(const double *acc, int row, int col)
{
    if (row < col) {
        const int t = row;
        row = col;
        col = t;
    }
    const int i = row >> 4, j = col >> 4, t = i * (i + 1) / 2 + j;
    const int rr = row & 15, l = (rr & 3) * 16 + (col & 15), r = rr >> 2;
    return acc[(size_t)t * 256 + r * 64 + l];
}

// Cholesky: blocked left-looking, panels of 16 columns, the matrix in LDS with an odd row stride.
// Per panel (a) every element of the panel's columns receives the contributions of all columns
// left of the panel (one thread per element, k ascending), (b) wave 0 factors the 16 x 16 diagonal
// block (lane = row, column by column, no workgroup barrier), (c) every row below the block
// finishes its 16 entries by forward substitution (one thread per row).  Three barriers per
// panel, 21 at d = 100, against three per COLUMN in the unblocked version this replaces (at d = 100
// that one lost to a host round trip: 212 vs 175 ms per 1000 iterations at the cfg2 size).  Every
// element still receives its subtractions for k = 0, 1, 2, ... in that order, one multiply and
// one subtract each, as the host loop (rmh_run_api.hip, cholesky_lower) does: same bits.
// Row stride of the LDS matrix: odd.  (Round 3 tried an even stride = 2 (mod 4) so that every row
// starts 16-byte aligned for the compiler's ds_read_b128 / ds_read2_b64 of consecutive k: the
// update went from 68 to ~140 us -- the one-thread-per-row phases read a column of 32..64 rows per
// instruction, and an even stride folds those onto a quarter of the banks.)
__host__ __device__ constexpr int acov_row_stride(int d) { return d | 1; }

__global__ void __launch_bounds__(1024) k_acov_update(AcovUpdateArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double A[]; // d x ds, row-major
    const int d = a.d, ds = acov_row_stride(d), tid = threadIdx.x, nth = blockDim.x;
    const double ntot = mom_at(a.acc, d, d);
    const double creg = a.creg;                                  // 1e-12 * mean(diag C0), adaptive.py:95-96
    // covariance from the moments, the expressions of the host version in the same order.  Division
    // throughput is what this step costs (four f64 divisions per element; round 2's flat loop over
    // all d^2 elements took 33 000 cycles of the kernel's 156 000, clock64): the matrix is symmetric
    // and the factorisation reads its lower triangle only, so thread (ty, tx) of 32 x 32 forms the
    // elements (ty + 32 a, tx + 32 b) with column <= row and stores each twice in the output --
    // half the divisions, no integer division per element, the column sums SX once through LDS.
    double *sx = A + d * ds;                                     // SX[j] = moments (j, d)
    for (int j = tid; j < d; j += nth) sx[j] = mom_at(a.acc, j, d);
    __syncthreads();
    {
        const int ty = tid >> 5, tx = tid & 31;
#pragma unroll
        for (int ai = 0; ai < 4; ++ai)
#pragma unroll
            for (int bj = 0; bj <= ai; ++bj) {
                const int i = ty + 32 * ai, j = tx + 32 * bj;
                if (i < d && j <= i) {
                    const double c0 = a.C0[i * d + j];
                    const double cs = (mom_at(a.acc, i, j) - sx[i] * sx[j] / ntot) / (ntot - 1);
                    double cv;
                    if (a.smooth) cv = (ntot * cs + a.t_adapt * c0) / (ntot + a.t_adapt);
                    else if (ntot < a.t_adapt) cv = c0;
                    else cv = cs + (i == j ? creg : 0.0);
                    if (a.marginalize && i != j) cv = 0.0;
                    cv = cv / a.f4;
                    A[i * ds + j] = cv;
                    a.Cout[i * d + j] = cv;
                    a.Cout[j * d + i] = cv;
                }
            }
    }
    __syncthreads();
    for (int p0 = 0; p0 < d; p0 += 16) {
        const int pw = (d - p0 < 16) ? d - p0 : 16;              // columns of this panel
        // (a) contributions of the columns left of the panel
        if (p0 > 0) {
            for (int e = tid; e < (d - p0) * pw; e += nth) {
                const int i = p0 + e / pw, c = p0 + e % pw;
                if (c <= i) {
                    double t = A[i * ds + c];
                    const double *ri = A + i * ds, *rc = A + c * ds;
#pragma unroll 8
                    for (int k = 0; k < p0; ++k) t -= ri[k] * rc[k];   // p0 is a multiple of 16
                    A[i * ds + c] = t;
                }
            }
            __syncthreads();
        }
        // (b) the diagonal block, wave 0: lane r holds row p0 + r of the block in registers; the
        // pivot row's entries reach the other lanes as scalars (v_readlane), no LDS round trips
        if (tid < 64) {
            const int r = tid & 15;                              // (lanes 16.. shadow lanes 0..15)
            double x[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) x[k] = (r < pw && k < pw) ? A[(p0 + r) * ds + p0 + k] : (r == k ? 1.0 : 0.0);
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                double t = x[c];
#pragma unroll
                for (int k = 0; k < c; ++k) t -= x[k] * readlane_f64(x[k], c); // A[r][k] * A[c][k]
                const double piv = readlane_f64(t, c);
                if (c < pw && !(piv > 0.0) && RMH_MOM_ABL == 0) *a.status = 1;
                const double ljj = sqrt(piv);
                x[c] = (r == c) ? ljj : t / ljj;                 // rows above the diagonal: unused
            }
            if (tid < 16 && r < pw) {
#pragma unroll
                for (int k = 0; k < 16; ++k)
                    if (k <= r && k < pw) A[(p0 + r) * ds + p0 + k] = x[k];
            }
        }
        __syncthreads();
        // (c) the rows below the block: forward substitution against it, the row's 16 entries in
        // registers, the block read from LDS (every lane the same address: a broadcast)
        if (pw == 16) {
            for (int i = p0 + 16 + tid; i < d; i += nth) {
                double x[16];
#pragma unroll
                for (int k = 0; k < 16; ++k) x[k] = A[i * ds + p0 + k];
#pragma unroll
                for (int c = 0; c < 16; ++c) {
                    double t = x[c];
                    const double *lc = A + (p0 + c) * ds + p0;
#pragma unroll
                    for (int k = 0; k < c; ++k) t -= x[k] * lc[k];
                    x[c] = t / lc[c];
                }
#pragma unroll
                for (int k = 0; k < 16; ++k) A[i * ds + p0 + k] = x[k];
            }
        }
        __syncthreads();
    }
    // L = chol / d^0.2 in place (one division per element), then the three output layouts from LDS
    {
        const int ty = tid >> 5, tx = tid & 31;
#pragma unroll
        for (int ai = 0; ai < 4; ++ai)
#pragma unroll
            for (int bj = 0; bj < 4; ++bj) {
                const int i = ty + 32 * ai, j = tx + 32 * bj;
                if (i < d && j < d) {
                    const double v = (j <= i) ? A[i * ds + j] / a.f2 : 0.0;
                    A[i * ds + j] = v;
                    a.Lout[i * d + j] = v;
                }
            }
    }
    __syncthreads();
    // tile image: tile (i, s), s <= 4i+3, at index 2i(i+1)+s; lane l <- L[16i + (l&15)][4s + (l>>4)]
    const int ntile = 2 * a.NT * (a.NT + 1);
    for (int idx = tid; idx < ntile * 64; idx += nth) {
        const int tl = idx >> 6, l = idx & 63;
        int i = 0;
        while (2 * (i + 1) * (i + 2) <= tl) ++i;
        const int sidx = tl - 2 * i * (i + 1);
        const int row = 16 * i + (l & 15), col = 4 * sidx + (l >> 4);
        double v = 0.0;
        if (row < d && col < d && col <= row) v = A[row * ds + col];
        a.Limg[idx] = a.img_scale * v;
    }
    // the four tail rows 16 NF .. for the VALU tail of k_run_rw_tail (d = 16 NF + 1 .. + 4), layout of
    // pack_tail (rmh_api.hip)
    if (a.Ltail) {
        const int NFt = (d - 1) / 16, r0 = 16 * NFt;
        for (int idx = tid; idx < (NFt + 1) * 64; idx += nth) {
            const int G = idx >> 6, g = (idx >> 4) & 3, e = idx & 15;   // [(G*4 + g)*16 + 4 q + r]
            const int sg = 4 * G + (e >> 2), r = e & 3;
            const int row = r0 + r, col = 4 * sg + g;
            double v = 0.0;
            if (sg <= 4 * NFt && row < d && col < d && col <= row) v = a.img_scale * A[row * ds + col];
            a.Ltail[idx] = v;
        }
    }
}

hipError_t launch_acov_update(const AcovUpdateArgs &a, hipStream_t st)
{
    const size_t lds = sizeof(double) * ((size_t)a.d * acov_row_stride(a.d) + 128);   // matrix + SX
    return launch(k_acov_update, 16, 16, lds, st, a); // one workgroup of 1024 threads
}

// ---- per-chain reductions of a recorded history (rmh_history_reduce) -----------------------------
// One wavefront = 64 chains x ONE parameter x one group of LG lags.  Pass 1: the chains' means over
// the entries; pass 2: the centred values go through a ring of the last lag0 + LG values in LDS
// (slot-major, lane-minor: conflict-free), and every step adds v_t v_{t-k} to the LG lag sums in
// registers -- sum_t (x_t - m)(x_{t+k} - m), which is what the zero-padded FFT autocorrelation of
// riemann_amd/diagnostics.py computes.  Lag group 0 also writes the chains' means and variances.
// Per wave the 64 chains' normalised autocorrelations rho(k) = acov(k) / acov(0) are summed as
// INTEGERS, round(rho 2^40): integer addition does not care about order or grouping, so the sum over
// chains is the same number whatever the shard boundaries of a multi-GPU job are (distributed.
// pool_history: the ranks' sums add up to the single-process sum bit for bit); the quantisation is
// 4.5e-13 per chain at most and averages out over the chains.  k_hist_acf_reduce adds the waves'
// partials up over the chain blocks.
constexpr double kAcfScale = 1099511627776.0;   // 2^40 (include/riemann_hip.h: RMH_ACF_SCALE)

template <int LG>
__global__ void __launch_bounds__(64) k_hist_acov(const double *hist, int64_t e0, int64_t e1, int rows, int64_t ld,
                                                  int64_t N, int d, int ngroups, int grp, double *mean_dn,
                                                  double *var_dn, long long *partial)
{
    extern __shared__ __attribute__((aligned(16))) double ring[];      // (lag0 + LG) x 64
    const int lane = threadIdx.x;
    const int64_t nblk = (N + 63) / 64;
    const int64_t wid = blockIdx.x;
    const int p = (int)(wid % d);
    const int64_t cb = wid / d;
    const int64_t c = cb * 64 + lane;
    const bool live = c < N;
    const int64_t cc = live ? c : N - 1;
    const int lag0 = grp * LG, R = lag0 + LG;
    const int64_t n = e1 - e0, estride = (int64_t)rows * ld;
    const double *src = hist + e0 * estride + (int64_t)p * ld + cc;
    double m = 0.0;
    for (int64_t t = 0; t < n; ++t) m += src[t * estride];
    m /= (double)n;
    double acc[LG], a0 = 0.0;
#pragma unroll
    for (int k = 0; k < LG; ++k) acc[k] = 0.0;
    for (int i = lane; i < R * 64; i += 64) ring[i] = 0.0;            // zeros before the series starts
    __builtin_amdgcn_wave_barrier();
    int slot = 0;                                                       // slot of v_t
    for (int64_t t = 0; t < n; ++t) {
        const double v = src[t * estride] - m;
        ring[slot * 64 + lane] = v;
        a0 = fma_(v, v, a0);
        int sk = slot - lag0;                                           // slot of v_{t - lag0}
        sk += sk < 0 ? R : 0;
#pragma unroll
        for (int k = 0; k < LG; ++k) {
            acc[k] = fma_(v, ring[sk * 64 + lane], acc[k]);
            sk = sk == 0 ? R - 1 : sk - 1;
        }
        slot = slot + 1 == R ? 0 : slot + 1;
    }
    if (grp == 0 && live) {
        mean_dn[(int64_t)p * N + c] = m;
        var_dn[(int64_t)p * N + c] = a0 / (double)n;
    }
    // sum over the wave's chains of acov(k) / acov(0) (a constant series counts as rho = 1, as
    // diagnostics.autocorr_1d does)
#pragma unroll
    for (int k = 0; k < LG; ++k) {
        const double r = live ? (a0 == 0.0 ? 1.0 : acc[k] / a0) : 0.0;
        long long q = __double2ll_rn(r * kAcfScale);
        for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
        if (lane == 0) partial[(cb * d + p) * (int64_t)(ngroups * LG) + lag0 + k] = q;
    }
}

// sums[k][p] = sum over the chain blocks of the waves' integer sums
__global__ void k_hist_acf_reduce(const long long *partial, int64_t nblk, int d, int K, long long *sums)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= d * K) return;
    const int p = idx / K, k = idx % K;
    long long s = 0;
    for (int64_t b = 0; b < nblk; ++b) s += partial[(b * d + p) * (int64_t)K + k];
    sums[(int64_t)k * d + p] = s;
}

hipError_t launch_hist_reduce(const double *hist, int64_t e0, int64_t e1, int rows, int64_t ld, int64_t N, int d,
                              int ngroups, double *mean_dn, double *var_dn, long long *partial, long long *sums,
                              hipStream_t st)
{
    constexpr int LG = 32;
    const int64_t nblk = (N + 63) / 64;
    const size_t ldsmax = sizeof(double) * (size_t)ngroups * LG * 64;
    if (ldsmax > 160 * 1024) return hipErrorInvalidValue;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_hist_acov<LG>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsmax);
    if (e != hipSuccess) return e;
    for (int grp = 0; grp < ngroups; ++grp) {   // one launch per lag group: its ring is (grp + 1) x 16 KB,
        //                                          so the low groups run at high occupancy
        const size_t lds = sizeof(double) * (size_t)(grp + 1) * LG * 64;
        hipLaunchKernelGGL(k_hist_acov<LG>, dim3((unsigned)(nblk * d)), dim3(64), lds, st, hist, e0, e1, rows, ld, N,
                           d, ngroups, grp, mean_dn, var_dn, partial);
        e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    const int K = ngroups * LG;
    hipLaunchKernelGGL(k_hist_acf_reduce, dim3((unsigned)((d * K + 255) / 256)), dim3(256), 0, st, partial, nblk, d, K,
                       sums);
    return hipGetLastError();
}

// ------------------------------------------------------------------ utilities
__global__ void k_rng_seed(uint64_t *rng, int64_t nslots, uint64_t seed, uint64_t chain0)
{
    const int64_t slot = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= nslots) return;
    const int64_t c = (slot >> 6) * 16 + (slot & 15);
    const int g = (int)((slot >> 4) & 3);
    Xo x;
    xo_init(x, seed, 4 * (chain0 + (uint64_t)c) + (uint64_t)g);
    xo_store(rng, nslots, slot, x);
}

hipError_t launch_rng_seed(uint64_t *rng, int64_t ld, uint64_t seed, uint64_t chain0,
                           hipStream_t st)
{
    const int64_t nslots = 4 * ld;
    hipLaunchKernelGGL(k_rng_seed, dim3((unsigned)((nslots + 255) / 256)), dim3(256), 0, st, rng,
                       nslots, seed, chain0);
    return hipGetLastError();
}

// xi (T, ld, d) and u (T, ld) exactly as the fused kernels draw them (nmode: the 64-bit normal mode)
__global__ void k_rng_stream(uint64_t *rng, int64_t ld, int d, int T, const double *icdf, int nmode,
                             double *xi, double *u)
{
    extern __shared__ __attribute__((aligned(16))) double tab[];
    icdf_stage(tab, icdf, nmode);
    __syncthreads();
    const int lane = threadIdx.x & 63, cl = lane & 15, g = lane >> 4;
    const int64_t blk = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (blk >= ld / 16) return;
    const int64_t c = blk * 16 + cl, slot = blk * 64 + lane, nslots = 4 * ld;
    Xo x = xo_load(rng, nslots, slot);
    const int Mw = (d + 7) / 8, ks = (d + 3) / 4;
    for (int t = 0; t < T; ++t) {
        if (nmode) {
            for (int s = 0; s < ks; ++s) { // word s of sub-stream g is coordinate 4s + g
                const uint64_t w[1] = {xo_next(x)};
                double z[1];
                normals64_lockstep<1>(w, tab, z);
                if (4 * s + g < d) xi[((int64_t)t * ld + c) * d + 4 * s + g] = z[0];
            }
        } else {
            for (int m = 0; m < Mw; ++m) {
                const uint64_t w = xo_next(x);
                const uint32_t hw[2] = {(uint32_t)(w >> 32), (uint32_t)w};
                double zw[2];
                normals_lockstep<2>(hw, tab, zw);
                const double zh = zw[0], zl = zw[1];
                if (8 * m + g < d) xi[((int64_t)t * ld + c) * d + 8 * m + g] = zh;
                if (8 * m + 4 + g < d) xi[((int64_t)t * ld + c) * d + 8 * m + 4 + g] = zl;
            }
        }
        if (u) { // null: the normals only
            const double uu = Tile<1>::accept_uniform(x, g, cl);
            if (g == 0) u[(int64_t)t * ld + c] = uu;
        }
    }
    xo_store(rng, nslots, slot, x);
}

hipError_t launch_rng_stream(uint64_t *rng, int64_t ld, int d, int T, const double *icdf, int nmode,
                             double *xi, double *u, hipStream_t st)
{
    const int64_t nblocks = ld / 16;
    hipLaunchKernelGGL(k_rng_stream, dim3((unsigned)((nblocks + 3) / 4)), dim3(256),
                       sizeof(double) * icdf_doubles(nmode), st, rng, ld, d, T, icdf, nmode, xi, u);
    return hipGetLastError();
}

// (N, d) row-major -> (d, ld) parameter-major, zero padding chains N..ld-1
__global__ void k_nd_to_dn(const double *src, double *dst, int64_t N, int d, int64_t ld)
{
    __shared__ double tile[32][33];
    const int64_t c0 = (int64_t)blockIdx.x * 32;
    const int p0 = blockIdx.y * 32;
    for (int r = threadIdx.y; r < 32; r += blockDim.y) {
        const int64_t c = c0 + r;
        const int p = p0 + threadIdx.x;
        tile[r][threadIdx.x] = (c < N && p < d) ? src[c * d + p] : 0.0;
    }
    __syncthreads();
    for (int r = threadIdx.y; r < 32; r += blockDim.y) {
        const int p = p0 + r;
        const int64_t c = c0 + threadIdx.x;
        if (p < d && c < ld) dst[(int64_t)p * ld + c] = tile[threadIdx.x][r];
    }
}

hipError_t launch_transpose_nd_to_dn(const double *src, double *dst, int64_t N, int d,
                                     int64_t ld, hipStream_t st)
{
    const dim3 grid((unsigned)((ld + 31) / 32), (unsigned)((d + 31) / 32));
    hipLaunchKernelGGL(k_nd_to_dn, grid, dim3(32, 8), 0, st, src, dst, N, d, ld);
    return hipGetLastError();
}

// `entries` stacked (d, ld) blocks -> (entries, N, d) row-major
__global__ void k_dn_to_nd(const double *src, double *dst, int64_t N, int d, int64_t ld)
{
    __shared__ double tile[32][33];
    const int64_t e = blockIdx.z;
    src += e * (int64_t)d * ld;
    dst += e * N * d;
    const int64_t c0 = (int64_t)blockIdx.x * 32;
    const int p0 = blockIdx.y * 32;
    for (int r = threadIdx.y; r < 32; r += blockDim.y) {
        const int p = p0 + r;
        const int64_t c = c0 + threadIdx.x;
        tile[r][threadIdx.x] = (p < d && c < N) ? src[(int64_t)p * ld + c] : 0.0;
    }
    __syncthreads();
    for (int r = threadIdx.y; r < 32; r += blockDim.y) {
        const int64_t c = c0 + r;
        const int p = p0 + threadIdx.x;
        if (c < N && p < d) dst[c * d + p] = tile[threadIdx.x][r];
    }
}

hipError_t launch_transpose_dn_to_nd(const double *src, double *dst, int64_t N, int d,
                                     int64_t ld, int64_t entries, hipStream_t st)
{
    for (int64_t e0 = 0; e0 < entries; e0 += 32768) {
        const int64_t ne = entries - e0 < 32768 ? entries - e0 : 32768;
        const dim3 grid((unsigned)((N + 31) / 32), (unsigned)((d + 31) / 32), (unsigned)ne);
        hipLaunchKernelGGL(k_dn_to_nd, grid, dim3(32, 8), 0, st, src + e0 * (int64_t)d * ld,
                           dst + e0 * N * d, N, d, ld);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

// one v_mfma_f64_16x16x4_f64 with row-major operands (layout self-test)
__global__ void k_debug_mfma(const double *A, const double *B, const double *C, double *D)
{
    const int l = threadIdx.x, m = l & 15, k = l >> 4;
    double4_t c;
    c.x = C[(k + 0) * 16 + m];
    c.y = C[(k + 4) * 16 + m];
    c.z = C[(k + 8) * 16 + m];
    c.w = C[(k + 12) * 16 + m];
    const double4_t d = mfma(A[m * 4 + k], B[k * 16 + m], c);
    D[(k + 0) * 16 + m] = d.x;
    D[(k + 4) * 16 + m] = d.y;
    D[(k + 8) * 16 + m] = d.z;
    D[(k + 12) * 16 + m] = d.w;
}

hipError_t launch_debug_mfma(const double *A, const double *B, const double *C, double *D,
                             hipStream_t st)
{
    hipLaunchKernelGGL(k_debug_mfma, dim3(1), dim3(64), 0, st, A, B, C, D);
    return hipGetLastError();
}

} // namespace rmh
