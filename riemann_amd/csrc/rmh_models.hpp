// rmh_models.hpp -- device Model policies for the MFMA kernels.
//
//  GaussModel<NT>  : MultiGaussianDist, riemann/models/gaussian.py:46-58
//  LogRegModel<NT> : logistic regression plugin against the Model ABC
//                    (riemann/models/model.py:4-64); not in the reference.
//
// A policy provides
//   lds_doubles(grad)      LDS it needs
//   setup(lds, args, grad) cooperative fill of the static part (caller syncs)
//   eval<GRAD>(...)        log posterior (and gradient) of the wave's 16 chains
//                          in the Tile<NT> register layout; every wave of the
//                          workgroup must call it (it may contain barriers).
#pragma once
#include "rmh_internal.hpp"
#include "rmh_mfma.hpp"

namespace rmh {

// ------------------------------------------------------------------ Gaussian
template <int NT>
struct GaussModel {
    using Args = GaussArgs;
    using T = Tile<NT>;
    static constexpr int S = T::S;
    static constexpr int kTailNF = 0;               // > 0: d = 16 NF + 1 .. 16 NF + 4 known at compile time
    static constexpr int kImg = tri_tiles(NT) * 64; // doubles per triangular image

    static constexpr __host__ __device__ size_t lds_doubles(bool grad) { return (size_t)kImg * (grad ? 2 : 1) + 4 * S; }

    static RMH_DEV void setup(double *lds, const Args &m, bool grad)
    {
        const int n = kImg, tid = threadIdx.x, nt = blockDim.x;
        for (int i = tid; i < n; i += nt) lds[i] = m.Wtiles[i];
        if (grad)
            for (int i = tid; i < n; i += nt) lds[n + i] = m.WTtiles[i];
        double *mu = lds + (grad ? 2 * n : n);
        for (int i = tid; i < 4 * S; i += nt) mu[i] = m.mu_pad[i];
    }

    // logL = -0.5 (u.u + d log 2pi + logdetC), u = L^-1 (theta - mu)   gaussian.py:49-52
    // grad = -L^-T u                                                     gaussian.py:54-58
    // log_prior = 0 (gaussian.py:46-47); log_posterior guard model.py:49-55
    template <bool GRAD, bool VALUE = true>
    static RMH_DEV void eval(const double *lds, const Args &m, bool grad_in_lds, int lane,
                             const double (&th)[S], double &lp, double (&grad)[S],
                             double (&parts)[2])
    {
        const int g = lane >> 4;
        const double *W = lds + lane;
        const double *mu = lds + (grad_in_lds ? 2 * kImg : kImg);
        double uu = 0.0;
        double4_t acc[NT];
        if (GRAD) {
            // u is needed whole (it is the B operand of the W^T product)
            double y[S];
#pragma unroll
            for (int s = 0; s < S; ++s) y[s] = th[s] - mu[4 * s + g];
#pragma unroll
            for (int i = 0; i < NT; ++i) acc[i] = double4_t{0.0, 0.0, 0.0, 0.0};
            T::mul_lower(W, y, acc, m.ks);
#pragma unroll
            for (int i = 0; i < NT; ++i)
                uu += acc[i].x * acc[i].x + acc[i].y * acc[i].y + acc[i].z * acc[i].z +
                      acc[i].w * acc[i].w;
        } else {
            // value only: two output tiles in flight (two independent MFMA chains),
            // theta - mu formed on the fly, so the phase needs ~2 accumulators
            // instead of NT plus a centred copy of theta
#pragma unroll
            for (int i0 = 0; i0 < NT; i0 += 2) {
                double4_t a0 = double4_t{0.0, 0.0, 0.0, 0.0}, a1 = a0;
#pragma unroll
                for (int s = 0; s < 4 * (i0 + 2); ++s) {
                    if (s >= S) break;
                    if (s < S - 4 || s < m.ks) { // zero padding k-steps skipped
                        const double y = th[s] - mu[4 * s + g];
                        if (s < 4 * (i0 + 1)) a0 = mfma(W[T::lo(i0, s) * 64], y, a0);
                        if (i0 + 1 < NT) a1 = mfma(W[T::lo(i0 + 1 < NT ? i0 + 1 : i0, s) * 64], y, a1);
                    }
                }
                uu += a0.x * a0.x + a0.y * a0.y + a0.z * a0.z + a0.w * a0.w;
                if (i0 + 1 < NT) uu += a1.x * a1.x + a1.y * a1.y + a1.z * a1.z + a1.w * a1.w;
            }
        }
        uu = sum4(uu);
        const double logL = -0.5 * ((uu + m.dlog2pi) + m.logdetC);
        lp = combine_post(0.0, logL);
        parts[0] = 0.0;
        parts[1] = logL;
        if (GRAD) {
            double u[S];
            T::unpack(acc, u);
            double4_t v[NT];
#pragma unroll
            for (int i = 0; i < NT; ++i) v[i] = double4_t{0.0, 0.0, 0.0, 0.0};
            T::mul_upper(lds + kImg + lane, u, v, m.ks);
            T::unpack(v, grad);
#pragma unroll
            for (int s = 0; s < S; ++s) grad[s] = -grad[s];
        }
    }
};

// ------------------------------------------------- mixture of Gaussians
// benchmarks.MixtureDist with MultiGaussianDist components (models/benchmarks.py:31-62):
//   log_likelihood = log(sum_k w_k exp(logL_k))   (:44-52, no log-sum-exp: underflow -> -inf)
//   log_prior      = log(sum_k w_k exp(0))        (:54-62, a constant)
// The components' tile images are read from global memory (L2): any number of
// components, no LDS.  No gradient (the reference has none either).
template <int NT>
struct MixModel {
    using Args = MixArgs;
    using T = Tile<NT>;
    static constexpr int S = T::S;
    static constexpr int kTailNF = 0;
    static constexpr int kImg = tri_tiles(NT) * 64;

    static constexpr __host__ __device__ size_t lds_doubles(bool) { return 0; }
    static RMH_DEV void setup(double *, const Args &, bool) {}

    // GRAD: d/dtheta log sum_k w_k N_k = sum_k e_k g_k / sum_k e_k with e_k = w_k exp(logL_k) and
    // g_k = -W_k^T W_k (theta - mu_k) (gaussian.py:54-58).  The reference's MixtureDist has no
    // gradient; this is the build's addition so that MALA / HMC can run on the mixture targets.
    template <bool GRAD, bool VALUE = true>
    static RMH_DEV void eval(const double *, const Args &m, bool, int lane, const double (&th)[S],
                             double &lp, double (&grad)[S], double (&parts)[2])
    {
        const int g = lane >> 4;
        double acc = 0.0;
        if (GRAD) {
#pragma unroll
            for (int s = 0; s < S; ++s) grad[s] = 0.0;
        }
        for (int k = 0; k < m.ncomp; ++k) {
            asm volatile("" ::: "memory");
            const double *W = m.Wtiles + (size_t)k * kImg;
            const double *mu = m.mu_pad + (size_t)k * (4 * S);
            double y[S];
#pragma unroll
            for (int s = 0; s < S; ++s) y[s] = th[s] - mu[4 * s + g];
            double4_t a[NT];
#pragma unroll
            for (int i = 0; i < NT; ++i) a[i] = double4_t{0.0, 0.0, 0.0, 0.0};
            tri_stream1<NT, false>(W, lane, y, a, m.ks);
            double uu = 0.0;
#pragma unroll
            for (int i = 0; i < NT; ++i)
                uu += a[i].x * a[i].x + a[i].y * a[i].y + a[i].z * a[i].z + a[i].w * a[i].w;
            uu = sum4(uu);
            const double logLk = -0.5 * ((uu + m.dlog2pi) + m.cst[k]); // gaussian.py:49-52
            const double ek = m.wts[k] * exp(logLk);
            acc += ek;
            if (GRAD) {
                double u[S];
                T::unpack(a, u);
                double4_t v[NT];
#pragma unroll
                for (int i = 0; i < NT; ++i) v[i] = double4_t{0.0, 0.0, 0.0, 0.0};
                tri_stream1<NT, true>(m.WTtiles + (size_t)k * kImg, lane, u, v, m.ks);
                double gk[S];
                T::unpack(v, gk);
#pragma unroll
                for (int s = 0; s < S; ++s) grad[s] = grad[s] + ek * (-gk[s]);
            }
        }
        if (GRAD) {
#pragma unroll
            for (int s = 0; s < S; ++s) grad[s] = grad[s] / acc;
        }
        const double logL = log_full(acc);
        lp = combine_post(m.logP0, logL);
        parts[0] = m.logP0;
        parts[1] = logL;
    }
};

// ------------------------------------------------------- logistic regression
//
// X streams through LDS in stages of CH = 2 row tiles (32 data rows).  One
// stage image ("chunk") = CH x [4*NT MFMA tiles of 16 rows x 4 columns, 512 B
// each, zero padded to 16*NT columns], then CH x [y[16], w[16]] (w = 1 for real
// rows, 0 for padding rows), padded to a whole KiB, so that a stage is moved by
// 1-KiB global_load_lds pieces (HBM/L2 -> LDS with no VGPR staging).
// Element (row i, column k) of a row tile sits in tile s = k>>2 at 8-byte slot
//     (k&3)*16 + (i ^ F),   F = ((s&3)<<2) | (k&3)
// The XOR makes BOTH operand fetches conflict-free reads of the same image:
//         eta = X beta   reads A[m=row][k=col]   (lane: row cl, col 4s+g)
//         g  += X^T r    reads A[m=col][k=row]   (lane: col 16i+cl, row 4r+g)
// under either LDS banking: ds_read_b64 (32-lane groups, 64 banks) and the
// ds_read2(st64)_b64 pairs the compiler merges them into (16-lane groups, 32
// banks) -- a first version with F = ((s&3)<<2)|((k&3)>>1<<1) was 2-way
// conflicted under the latter (SQ_LDS_BANK_CONFLICT = 30 % of LDS cycles).
// So X is stored, moved and staged exactly once.

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_void_t;

// GT ("gradient tail", d = 16 NF + 1 .. 16 NF + 4 with NF = NT - 1; round 2: d = 97..100 only): the last
// 16-column tile of the gradient product holds at most 4 real columns; v_mfma_f64 and the VALU share
// the FP64 units on this chip, so those 4 columns
// are cheaper as 16 DPP fmas per row tile than as 4 MFMAs (as the four tail rows of
// k_run_rw_tail, rmh_run.hpp): lane (g, cl = 4r + c) reads X[row 4r+g][column 16 NF + c] with ONE
// ds_read_b64, the fmas take it by row broadcast, and every lane accumulates its own data rows'
// share of the four columns; the chain's four lanes meet once per evaluation (sum4_scatter).
// FULL (round 3): d = 16 NT exactly, every k-step of eta is real -- the generic kernel decides the last four
// k-steps of every eta chain by wave-uniform tests (four branches inside a chain of 4 NT MFMAs); as a
// compile-time fact the chain is straight-line code (d = 32: 2.15 -> 1.97 ms per iteration, d = 64: 3.38 -> 3.16).
template <int NT, bool GT = false, bool FULL = false>
struct LogRegModel {
    static_assert(!GT || NT >= 2, "the gradient tail needs at least one full column tile");
    static_assert(!(GT && FULL), "a dimension is either 16 NT or 16 (NT - 1) + 1..4");
    using Args = LogRegArgs;
    using T = Tile<NT>;
    static constexpr int S = T::S;
    static constexpr int CH = kLogRegCH;
    static constexpr int XT = 256 * NT;            // doubles of one row tile's X tiles
    static constexpr int CHD = kChunkDoubles(NT);  // doubles per stage
    static constexpr int NP = CHD / 128;           // 1-KiB DMA pieces per stage
    static constexpr int kTailNF = GT ? NT - 1 : 0; // > 0: d = 16 NF + 1 .. 16 NF + 4 is a compile-time fact
    static constexpr int NG = GT ? NT - 1 : NT;    // gradient row tiles on the matrix core
    static constexpr int KS = GT ? 4 * (NT - 1) + 1 : FULL ? 4 * NT : 0;   // k-steps of eta when known at compile time

    static constexpr __host__ __device__ size_t lds_doubles(bool) { return (size_t)2 * CHD + kLogiDoubles; }
    // LDS: [stage 0][stage 1][link-function table]
    static RMH_DEV void setup(double *lds, const Args &m, bool)
    {
        for (int i = threadIdx.x; i < kLogiDoubles; i += blockDim.x) lds[2 * CHD + i] = m.logi[i];
    }

    // asynchronous copy of stage `chunk` into `dst` (LDS): wave w moves pieces
    // w, w+NW, ...; completion = s_waitcnt vmcnt(0), which __syncthreads() issues
    static RMH_DEV void stage(const Args &m, int64_t chunk, double *dst)
    {
        // (the wave index as a SCALAR: piece counter and LDS destination are then scalar arithmetic -- as a vector
        // value the loop state was spilled in the generic NT = 6, 7 kernels and reloaded inside the stage loop)
        const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), nw = blockDim.x >> 6;
        const double *src = m.Ximg + chunk * CHD;
        for (int j = wave; j < NP; j += nw)
            __builtin_amdgcn_global_load_lds((glb_void_t *)(src + j * 128 + lane * 2),
                                             (lds_void_t *)(dst + j * 128), 16, 0, 0);
    }

    static RMH_DEV double4_t eta_tile(const double *img, const int (&eoff)[4], int ks,
                                      const double (&th)[S])
    {
        double4_t eta = double4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < S; ++s) // k-steps beyond ceil(d/4) multiply zero columns: skipped
            if (KS > 0 ? s < KS : (s < S - 4 || s < ks)) eta = mfma(img[s * 64 + eoff[s & 3]], th[s], eta);
        return eta;
    }

    struct GradTail {
        double t0 = 0.0, t1 = 0.0, t2 = 0.0, t3 = 0.0;
        // columns 96..99 of X^T res for the lane's four data rows; xv = this lane's entry
        RMH_DEV void add(double xv, const double (&res)[4])
        {
            fmac_bcast<0>(t0, xv, res[0]);  fmac_bcast<1>(t1, xv, res[0]);
            fmac_bcast<2>(t2, xv, res[0]);  fmac_bcast<3>(t3, xv, res[0]);
            fmac_bcast<4>(t0, xv, res[1]);  fmac_bcast<5>(t1, xv, res[1]);
            fmac_bcast<6>(t2, xv, res[1]);  fmac_bcast<7>(t3, xv, res[1]);
            fmac_bcast<8>(t0, xv, res[2]);  fmac_bcast<9>(t1, xv, res[2]);
            fmac_bcast<10>(t2, xv, res[2]); fmac_bcast<11>(t3, xv, res[2]);
            fmac_bcast<12>(t0, xv, res[3]); fmac_bcast<13>(t1, xv, res[3]);
            fmac_bcast<14>(t2, xv, res[3]); fmac_bcast<15>(t3, xv, res[3]);
        }
    };

    // G += X^T res for the NG column tiles on the matrix core: 4 k-steps (the lane's data rows
    // 4r + g) x NG MFMAs.  The A operands of k-step r + 1 are read while the MFMAs of k-step r
    // issue (a0 = those of k-step 0, read by the caller before the link function): left to
    // itself the compiler reads every pair of operands right before its two MFMAs and waits
    // for the LDS twelve times per row tile.  sched_barrier pins the order.
    static RMH_DEV void grad_tile(const double *img, const int (&goff)[4], const double (&res)[4],
                                  const double (&a0)[NG], double4_t (&G)[NG])
    {
        double a[2][NG];
#pragma unroll
        for (int i = 0; i < NG; ++i) a[0][i] = a0[i];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (r < 3) {
#pragma unroll
                for (int i = 0; i < NG; ++i) a[(r + 1) & 1][i] = img[i * 256 + goff[r + 1]];
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int i = 0; i < NG; ++i) G[i] = mfma(a[r & 1][i], res[r], G[i]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // One stage = CH row tiles, one after the other: eta (4*NT MFMAs), link function
    // of the lane's 4 values (in lockstep), gradient (4*NT MFMAs).  Measured alternatives that
    // were slower on MI355X: all eta / all link / all gradient phases (0.48 vs 0.51 of the
    // roof; 8 values in flight spill into AGPR traffic) and a hand pipeline that
    // parks the table gathers across the next MFMA block (0.48; 160 more live VGPRs).
    // Per row tile the y / weight block holds 16 (y, w) pairs: w = 1 for real rows, 0 for padding.
    // Everything a row tile reads that does not depend on its own arithmetic -- the (y, w) pairs,
    // the tail entry, the first gradient operands -- is read BEFORE the eta chain.
    template <bool GRAD, bool VALUE>
    static RMH_DEV void stage_compute(const double *cur, const double *ltab, const int (&eoff)[4],
                                      const int (&goff)[4], int toff, int g, int ks, const double (&th)[S],
                                      double &ll, double4_t (&G)[NG], GradTail &gt)
    {
        typedef double d2_t __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            const double *img = cur + k * XT;
            const d2_t *yw = static_cast<const d2_t *>(__builtin_assume_aligned(cur + CH * XT + k * 32, 16));
            d2_t p[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {                               // (y - 1/2, w / 16) of data row 4r + g
                // gradient only: the weight is not used.  Read as half of a 16-byte pair its dead upper
                // registers are handed to the next read at once, which then has to wait for this one:
                // four serialised LDS round trips at the head of every row tile
                if (VALUE) p[r] = yw[4 * r + g];
                else p[r].x = reinterpret_cast<const double *>(yw)[2 * (4 * r + g)];
            }
            double xv = 0.0, a0[NG];
            if (GRAD) {
                if (GT) xv = img[(4 * (NT - 1)) * 64 + toff];   // the tile of k-step 4 NF: columns 16 NF ..
#pragma unroll
                for (int i = 0; i < NG; ++i) a0[i] = img[i * 256 + goff[0]];
            }
            __builtin_amdgcn_sched_barrier(0);
            const double4_t eta = eta_tile(img, eoff, ks, th);
            __builtin_amdgcn_sched_barrier(0);
            const double e[4] = {eta.x, eta.y, eta.z, eta.w};
            double lse[4], sg[4], res[4];
            if (VALUE) { // twelve coefficient pairs per value: two values at a time (registers)
                const double e01[2] = {e[0], e[1]}, e23[2] = {e[2], e[3]};
                double l01[2], s01[2], l23[2], s23[2];
                logistic_lockstep<2, true>(e01, ltab, l01, s01);
                logistic_lockstep<2, true>(e23, ltab, l23, s23);
                lse[0] = l01[0]; lse[1] = l01[1]; lse[2] = l23[0]; lse[3] = l23[1];
                sg[0] = s01[0];  sg[1] = s01[1];  sg[2] = s23[0];  sg[3] = s23[1];
            } else {
                logistic_lockstep<4, false>(e, ltab, lse, sg);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                // e = 16 eta (the image holds 16 X):  16 (y eta - log(1 + exp eta)) = (y - 1/2) e - hv, weight
                // 1/16 (0 on padding rows);  y - sigmoid(eta) = (y - 1/2) - qs needs no weight: padding rows
                // are zero rows of X, whatever multiplies them
                if (VALUE) ll = fma_(p[r].y, fma_(p[r].x, e[r], -lse[r]), ll);
                res[r] = p[r].x - sg[r];
            }
            if (GRAD) {
                __builtin_amdgcn_sched_barrier(0);
                if (GT) gt.add(xv, res);
                grad_tile(img, goff, res, a0, G);
            }
        }
    }

    // logL = sum_i y_i eta_i - log(1 + exp(eta_i)),  eta = X beta
    // logP = -beta.beta / (2 tau2);  grad = X^T (y - sigmoid(eta)) - beta / tau2
    // VALUE = false: gradient only (lp and parts are then meaningless)
    template <bool GRAD, bool VALUE = true>
    static RMH_DEV void eval(double *lds, const Args &m, bool, int lane, const double (&th)[S],
                             double &lp, double (&grad)[S], double (&parts)[2])
    {
        const int cl = lane & 15, g = lane >> 4;
        int eoff[4], goff[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            eoff[k] = g * 16 + (cl ^ ((k << 2) | g));
            goff[k] = (cl >> 2) * 64 + (cl & 3) * 16 + ((4 * k + g) ^ cl);
        }
        // gradient tail: lane (g, cl = 4r + c) holds X[row 4r + g][column 16 NF + c] = tile 4 NF,
        // slot c*16 + (row ^ F) with F = c (image layout above: s & 3 = 0 for s = 4 NF)
        const int toff = (cl & 3) * 16 + ((4 * (cl >> 2) + g) ^ (cl & 3));
        double4_t G[NG];
#pragma unroll
        for (int i = 0; i < NG; ++i) G[i] = double4_t{0.0, 0.0, 0.0, 0.0};
        // (an opaque zero: as a plain constant the compiler materialises these five initial values
        // once at kernel entry and keeps them in scratch until every evaluation)
        double zero = 0.0;
        asm volatile("" : "+v"(zero));
        GradTail gt;
        gt.t0 = gt.t1 = gt.t2 = gt.t3 = zero;
        double ll = zero;
        __syncthreads(); // previous users of the staging buffers are done
        stage(m, 0, lds);
        __syncthreads();
        for (int64_t c = 0; c < m.nchunks; ++c) {
            const double *cur = lds + (c & 1) * CHD;
            if (c + 1 < m.nchunks) stage(m, c + 1, lds + ((c + 1) & 1) * CHD);
            stage_compute<GRAD, VALUE>(cur, lds + 2 * CHD, eoff, goff, toff, g, m.ks, th, ll, G, gt);
            __syncthreads();
        }
        ll = sum4(ll);
        double bb = 0.0;
#pragma unroll
        for (int s = 0; s < S; ++s) bb += th[s] * th[s];
        bb = sum4(bb);
        const double logP = -bb / m.two_tau2;
        lp = combine_post(logP, ll);
        parts[0] = logP;
        parts[1] = ll;
        if (GRAD) {
#pragma unroll
            for (int i = 0; i < NG; ++i) {
                grad[4 * i + 0] = G[i].x;
                grad[4 * i + 1] = G[i].y;
                grad[4 * i + 2] = G[i].z;
                grad[4 * i + 3] = G[i].w;
            }
            if (GT) { // rows 16 NF ..: lane group g keeps row 16 NF + g (tile NF, register 0)
                grad[4 * (NT - 1)] = sum4_scatter(gt.t0, gt.t1, gt.t2, gt.t3);
                grad[4 * (NT - 1) + 1] = grad[4 * (NT - 1) + 2] = grad[4 * (NT - 1) + 3] = 0.0;
            }
#pragma unroll
            for (int s = 0; s < S; ++s) grad[s] = grad[s] * 0.0625 - th[s] * m.inv_tau2;   // (16 X)^T res / 16: exact
        }
    }
};

} // namespace rmh
