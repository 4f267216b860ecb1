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
            const double *W = m.Wtiles + (size_t)k * kImg + lane;
            const double *mu = m.mu_pad + (size_t)k * (4 * S);
            double y[S];
#pragma unroll
            for (int s = 0; s < S; ++s) y[s] = th[s] - mu[4 * s + g];
            double4_t a[NT];
#pragma unroll
            for (int i = 0; i < NT; ++i) a[i] = double4_t{0.0, 0.0, 0.0, 0.0};
            T::template mul_lower<true>(W, y, a, m.ks);
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
                T::template mul_upper<true>(m.WTtiles + (size_t)k * kImg + lane, u, v, m.ks);
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

template <int NT>
struct LogRegModel {
    using Args = LogRegArgs;
    using T = Tile<NT>;
    static constexpr int S = T::S;
    static constexpr int CH = kLogRegCH;
    static constexpr int XT = 256 * NT;            // doubles of one row tile's X tiles
    static constexpr int CHD = kChunkDoubles(NT);  // doubles per stage
    static constexpr int NP = CHD / 128;           // 1-KiB DMA pieces per stage

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
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
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
            if (s < S - 4 || s < ks) eta = mfma(img[s * 64 + eoff[s & 3]], th[s], eta);
        return eta;
    }

    static RMH_DEV void grad_tile(const double *img, const int (&goff)[4], const double (&res)[4],
                                  double4_t (&G)[NT])
    {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int i = 0; i < NT; ++i) G[i] = mfma(img[i * 256 + goff[r]], res[r], G[i]);
        }
    }

    static RMH_DEV double comp(const double4_t &v, int r)
    {
        return (r == 0) ? v.x : (r == 1) ? v.y : (r == 2) ? v.z : v.w;
    }

    // One stage = CH row tiles, one after the other: eta (4*NT MFMAs), link function
    // of the lane's 4 values, gradient (4*NT MFMAs).  Measured alternatives that were
    // slower on MI355X: all eta / all link / all gradient phases (0.48 vs 0.51 of the
    // roof; 8 values in flight spill into AGPR traffic) and a hand pipeline that
    // parks the table gathers across the next MFMA block (0.48; 160 more live VGPRs).
    template <bool GRAD, bool VALUE>
    static RMH_DEV void stage_compute(const double *cur, const double *ltab, const int (&eoff)[4],
                                      const int (&goff)[4], int g, int ks, const double (&th)[S],
                                      double &ll, double4_t (&G)[NT])
    {
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            const double *img = cur + k * XT, *yw = cur + CH * XT + k * 32;
            const double4_t eta = eta_tile(img, eoff, ks, th);
            double res[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double e = comp(eta, r), yv = yw[4 * r + g], wv = yw[16 + 4 * r + g];
                double lse, sg;
                if (VALUE) {
                    logistic_terms(e, ltab, lse, sg);
                    ll += wv * (yv * e - lse);
                } else { // gradient-only evaluation: the log-likelihood value is not needed
                    sg = logistic_sigmoid(e, ltab);
                }
                res[r] = wv * (yv - sg);
            }
            if (GRAD) grad_tile(img, goff, res, G);
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
        double4_t G[NT];
#pragma unroll
        for (int i = 0; i < NT; ++i) G[i] = double4_t{0.0, 0.0, 0.0, 0.0};
        double ll = 0.0;
        __syncthreads(); // previous users of the staging buffers are done
        stage(m, 0, lds);
        __syncthreads();
        for (int64_t c = 0; c < m.nchunks; ++c) {
            const double *cur = lds + (c & 1) * CHD;
            if (c + 1 < m.nchunks) stage(m, c + 1, lds + ((c + 1) & 1) * CHD);
            stage_compute<GRAD, VALUE>(cur, lds + 2 * CHD, eoff, goff, g, m.ks, th, ll, G);
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
            T::unpack(G, grad);
#pragma unroll
            for (int s = 0; s < S; ++s) grad[s] = grad[s] - th[s] * m.inv_tau2;
        }
    }
};

} // namespace rmh
