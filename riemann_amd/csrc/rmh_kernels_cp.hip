// rmh_kernels_cp.hip -- changepoint regression at fixed k (BASELINE.json cfg 5):
//   ChangepointRegression1D.log_prior / log_likelihood / predict,
//   riemann/models/changepoint.py:106-160, 175-181, log_gamma_dist :18-19,
// sampled by a full-vector MetropolisRandomWalk (proposals/randomwalk.py:21-26)
// whose covariance is the block diagonal of riemann/test_changepoint.py:36-38.
//
// Layout: the same four-lanes-per-chain wavefront tile as the MFMA kernels
// (lane = 16*g + cl: chain cl of the wave's 16, sub-stream g), because the
// random stream is defined that way.  theta (d = 2k+2 <= 18 doubles) is
// replicated in the four lanes; the lanes split the work that dominates:
//   normals    lane g draws coordinates j = g (mod 4), then a 4-lane exchange
//   prior      lane g takes the logs of elements j = g (mod 4)
//   likelihood lane g takes the data points i = g (mod 4); x, y sit in LDS
// and partial sums meet in a 2-step xor butterfly.  HBM-bound on paper (104 B
// per recorded sample), instruction-issue bound in practice.
#include "rmh_run.hpp"


namespace rmh {

// an opaque copy: keeps a value in a VGPR as far as the optimiser can tell
RMH_DEV double reg(double x)
{
    asm volatile("" : "+v"(x));
    return x;
}

// (lo, hi) <- (a < x) ? y : (lo, hi) on the two halves of a double, spelled out: as a C++ select
// over a register array LLVM turns a chain of these into a select of ADDRESSES plus one scratch
// load per datum (cfg5: 23.2 ms), and with opaque copies in the conditional operands every
// comparison becomes a branch (22.5 ms; this form: 21.6 ms).  One asm per comparison, so that the
// two wait states a VALU-written lane mask needs before a VALU read are there by construction
// (the compiler pads such pairs itself but does not look inside asm).
// "This value has arrived": an empty asm that reads x.  gfx9 counts loads and stores in ONE in-order
// vmcnt queue, and the compiler puts the wait for a value loaded before a loop at its first use INSIDE
// the loop -- where, from the second iteration on, it waits for nothing but the previous iteration's
// history stores (a drain of ~1000 cycles per iteration in the one-wave-per-SIMD kernels below).
// Touching every such value before the loop moves the wait there.
RMH_DEV void arrived(double x) { asm volatile("" ::"v"(x)); }
RMH_DEV void arrived(uint64_t x) { asm volatile("" ::"v"(x)); }
RMH_DEV void arrived(const Xo &x)
{
    arrived(x.s0);
    arrived(x.s1);
    arrived(x.s2);
    arrived(x.s3);
}

RMH_DEV void sel_lt(uint32_t &lo, uint32_t &hi, double a, double x, double y)
{
    const uint64_t u = (uint64_t)__double_as_longlong(y);
    asm("v_cmp_lt_f64 vcc, %2, %3\n\ts_nop 1\n\tv_cndmask_b32 %0, %0, %4, vcc\n\tv_cndmask_b32 %1, %1, %5, vcc"
        : "+v"(lo), "+v"(hi)
        : "v"(a), "v"(x), "v"((uint32_t)u), "v"((uint32_t)(u >> 32))
        : "vcc");
}

template <int N>
RMH_DEV double pick(const double (&a)[N], int base, int g)
{
    // a[base + g] for g in 0..3 with static register indices; 0 beyond N
    const double v0 = (base + 0 < N) ? a[base + 0 < N ? base + 0 : 0] : 0.0;
    const double v1 = (base + 1 < N) ? a[base + 1 < N ? base + 1 : 0] : 0.0;
    const double v2 = (base + 2 < N) ? a[base + 2 < N ? base + 2 : 0] : 0.0;
    const double v3 = (base + 3 < N) ? a[base + 3 < N ? base + 3 : 0] : 0.0;
    return g == 0 ? v0 : g == 1 ? v1 : g == 2 ? v2 : v3;
}

template <int KC>
struct CpModel {
    static constexpr int D = 2 * KC + 2; // cpx[KC], cpv[KC+1], sig
    static constexpr int KP = KC + 1;    // pieces ("k" of changepoint.py:133)

    // log_prior, changepoint.py:128-160.  Returns NaN-mapped (-inf) value.
    static RMH_DEV double log_prior(const CpArgs &m, const double (&th)[D], int g)
    {
        double cpv[KP], ds[KP];
#pragma unroll
        for (int j = 0; j < KP; ++j) cpv[j] = th[KC + j];
#pragma unroll
        for (int j = 0; j < KP; ++j) {
            const double lo = (j == 0) ? m.xmin : th[j - 1 < 0 ? 0 : j - 1];
            const double hi = (j == KC) ? m.xmax : th[j < KC ? j : 0];
            ds[j] = hi - lo; // s[1:] - s[:-1], :142-143
        }
        double pv = 0.0, ps = 0.0;
#pragma unroll
        for (int b = 0; b < KP; b += 4) {
            const bool valid = b + g < KP;
            const double v = valid ? pick<KP>(cpv, b, g) : 1.0;
            const double dx = valid ? pick<KP>(ds, b, g) : 1.0;
            // log_gamma_dist (:18-19): alpha log(beta) + (alpha-1) log(x) - beta x - lgamma(alpha)
            const double tv = ((m.c_v + m.alpha_m1 * log_full(v)) - m.beta * v) - m.lgam_alpha;
            const double ts = log_full(dx);
            pv += valid ? tv : 0.0;
            ps += valid ? ts : 0.0;
        }
        pv = sum4(pv);
        ps = sum4(ps);
        const double sig = th[D - 1];
        double p2 = log_full(1.0 / (sig * sig)); // :145
        if (sig < 0.0) p2 = __builtin_nan("");   // :146-147
        double logP = ((m.c_k + pv) + ((m.c_s + ps) - m.c_sL)) + p2; // :156
        if (logP != logP) logP = -__builtin_huge_val();             // :158-159
        return logP;
    }

    // predict (:181) for one datum, valid when cpx is sorted (prior finite): a chain of
    // v = (cpx[j] < x) ? cpv[j+1] : v  (sel_lt above)
    static RMH_DEV double predict_sorted(const double (&th)[D], double x)
    {
        const uint64_t v = (uint64_t)__double_as_longlong(th[KC]);
        uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
#pragma unroll
        for (int j = 0; j < KC; ++j) sel_lt(lo, hi, th[j], x, th[KC + 1 + j]);
        return __longlong_as_double((long long)((uint64_t)lo | ((uint64_t)hi << 32)));
    }

    // numpy searchsorted(cpx, x) side='left' as a bisection with static
    // register indices (any cpx, sorted or not); returns cpv[index]
    static RMH_DEV double predict_any(const double (&th)[D], double x, int &idx)
    {
        int lo = 0, hi = KC;
#pragma unroll
        for (int it = 0; it < 5; ++it) { // 2^5 > KC+1 for KC <= 16
            if (lo < hi) {
                const int mid = lo + ((hi - lo) >> 1);
                double a = reg(th[0]);
#pragma unroll
                for (int j = 1; j < KC; ++j) a = (mid == j) ? reg(th[j]) : a;
                if (a < x) lo = mid + 1; else hi = mid;
            }
        }
        idx = lo;
        double v = reg(th[KC]);
#pragma unroll
        for (int j = 1; j < KP; ++j) v = (lo == j) ? reg(th[KC + j]) : v;
        return v;
    }

    // log_likelihood, changepoint.py:106-126
    template <bool SORTED>
    static RMH_DEV double log_lik(const double *xy, const CpArgs &m, const double (&th)[D], int g)
    {
        const double sig = th[D - 1], sig2 = sig * sig;
        double ss = 0.0;
        for (int i = g; i < m.n; i += 4) {
            const double x = xy[i], y = xy[m.n + i];
            int idx;
            const double v = SORTED ? predict_sorted(th, x) : predict_any(th, x, idx);
            const double r = v - y;
            ss += r * r;
        }
        ss = sum4(ss) / sig2;
        double logL = -0.5 * ((ss + m.n * log_full(sig2)) + m.nlog2pi); // :118-121
        if (logL != logL) logL = -__builtin_huge_val();                 // :124-125
        return logL;
    }
};

// The changepoint model as a Model policy of the generic fused kernels (rmh_run.hpp): slower
// than k_cp_rw below (256-thread workgroups, the chain's parameters gathered from the tile
// layout for every evaluation) but it brings everything those kernels can do -- scale
// adaptation, pCN, covariance adaptation, parallel tempering -- to this model.
constexpr int kCpPolicyMaxN = 1024; // data points the policy's LDS area holds

template <int KC>
struct CpPolicy {
    using Args = CpArgs;
    using CM = CpModel<KC>;
    static constexpr int D = CM::D, NT = (D + 15) / 16;
    using T = Tile<NT>;
    static constexpr int S = T::S;

    static constexpr __host__ __device__ size_t lds_doubles(bool) { return 2 * kCpPolicyMaxN; }
    static RMH_DEV void setup(double *lds, const Args &m, bool)
    {
        for (int i = threadIdx.x; i < 2 * m.n; i += blockDim.x) lds[i] = m.xy[i];
    }

    template <bool GRAD, bool VALUE = true>
    static RMH_DEV void eval(const double *lds, const Args &m, bool, int lane, const double (&th)[S],
                             double &lp, double (&grad)[S], double (&parts)[2])
    {
        static_assert(!GRAD, "the changepoint model has no gradient");
        const int cl = lane & 15, g = lane >> 4;
        double full[D];
#pragma unroll
        for (int q = 0; q < (D + 3) / 4; ++q) {
            double o[4];
            gather4(th[q], o);
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (4 * q + r < D) full[4 * q + r] = o[r];
        }
        const double logP = CM::log_prior(m, full, g);
        double logL = -__builtin_huge_val();
        // the sorted-cpx likelihood is valid wherever the prior is finite; elsewhere the
        // posterior is -inf whatever it returns
        if (__any(logP > -__builtin_huge_val())) logL = CM::template log_lik<true>(lds, m, full, g);
        lp = combine_post(logP, logL);
        parts[0] = logP;
        parts[1] = logL;
        (void)grad;
    }
};

template <int KC, int D>
RMH_DEV void cp_load(const double *theta, int64_t ld, int64_t c, double (&th)[D])
{
#pragma unroll
    for (int p = 0; p < D; ++p) th[p] = theta[(int64_t)p * ld + c];
}

// fused Sampler.run (samplers/sampler.py:44-90)
// 512-thread workgroups, two per CU (LDS: 47.5 KB each) = 4 waves per SIMD: the
// kernel is latency-bound (41 % of wave cycles parked on s_waitcnt at 2 waves per
// SIMD), so occupancy pays: +22 % over 256-thread workgroups.
// BLOCK: the block-mixture proposal of ChangepointRegression1DProp.propose at fixed k
// (riemann/test_changepoint.py:43-56): after the normals every sub-stream draws one more
// word -- sub-stream 0's is the accept uniform, those of sub-streams 1..3 the chained block
// selectors (:47-51) -- and only the picked block's normals enter theta + L xi (L is block
// diagonal); no block picked (the reference's trans-dimensional branch): theta' = theta.
template <int KC, bool BLOCK>
__global__ void __launch_bounds__(512, 4) k_cp_rw(CpArgs m, RwArgs pr, ChainArgs ch, RunArgs run)
{
    using M = CpModel<KC>;
    constexpr int D = M::D;
    constexpr int NT = (D + 15) / 16, S = 4 * NT;
    using T = Tile<NT>;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double *xy = lds;                 // x[n], y[n]
    double *Limg = lds + 2 * m.n;     // tile image of scale*L
    double *tab = Limg + tri_tiles(NT) * 64;
    for (int i = threadIdx.x; i < 2 * m.n; i += blockDim.x) xy[i] = m.xy[i];
    for (int i = threadIdx.x; i < tri_tiles(NT) * 64; i += blockDim.x) Limg[i] = pr.Ltiles[i];
    icdf_stage(tab, ch.icdf);
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cl = lane & 15, g = lane >> 4;
    const int64_t blk = (int64_t)blockIdx.x * (blockDim.x >> 6) + wave;
    if (blk >= ch.nblocks) return;
    const int64_t c = blk * 16 + cl, slot = blk * 64 + lane;

    double th[D];
    cp_load<KC, D>(ch.theta, ch.ld, c, th);
    double lp = ch.logpost[c];
    Xo rng = xo_load(ch.rng, ch.nslots, slot);

    int64_t knext = run.nburn, kentry = 0;
    // raw buffer stores (descriptor per entry in SGPRs, row offset scalar, one VGPR lane
    // offset): flat addressing spent a 64-bit multiply-add per stored row
    const uint32_t hvoff = (uint32_t)((int64_t)g * ch.ld + c) * 8u, hrow = (uint32_t)ch.ld * 32u;
    auto rec = [&](int64_t e) {
        if (e != knext) return;
        knext += run.nthin;
        const int64_t k = kentry++;
        if (!run.h_theta) return;
        const __amdgpu_buffer_rsrc_t r =
            rsrc(run.h_theta + k * (int64_t)(4 * ch.ks) * ch.ld, (int64_t)(4 * ch.ks) * ch.ld);
#pragma unroll
        for (int s = 0; s < S; ++s)
            if (4 * s + g < D) buf_store(r, hvoff, hrow * (uint32_t)s, pick<D>(th, 4 * s, g));
        if (g == 0) run.h_logpost[k * ch.ld + c] = lp;
    };
    rec(0);

    for (int64_t t = 0; t < run.T; ++t) {
        // theta' = theta + scale L xi  (randomwalk.py:26) on the matrix core, in the
        // 4-lanes-per-chain tile; then the chain's lanes exchange their quarters
        double xi[S];
        T::template normals_fixed<(D + 7) / 8, D, 1>(rng, D, g, tab, xi);   // D is a compile-time constant
        double ublk = 0.0, ub[4] = {0.0, 0.0, 0.0, 0.0};
        if (BLOCK) {
            ublk = u01(xo_next(rng));
            gather4(ublk, ub);
            const double u0 = ub[1], u1 = ub[2], u2 = ub[3];
            int pk = -1;
            if (pr.nblk > 2 && u2 < pr.blk_p[2]) pk = 2;
            if (pr.nblk > 1 && u1 < pr.blk_p[1]) pk = 1;
            if (pr.nblk > 0 && u0 < pr.blk_p[0]) pk = 0;
            const int lo = pk <= 0 ? 0 : pr.blk_end[pk == 2 ? 1 : 0];
            const int hi = pk < 0 ? 0 : pr.blk_end[pk];
#pragma unroll
            for (int s = 0; s < S; ++s) xi[s] = (4 * s + g >= lo && 4 * s + g < hi) ? xi[s] : 0.0;
        }
        double4_t acc[NT];
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            acc[i].x = pick<D>(th, 16 * i + 0, g);
            acc[i].y = pick<D>(th, 16 * i + 4, g);
            acc[i].z = pick<D>(th, 16 * i + 8, g);
            acc[i].w = pick<D>(th, 16 * i + 12, g);
        }
        T::mul_lower(Limg + lane, xi, acc, ch.ks);
        double loc[S], thp[D];
        T::unpack(acc, loc);
#pragma unroll
        for (int q = 0; q < (D + 3) / 4; ++q) { // row swaps (VALU), not ds_bpermute: rmh_device.hpp
            double o[4];
            gather4(loc[q], o);
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (4 * q + r < D) thp[4 * q + r] = o[r];
        }

        const double logP = M::log_prior(m, thp, g);
        double lpp = -__builtin_huge_val();
        // wave-level skip: only evaluate the likelihood where some chain needs it
        if (__any(logP > -__builtin_huge_val())) {
            const double logL = M::template log_lik<true>(xy, m, thp, g);
            lpp = combine_post(logP, logL); // model.py:49-55
        }
        const double u = BLOCK ? ub[0] : T::accept_uniform(rng, g, cl);
        const bool ok = mh_accept(lpp, lp, 0.0, u);
        if (ok) {
#pragma unroll
            for (int p = 0; p < D; ++p) th[p] = thp[p];
            lp = lpp;
        }
        if (run.h_accept && g == 0) run.h_accept[t * ch.ld + c] = ok ? 1 : 0;
        rec(t + 1);
    }
    if (g == 0) {
#pragma unroll
        for (int p = 0; p < D; ++p) ch.theta[(int64_t)p * ch.ld + c] = th[p];
        ch.logpost[c] = lp;
    }
    xo_store(ch.rng, ch.nslots, slot, rng);
}

// ------------------------------------------------------------------------------------------
// BASELINE cfg 5 proper: plain random walk with a DIAGONAL proposal factor (the block proposal of
// riemann/test_changepoint.py:36-38 written as one matrix) and alpha = 1.  Same sampler, same
// stream, same accept rule as k_cp_rw; what changes is where the arithmetic of an iteration goes:
//   * likelihood from sufficient statistics.  k_cp_rw spends O(n) compare/select/fma steps per
//     chain and iteration on sum_i (ypred_i - y_i)^2 (changepoint.py:181, :118).  The data are fixed:
//     with x sorted once (host), the data of piece p are the run [B_{p-1}, B_p) with
//     B_p = #{i : x_i <= cpx_p} (searchsorted side='left' of predict, :181, seen from the data's
//     side), and with P1 = prefix sums of y - ym
//         sum_i (ypred_i - y_i)^2 = sum (y_i - ym)^2 + sum_p vc_p (m_p vc_p - 2 (P1[B_p] - P1[B_{p-1}])),
//     vc_p = cpv_p - ym, m_p = B_p - B_{p-1}: k branch-free binary searches of floor(log2 n) + 1
//     steps (x padded with +inf to 2^steps - 1 entries in LDS) instead of n predictions.  Centring
//     by ym keeps the cancellation harmless (the sums are O(n var y), not O(n y^2)).
//   * the chain's four lanes share the work through LDS, not through register selects: lane g
//     owns coordinates g, 4+g, 8+g (.. the B-operand layout of the stream), proposes them with one
//     fma each (diagonal factor: no matrix product), writes them to a per-wave LDS row and every
//     lane reads the whole candidate back; lane g searches boundary g (and 4+g), the lanes publish
//     (B, P1[B]) pairs the same way.  The cross-lane traffic costs LDS instructions, which the
//     four resident waves per SIMD cover, instead of VALU issue slots (k_cp_rw: three all-gathers
//     by row swaps, ~170 cycles, plus ~90 v_cndmask of static-index selects per iteration).
//   * prior (changepoint.py:128-160) with alpha = 1: (alpha-1) log(cpv) is 0 for cpv > 0 and NaN
//     otherwise, exactly as numpy evaluates it, so no logarithm per height; sum_j log(ds_j) =
//     log(prod_j ds_j) guarded by min_j ds_j > 0 (two negative gaps must not cancel); log(1/sig^2)
//     = -log(sig^2), which the likelihood needs anyway: two logarithms per iteration instead of six.
// The sums are formed in another order than numpy's, so log densities agree with the oracle to
// rounding (1e-13 observed, 1e-9 asserted), not bit for bit; accept masks agree exactly on every
// fixture and comparison run.
// STEPS > 0: the search depth as a compile-time constant (the probe offsets become instruction
// immediates: 3 instead of 5.5 vector instructions per step); 0: m.ss_steps at run time.
// BLOCK: the block-mixture proposal of ChangepointRegression1DProp at fixed k (test_changepoint.py:18-56,
// as in k_cp_rw: one more word per sub-stream after the normals, block picked by chained uniforms, the
// normals outside the block zeroed -- with a diagonal factor the other coordinates then stay put exactly).
template <int KC, int STEPS, bool BLOCK = false>
__global__ void __launch_bounds__(512, 4) k_cp_rw_ss(CpArgs m, RwArgs pr, ChainArgs ch, RunArgs run)
{
    constexpr int D = 2 * KC + 2, KP = KC + 1;
    constexpr int NT = (D + 15) / 16, S = 4 * NT; // the lane owns coordinates 4s + g
    constexpr int NS = (D + 3) / 4;              // k-steps that hold coordinates
    constexpr int TS = 2 * (((D + 1) / 2) | 1);  // doubles per chain row of the candidate exchange (an odd
                                                 // number of 16-byte units: the 16 chains' reads fall on disjoint banks)
    constexpr int BS = 2 * (KC | 1);             // doubles per chain row of the (B, P1[B]) exchange
    using T = Tile<NT>;
    typedef double d2_t __attribute__((ext_vector_type(2)));
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int np = (1 << m.ss_steps) - 1;        // padded x entries; P1 has np + 1
    double *xs = lds, *P1 = lds + np;
    double *tab = P1 + np + 2;                   // 2 np + 2 doubles: the table and the rows stay 16-byte aligned
    double *xch = tab + kIcdfDoubles;            // per wave: 16 x TS candidate rows, 16 x BS boundary rows
    for (int i = threadIdx.x; i < 2 * np + 1; i += blockDim.x) lds[i] = m.ss_xp[i];
    icdf_stage(tab, ch.icdf);
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cl = lane & 15, g = lane >> 4;
    const int64_t blk = (int64_t)blockIdx.x * (blockDim.x >> 6) + wave;
    if (blk >= ch.nblocks) return;
    const int64_t c = blk * 16 + cl, slot = blk * 64 + lane;
    double *trow = xch + wave * (16 * (TS + BS)) + cl * TS;
    double *brow = xch + wave * (16 * (TS + BS)) + 16 * TS + cl * BS;

    // this lane's coordinates 4s + g of the current state and of the proposal factor
    double th[NS], ld[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const bool in = 4 * s + g < D;
        th[s] = in ? ch.theta[(int64_t)(4 * s + g) * ch.ld + c] : 0.0;
        ld[s] = in ? pr.Ldiag[4 * s + g] : 0.0;
    }
    double lp = ch.logpost[c];
    Xo rng = xo_load(ch.rng, ch.nslots, slot);

    int64_t knext = run.nburn, kentry = 0;
    const uint32_t hvoff = (uint32_t)((int64_t)g * ch.ld + c) * 8u, hrow = (uint32_t)ch.ld * 32u;
    auto rec = [&](int64_t e) {
        if (e != knext) return;
        knext += run.nthin;
        const int64_t k = kentry++;
        if (!run.h_theta) return;
        const __amdgpu_buffer_rsrc_t r =
            rsrc(run.h_theta + k * (int64_t)(4 * ch.ks) * ch.ld, (int64_t)(4 * ch.ks) * ch.ld);
#pragma unroll
        for (int s = 0; s < NS; ++s)
            if (4 * s + g < D) buf_store(r, hvoff, hrow * (uint32_t)s, th[s]);   // rows 4s + g: the history layout
        if (g == 0) buf_store(rsrc(run.h_logpost + k * ch.ld, ch.ld), (uint32_t)c * 8u, 0u, lp);
    };
    rec(0);

    const double kpc = (double)KP * (m.c_v - m.lgam_alpha);   // alpha = 1: sum_j (alpha log beta - lgamma(alpha))
    for (int64_t t = 0; t < run.T; ++t) {
        asm volatile("" ::: "memory");
        // ---- propose: theta' = theta + scale L xi, L diagonal (randomwalk.py:26)
        double xi[S];
        T::template normals_fixed<(D + 7) / 8, D, 1>(rng, D, g, tab, xi);
        double ub[4] = {0.0, 0.0, 0.0, 0.0};
        if (BLOCK) { // one word per sub-stream: (accept uniform, u_cpx, u_cpv, u_sig)
            gather4(u01(xo_next(rng)), ub);
            int pk = -1;
            if (pr.nblk > 2 && ub[3] < pr.blk_p[2]) pk = 2;
            if (pr.nblk > 1 && ub[2] < pr.blk_p[1]) pk = 1;
            if (pr.nblk > 0 && ub[1] < pr.blk_p[0]) pk = 0;
            const int lo = pk <= 0 ? 0 : pr.blk_end[pk == 2 ? 1 : 0];
            const int hi = pk < 0 ? 0 : pr.blk_end[pk];
#pragma unroll
            for (int s = 0; s < NS; ++s) xi[s] = (4 * s + g >= lo && 4 * s + g < hi) ? xi[s] : 0.0;
        }
        double tp[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            tp[s] = fma_(ld[s], xi[s], th[s]);
            if (4 * s + g < D) trow[4 * s + g] = tp[s];
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the wave's own LDS writes are in order
        double cand[2 * ((D + 1) / 2)];
#pragma unroll
        for (int q = 0; q < (D + 1) / 2; ++q) {
            const d2_t v = reinterpret_cast<const d2_t *>(trow - 0)[q];
            cand[2 * q] = v.x;
            cand[2 * q + 1] = v.y;
        }
        // ---- prior (changepoint.py:128-160), every lane for itself
        double prod = 1.0, dmin = __builtin_huge_val(), vsum = 0.0, vmin = __builtin_huge_val();
#pragma unroll
        for (int j = 0; j < KP; ++j) {
            const double lo = (j == 0) ? m.xmin : cand[j - 1 < 0 ? 0 : j - 1];
            const double hi = (j == KC) ? m.xmax : cand[j < KC ? j : 0];
            const double dx = hi - lo;                            // s[1:] - s[:-1], :142-143
            prod *= dx;
            dmin = __builtin_fmin(dmin, dx);                      // (a NaN gap leaves dmin alone and poisons prod)
            const double v = cand[KC + j];
            vsum += v;
            vmin = __builtin_fmin(vmin, v);
        }
        const double nan = __builtin_nan("");
        const double sig = cand[D - 1], sig2 = sig * sig;
        // The iteration's three logarithms -- log(u) of the accept test (sampler.py:84), log(prod ds)
        // and log(sig^2) -- are ONE evaluation: lane group 0 (the owner of the accept uniform's
        // sub-stream) takes u, group 1 the product, group 2 sig^2, and an all-gather over the chain's
        // four lanes hands every lane the three results.  log_fast is log_core, operation for
        // operation the accept test's logarithm, for every positive normal argument.
        const double uown = BLOCK ? ub[0] : u01(xo_next_if(rng, g == 0));
        const double larg = g == 0 ? uown : g == 1 ? prod : g == 2 ? sig2 : 1.0;
        double lg[4];
        gather4(log_fast(larg), lg);
        const double lu = lg[0], lsig2 = lg[2];
        const double ps = (dmin > 0.0) ? lg[1] : (dmin == 0.0 ? -__builtin_huge_val() : nan);
        const double pv = (vmin > 0.0) ? fma_(-m.beta, vsum, kpc) : nan;     // :19 with alpha = 1
        const double p2 = (sig < 0.0) ? nan : -lsig2;                        // :145-147
        double logP = ((m.c_k + pv) + ((m.c_s + ps) - m.c_sL)) + p2;         // :156
        if (logP != logP) logP = -__builtin_huge_val();                      // :158-159
        double lpp = -__builtin_huge_val();
        if (__any(logP > -__builtin_huge_val())) {
            // ---- likelihood: lane g searches B_j = #{x <= cpx'_j} for j = g (and 4 + g)
#pragma unroll
            for (int jj = 0; jj < (KC + 3) / 4; ++jj) {
                const int j = 4 * jj + g;
                if (j < KC) {
                    const double key = trow[j];
                    int lo8 = 0;                                  // 8 * (entries known to be <= key)
                    if (STEPS > 0) {
#pragma unroll
                        for (int st = STEPS - 1; st >= 0; --st) {
                            const double xv = *reinterpret_cast<const double *>(
                                reinterpret_cast<const char *>(xs) + lo8 + ((8 << st) - 8));
                            lo8 += (xv <= key) ? (8 << st) : 0;
                        }
                    } else {
                        for (int st = m.ss_steps - 1; st >= 0; --st) {
                            const int step8 = 8 << st;
                            const double xv = *reinterpret_cast<const double *>(
                                reinterpret_cast<const char *>(xs) + lo8 + step8 - 8);
                            lo8 += (xv <= key) ? step8 : 0;
                        }
                    }
                    const double p1 = *reinterpret_cast<const double *>(reinterpret_cast<const char *>(P1) + lo8);
                    d2_t pair;
                    pair.x = (double)(lo8 >> 3);
                    pair.y = p1;
                    reinterpret_cast<d2_t *>(brow)[j] = pair;
                }
            }
            __builtin_amdgcn_wave_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            double acc = 0.0, bprev = 0.0, pprev = 0.0;
#pragma unroll
            for (int pz = 0; pz < KP; ++pz) {
                double bcur = (double)m.n, pcur = 0.0;            // P1[n] = 0 up to rounding: the y are centred
                if (pz < KC) {
                    const d2_t pair = reinterpret_cast<const d2_t *>(brow)[pz];
                    bcur = pair.x;
                    pcur = pair.y;
                } else {
                    pcur = P1[m.n];
                }
                const double vc = cand[KC + pz] - m.ss_ym;
                acc = fma_(vc, fma_(bcur - bprev, vc, -2.0 * (pcur - pprev)), acc);
                bprev = bcur;
                pprev = pcur;
            }
            const double ss = (m.ss_p2 + acc) / sig2;
            double logL = -0.5 * ((ss + m.n * lsig2) + m.nlog2pi);            // :118-121
            if (logL != logL) logL = -__builtin_huge_val();                   // :124-125
            lpp = combine_post(logP, logL);                                   // model.py:49-55
        }
        // Sampler.sample, sampler.py:83-84 (mh_accept with the logarithm taken above; log(0) = -inf)
        const double delta = lpp - lp - 0.0;
        const bool ok = lu < ((delta < 0.0) ? delta : 0.0);
#pragma unroll
        for (int s = 0; s < NS; ++s) th[s] = ok ? tp[s] : th[s];
        lp = ok ? lpp : lp;
        if (run.h_accept && g == 0) run.h_accept[t * ch.ld + c] = ok ? 1 : 0;
        rec(t + 1);
    }
#pragma unroll
    for (int s = 0; s < NS; ++s)
        if (4 * s + g < D) ch.theta[(int64_t)(4 * s + g) * ch.ld + c] = th[s];
    if (g == 0) ch.logpost[c] = lp;
    xo_store(ch.rng, ch.nslots, slot, rng);
}

// ---- the same sampler with ONE lane per chain for everything but the draws -------------------
// k_cp_rw_ss keeps the chain in four lanes throughout, because the random stream is defined over
// four sub-streams per chain -- but only the normals need that.  Here a wavefront owns 64 chains:
// it draws in four passes of 16 chains (lane group g = sub-stream g, as the stream's spec wants; the
// lane keeps the four passes' xoshiro states), parks normals and accept uniforms in an LDS row per
// chain, and then lane l alone carries chain l through proposal, prior, the k boundary searches
// (independent, interleaved), likelihood, accept and record.  Nothing is computed four times and
// no lane idles through a search it has no boundary for: per chain and iteration 14 vector
// instructions where k_cp_rw_ss issues 26 (413 per 16 chains) and k_cp_rw 69.  History rows are
// written by 64 consecutive chains per instruction (whole 512-byte lines).
// S2 > 0: a bucket table in front of the searches (CpArgs::ss_nb, rmh_internal.hpp): one multiply-and-
// truncate and one table read give #{x <= key} up to the (at most 2^S2 - 1) points of one bucket, which
// S2 search steps resolve -- 1 + S2 dependent LDS round trips per boundary instead of floor(log2 n) + 1,
// and this kernel's single wave per SIMD waits out every one of them.  Same counts, hence the same bits.
// BLK (round 3): the reference's own fixed-k proposal, the block mixture of ChangepointRegression1DProp
// (riemann/test_changepoint.py:18-56), on this layout -- as in k_cp_rw_ss<.., BLOCK> every sub-stream
// draws one more word after the normals (accept uniform, u_cpx, u_cpv, u_sig), all four parked in the
// chain's LDS row; the lane that carries the chain picks the block by the chained uniforms and scales
// the diagonal factor of the other blocks' coordinates by 0 -- fma(0, xi, theta) = theta, the bits the
// zeroed normals of the four-lane kernel give.  The three blocks are the reference's (k changepoint
// positions, k + 1 heights, sigma): their ends are compile-time constants here (the launcher checks).
// ADAPT (end of round 3): AdaptScaleRandomWalk, the chain's own scale (AdaptScale, rmh_run.hpp) -- one lane is one chain here,
// so the three state values sit in the lane: theta' = theta + scale (L xi) with the unscaled diagonal, operation for
// operation what k_run_rw<CpPolicy, ..., ADAPT> does (5.5e9 it/s there, the generic four-lanes-per-chain kernel).
template <int KC, int STEPS, int S2 = 0, bool BLK = false, bool ADAPT = false>
__global__ void __launch_bounds__(256) k_cp_rw_ss64(CpArgs m, RwArgs pr, ChainArgs ch, RunArgs run)
{
    static_assert(!(BLK && ADAPT), "the block mixture has no scale adaptation");
    constexpr int D = 2 * KC + 2, KP = KC + 1;
    constexpr int NT = (D + 15) / 16, S = 4 * NT, NS = (D + 3) / 4;
    constexpr int DE = 2 * ((D + (BLK ? 5 : 2)) / 2);   // doubles read back per chain: D normals + the uniform(s), even
    constexpr int TS = 2 * ((DE / 2) | 1);       // row stride: an odd number of 16-byte units (bank spread)
    using T = Tile<NT>;
    typedef double d2_t __attribute__((ext_vector_type(2)));
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int np = (1 << m.ss_steps) - 1;
    const int bkd = S2 > 0 ? (((m.ss_nb + 2) / 2 + 1) & ~1) : 0;   // doubles of the bucket table (int32 entries)
    double *xs = lds, *P1 = lds + np;
    const int32_t *bk = reinterpret_cast<const int32_t *>(lds + 2 * np + 2);
    double *tab = P1 + np + 2 + bkd;
    double *xch = tab + kIcdfDoubles;            // per wave: 64 rows of TS doubles
    for (int i = threadIdx.x; i < 2 * np + 1; i += blockDim.x) lds[i] = m.ss_xp[i];
    if (S2 > 0)
        for (int i = threadIdx.x; i < (m.ss_nb + 2) / 2; i += blockDim.x) lds[2 * np + 2 + i] = m.ss_xp[2 * np + 2 + i];
    icdf_stage(tab, ch.icdf);
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cl = lane & 15, g = lane >> 4;
    const int64_t grp = (int64_t)blockIdx.x * (blockDim.x >> 6) + wave;   // group of 64 chains
    const int64_t blk0 = grp * 4;                                          // its first 16-chain block
    if (blk0 >= ch.nblocks) return;
    double *rows = xch + wave * (64 * TS);
    const int64_t c = grp * 64 + lane;            // the chain this lane evaluates
    const bool live = c < ch.ld;

    double th[D];
#pragma unroll
    for (int j = 0; j < D; ++j) th[j] = live ? ch.theta[(int64_t)j * ch.ld + c] : 0.0;
    double lp = live ? ch.logpost[c] : 0.0;
    AdaptScale ad{0.0, 0.0, 1.0};
    if (ADAPT && live) ad.load(ch, c);
    Xo st[4];                                      // pass b: chain 16 (blk0 + b) + cl, sub-stream g
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const int64_t slot = ((blk0 + b < ch.nblocks) ? blk0 + b : blk0) * 64 + lane;
        st[b] = xo_load(ch.rng, ch.nslots, slot);
    }

    // the proposal's diagonal factor in registers (round 2 read it inside the loop: behind the memory
    // fence below the compiler re-issued six vector loads per iteration, and their vmcnt wait drained
    // the previous iteration's history stores)
    double ldg[D];
#pragma unroll
    for (int j = 0; j < D; ++j) ldg[j] = pr.Ldiag[j];

    int64_t knext = run.nburn, kentry = 0;
    auto rec = [&](int64_t e) {
        if (e != knext) return;
        knext += run.nthin;
        const int64_t k = kentry++;
        if (!run.h_theta || !live) return;
        const __amdgpu_buffer_rsrc_t r =
            rsrc(run.h_theta + k * (int64_t)(4 * ch.ks) * ch.ld, (int64_t)(4 * ch.ks) * ch.ld);
#pragma unroll
        for (int j = 0; j < D; ++j) buf_store(r, (uint32_t)c * 8u, (uint32_t)ch.ld * 8u * (uint32_t)j, th[j]);
        buf_store(rsrc(run.h_logpost + k * ch.ld, ch.ld), (uint32_t)c * 8u, 0u, lp);
    };
    rec(0);
    // everything loaded so far has arrived before the loop starts (see `arrived`)
#pragma unroll
    for (int j = 0; j < D; ++j) {
        arrived(th[j]);
        arrived(ldg[j]);
    }
    arrived(lp);
    if (ADAPT) {
        arrived(ad.n);
        arrived(ad.acc);
        arrived(ad.scale);
    }
#pragma unroll
    for (int b = 0; b < 4; ++b) arrived(st[b]);

    const double kpc = (double)KP * (m.c_v - m.lgam_alpha);
    for (int64_t t = 0; t < run.T; ++t) {
        asm volatile("" ::: "memory");
        // ---- the draws: four passes of 16 chains x 4 sub-streams.  One wave per SIMD has nobody to
        // hide its latencies behind, so the passes are interleaved by hand: the four xoshiro steps of
        // a word side by side, then the normals of two passes in lockstep (all table reads first,
        // the Horner chains alternating; same operations per normal, the same bits).
        constexpr int MW = (D + 7) / 8;
        uint64_t w[4][MW];
#pragma unroll
        for (int mw = 0; mw < MW; ++mw)
#pragma unroll
            for (int b = 0; b < 4; ++b) w[b][mw] = xo_next(st[b]);
        double ua[4];
#pragma unroll
        for (int b = 0; b < 4; ++b)   // the accept word: sub-stream 0; BLK: one word of every sub-stream
            ua[b] = BLK ? u01(xo_next(st[b])) : u01(xo_next_if(st[b], g == 0));
        constexpr int PL = 2;                     // passes evaluated in lockstep
#pragma unroll
        for (int b0 = 0; b0 < 4; b0 += PL) {
            uint32_t h[PL * NS];
            double z[PL * NS];
#pragma unroll
            for (int q = 0; q < PL; ++q)
#pragma unroll
                for (int sk = 0; sk < NS; ++sk)
                    h[q * NS + sk] = (sk & 1) ? (uint32_t)w[b0 + q][sk / 2] : (uint32_t)(w[b0 + q][sk / 2] >> 32);
            normals_lockstep<PL * NS>(h, tab, z);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < PL; ++q) {
                double *row = rows + (16 * (b0 + q) + cl) * TS;
#pragma unroll
                for (int sk = 0; sk < NS; ++sk)
                    if (4 * sk + g < D) row[4 * sk + g] = z[q * NS + sk];
                if (BLK) row[D + g] = ua[b0 + q];
                else if (g == 0) row[D] = ua[b0 + q];
            }
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        double dr[DE];
#pragma unroll
        for (int q = 0; q < DE / 2; ++q) {
            const d2_t v = reinterpret_cast<const d2_t *>(rows + lane * TS)[q];
            dr[2 * q] = v.x;
            dr[2 * q + 1] = v.y;
        }
        // ---- propose: theta' = theta + scale L xi, L diagonal (randomwalk.py:26)
        double tp[D];
        if (BLK) { // block picked by the chained uniforms (test_changepoint.py:43-56, as k_cp_rw_ss)
            int pk = -1;
            if (dr[D + 3] < pr.blk_p[2]) pk = 2;
            if (dr[D + 2] < pr.blk_p[1]) pk = 1;
            if (dr[D + 1] < pr.blk_p[0]) pk = 0;
            const double mb[3] = {pk == 0 ? 1.0 : 0.0, pk == 1 ? 1.0 : 0.0, pk == 2 ? 1.0 : 0.0};
#pragma unroll
            for (int j = 0; j < D; ++j) tp[j] = fma_(ldg[j] * mb[j < KC ? 0 : (j < D - 1 ? 1 : 2)], dr[j], th[j]);
        } else if (ADAPT) { // theta + scale * (L xi): product, product, sum -- as the generic kernel rounds them
#pragma unroll
            for (int j = 0; j < D; ++j) {
                const double lx = ldg[j] * dr[j];
                tp[j] = th[j] + ad.scale * lx;
            }
        } else {
#pragma unroll
            for (int j = 0; j < D; ++j) tp[j] = fma_(ldg[j], dr[j], th[j]);
        }
        // ---- prior (changepoint.py:128-160, alpha = 1: see k_cp_rw_ss)
        double prod = 1.0, dmin = __builtin_huge_val(), vsum = 0.0, vmin = __builtin_huge_val();
#pragma unroll
        for (int j = 0; j < KP; ++j) {
            const double lo = (j == 0) ? m.xmin : tp[j - 1 < 0 ? 0 : j - 1];
            const double hi = (j == KC) ? m.xmax : tp[j < KC ? j : 0];
            const double dx = hi - lo;
            prod *= dx;
            dmin = __builtin_fmin(dmin, dx);
            vsum += tp[KC + j];
            vmin = __builtin_fmin(vmin, tp[KC + j]);
        }
        const double nan = __builtin_nan("");
        const double sig = tp[D - 1], sig2 = sig * sig;
        // the iteration's three logarithms side by side: log(sig^2), log(prod ds) and log(u) of the
        // accept test (sampler.py:84; log_core is that test's logarithm, log(0) = -inf)
        double lsig2, lprod, lu;
        log_fast3(sig2, prod, dr[D], lsig2, lprod, lu);
        const double ps = (dmin > 0.0) ? lprod : (dmin == 0.0 ? -__builtin_huge_val() : nan);
        const double pv = (vmin > 0.0) ? fma_(-m.beta, vsum, kpc) : nan;
        const double p2 = (sig < 0.0) ? nan : -lsig2;
        double logP = ((m.c_k + pv) + ((m.c_s + ps) - m.c_sL)) + p2;
        if (logP != logP) logP = -__builtin_huge_val();
        double lpp = -__builtin_huge_val();
        if (__any(logP > -__builtin_huge_val())) {
            // ---- likelihood: the KC boundary searches of this chain, step by step side by side
            int lo8[KC];
#pragma unroll
            for (int j = 0; j < KC; ++j) lo8[j] = 0;
            if (S2 > 0) {
                const double nbm1 = (double)(m.ss_nb - 1);
#pragma unroll
                for (int j = 0; j < KC; ++j) {
                    const double fb = __builtin_fmin(__builtin_fmax((tp[j] - m.ss_x0) * m.ss_invh, 0.0), nbm1);
                    lo8[j] = bk[(int)fb];                         // 8 L[bucket]
                }
#pragma unroll
                for (int sp = S2 - 1; sp >= 0; --sp)
#pragma unroll
                    for (int j = 0; j < KC; ++j) {
                        const double xv = *reinterpret_cast<const double *>(
                            reinterpret_cast<const char *>(xs) + lo8[j] + ((8 << sp) - 8));
                        lo8[j] += (xv <= tp[j]) ? (8 << sp) : 0;
                    }
            } else if (STEPS > 0) {
#pragma unroll
                for (int sp = STEPS - 1; sp >= 0; --sp)
#pragma unroll
                    for (int j = 0; j < KC; ++j) {
                        const double xv = *reinterpret_cast<const double *>(
                            reinterpret_cast<const char *>(xs) + lo8[j] + ((8 << sp) - 8));
                        lo8[j] += (xv <= tp[j]) ? (8 << sp) : 0;
                    }
            } else {
                for (int sp = m.ss_steps - 1; sp >= 0; --sp) {
                    const int step8 = 8 << sp;
#pragma unroll
                    for (int j = 0; j < KC; ++j) {
                        const double xv = *reinterpret_cast<const double *>(
                            reinterpret_cast<const char *>(xs) + lo8[j] + step8 - 8);
                        lo8[j] += (xv <= tp[j]) ? step8 : 0;
                    }
                }
            }
            double acc = 0.0, bprev = 0.0, pprev = 0.0;
#pragma unroll
            for (int pz = 0; pz < KP; ++pz) {
                const int b8 = pz < KC ? lo8[pz < KC ? pz : 0] : 8 * m.n;
                const double bcur = (double)(b8 >> 3);
                const double pcur = *reinterpret_cast<const double *>(reinterpret_cast<const char *>(P1) + b8);
                const double vc = tp[KC + pz] - m.ss_ym;
                acc = fma_(vc, fma_(bcur - bprev, vc, -2.0 * (pcur - pprev)), acc);
                bprev = bcur;
                pprev = pcur;
            }
            const double ss = (m.ss_p2 + acc) / sig2;
            double logL = -0.5 * ((ss + m.n * lsig2) + m.nlog2pi);
            if (logL != logL) logL = -__builtin_huge_val();
            lpp = combine_post(logP, logL);
        }
        // ---- accept (sampler.py:83-84)
        const double delta = lpp - lp - 0.0;
        const bool ok = lu < ((delta < 0.0) ? delta : 0.0);
#pragma unroll
        for (int j = 0; j < D; ++j) th[j] = ok ? tp[j] : th[j];
        lp = ok ? lpp : lp;
        if (ADAPT) ad.adapt(ok, ch.adapt_target);   // adapt() after the step, sampler.py:88
        if (run.h_accept && live) run.h_accept[t * ch.ld + c] = ok ? 1 : 0;
        rec(t + 1);
    }
    if (live) {
#pragma unroll
        for (int j = 0; j < D; ++j) ch.theta[(int64_t)j * ch.ld + c] = th[j];
        ch.logpost[c] = lp;
        if (ADAPT) ad.store(ch, c);
    }
#pragma unroll
    for (int b = 0; b < 4; ++b)
        if (blk0 + b < ch.nblocks) xo_store(ch.rng, ch.nslots, (blk0 + b) * 64 + lane, st[b]);
}

// adapt: the AdaptScale variant exists on the one-lane-per-chain kernel only (hipErrorNotReady: not applicable here,
// the caller takes the generic kernel)
template <int KC>
static hipError_t cp_rw_ss_launch(const CpArgs &m, const RwArgs &p, const ChainArgs &c,
                                  const RunArgs &r, hipStream_t st, bool adapt)
{
    constexpr int D = 2 * KC + 2;
    const int np = (1 << m.ss_steps) - 1;
    // one lane per chain (k_cp_rw_ss64: 4 waves x 64 chains per workgroup) once there are enough
    // chains to give every SIMD of the chip a wavefront of 64 (65 536: one each); below that the
    // four-lanes-per-chain kernel keeps four times as many waves in flight
    // (block mixture: the reference's three blocks -- k positions, k + 1 heights, sigma -- whose ends the
    // one-lane kernel holds as constants; any other block structure stays on the four-lane kernel)
    const bool blk = p.nblk > 0;
    const bool blk_ok = !blk || (p.nblk == 3 && p.blk_end[0] == KC && p.blk_end[1] == D - 1 && p.blk_end[2] == D);
    if (blk_ok && (policy(RMH_POLICY_CP_SS64) || (c.nblocks >= 4096 && !policy(RMH_POLICY_CP_SS16)))) {
        const int DE = 2 * ((D + (blk ? 5 : 2)) / 2), TS64 = 2 * ((DE / 2) | 1);
        const int nw = 4;
        const bool buckets = m.ss_nb > 0 && !policy(RMH_POLICY_CP_NO_BUCKETS);
        const size_t bkd = buckets ? (size_t)(((m.ss_nb + 2) / 2 + 1) & ~1) : 0;
        const size_t lds = sizeof(double) * ((size_t)2 * np + 2 + bkd + kIcdfDoubles + (size_t)nw * 64 * TS64);
        if (lds <= 160 * 1024) {
            const int64_t ngrp = (c.nblocks + 3) / 4;
            const dim3 grid((unsigned)((ngrp + nw - 1) / nw));
#define RMH_CP_SS64(...)                                                                                 \
    {                                                                                                    \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_cp_rw_ss64<KC, __VA_ARGS__>), \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);        \
        if (e != hipSuccess) return e;                                                                   \
        hipLaunchKernelGGL((k_cp_rw_ss64<KC, __VA_ARGS__>), grid, dim3(64 * nw), lds, st, m, p, c, r);    \
        return hipGetLastError();                                                                        \
    }
            if (adapt) { // (and here)
                if (blk) return hipErrorNotReady;
                if (buckets && m.ss_s2 == 2) RMH_CP_SS64(0, 2, false, true)
                if (buckets && m.ss_s2 == 3) RMH_CP_SS64(0, 3, false, true)
                RMH_CP_SS64(0, 0, false, true)
            }
            if (blk) { // fewer instantiations: the bucket table when the model has one, else run-time depth
                if (buckets && m.ss_s2 == 2) RMH_CP_SS64(0, 2, true)
                if (buckets && m.ss_s2 == 3) RMH_CP_SS64(0, 3, true)
                RMH_CP_SS64(0, 0, true)
            }
            if (buckets && m.ss_s2 == 2) RMH_CP_SS64(0, 2)
            if (buckets && m.ss_s2 == 3) RMH_CP_SS64(0, 3)
            switch (m.ss_steps) {
            case 6: RMH_CP_SS64(6)
            case 7: RMH_CP_SS64(7)
            case 8: RMH_CP_SS64(8)
            default: RMH_CP_SS64(0)
            }
#undef RMH_CP_SS64
        }
    }
    if (adapt) return hipErrorNotReady;
    constexpr int TS = 2 * (((D + 1) / 2) | 1), BS = 2 * (KC | 1);
    const int nw = 8;
    const size_t lds = sizeof(double) * ((size_t)2 * np + 2 + kIcdfDoubles + (size_t)nw * 16 * (TS + BS));
    if (lds > 80 * 1024) return hipErrorInvalidValue;
    const dim3 grid((unsigned)((c.nblocks + nw - 1) / nw));
#define RMH_CP_SS(STEPS, BLK)                                                                          \
    {                                                                                                  \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_cp_rw_ss<KC, STEPS, BLK>), \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);      \
        if (e != hipSuccess) return e;                                                                 \
        hipLaunchKernelGGL((k_cp_rw_ss<KC, STEPS, BLK>), grid, dim3(64 * nw), lds, st, m, p, c, r);    \
        return hipGetLastError();                                                                      \
    }
    if (p.nblk > 0) { // block-mixture proposal: the cfg-5 data size as a constant, anything else at run time
        if (m.ss_steps == 7) RMH_CP_SS(7, true)
        RMH_CP_SS(0, true)
    }
    switch (m.ss_steps) {   // n = 32..63, 64..127 (cfg 5: n = 100), 128..255; anything else at run time
    case 6: RMH_CP_SS(6, false)
    case 7: RMH_CP_SS(7, false)
    case 8: RMH_CP_SS(8, false)
    default: RMH_CP_SS(0, false)
    }
#undef RMH_CP_SS
}

// Model.log_prior / log_likelihood / log_posterior for the states in ch.theta
template <int KC>
__global__ void __launch_bounds__(256) k_cp_eval(CpArgs m, ChainArgs ch, double *lprior,
                                                  double *llik, double *lpost)
{
    using M = CpModel<KC>;
    constexpr int D = M::D;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    for (int i = threadIdx.x; i < 2 * m.n; i += blockDim.x) lds[i] = m.xy[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cl = lane & 15, g = lane >> 4;
    const int64_t blk = (int64_t)blockIdx.x * (blockDim.x >> 6) + wave;
    if (blk >= ch.nblocks) return;
    const int64_t c = blk * 16 + cl;
    double th[D];
    cp_load<KC, D>(ch.theta, ch.ld, c, th);
    const double logP = M::log_prior(m, th, g);
    const double logL = M::template log_lik<false>(lds, m, th, g);
    if (g == 0) {
        if (lprior) lprior[c] = logP;
        if (llik) llik[c] = logL;
        if (lpost) lpost[c] = combine_post(logP, logL);
    }
}

template <int KC>
static hipError_t cp_rw_launch(const CpArgs &m, const RwArgs &p, const ChainArgs &c,
                               const RunArgs &r, hipStream_t st)
{
    constexpr int D = 2 * KC + 2, NT = (D + 15) / 16;
    const size_t lds = sizeof(double) * (2 * (size_t)m.n + tri_tiles(NT) * 64 + kIcdfDoubles);
    const int nw = 8;
    const dim3 grid((unsigned)((c.nblocks + nw - 1) / nw));
    if (p.nblk > 0) hipLaunchKernelGGL((k_cp_rw<KC, true>), grid, dim3(64 * nw), lds, st, m, p, c, r);
    else hipLaunchKernelGGL((k_cp_rw<KC, false>), grid, dim3(64 * nw), lds, st, m, p, c, r);
    return hipGetLastError();
}

template <int KC>
static hipError_t cp_eval_launch(const CpArgs &m, const ChainArgs &c, double *a, double *b,
                                 double *d, hipStream_t st)
{
    const size_t lds = sizeof(double) * 2 * (size_t)m.n;
    const int nw = 4;
    const dim3 grid((unsigned)((c.nblocks + nw - 1) / nw));
    hipLaunchKernelGGL(k_cp_eval<KC>, grid, dim3(64 * nw), lds, st, m, c, a, b, d);
    return hipGetLastError();
}

// --------------------------------------------------------------------------------------
// Trans-dimensional sampling over the number of changepoints:
//   ChangepointRegression1DProp.propose WITH its birth/death branch,
//   riemann/test_changepoint.py:43-73, on add_changepoint / subtract_changepoint,
//   riemann/models/changepoint.py:193-240 (height map :48-69), and the model's
//   log_prior / log_likelihood (:106-160) for any number of changepoints.
// Same four-lanes-per-chain tile.  KD = compile-time size of the register arrays (6 or 14),
// KM <= KD the capacity (the model's kmax) that fixes the padded layout
//   [cpx (KM, +inf beyond k), cpv (KM+1, 1.0 beyond k+1), sig, k]
// and the normals' coordinates (cpx_j: j, cpv_j: KM + j, sig: 2 KM + 1).  The normals are
// parked in LDS so that "coordinate KM + j" is an address, not a register index.  Stream per
// iteration: the normals, then two more words from every sub-stream -- (accept, U1, U2, U3)
// and (U4, location, height split, index) -- DESIGN.md, "Trans-dimensional changepoint sampling".
// logqratio of a birth is log|det J| of (h, u) -> (h/f, h f), f = sqrt((1-u)/u): log(h / (u (1-u)))
// (the reference differentiates the map with autograd), of a death its negative.
template <int KD>
struct CpRj {
    static constexpr int KP = KD + 1;

    // the padded state of one chain, replicated in its four lanes
    struct State {
        double cpx[KD], cpv[KP], sig;
        int k;
    };

    static RMH_DEV double sel(const double (&a)[KP], int idx)
    {
        double v = reg(a[0]);
#pragma unroll
        for (int j = 1; j < KP; ++j) v = (idx == j) ? reg(a[j]) : v;
        return v;
    }

    template <bool SORTED>
    static RMH_DEV double log_lik(const double *xy, const CpRjArgs &m, const State &s, int g)
    {
        const CpArgs &b = m.base;
        const double sig2 = s.sig * s.sig;
        double ss = 0.0;
        for (int i = g; i < b.n; i += 4) {
            const double x = xy[i], y = xy[b.n + i];
            double v;
            if (SORTED) { // cpx sorted (finite prior), +inf beyond k: cpv[#{cpx < x}]
                const uint64_t v0 = (uint64_t)__double_as_longlong(s.cpv[0]);
                uint32_t lo = (uint32_t)v0, hi = (uint32_t)(v0 >> 32);
#pragma unroll
                for (int j = 0; j < KD; ++j) sel_lt(lo, hi, s.cpx[j], x, s.cpv[j + 1]);
                v = __longlong_as_double((long long)((uint64_t)lo | ((uint64_t)hi << 32)));
            } else { // numpy searchsorted on the first k entries, any order
                int lo = 0, hi = s.k;
#pragma unroll
                for (int it = 0; it < 5; ++it) {
                    if (lo < hi) {
                        const int mid = lo + ((hi - lo) >> 1);
                        double a = reg(s.cpx[0]);
#pragma unroll
                        for (int j = 1; j < KD; ++j) a = (mid == j) ? reg(s.cpx[j]) : a;
                        if (a < x) lo = mid + 1; else hi = mid;
                    }
                }
                v = sel(s.cpv, lo);
            }
            const double r = v - y;
            ss += r * r;
        }
        ss = sum4(ss) / sig2;
        double logL = -0.5 * ((ss + b.n * log_full(sig2)) + b.nlog2pi);
        if (logL != logL) logL = -__builtin_huge_val();
        return logL;
    }

    // The same likelihood from sufficient statistics (k_cp_rw_ss above has the algebra): the data of
    // piece p are the run [B_{p-1}, B_p) of the sorted x, B_j = #{x_i <= cpx_j}; lane g searches the
    // boundaries j = g (mod 4) and publishes (B_j, P1[B_j]) in the chain's LDS row, then every lane
    // adds up the k + 1 pieces.  Boundaries from j = k on are clamped to (n, P1[n]): the pieces beyond
    // k are empty by construction and need no mask (cpx beyond k is +inf, cpv beyond k + 1 is 1).
    // O(k log n) instead of O(n k) per evaluation; sorted cpx only (an unsorted candidate has
    // log prior -inf and its likelihood is never used).
    template <int STEPS>
    static RMH_DEV double log_lik_ss(const double *xs, const double *P1, double *brow, const CpRjArgs &m,
                                     const State &s, int g)
    {
        typedef double d2_t __attribute__((ext_vector_type(2)));
        const CpArgs &b = m.base;
#pragma unroll
        for (int j0 = 0; j0 < KD; j0 += 4) {
            double key = __builtin_huge_val();
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (j0 + q < KD && g == q) key = s.cpx[j0 + q < KD ? j0 + q : 0];
            int lo8 = 0;                                      // 8 * (entries known to be <= key)
            if (STEPS > 0) {
#pragma unroll
                for (int st = STEPS - 1; st >= 0; --st) {
                    const double xv = *reinterpret_cast<const double *>(
                        reinterpret_cast<const char *>(xs) + lo8 + ((8 << st) - 8));
                    lo8 += (xv <= key) ? (8 << st) : 0;
                }
            } else {
                for (int st = b.ss_steps - 1; st >= 0; --st) {
                    const int step8 = 8 << st;
                    const double xv = *reinterpret_cast<const double *>(
                        reinterpret_cast<const char *>(xs) + lo8 + step8 - 8);
                    lo8 += (xv <= key) ? step8 : 0;
                }
            }
            d2_t pair;
            pair.x = (double)(lo8 >> 3);
            pair.y = *reinterpret_cast<const double *>(reinterpret_cast<const char *>(P1) + lo8);
            if (j0 + g < KD) reinterpret_cast<d2_t *>(brow)[j0 + g] = pair;
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the wave's own LDS writes are in order
        const double bn = (double)b.n, pn = P1[b.n];
        double acc = 0.0, bprev = 0.0, pprev = 0.0;
#pragma unroll
        for (int pz = 0; pz < KP; ++pz) {
            double bcur = bn, pcur = pn;
            if (pz < KD) {
                const d2_t pair = reinterpret_cast<const d2_t *>(brow)[pz];
                bcur = pz < s.k ? pair.x : bn;
                pcur = pz < s.k ? pair.y : pn;
            }
            const double vc = s.cpv[pz] - b.ss_ym;
            acc = fma_(vc, fma_(bcur - bprev, vc, -2.0 * (pcur - pprev)), acc);
            bprev = bcur;
            pprev = pcur;
            // four pairs in flight at a time: read all at once they cost 2 KP registers next to the
            // two padded states the iteration already holds
            if ((pz & 3) == 3) asm volatile("" ::: "memory");
        }
        __builtin_amdgcn_wave_barrier();                      // the row is rewritten by the next evaluation
        const double sig2 = s.sig * s.sig;
        const double ss = (b.ss_p2 + acc) / sig2;
        double logL = -0.5 * ((ss + b.n * log_full(sig2)) + b.nlog2pi);
        if (logL != logL) logL = -__builtin_huge_val();
        return logL;
    }

    // log_prior for k changepoints (k' = k + 1 pieces), changepoint.py:128-160, with the lane walking all
    // pieces itself, in order.  sum_j log(ds_j) = log(prod_j ds_j), guarded by min_j ds_j > 0 (two
    // negative gaps must not cancel), as in k_cp_rw_ss: one logarithm instead of k + 1; pieces beyond k
    // contribute a factor 1.  alpha = 1: (alpha - 1) log(v) is +-0 for 0 < v < inf and NaN otherwise,
    // exactly as numpy evaluates it -- no logarithm per height either.  ONE statement for the four-lane
    // kernel (every lane of a chain evaluates it: the state is replicated there), the one-lane kernel and
    // the evaluation entry: a shard that runs on one kernel and the whole job that runs on the other must
    // agree bit for bit (round 2's four-lane version split the pieces over the chain's lanes, four
    // logarithms per lane and a sum across lanes).
    static RMH_DEV double log_prior1(const CpRjArgs &m, const double *ktab, const State &s, int KM)
    {
        const CpArgs &b = m.base;
        double pv = 0.0, prod = 1.0, dmin = __builtin_huge_val();
        bool vbad = false;
#pragma unroll
        for (int j = 0; j < KP; ++j) {
            if (j <= KM) { // (wave-uniform: the model's capacity)
                const bool valid = j <= s.k;
                const double lo = (j == 0) ? b.xmin : s.cpx[j >= 1 ? j - 1 : 0];
                double hi = (j < KD) ? s.cpx[j < KD ? j : 0] : b.xmax;
                hi = (j == s.k) ? b.xmax : hi;        // the last piece ends at xmax (cpx beyond k is +inf)
                const double dx = valid ? hi - lo : 1.0, vv = valid ? s.cpv[j] : 1.0;
                prod *= dx;
                dmin = __builtin_fmin(dmin, dx);      // (a NaN gap leaves dmin alone and poisons prod)
                double la = 0.0;
                if (b.alpha_m1 == 0.0) vbad = vbad || !(vv > 0.0 && vv < __builtin_huge_val());
                else la = b.alpha_m1 * log_full(vv);
                const double tv = ((b.c_v + la) - b.beta * vv) - b.lgam_alpha;
                pv += valid ? tv : 0.0;
            }
        }
        const double nan = __builtin_nan("");
        const double sig2 = s.sig * s.sig;
        double lprod, lsig2;
        {
            double dummy;
            log_fast3(prod, sig2, 1.0, lprod, lsig2, dummy);
        }
        const double ps = (dmin > 0.0) ? lprod : (dmin == 0.0 ? -__builtin_huge_val() : nan);
        if (vbad) pv = nan;
        const double *kt = ktab + 4 * (s.k + 1);
        const double p2 = (s.sig < 0.0) ? nan : -lsig2;          // log(1 / sig^2)
        double logP = ((kt[0] + pv) + ((kt[1] + ps) - kt[2])) + p2;
        if (logP != logP) logP = -__builtin_huge_val();
        return logP;
    }

    // ---- one lane per chain (k_cprj_run64) --------------------------------------------------------
    // the likelihood from sufficient statistics, all boundary searches by this lane, step by step side
    // by side (independent chains of dependent LDS reads: the wave's only latency cover); the pieces are
    // summed exactly as log_lik_ss sums them
    // S2 > 0: the bucket table of k_cp_rw_ss64 in front of the searches (CpArgs::ss_nb): one multiply-and-
    // truncate and one table read, then S2 search steps -- 1 + S2 dependent LDS round trips per boundary
    // instead of floor(log2 n) + 1.  Same counts, hence the same bits.
    template <int STEPS, int S2>
    static RMH_DEV double log_lik_ss1(const double *xs, const double *P1, const int32_t *bk, const CpRjArgs &m,
                                      const State &s, int KM)
    {
        const CpArgs &b = m.base;
        int lo8[KD];
#pragma unroll
        for (int j = 0; j < KD; ++j) lo8[j] = 0;
        if (S2 > 0) {
            const double nbm1 = (double)(b.ss_nb - 1);
#pragma unroll
            for (int j = 0; j < KD; ++j)
                if (j < KM) {
                    // (+inf beyond k: the product clamps to the last bucket, the steps run into the +inf padding)
                    const double fb = __builtin_fmin(__builtin_fmax((s.cpx[j] - b.ss_x0) * b.ss_invh, 0.0), nbm1);
                    lo8[j] = bk[(int)fb];                         // 8 L[bucket]
                }
#pragma unroll
            for (int sp = S2 - 1; sp >= 0; --sp)
#pragma unroll
                for (int j = 0; j < KD; ++j)
                    if (j < KM) {
                        const double xv = *reinterpret_cast<const double *>(
                            reinterpret_cast<const char *>(xs) + lo8[j] + ((8 << sp) - 8));
                        lo8[j] += (xv <= s.cpx[j]) ? (8 << sp) : 0;
                    }
        } else if (STEPS > 0) {
#pragma unroll
            for (int st = STEPS - 1; st >= 0; --st)
#pragma unroll
                for (int j = 0; j < KD; ++j)
                    if (j < KM) {
                        const double xv = *reinterpret_cast<const double *>(
                            reinterpret_cast<const char *>(xs) + lo8[j] + ((8 << st) - 8));
                        lo8[j] += (xv <= s.cpx[j]) ? (8 << st) : 0;
                    }
        } else {
            for (int st = b.ss_steps - 1; st >= 0; --st) {
                const int step8 = 8 << st;
#pragma unroll
                for (int j = 0; j < KD; ++j)
                    if (j < KM) {
                        const double xv = *reinterpret_cast<const double *>(
                            reinterpret_cast<const char *>(xs) + lo8[j] + step8 - 8);
                        lo8[j] += (xv <= s.cpx[j]) ? step8 : 0;
                    }
            }
        }
        const double bn = (double)b.n, pn = P1[b.n];
        double acc = 0.0, bprev = 0.0, pprev = 0.0;
#pragma unroll
        for (int pz = 0; pz < KP; ++pz) {
            // (all KP pieces, as log_lik_ss: beyond the capacity cpv is 1 and the run is empty)
            double bcur = bn, pcur = pn;
            if (pz < KD) {
                const int b8 = lo8[pz < KD ? pz : 0];
                const bool in = pz < s.k;   // boundaries from k on are clamped to (n, P1[n]): the pieces beyond k are empty
                bcur = in ? (double)(b8 >> 3) : bn;
                pcur = in ? *reinterpret_cast<const double *>(reinterpret_cast<const char *>(P1) + (in ? b8 : 0)) : pn;
            }
            const double vc = s.cpv[pz] - b.ss_ym;
            acc = fma_(vc, fma_(bcur - bprev, vc, -2.0 * (pcur - pprev)), acc);
            bprev = bcur;
            pprev = pcur;
        }
        const double sig2 = s.sig * s.sig;
        const double ss = (b.ss_p2 + acc) / sig2;
        double logL = -0.5 * ((ss + b.n * log_full(sig2)) + b.nlog2pi);
        if (logL != logL) logL = -__builtin_huge_val();
        return logL;
    }

    // all rows of the padded layout by this lane (64 consecutive chains per instruction)
    static RMH_DEV void store1(double *dst, int64_t ld, int64_t c, int KM, const State &s)
    {
#pragma unroll
        for (int j = 0; j < KD; ++j)
            if (j < KM) dst[(int64_t)j * ld + c] = s.cpx[j];
#pragma unroll
        for (int j = 0; j < KP; ++j)
            if (j <= KM) dst[(int64_t)(KM + j) * ld + c] = s.cpv[j];
        dst[(int64_t)(2 * KM + 1) * ld + c] = s.sig;
        dst[(int64_t)(2 * KM + 2) * ld + c] = (double)s.k;
    }

    static RMH_DEV void load(const double *theta, int64_t ld, int64_t c, int KM, State &s)
    {
#pragma unroll
        for (int j = 0; j < KD; ++j) s.cpx[j] = (j < KM) ? theta[(int64_t)j * ld + c] : __builtin_huge_val();
#pragma unroll
        for (int j = 0; j < KP; ++j) s.cpv[j] = (j <= KM) ? theta[(int64_t)(KM + j) * ld + c] : 1.0;
        s.sig = theta[(int64_t)(2 * KM + 1) * ld + c];
        s.k = (int)theta[(int64_t)(2 * KM + 2) * ld + c];
    }

    // rows of the padded layout; lane g stores rows = g (mod 4)
    static RMH_DEV void store(double *dst, int64_t ld, int64_t c, int KM, int g, const State &s)
    {
#pragma unroll
        for (int j = 0; j < KD; ++j)
            if (j < KM && (j & 3) == g) dst[(int64_t)j * ld + c] = s.cpx[j];
#pragma unroll
        for (int j = 0; j < KP; ++j)
            if (j <= KM && ((KM + j) & 3) == g) dst[(int64_t)(KM + j) * ld + c] = s.cpv[j];
        if (((2 * KM + 1) & 3) == g) dst[(int64_t)(2 * KM + 1) * ld + c] = s.sig;
        if (((2 * KM + 2) & 3) == g) dst[(int64_t)(2 * KM + 2) * ld + c] = (double)s.k;
    }
};

// SS < 0: the O(n k) likelihood over the data in LDS; SS >= 0: from sufficient statistics
// (CpRj::log_lik_ss; SS > 0 = the search depth as a compile-time constant, 0 = m.base.ss_steps)
template <int KD, int SS>
__global__ void __launch_bounds__(256, (KD <= 6 ? 3 : 2)) k_cprj_run(CpRjArgs m, ChainArgs ch, RunArgs run)
{
    using R = CpRj<KD>;
    constexpr int KP = KD + 1;
    constexpr int BS = 2 * (KD | 1);               // doubles per chain row of the (B, P1[B]) exchange
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int KM = m.KM, Dn = 2 * KM + 2, n = m.base.n;
    const int np = (1 << m.base.ss_steps) - 1;     // SS >= 0: sorted x padded to np entries, then P1 (np + 1)
    const int ndata = SS >= 0 ? 2 * np + 2 : 2 * n;
    double *xy = lds;                              // x[n], y[n]  (SS >= 0: xs[np], P1[np + 1])
    double *ktab = lds + ndata;                    // (KM+2) x 4
    double *tab = ktab + 4 * (KM + 2);             // inverse-CDF table (16-byte aligned: n even or padded by the host)
    double *xi_all = tab + kIcdfDoubles;           // per wave: 16 chains x 32 coordinates
    double *b_all = xi_all + 4 * 16 * 32;          // SS >= 0: per wave 16 chains x BS
    if (SS >= 0)
        for (int i = threadIdx.x; i < 2 * np + 1; i += blockDim.x) xy[i] = m.base.ss_xp[i];
    else
        for (int i = threadIdx.x; i < 2 * n; i += blockDim.x) xy[i] = m.base.xy[i];
    for (int i = threadIdx.x; i < 4 * (KM + 2); i += blockDim.x) ktab[i] = m.ktab[i];
    icdf_stage(tab, ch.icdf);
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cl = lane & 15, g = lane >> 4;
    const int64_t blk = (int64_t)blockIdx.x * (blockDim.x >> 6) + wave;
    if (blk >= ch.nblocks) return;
    const int64_t c = blk * 16 + cl, slot = blk * 64 + lane;
    double *xw = xi_all + wave * (16 * 32) + cl * 32;   // this chain's normals by coordinate
    double *brow = b_all + wave * (16 * BS) + cl * BS;

    typename R::State th;
    R::load(ch.theta, ch.ld, c, KM, th);
    double lp = ch.logpost[c];
    Xo rng = xo_load(ch.rng, ch.nslots, slot);
    const int rows = 4 * ch.ks;

    int64_t knext = run.nburn, kentry = 0;
    auto rec = [&](int64_t e) {
        if (e != knext) return;
        knext += run.nthin;
        const int64_t kk = kentry++;
        if (!run.h_theta) return;
        R::store(run.h_theta + kk * (int64_t)rows * ch.ld, ch.ld, c, KM, g, th);
        if (g == 0) run.h_logpost[kk * ch.ld + c] = lp;
    };
    rec(0);

    for (int64_t t = 0; t < run.T; ++t) {
        asm volatile("" ::: "memory");
        // normals of the padded coordinates: the halves of word mm of sub-stream g are the
        // coordinates 8 mm + g and 8 mm + 4 + g
        for (int mm = 0; mm < (Dn + 7) / 8; ++mm) { // the normals cover the 2 KM + 2 padded coordinates
            const uint64_t w = xo_next(rng);
            const uint32_t hw[2] = {(uint32_t)(w >> 32), (uint32_t)w};
            double zw[2];
            normals_lockstep<2>(hw, tab, zw);
            const double zh = zw[0], zl = zw[1];
            if (8 * mm + g < Dn) xw[8 * mm + g] = zh;
            if (8 * mm + 4 + g < Dn) xw[8 * mm + 4 + g] = zl;
        }
        const double w1 = u01(xo_next(rng)), w2 = u01(xo_next(rng));
        double g1[4], g2[4];
        gather4(w1, g1);
        gather4(w2, g2);
        const double uacc = g1[0], U1 = g1[1], U2 = g1[2], U3 = g1[3];
        const double U4 = g2[0], us = g2[1], uu = g2[2], un = g2[3];
        __builtin_amdgcn_wave_barrier(); // the wave's LDS writes above are read below (same wave: in order)

        typename R::State tp = th;
        double logq = 0.0;
        int mv;
        const int k = th.k;
        if (U1 < m.pcum[0]) { // cpx block, test_changepoint.py:47-49
            mv = 0;
            const double sx = ktab[4 * (k + 1) + 3];
#pragma unroll
            for (int j = 0; j < KD; ++j)
                if (j < k) tp.cpx[j] = th.cpx[j] + sx * xw[j];
        } else if (U2 < m.pcum[1]) { // cpv block, :50-52
            mv = 1;
#pragma unroll
            for (int j = 0; j < KP; ++j)
                if (j <= k) tp.cpv[j] = th.cpv[j] + m.sv * xw[KM + j];
        } else if (U3 < m.pcum[2]) { // sig, :53-55
            mv = 2;
            tp.sig = th.sig + m.ss * xw[2 * KM + 1];
        } else if (k == 0 || U4 > 0.5) { // birth, :58-63, changepoint.py:193-218
            mv = 3;
            const double s = m.base.xmin + (m.base.xmax - m.base.xmin) * us;
            const double u = 0.5 + (-0.1 + 0.2 * uu) / sqrt((double)n);
            if (k >= KM) {
                mv = 5; // no room: refused
            } else {
                int nn = 0;
#pragma unroll
                for (int j = 0; j < KD; ++j) nn += (th.cpx[j] < s) ? 1 : 0; // +inf beyond k
                const double h = R::sel(th.cpv, nn), ufac = sqrt((1.0 - u) / u);
                const double h1 = h / ufac, h2 = h * ufac;
#pragma unroll
                for (int j = 0; j < KD; ++j)
                    tp.cpx[j] = (j < nn) ? th.cpx[j] : (j == nn) ? s : th.cpx[j > 0 ? j - 1 : 0];
#pragma unroll
                for (int j = 0; j < KP; ++j)
                    tp.cpv[j] = (j < nn) ? th.cpv[j] : (j == nn) ? h1 : (j == nn + 1) ? h2 : th.cpv[j > 0 ? j - 1 : 0];
                tp.k = k + 1;
                logq = log_full(__builtin_fabs(h / (u * (1.0 - u))));
            }
        } else { // death, :65-71, changepoint.py:220-240
            mv = 4;
            int nn = (int)(un * k);
            nn = nn >= k ? k - 1 : nn;
            const double h1 = R::sel(th.cpv, nn), h2 = R::sel(th.cpv, nn + 1);
            const double h = sqrt(h1 * h2), u = 1.0 / (1.0 + h2 / h1);
#pragma unroll
            for (int j = 0; j < KD; ++j)
                tp.cpx[j] = (j < nn) ? th.cpx[j] : (j + 1 < KD ? th.cpx[j + 1 < KD ? j + 1 : 0] : __builtin_huge_val());
#pragma unroll
            for (int j = 0; j < KP; ++j)
                tp.cpv[j] = (j < nn) ? th.cpv[j] : (j == nn) ? h : (j + 1 < KP ? th.cpv[j + 1 < KP ? j + 1 : 0] : 1.0);
            tp.k = k - 1;
            logq = log_full(__builtin_fabs(u * (1.0 - u) / h));
        }
        double lpp = -__builtin_huge_val();
        if (mv != 5) {
            const double logP = R::log_prior1(m, ktab, tp, KM);
            if (__any(logP > -__builtin_huge_val())) {
                const double logL = SS >= 0 ? R::template log_lik_ss<SS>(xy, xy + np, brow, m, tp, g)
                                            : R::template log_lik<true>(xy, m, tp, g);
                lpp = combine_post(logP, logL);
            }
        }
        const bool ok = mh_accept(lpp, lp, logq, uacc);
        if (ok) {
            th = tp;
            lp = lpp;
        }
        // accept byte: bit 0 accepted, bits 1..3 the move (0 cpx, 1 cpv, 2 sig, 3 birth, 4 death, 5 refused)
        if (run.h_accept && g == 0) run.h_accept[t * ch.ld + c] = (uint8_t)((ok ? 1 : 0) | (mv << 1));
        rec(t + 1);
    }
    R::store(ch.theta, ch.ld, c, KM, g, th);
    if (g == 0) ch.logpost[c] = lp;
    xo_store(ch.rng, ch.nslots, slot, rng);
}

// ---- the trans-dimensional sampler with ONE lane per chain (round 3) --------------------------
// k_cprj_run keeps a chain in four lanes and replicates the move logic -- five divergent branches that
// rebuild the padded state through static-index selects -- in all of them.  Only the draws need four
// lanes per chain (the stream's four sub-streams).  As k_cp_rw_ss64: a wavefront owns 64 chains, draws in
// four passes of 16 chains x 4 sub-streams (normals of the 2 KM + 2 padded coordinates, then the two
// uniform words of every sub-stream), parks them in an LDS row per chain, and lane l alone carries chain l
// through the move, the prior, the KM boundary searches, the likelihood, the accept test and the record.
// Same stream, same moves, same accept rule, the same arithmetic as the four-lane kernel (CpRj::
// log_prior1 is shared, log_lik_ss1 sums the pieces as log_lik_ss does): a shard on one kernel equals
// those chains of a job on the other bit for bit.
template <int KD, int SS, int S2 = 0>
__global__ void __launch_bounds__(256) k_cprj_run64(CpRjArgs m, ChainArgs ch, RunArgs run)
{
    using R = CpRj<KD>;
    constexpr int KP = KD + 1;
    static_assert(SS >= 0, "sufficient statistics only");
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int KM = m.KM, Dn = 2 * KM + 2, n = m.base.n;
    const int np = (1 << m.base.ss_steps) - 1;
    const int TS = 2 * (((Dn + 9) / 2) | 1);       // row: Dn normals + 8 uniforms, an odd number of 16-byte units
    const int bkd = S2 > 0 ? (((m.base.ss_nb + 2) / 2 + 1) & ~1) : 0;   // doubles of the bucket table (int32 entries)
    double *xs = lds, *P1 = lds + np;
    const int32_t *bk = reinterpret_cast<const int32_t *>(lds + 2 * np + 2);
    double *ktab = lds + 2 * np + 2 + bkd;         // (KM+2) x 4
    double *tab = ktab + 4 * (KM + 2);
    double *xch = tab + kIcdfDoubles;              // per wave: 64 rows of TS doubles
    for (int i = threadIdx.x; i < 2 * np + 1; i += blockDim.x) lds[i] = m.base.ss_xp[i];
    if (S2 > 0)
        for (int i = threadIdx.x; i < (m.base.ss_nb + 2) / 2; i += blockDim.x)
            lds[2 * np + 2 + i] = m.base.ss_xp[2 * np + 2 + i];
    for (int i = threadIdx.x; i < 4 * (KM + 2); i += blockDim.x) ktab[i] = m.ktab[i];
    icdf_stage(tab, ch.icdf);
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cl = lane & 15, g = lane >> 4;
    const int64_t grp = (int64_t)blockIdx.x * (blockDim.x >> 6) + wave;   // group of 64 chains
    const int64_t blk0 = grp * 4;
    if (blk0 >= ch.nblocks) return;
    double *rows = xch + wave * (64 * TS);
    const double *xw = rows + lane * TS;           // this lane's chain: normals by coordinate, then the uniforms
    const int64_t c = grp * 64 + lane;
    const bool live = c < ch.ld;
    const int64_t cs = live ? c : ch.ld - 1;       // (a lane beyond the padded chains shadows the last one)

    typename R::State th;
    R::load(ch.theta, ch.ld, cs, KM, th);
    double lp = ch.logpost[cs];
    Xo st[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const int64_t slot = ((blk0 + b < ch.nblocks) ? blk0 + b : blk0) * 64 + lane;
        st[b] = xo_load(ch.rng, ch.nslots, slot);
    }
    const int rowsH = 4 * ch.ks;

    int64_t knext = run.nburn, kentry = 0;
    auto rec = [&](int64_t e) {
        if (e != knext) return;
        knext += run.nthin;
        const int64_t kk = kentry++;
        if (!run.h_theta || !live) return;
        R::store1(run.h_theta + kk * (int64_t)rowsH * ch.ld, ch.ld, c, KM, th);
        run.h_logpost[kk * ch.ld + c] = lp;
    };
    rec(0);
    // everything loaded so far has arrived before the loop starts (see `arrived`)
#pragma unroll
    for (int j = 0; j < KD; ++j) arrived(th.cpx[j]);
#pragma unroll
    for (int j = 0; j < KP; ++j) arrived(th.cpv[j]);
    arrived(th.sig);
    arrived(lp);
#pragma unroll
    for (int b = 0; b < 4; ++b) arrived(st[b]);

    const int MW = (Dn + 7) / 8;
    for (int64_t t = 0; t < run.T; ++t) {
        asm volatile("" ::: "memory");
        // ---- the draws: pass b = chains 16 b + cl of the group, sub-stream g.  One wave per SIMD has
        // nobody to hide latencies behind: the four passes' xoshiro steps of a word side by side, the
        // normals of two passes (four half words) in lockstep
        double *row0 = rows + cl * TS;
        for (int mm = 0; mm < MW; ++mm) {
            uint64_t w[4];
#pragma unroll
            for (int b = 0; b < 4; ++b) w[b] = xo_next(st[b]);
            const bool hi_in = 8 * mm + g < Dn, lo_in = 8 * mm + 4 + g < Dn;
#pragma unroll
            for (int b0 = 0; b0 < 4; b0 += 2) {
                const uint32_t hw[4] = {(uint32_t)(w[b0] >> 32), (uint32_t)w[b0], (uint32_t)(w[b0 + 1] >> 32),
                                        (uint32_t)w[b0 + 1]};
                double zw[4];
                normals_lockstep<4>(hw, tab, zw);
                double *ra = row0 + 16 * b0 * TS + 8 * mm + g, *rb = ra + 16 * TS;
                if (hi_in) {
                    ra[0] = zw[0];
                    rb[0] = zw[2];
                }
                if (lo_in) {
                    ra[4] = zw[1];
                    rb[4] = zw[3];
                }
            }
        }
        {
            double u1[4], u2[4];
#pragma unroll
            for (int b = 0; b < 4; ++b) u1[b] = u01(xo_next(st[b]));       // (accept, U1, U2, U3)
#pragma unroll
            for (int b = 0; b < 4; ++b) u2[b] = u01(xo_next(st[b]));       // (U4, location, height split, index)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                row0[16 * b * TS + Dn + g] = u1[b];
                row0[16 * b * TS + Dn + 4 + g] = u2[b];
            }
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the wave's own LDS writes are in order
        const double uacc = xw[Dn], U1 = xw[Dn + 1], U2 = xw[Dn + 2], U3 = xw[Dn + 3];
        const double U4 = xw[Dn + 4], us = xw[Dn + 5], uu = xw[Dn + 6], un = xw[Dn + 7];

        typename R::State tp = th;
        double logq = 0.0;
        int mv;
        const int k = th.k;
        if (U1 < m.pcum[0]) { // cpx block, test_changepoint.py:47-49
            mv = 0;
            const double sx = ktab[4 * (k + 1) + 3];
#pragma unroll
            for (int j = 0; j < KD; ++j)
                if (j < KM) tp.cpx[j] = (j < k) ? th.cpx[j] + sx * xw[j] : th.cpx[j];
        } else if (U2 < m.pcum[1]) { // cpv block, :50-52
            mv = 1;
#pragma unroll
            for (int j = 0; j < KP; ++j)
                if (j <= KM) tp.cpv[j] = (j <= k) ? th.cpv[j] + m.sv * xw[KM + j] : th.cpv[j];
        } else if (U3 < m.pcum[2]) { // sig, :53-55
            mv = 2;
            tp.sig = th.sig + m.ss * xw[2 * KM + 1];
        } else if (k == 0 || U4 > 0.5) { // birth, :58-63, changepoint.py:193-218
            mv = 3;
            const double s = m.base.xmin + (m.base.xmax - m.base.xmin) * us;
            const double u = 0.5 + (-0.1 + 0.2 * uu) / sqrt((double)n);
            if (k >= KM) {
                mv = 5; // no room: refused
            } else {
                int nn = 0;
#pragma unroll
                for (int j = 0; j < KD; ++j) nn += (th.cpx[j] < s) ? 1 : 0; // +inf beyond k
                const double h = R::sel(th.cpv, nn), ufac = sqrt((1.0 - u) / u);
                const double h1 = h / ufac, h2 = h * ufac;
#pragma unroll
                for (int j = 0; j < KD; ++j)
                    tp.cpx[j] = (j < nn) ? th.cpx[j] : (j == nn) ? s : th.cpx[j > 0 ? j - 1 : 0];
#pragma unroll
                for (int j = 0; j < KP; ++j)
                    tp.cpv[j] = (j < nn) ? th.cpv[j] : (j == nn) ? h1 : (j == nn + 1) ? h2 : th.cpv[j > 0 ? j - 1 : 0];
                tp.k = k + 1;
                logq = log_full(__builtin_fabs(h / (u * (1.0 - u))));
            }
        } else { // death, :65-71, changepoint.py:220-240
            mv = 4;
            int nn = (int)(un * k);
            nn = nn >= k ? k - 1 : nn;
            const double h1 = R::sel(th.cpv, nn), h2 = R::sel(th.cpv, nn + 1);
            const double h = sqrt(h1 * h2), u = 1.0 / (1.0 + h2 / h1);
#pragma unroll
            for (int j = 0; j < KD; ++j)
                tp.cpx[j] = (j < nn) ? th.cpx[j] : (j + 1 < KD ? th.cpx[j + 1 < KD ? j + 1 : 0] : __builtin_huge_val());
#pragma unroll
            for (int j = 0; j < KP; ++j)
                tp.cpv[j] = (j < nn) ? th.cpv[j] : (j == nn) ? h : (j + 1 < KP ? th.cpv[j + 1 < KP ? j + 1 : 0] : 1.0);
            tp.k = k - 1;
            logq = log_full(__builtin_fabs(u * (1.0 - u) / h));
        }
        double lpp = -__builtin_huge_val();
        double logP = -__builtin_huge_val();
        if (mv != 5) logP = R::log_prior1(m, ktab, tp, KM);
        if (__any(logP > -__builtin_huge_val())) {
            const double logL = R::template log_lik_ss1<SS, S2>(xs, P1, bk, m, tp, KM);
            lpp = combine_post(logP, logL);
        }
        const bool ok = mh_accept(lpp, lp, logq, uacc);
        if (ok) {
            th = tp;
            lp = lpp;
        }
        // accept byte: bit 0 accepted, bits 1..3 the move (0 cpx, 1 cpv, 2 sig, 3 birth, 4 death, 5 refused)
        if (run.h_accept && live) run.h_accept[t * ch.ld + c] = (uint8_t)((ok ? 1 : 0) | (mv << 1));
        rec(t + 1);
        __builtin_amdgcn_wave_barrier();                      // the rows are rewritten by the next draws
    }
    if (live) {
        R::store1(ch.theta, ch.ld, c, KM, th);
        ch.logpost[c] = lp;
    }
#pragma unroll
    for (int b = 0; b < 4; ++b)
        if (blk0 + b < ch.nblocks) xo_store(ch.rng, ch.nslots, (blk0 + b) * 64 + lane, st[b]);
}

template <int KD>
__global__ void __launch_bounds__(256) k_cprj_eval(CpRjArgs m, ChainArgs ch, double *lprior, double *llik,
                                                   double *lpost)
{
    using R = CpRj<KD>;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int n = m.base.n, KM = m.KM;
    double *xy = lds, *ktab = lds + 2 * n;
    for (int i = threadIdx.x; i < 2 * n; i += blockDim.x) xy[i] = m.base.xy[i];
    for (int i = threadIdx.x; i < 4 * (KM + 2); i += blockDim.x) ktab[i] = m.ktab[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cl = lane & 15, g = lane >> 4;
    const int64_t blk = (int64_t)blockIdx.x * (blockDim.x >> 6) + wave;
    if (blk >= ch.nblocks) return;
    const int64_t c = blk * 16 + cl;
    typename R::State th;
    R::load(ch.theta, ch.ld, c, KM, th);
    th.k = th.k < 0 ? 0 : th.k > KM ? KM : th.k;
    const double logP = R::log_prior1(m, ktab, th, KM);
    const double logL = R::template log_lik<false>(xy, m, th, g);
    if (g == 0) {
        if (lprior) lprior[c] = logP;
        if (llik) llik[c] = logL;
        if (lpost) lpost[c] = combine_post(logP, logL);
    }
}

hipError_t launch_cprj_run(const CpRjArgs &m, const ChainArgs &c, const RunArgs &r, hipStream_t st)
{
    // likelihood from sufficient statistics when the model has them (n <= 2047); RMH_POLICY_CPRJ_NO_SS
    // keeps the O(n k) loop over the data (measurements, cross-checks)
    const bool ss = m.base.ss_xp && !policy(RMH_POLICY_CPRJ_NO_SS);
    const int np = (1 << m.base.ss_steps) - 1;
    // one lane per chain once every SIMD of the chip gets a wavefront of 64 chains (as k_cp_rw_ss64)
    if (ss && m.KM <= 14 && (policy(RMH_POLICY_CP_SS64) || (c.nblocks >= 4096 && !policy(RMH_POLICY_CP_SS16)))) {
        const int Dn = 2 * m.KM + 2, TS = 2 * (((Dn + 9) / 2) | 1), nw = 4;
        const bool buckets = m.base.ss_nb > 0 && (m.base.ss_s2 == 2 || m.base.ss_s2 == 3) && !policy(RMH_POLICY_CP_NO_BUCKETS);
        const size_t bkd = buckets ? (size_t)(((m.base.ss_nb + 2) / 2 + 1) & ~1) : 0;
        const size_t lds64 = sizeof(double) * ((size_t)2 * np + 2 + bkd + 4 * (m.KM + 2) + kIcdfDoubles + (size_t)nw * 64 * TS);
        if (lds64 <= 160 * 1024) {
            const int64_t ngrp = (c.nblocks + 3) / 4;
            const dim3 grid64((unsigned)((ngrp + nw - 1) / nw));
#define RMH_CPRJ64(...)                                                                                   \
    {                                                                                                     \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_cprj_run64<__VA_ARGS__>),     \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds64);       \
        if (e != hipSuccess) return e;                                                                    \
        hipLaunchKernelGGL((k_cprj_run64<__VA_ARGS__>), grid64, dim3(64 * nw), lds64, st, m, c, r);       \
        return hipGetLastError();                                                                         \
    }
#define RMH_CPRJ64_KD(KD)                                         \
    {                                                             \
        if (buckets && m.base.ss_s2 == 2) RMH_CPRJ64(KD, 0, 2)    \
        if (buckets && m.base.ss_s2 == 3) RMH_CPRJ64(KD, 0, 3)    \
        if (m.base.ss_steps == 7) RMH_CPRJ64(KD, 7, 0)            \
        RMH_CPRJ64(KD, 0, 0)                                      \
    }
            if (m.KM <= 6) RMH_CPRJ64_KD(6)
            if (m.KM <= 10) RMH_CPRJ64_KD(10)   // the reference script's capacity, 2 Ncpx = 10 (test_changepoint.py)
            RMH_CPRJ64_KD(14)
#undef RMH_CPRJ64_KD
#undef RMH_CPRJ64
        }
    }
    const size_t ndata = ss ? (size_t)2 * np + 2 : (size_t)2 * m.base.n;
    const size_t lds = sizeof(double) * (ndata + 4 * (m.KM + 2) + kIcdfDoubles + 4 * 16 * 32 +
                                         (ss ? 4 * 16 * 2 * (14 | 1) : 0));
    const dim3 grid((unsigned)((c.nblocks + 3) / 4)), block(256);
#define RMH_CPRJ_LAUNCH(KD, SS)                                                                      \
    do {                                                                                             \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_cprj_run<KD, SS>),       \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);    \
        if (e != hipSuccess) return e;                                                               \
        hipLaunchKernelGGL((k_cprj_run<KD, SS>), grid, block, lds, st, m, c, r);                     \
    } while (0)
    if (m.KM <= 6) {
        if (!ss) RMH_CPRJ_LAUNCH(6, -1);
        else if (m.base.ss_steps == 7) RMH_CPRJ_LAUNCH(6, 7);
        else RMH_CPRJ_LAUNCH(6, 0);
    } else if (m.KM <= 14) {
        if (!ss) RMH_CPRJ_LAUNCH(14, -1);
        else if (m.base.ss_steps == 7) RMH_CPRJ_LAUNCH(14, 7);
        else RMH_CPRJ_LAUNCH(14, 0);
    } else {
        return hipErrorInvalidValue;
    }
#undef RMH_CPRJ_LAUNCH
    return hipGetLastError();
}

hipError_t launch_cprj_eval(const CpRjArgs &m, const ChainArgs &c, double *lprior, double *llik,
                            double *lpost, hipStream_t st)
{
    const size_t lds = sizeof(double) * ((size_t)2 * m.base.n + 4 * (m.KM + 2));
    const dim3 grid((unsigned)((c.nblocks + 3) / 4)), block(256);
    if (m.KM <= 6) hipLaunchKernelGGL(k_cprj_eval<6>, grid, block, lds, st, m, c, lprior, llik, lpost);
    else if (m.KM <= 14) hipLaunchKernelGGL(k_cprj_eval<14>, grid, block, lds, st, m, c, lprior, llik, lpost);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

#define RMH_CP_DISPATCH(fn, ...)                       \
    switch (k) {                                       \
    case 1: return fn<1>(__VA_ARGS__);                 \
    case 2: return fn<2>(__VA_ARGS__);                 \
    case 3: return fn<3>(__VA_ARGS__);                 \
    case 4: return fn<4>(__VA_ARGS__);                 \
    case 5: return fn<5>(__VA_ARGS__);                 \
    case 6: return fn<6>(__VA_ARGS__);                 \
    case 7: return fn<7>(__VA_ARGS__);                 \
    case 8: return fn<8>(__VA_ARGS__);                 \
    default: return hipErrorInvalidValue;              \
    }

hipError_t launch_cp_rw(int k, const CpArgs &m, const RwArgs &p, const ChainArgs &c,
                        const RunArgs &r, hipStream_t st)
{
    RMH_CP_DISPATCH(cp_rw_launch, m, p, c, r, st)
}

hipError_t launch_cp_rw_ss(int k, const CpArgs &m, const RwArgs &p, const ChainArgs &c,
                           const RunArgs &r, hipStream_t st, bool adapt)
{
    RMH_CP_DISPATCH(cp_rw_ss_launch, m, p, c, r, st, adapt)
}

// the generic fused kernels over the changepoint policy: variant 1 = AdaptScale random walk,
// 2 = pCN, 3 = AdaptScalepCN, 4 = parallel tempering, 5 = pCN within parallel tempering, 0 = plain (used by the covariance adaptation)
template <int KC>
static hipError_t cp_generic_launch(const CpArgs &m, const RwArgs &p, const ChainArgs &c,
                                    const RunArgs &r, int variant, hipStream_t st)
{
    using M = CpPolicy<KC>;
    if (m.n > kCpPolicyMaxN) return hipErrorInvalidValue;
    const size_t lds = sizeof(double) * (tri_tiles(M::NT) * 64 + M::lds_doubles(false) + icdf_doubles(c.nmode));
    const dim3 grid((unsigned)((c.nblocks + 3) / 4)), block(256);
    hipError_t e = hipSuccess;
#define RMH_CP_GEN(...)                                                                          \
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_run_rw<M, 4, __VA_ARGS__>),         \
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);               \
    if (e != hipSuccess) return e;                                                               \
    hipLaunchKernelGGL((k_run_rw<M, 4, __VA_ARGS__>), grid, block, lds, st, m, p, c, r)
    switch (variant) {
    case 0: RMH_CP_GEN(false, false, false); break;
    case 1: RMH_CP_GEN(true, false, false); break;
    case 2: RMH_CP_GEN(false, true, false); break;
    case 3: RMH_CP_GEN(true, true, false); break;
    case 4: RMH_CP_GEN(false, false, true); break;
    case 5: RMH_CP_GEN(false, true, true); break;
    default: return hipErrorInvalidValue;
    }
#undef RMH_CP_GEN
    return hipGetLastError();
}

hipError_t launch_cp_generic(int k, const CpArgs &m, const RwArgs &p, const ChainArgs &c,
                             const RunArgs &r, int variant, hipStream_t st)
{
    RMH_CP_DISPATCH(cp_generic_launch, m, p, c, r, variant, st)
}

hipError_t launch_cp_eval(int k, const CpArgs &m, const ChainArgs &c, double *lprior,
                          double *llik, double *lpost, hipStream_t st)
{
    RMH_CP_DISPATCH(cp_eval_launch, m, c, lprior, llik, lpost, st)
}

} // namespace rmh
