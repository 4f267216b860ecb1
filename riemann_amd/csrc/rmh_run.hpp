// rmh_run.hpp -- the fused Sampler.run kernels over a device Model policy.
//
// One launch = T iterations of Sampler.sample (riemann/samplers/sampler.py:72-90)
// for every chain: propose (proposals/randomwalk.py:21-26 or
// proposals/hamiltonian.py:13-52,76-91), Model.log_posterior
// (models/model.py:43-55), accept/reject (sampler.py:83-87), record
// (sampler.py:62-70 with the [Nburn::Nthin] slice of :53-54).
//
// A workgroup of NW wavefronts owns 16*NW chains; LDS holds the (scaled)
// Cholesky tile image of the proposal and whatever the Model policy needs.
#pragma once
#include "rmh_models.hpp"

namespace rmh {

// History bookkeeping of Sampler.run (riemann/samplers/sampler.py:49-54): state
// number e (0 = state on entry) is kept iff e >= nburn and (e - nburn) % nthin
// == 0, at entry (e - nburn) / nthin.  States arrive in order, so the next kept
// number and its entry are tracked incrementally (no 64-bit division per step).
struct Keep {
    int64_t next, entry;
    RMH_DEV explicit Keep(const RunArgs &run) : next(run.nburn), entry(0) {}

    template <int NT, int KS = 0>   // KS > 0: ch.ks is that constant (Tile::store_rows)
    RMH_DEV void record(const RunArgs &run, const ChainArgs &ch, int64_t e, int64_t c, int g,
                        bool active, const double (&th)[4 * NT], double lp)
    {
        if (e != next) return;
        next += run.nthin;
        const int64_t k = entry++;
        if (!run.h_theta || !active) return;
        double *dst = run.h_theta + k * (int64_t)(4 * ch.ks) * ch.ld;
        Tile<NT>::template store_rows<KS>(dst, ch.ld, c, g, KS > 0 ? KS : ch.ks, th);
        // (raw buffer store: a flat one keeps a 64-bit address per lane alive across the iteration; the
        // lane offset is the state rows' own -- (g ld + c) 8 = 8 c on the lanes that store -- so that no
        // second offset register lives through the loop)
        if (g == 0) buf_store(rsrc(run.h_logpost + k * ch.ld, ch.ld), (uint32_t)(g * ch.ld + c) * 8u, 0u, lp);
    }
};

// v <- T v for a triangular tile image T in global memory (lower, or upper when UPPER), in place
template <int NT, bool UPPER>
RMH_DEV void tri_apply(const double *img, int lane, double (&v)[4 * NT], int ks)
{
    asm volatile("" ::: "memory"); // the image is loop-invariant: keep its loads here (no LICM)
    double4_t acc[1][NT];
    double B[1][4 * NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) acc[0][i] = double4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < 4 * NT; ++s) B[0][s] = v[s];
    tri_stream<NT, UPPER, 1>(img, lane, B, acc, ks);
    Tile<NT>::unpack(acc[0], v);
}

// acc += L xi for the proposal's image in LDS, or in global memory when GLOBAL
template <int NT, bool GLOBAL>
RMH_DEV void mul_lower_any(const double *img, int lane, const double (&B)[4 * NT], double4_t (&acc)[NT], int ks)
{
    if constexpr (GLOBAL) tri_stream1<NT, false>(img, lane, B, acc, ks);
    else Tile<NT>::mul_lower(img + lane, B, acc, ks);
}

// the same for two vectors at once: every slice of the image is loaded once and feeds two MFMAs
template <int NT, bool UPPER>
RMH_DEV void tri_apply2(const double *img, int lane, double (&v0)[4 * NT], double (&v1)[4 * NT], int ks)
{
    asm volatile("" ::: "memory");
    double4_t acc[2][NT];
    double B[2][4 * NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) acc[0][i] = acc[1][i] = double4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < 4 * NT; ++s) {
        B[0][s] = v0[s];
        B[1][s] = v1[s];
    }
    tri_stream<NT, UPPER, 2>(img, lane, B, acc, ks);
    Tile<NT>::unpack(acc[0], v0);
    Tile<NT>::unpack(acc[1], v1);
}

// AdaptScaleProposal (riemann/proposals/adaptive.py:11-35) per chain: the adapt()
// hook of sampler.py:88 after every step.  State in ch.adapt: rows Nsamples,
// Naccepts, scale.  The first call compares with last_theta = None and therefore
// always counts as a move (adaptive.py:27 with :22).
struct AdaptScale {
    double n, acc, scale;
    RMH_DEV void load(const ChainArgs &ch, int64_t c)
    {
        n = ch.adapt[c];
        acc = ch.adapt[ch.ld + c];
        scale = ch.adapt[2 * ch.ld + c];
    }
    RMH_DEV void store(const ChainArgs &ch, int64_t c) const
    {
        ch.adapt[c] = n;
        ch.adapt[ch.ld + c] = acc;
        ch.adapt[2 * ch.ld + c] = scale;
    }
    RMH_DEV void adapt(bool accepted, double target)
    {
        const double moved = (n == 0.0 || accepted) ? 1.0 : 0.0;
        n += 1.0;
        acc += moved;
        const double ratio = exp(1.0 / n);
        scale = (acc / n > target) ? scale * ratio : scale / ratio;
    }
};

// One lane's share of a parallel-tempering sweep (PTSampler, samplers/ptsampler.py:42-127;
// layout and rules in the header of k_run_rw below).  Shared by the random-walk kernels.
struct PtLane {
    int rung = 0, ebase = 0;
    bool valid = false, init = false, partner = false;
    double beta = 1.0, sP = 0.0, sL = 0.0, vswap = 0.0;

    RMH_DEV void setup(const RunArgs &run, int cl, int64_t c)
    {
        const int K = run.pt_K;
        rung = cl % K;
        ebase = cl - rung;
        valid = cl < (16 / K) * K;
        beta = run.pt_beta[c];
    }
    // log posterior of the tempered model at (prior, likelihood) parts (ptsampler.py:35-39)
    RMH_DEV double tempered(const double (&parts)[2]) const { return combine_post(parts[0], beta * parts[1]); }
    // first words of the sweep: selector (sub-stream 0) and swap uniform (sub-stream 1), then the
    // sequential "already swapped" rule of ptsampler.py:104-107 from one ballot
    RMH_DEV void select(const RunArgs &run, Xo &rng, int cl)
    {
        const double uw = u01(xo_next(rng));
        const double usel = __shfl(uw, cl);
        vswap = __shfl(uw, 16 + cl);
        // ptsampler.py:107: within-chain iff u > Pswap or last rung
        const bool wants = valid && rung < run.pt_K - 1 && !(usel > run.pt_pswap);
        const unsigned wm = (unsigned)(__ballot(wants) & 0xFFFFu) >> ebase;
        bool prev = false;
        init = partner = false;
        for (int j = 0; j < run.pt_K; ++j) { // :104-106, rungs in order
            const bool in = ((wm >> j) & 1u) && !prev;
            if (j == rung) {
                init = in;
                partner = prev;
            }
            prev = in;
        }
    }
    RMH_DEV bool within(bool acc_ok) const { return acc_ok && valid && !init && !partner; }
    // after the within-chain step: swap proposal between rung i (initiator) and i+1,
    // ptsampler.py:110-125; returns the accept code of this slot
    template <int S>
    RMH_DEV int swap(int lane, bool acc_ok, const double (&parts)[2], double (&th)[S], double &lp)
    {
        sP = acc_ok ? parts[0] : sP;
        sL = acc_ok ? parts[1] : sL;
        const bool swapper = init || partner;
        const int src = init ? lane + 1 : partner ? lane - 1 : lane;
        const double oP = __shfl(sP, src), oL = __shfl(sL, src), ob = __shfl(beta, src);
        const double olp = __shfl(lp, src), ov = __shfl(vswap, src);
        const double cross_me = combine_post(oP, beta * oL);  // my model at the other state
        const double cross_ot = combine_post(sP, ob * sL);    // the other model at my state
        // (logpost_ji + logpost_ij) - (logpost_ii + logpost_jj): the same bits on both lanes
        const double delta = (cross_ot + cross_me) - (init ? lp + olp : olp + lp);
        // U < min(1, exp(D)) restated as log(U) < min(0, D)  (the initiator's uniform)
        const bool sw_ok = swapper && mh_accept(delta, 0.0, 0.0, partner ? ov : vswap);
        if (__any(sw_ok)) {
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const double o = __shfl(th[s], src);
                th[s] = sw_ok ? o : th[s];
            }
        }
        sP = sw_ok ? oP : sP;
        sL = sw_ok ? oL : sL;
        lp = sw_ok ? cross_me : lp;
        return swapper ? (sw_ok ? 3 : 2) : (acc_ok ? 1 : 0);
    }
};

// ------------------------------------------------------------- random walk --
// On MI355X v_mfma_f64 and the VALU do not overlap across wavefronts
// (tools/microbench2.hip): a second resident wave cannot hide VALU work behind
// MFMAs, it only covers s_waitcnt stalls.  The chain state, the candidate and the
// normals all stay in registers (<= 256 at two waves per SIMD) and the only HBM
// traffic of an iteration is the recorded sample.
// PCN: preconditioned Crank-Nicolson (proposals/randomwalk.py:78-101):
//   theta' = rho theta + rho_c L xi;  logqratio = -(|u_fwd|^2 - |u_rev|^2)/2 with
//   u = (rho_c L)^-1 dtheta for dtheta_fwd = theta' - rho theta, dtheta_rev = theta - rho theta'.
//
// PT: parallel tempering (PTSampler, samplers/ptsampler.py:42-127).  A wavefront's 16
// chain slots hold 16/K ensembles of K replicas, rung-minor (slots >= K*(16/K) idle);
// replica of rung i samples TemperedModel(model, beta_i) (:11-39).  A sweep (:100-125)
// = one selector uniform per rung (sub-stream 0) deciding between the within-chain MH
// step and a swap proposal with the next rung (swap uniform: sub-stream 1); the
// sequential "already swapped" rule (:104-106) is resolved per ensemble from one
// ballot.  An accepted swap exchanges the two replicas' registers with ds_bpermute
// (neighbouring lanes of the same 16-lane row): no memory traffic.
// LGLOBAL: the proposal's tile image is read from global memory (L2) instead of LDS
// (models whose own LDS use leaves no room for it: the logistic regression).
// FULL: d = 16 NT exactly (Gaussian model).  The dimension's k-step and word counts are then compile-time
// constants: no wave-uniform test in front of the last four k-steps of every triangular product, no
// coordinate selects in the normals (as the VALU-tail kernels have it for d = 16 NF + 1..4).
// KSC (last parameter; FULL = true is KSC = 4 NT): the k-step count alone as a compile-time constant, e.g. 4 (NT - 1) + 1
// at d = 16 (NT - 1) + 1 .. 4 for the models and proposals that have no VALU-tail kernel (launch_rw_ks, rmh_launch.hpp)
template <class M, int NW, bool ADAPT, bool PCN = false, bool PT = false, bool LGLOBAL = false, bool FULL = false, int KSC = 0>
__global__ void __launch_bounds__(64 * NW) k_run_rw(typename M::Args m_in, RwArgs pr, ChainArgs ch_in, RunArgs run)
{
    constexpr int NT = M::T::S / 4, S = M::T::S;
    using T = Tile<NT>;
    static_assert(!(FULL && KSC > 0), "FULL is KSC = 4 NT with d and the word count on top");
    typename M::Args m = m_in;
    ChainArgs ch = ch_in;
    if constexpr (FULL) {
        ch.d = 4 * S;
        ch.ks = S;
        ch.M = S / 2;
        m.ks = S;
    } else if constexpr (KSC > 0) {
        ch.ks = KSC;
        m.ks = KSC;
    }
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const double *Limg = LGLOBAL ? pr.Ltiles : lds;
    double *mlds = lds + (LGLOBAL ? 0 : tri_tiles(NT) * 64);
    double *tab = mlds + M::lds_doubles(false);
    if (!LGLOBAL)
        for (int i = threadIdx.x; i < tri_tiles(NT) * 64; i += blockDim.x) lds[i] = pr.Ltiles[i];
    M::setup(mlds, m, false);
    icdf_stage(tab, ch.icdf, ch.nmode);
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cl = lane & 15, g = lane >> 4;
    int64_t blk = (int64_t)blockIdx.x * (blockDim.x >> 6) + wave;
    const bool active = blk < ch.nblocks;
    if (!active) blk = ch.nblocks - 1;
    const int64_t c = blk * 16 + cl;
    const int64_t slot = blk * 64 + lane;

    double th[S];
    T::load_state(ch.theta, ch.ld, c, g, th);
    double lp = ch.logpost[c];
    Xo rng = xo_load(ch.rng, ch.nslots, slot);
    PtLane pt;
    if (PT) {
        pt.setup(run, cl, c);
        double dummy0[S], parts0[2], lp0;
        M::template eval<false>(mlds, m, false, lane, th, lp0, dummy0, parts0);
        pt.sP = parts0[0];
        pt.sL = parts0[1];
        lp = pt.tempered(parts0);   // Sampler.__init__ of the rung, sampler.py:42
    }
    Keep keep(run);
    keep.record<NT>(run, ch, 0, c, g, active, th, lp);
    AdaptScale ad{0.0, 0.0, 1.0};
    if (ADAPT) ad.load(ch, c);
    // AdaptScalepCN (randomwalk.py:104-119): rho <- tanh(rho / scale) before every proposal,
    // a recursion on the chain's own rho (4th row of the adaptation state; NaN = not started);
    // rho_c and with it the tile image keep their initial value, as in the reference
    double rho = pr.rho;
    if (PCN && ADAPT) {
        const double aux = ch.adapt[3 * ch.ld + c];
        rho = (aux != aux) ? pr.rho : aux;
    }

    for (int64_t t = 0; t < run.T; ++t) {
        // the tile images never change, so without this fence LICM hoists their
        // ds_reads out of the loop and spends the whole register file on them
        asm volatile("" ::: "memory");
        if (PT) pt.select(run, rng, cl);
        double xi[S];
        T::normals_mode(ch.nmode, rng, ch.M, ch.ks, ch.d, g, tab, xi);
        // theta' = theta + scale * L xi        (randomwalk.py:26; logqratio = 0.0)
        double4_t acc[NT];
        double thp[S], lpp, dummy[S], parts[2], logq = 0.0;
        if (PCN) {
            if (ADAPT) rho = tanh(rho / ad.scale);       // randomwalk.py:118
#pragma unroll
            for (int s = 0; s < S; ++s) thp[s] = rho * th[s];
            T::pack(thp, acc);
            mul_lower_any<NT, LGLOBAL>(Limg, lane, xi, acc, ch.ks);   // image holds rho_c * L
            T::unpack(acc, thp);
            double ufw[S], urv[S];
#pragma unroll
            for (int s = 0; s < S; ++s) {
                ufw[s] = thp[s] - rho * th[s];           // randomwalk.py:95-96
                urv[s] = th[s] - rho * thp[s];
            }
            tri_apply2<NT, false>(pr.Vtiles, lane, ufw, urv, ch.ks);
            double sf = 0.0, sr = 0.0;
#pragma unroll
            for (int s = 0; s < S; ++s) {
                sf += ufw[s] * ufw[s];
                sr += urv[s] * urv[s];
            }
            logq = -0.5 * (sum4(sf) - sum4(sr));         // :99-100
        } else if (ADAPT && !PCN) { // the chain's own adapted scale: theta + scale * (L xi), image unscaled
#pragma unroll
            for (int i = 0; i < NT; ++i) acc[i] = double4_t{0.0, 0.0, 0.0, 0.0};
            mul_lower_any<NT, LGLOBAL>(Limg, lane, xi, acc, ch.ks);
            T::unpack(acc, thp);
#pragma unroll
            for (int s = 0; s < S; ++s) thp[s] = th[s] + ad.scale * thp[s];
        } else {
            T::pack(th, acc);
            mul_lower_any<NT, LGLOBAL>(Limg, lane, xi, acc, ch.ks);
            T::unpack(acc, thp);
        }
        M::template eval<false>(mlds, m, false, lane, thp, lpp, dummy, parts);
        if (PT) lpp = pt.tempered(parts); // TemperedModel, ptsampler.py:35-39
        const double u = T::accept_uniform(rng, g, cl);
        bool acc_ok = mh_accept(lpp, lp, logq, u);
        if (PT) acc_ok = pt.within(acc_ok);
#pragma unroll
        for (int s = 0; s < S; ++s) th[s] = acc_ok ? thp[s] : th[s];
        lp = acc_ok ? lpp : lp;
        int code = acc_ok ? 1 : 0;
        if (PT) code = pt.swap(lane, acc_ok, parts, th, lp);
        if (ADAPT) ad.adapt(acc_ok, ch.adapt_target);
        if (run.h_accept && active && g == 0) run.h_accept[t * ch.ld + c] = (uint8_t)code;
        keep.record<NT>(run, ch, t + 1, c, g, active, th, lp);
    }
    if (active) {
        T::store_state(ch.theta, ch.ld, c, g, th);
        if (g == 0) ch.logpost[c] = lp;
        if (ADAPT && g == 0) ad.store(ch, c);
        if (PCN && ADAPT && g == 0) ch.adapt[3 * ch.ld + c] = rho;
        xo_store(ch.rng, ch.nslots, slot, rng);
    }
}

// -------------------------------------- random walk, d = 97..100 (BASELINE cfg 2) --
// With 16-row tiles a 100-dimensional triangular product spends 25 of its 109 MFMAs on the
// seventh row tile, which holds 4 real rows.  v_mfma_f64 and the VALU share the FP64 units on
// this chip (rmh_run.hpp header), so those 4 rows are cheaper as plain FMAs: every lane forms
// the partial dot products of rows 96..99 over its own 25 coordinates, the chain's four lanes
// add them up (sum4_scatter), and lane group g keeps row 96+g -- exactly the register the MFMA
// would have written (tile 6, register 0).  1600 MFMA cycles become ~440 VALU cycles per product.
// MetropolisRandomWalk on MultiGaussianDist: plain, with the per-chain adapted scale (ADAPT) or
// within parallel tempering (PT); same results as k_run_rw up to the summation order inside
// those four rows.
// Coefficient layout in LDS: for each group S of four k-steps and lane group g one 16-double row,
// entry 4 q + r = row 96+r, column 4 (4S+q) + g.  A lane reads ONE entry per group (a conflict-free
// ds_read_b64 for the wavefront) and the FMAs take the entry they need from the lane that holds
// it with the DPP row broadcast of v_fmac_f64 (row_newbcast: lane n of the own 16-lane row).
// The 16 lanes of a row share all their coefficients, so reading four per lane and k-step
// (2 ds_read_b128, the first version) moved 16x the bytes: those reads cost 2.0 of 34.8 ms.
// The DPP source must not have been written by the VALU in the two preceding wait states; the
// compiler does not see inside the asm, tools/check_dpp_hazards.py (a CPU test) checks the build.
// Round 3: the same for every d = 16 NF + 1 .. 16 NF + 4 (NF = 1..6 full row tiles): NF as a template
// parameter of the tail pieces and of the kernel -- d = 36 no longer pays a whole padded row tile.
constexpr int tail_ksteps(int NF) { return 4 * NF + 1; }   // k-steps 0 .. 4 NF carry coordinates
constexpr int tail_doubles(int NF) { return (NF + 1) * 64; }
constexpr int kTailDoubles = tail_doubles(6);

template <int NF>
struct TailAcc {
    static constexpr int kS = tail_ksteps(NF), kG = NF + 1;
    double t0 = 0.0, t1 = 0.0, t2 = 0.0, t3 = 0.0;
    double c[kG];
    // this lane's entry of group S (lt: the image, g/cl: lane group and lane in the row)
    RMH_DEV void load(const double *lt, int g, int cl, int S) { c[S] = lt[(S * 4 + g) * 16 + cl]; }
    // the four rows' FMAs of k-step s = 4 S + Q
    template <int S, int Q>
    RMH_DEV void step(double x)
    {
        fmac_bcast<4 * Q + 0>(t0, c[S], x);
        fmac_bcast<4 * Q + 1>(t1, c[S], x);
        fmac_bcast<4 * Q + 2>(t2, c[S], x);
        fmac_bcast<4 * Q + 3>(t3, c[S], x);
    }
    // k-steps 4S .. 4S+3 (the last group holds k-step 4 NF only)
    template <int S>
    RMH_DEV void group(const double (&v)[4 * (NF + 1)])
    {
        step<S, 0>(v[4 * S]);
        if (4 * S + 1 < kS) {
            step<S, 1>(v[4 * S + 1]);
            step<S, 2>(v[4 * S + 2]);
            step<S, 3>(v[4 * S + 3]);
        }
    }
    template <int S0>
    RMH_DEV void groups_from(const double (&v)[4 * (NF + 1)])
    {
        if constexpr (S0 < kG) {
            group<S0>(v);
            groups_from<S0 + 1>(v);
        }
    }
    // row g of the result on lane group g
    RMH_DEV double finish() { return sum4_scatter(t0, t1, t2, t3); }
};

template <int NF>
RMH_DEV double tail_rows(const double *lt, int g, int cl, const double (&v)[4 * (NF + 1)])
{
    TailAcc<NF> tl;
#pragma unroll
    for (int S = 0; S < NF + 1; ++S) tl.load(lt, g, cl, S);
    tl.template groups_from<0>(v);
    return tl.finish();
}

// log N(theta | mu, C) from y = theta - mu for d = 16 NF + 1 .. 16 NF + 4: |W y|^2 with the NF full row
// tiles on the matrix core (two in flight) and the four tail rows on the VALU
template <int NF>
RMH_DEV double gauss_loglik(const double *Wimg, const double *Wt, const GaussArgs &m, int lane, int g,
                            int cl, const double (&y)[4 * (NF + 1)])
{
    using T = Tile<NF + 1>;
    double uu = 0.0;
#pragma unroll
    for (int i0 = 0; i0 < NF; i0 += 2) {
        double4_t a0 = double4_t{0.0, 0.0, 0.0, 0.0}, a1 = a0;
        constexpr bool kDummy = false;
        (void)kDummy;
        const bool two = i0 + 1 < NF;
#pragma unroll
        for (int s = 0; s < 4 * (i0 + 2); ++s) {
            if (s < 4 * (i0 + 1)) a0 = mfma(Wimg[T::lo(i0, s) * 64 + lane], y[s], a0);
            if (two) a1 = mfma(Wimg[T::lo(i0 + 1 < NF ? i0 + 1 : i0, s) * 64 + lane], y[s], a1);
        }
        uu += a0.x * a0.x + a0.y * a0.y + a0.z * a0.z + a0.w * a0.w;
        if (two) uu += a1.x * a1.x + a1.y * a1.y + a1.z * a1.z + a1.w * a1.w;
    }
    // (with the first tail version -- four coefficients per lane and k-step from LDS -- spreading
    // this tail between the MFMAs spilled at the 256-register cap, 37.5 -> 42..59 ms; with one
    // coefficient per four k-steps there is no read latency left worth hiding)
    const double ut = tail_rows<NF>(Wt, g, cl, y);
    uu += ut * ut;
    uu = sum4(uu);
    return -0.5 * ((uu + m.dlog2pi) + m.logdetC);
}

// the proposal product of k_run_rw_tail: group SG of four k-steps -- its MFMAs over the full row tiles,
// then its tail FMAs; the next group's tail coefficient is read before the MFMAs that cover the latency
template <int NF, int SG>
RMH_DEV void tail_prop_groups(const double *Limg, const double *Lt, TailAcc<NF> &tl, const double (&xi)[4 * (NF + 1)],
                              double4_t (&acc)[NF], int lane, int g, int cl)
{
    using T = Tile<NF + 1>;
    if constexpr (SG < NF) {
        tl.load(Lt, g, cl, SG + 1);
#pragma unroll
        for (int s = 4 * SG; s < 4 * SG + 4; ++s) {
#pragma unroll
            for (int i = SG; i < NF; ++i) acc[i] = mfma(Limg[T::lo(i, s) * 64 + lane], xi[s], acc[i]);
        }
        tl.template group<SG>(xi);
        tail_prop_groups<NF, SG + 1>(Limg, Lt, tl, xi, acc, lane, g, cl);
    }
}

// ---- pooled second moments inside the sampling kernel (AdaptCov) --------------------------
// AdaptCovProposal.adapt (riemann/proposals/adaptive.py:73-76) adds theta and theta theta^T of
// the chain's state after EVERY step; pooled over the chains that is S += sum_c z_c z_c^T with
// z = (theta, 1), a (d+1) x (d+1) symmetric product per iteration.  Round 2 wrote every state
// of a stretch to a scratch history and re-read it in k_moments (2 x 53 GB per 1000 iterations
// at the cfg2 size, two more launches per update).  Here the workgroup forms the product
// itself, right after the accept step, from the states its eight wavefronts hold in registers:
//  * a state tile is (parameter 4s+g, chain cl) per lane; the matrix core wants the CHAIN as the
//    contraction index -- lane (m, k) <- z[16i + m][chain 4q + k] for both operands -- so every
//    wavefront passes its 16 chains through a small LDS panel (112 rows, stride 17 doubles:
//    conflict-free for ds_read_b64 up to one bank pair, and for the ds_read2_b64 the compiler may
//    merge two gathers into -- banked modulo 32 dwords over 16-lane groups -- where an even
//    stride is 2-way);
//  * the 28 lower tile pairs of the 7 x 7 tile grid are dealt to the eight wavefronts as 2 x 2
//    blocks {row tiles X2, X3} x {column tiles X0, X1}: four operand reads for four MFMAs per
//    k-block and ONE code path for all wavefronts (kMomTiles below; a first version with two MFMA
//    patterns behind a wave-uniform branch made the compiler copy all accumulators between the two
//    register assignments every round, draining the matrix pipe each time);
//  * 16 accumulator doubles per lane, kept in this workgroup's slice of run.mom_partial between
//    iterations (L2 resident; the sampling code is at the 256-register cap and would spill them
//    to scratch anyway), read back before the accept step;
//  * eight rounds per iteration, round k = the chains of wavefront k, software-pipelined: after
//    the barrier that publishes panel k+1 every wavefront interleaves ITS reads of panel k+1 with
//    the MFMAs of round k (operands in registers since the previous round) -- eight reads, four
//    MFMAs, eight reads ...: issued up front, the reads of eight wavefronts fill the LDS queue and
//    hold the matrix pipe back -- and wavefront k+2 refills panel k's buffer in the middle of the
//    round.  ONE workgroup barrier per round, and it waits for LDS traffic only (lds_barrier).
// The per-workgroup partials are added up in a fixed order by k_moments_reduce (reproducible).
constexpr int kMomPS = 17, kMomPanel = 112 * kMomPS;
#ifndef RMH_MOM_ABL
#define RMH_MOM_ABL 0   // ablation builds of tools/mom_ablation.sh (timing only, wrong moments)
#endif

// workgroup barrier for LDS traffic only: __syncthreads() also waits for vmcnt(0), i.e. for the
// history stores of the record step to be acknowledged
RMH_DEV void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// operand tiles of wavefront w: column tiles X0, X1 and row tiles X2, X3; its four MFMAs per k-block
// are the 2 x 2 block (X2,X0) (X3,X0) (X2,X1) (X3,X1) -- ONE code path for all wavefronts, four
// operand reads for four MFMAs.  Waves 0..5 cover 24 pairs (the free row slot beside tile 6 takes
// the diagonal blocks' pairs); waves 6, 7 form two diagonal pairs each and waste two slots (bit
// mask kMomKeep).  Together: all 28 pairs i >= j exactly once.
__device__ constexpr int kMomTiles[8][4] = {{0, 1, 2, 3}, {0, 1, 4, 5}, {2, 3, 4, 5}, {0, 1, 6, 1},
                                            {2, 3, 6, 3}, {4, 5, 6, 5}, {0, 2, 0, 2}, {4, 6, 4, 6}};
__device__ constexpr int kMomKeep[8] = {15, 15, 15, 15, 15, 15, 9, 9};

struct MomFused {
    static constexpr int NT = 7, NTRI = 28, TPW = 4;
    int xo[4];              // byte offsets of the wave's operand tiles in a panel (wave-uniform)
    int po[TPW];            // offsets of its accumulator tiles in the workgroup's partial, -1: unused slot
    double4_t acc[TPW];
    bool all_live = true;   // every chain of the wave counts (wave-uniform)
#if RMH_MOM_ABL == 9
    int dbg_it = 0;   // iteration number, for the clock64 print of the ablation build
#endif

    RMH_DEV void setup(int wave, bool live)
    {
        all_live = __all(live);
        int tl[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            tl[k] = kMomTiles[wave][k];
            xo[k] = 16 * tl[k] * kMomPS * 8;
        }
        const int keep = kMomKeep[wave];
#pragma unroll
        for (int tt = 0; tt < TPW; ++tt) {
            const int r = tl[2 + (tt & 1)], c = tl[tt >> 1];     // slot tt: row tile X2 / X3, column tile X0 / X1
            po[tt] = ((keep >> tt) & 1) ? (r * (r + 1) / 2 + c) * 256 : -1;
        }
    }

    // this wave's 16 chains into a panel; z = (theta, 1) for a live chain, 0 otherwise.
    // * While MFMAs are in flight on a SIMD every other vector instruction of its two wavefronts
    //   waits for a whole MFMA (64 cycles; measured with clock64: a round with a writer took
    //   2048 + 64 x [vector instructions of the write] cycles), and the other seven wavefronts wait
    //   for the writer at the next barrier.  So the write is LDS instructions only: addresses are
    //   formed once per iteration before the first barrier (`wa`: row g, column cl of panel 0;
    //   `ca`: row 0, column cl), everything else is an immediate offset.
    // * Single ds_write_b64 with immediate offsets, as inline asm: left to itself the compiler pairs the
    //   stores into ds_write2_b64, whose 8-bit offsets do not reach across the panel, and materialises
    //   ten row addresses (v_add_u32) plus a spilled base (scratch_load + vmcnt(0)) in front of them.
    // * Common path (every chain of the wave live): 28 stores of the state registers (rows > d of
    //   the state are zero), then the constant 1 into row d by all four lanes of a chain (same
    //   address, same value).  Only a wave with padding chains pays selects.
    template <int OFF>
    RMH_DEV static void lds_write(uint32_t addr, double v)
    {
        asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(OFF) : "memory");
    }
    template <int BUF, int S0>
    RMH_DEV static void write_rows(uint32_t wa, bool live, bool all_live, const double (&th)[28])
    {
        if constexpr (S0 < 28) {
            lds_write<BUF * kMomPanel * 8 + S0 * 4 * kMomPS * 8>(wa, all_live ? th[S0] : (live ? th[S0] : 0.0));
            write_rows<BUF, S0 + 1>(wa, live, all_live, th);
        }
    }
    template <int BUF>
    RMH_DEV void write(uint32_t wa, uint32_t ca, int d, bool live, double one, const double (&th)[28]) const
    {
        if (RMH_MOM_ABL == 7) return;   // barriers only
        if (all_live) write_rows<BUF, 0>(wa, true, true, th);
        else write_rows<BUF, 0>(wa, live, false, th);
        // row d <- 1 (0 for a padding chain; `one` holds that), after the zero the loop put there
        if (d == 100) lds_write<BUF * kMomPanel * 8 + 100 * kMomPS * 8>(ca, one);
        else if (d == 99) lds_write<BUF * kMomPanel * 8 + 99 * kMomPS * 8>(ca, one);
        else if (d == 98) lds_write<BUF * kMomPanel * 8 + 98 * kMomPS * 8>(ca, one);
        else lds_write<BUF * kMomPanel * 8 + 97 * kMomPS * 8>(ca, one);
    }

    // the running sums back from the workgroup's partial (L2): issued before the accept step so that
    // the round trip is over when the first round needs them
    RMH_DEV void begin(const double *partial_wg, bool first, int lane)
    {
#pragma unroll
        for (int tt = 0; tt < TPW; ++tt) {
            acc[tt] = double4_t{0.0, 0.0, 0.0, 0.0};
            if (!first && po[tt] >= 0) {
                const double *src = partial_wg + po[tt] + lane;
                acc[tt].x = src[0];
                acc[tt].y = src[64];
                acc[tt].z = src[128];
                acc[tt].w = src[192];
            }
        }
    }

    // operands of one round: tile slot x, k-block q
    struct Ops {
        double v[4][4];
    };
    // One ds_read_b64 with an immediate offset: no address arithmetic between the MFMAs (every
    // vector instruction of the phase is taken from the matrix pipe's time), and no merging into
    // ds_read2_b64 (half the LDS rate).  The compiler does not see the read: `lds_barrier` and the
    // waits below order it.
    template <int OFF>
    RMH_DEV static double lds_read(uint32_t addr)
    {
        double v;
        asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
        return v;
    }
    template <int BUF, int Q>
    RMH_DEV static void fetch_q(const uint32_t (&base)[4], Ops &o)
    {
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            if (RMH_MOM_ABL == 2 || RMH_MOM_ABL == 7) o.v[x][Q] = 0.0;
            else o.v[x][Q] = lds_read<BUF * kMomPanel * 8 + 32 * Q>(base[x]);
        }
    }
    template <int Q>
    RMH_DEV void multiply_q(const Ops &o)
    {
        if (RMH_MOM_ABL == 2 || RMH_MOM_ABL == 3 || RMH_MOM_ABL == 7) {
#pragma unroll
            for (int x = 0; x < 4; ++x) asm volatile("" ::"v"(o.v[x][Q]));
            return;
        }
        acc[0] = mfma(o.v[2][Q], o.v[0][Q], acc[0]);
        acc[1] = mfma(o.v[3][Q], o.v[0][Q], acc[1]);
        acc[2] = mfma(o.v[2][Q], o.v[1][Q], acc[2]);
        acc[3] = mfma(o.v[3][Q], o.v[1][Q], acc[3]);
    }

    // round k: multiply from `cur` while the next round's operands (buffer NB, when PREFETCH)
    // arrive in `nxt` -- four reads, four MFMAs, four reads ...; the writer of round k+2 refills
    // the buffer round k's operands came from (buffer 1 - NB) in the middle of the round
    template <bool PREFETCH, int NB>
    RMH_DEV void round(Ops &cur, Ops &nxt, const uint32_t (&base)[4], bool writer, uint32_t wa, uint32_t ca, int d,
                       bool live, double one, const double (&th)[28])
    {
        // the operands of this round were read (inline asm) during the previous one and waited for
        // by the s_waitcnt of the barrier in between; tell the compiler they are defined here
#pragma unroll
        for (int x = 0; x < 4; ++x)
#pragma unroll
            for (int q = 0; q < 4; ++q) asm volatile("" : "+v"(cur.v[x][q]));
        if (PREFETCH) fetch_q<NB, 0>(base, nxt);
        multiply_q<0>(cur);
        if (PREFETCH) fetch_q<NB, 1>(base, nxt);
        multiply_q<1>(cur);
        if (writer) write<1 - NB>(wa, ca, d, live, one, th);
        if (PREFETCH) fetch_q<NB, 2>(base, nxt);
        multiply_q<2>(cur);
        if (PREFETCH) fetch_q<NB, 3>(base, nxt);
        multiply_q<3>(cur);
    }

    // one iteration: add sum over the workgroup's live chains of z z^T to its partial
    RMH_DEV void step(double *panels, double *partial_wg, int wave, int lane, int d, bool live,
                      const double (&th)[28])
    {
        if (RMH_MOM_ABL == 1) return;
        if (RMH_MOM_ABL == 6) { // one barrier per iteration, nothing else: what lockstep costs the sampling phase
            lds_barrier();
            return;
        }
        const int g = lane >> 4, cl = lane & 15;
#if RMH_MOM_ABL == 9
        long long ts[12];
        int nts = 0;
#define RMH_TS() ts[nts++] = clock64()
#else
#define RMH_TS()
#endif
        RMH_TS();
        // every address of the phase, formed here -- before the first barrier, while the early waves
        // wait for the late ones anyway: gather (row m = cl, chain k = g) of the wave's four tiles,
        // store (row g, column cl) and the constant's column, as LDS byte addresses of panel 0
        const uint32_t pl = (uint32_t)(uintptr_t)panels;
        const uint32_t p0 = pl + (uint32_t)(cl * kMomPS + g) * 8u;
        uint32_t base[4];
#pragma unroll
        for (int x = 0; x < 4; ++x) base[x] = p0 + (uint32_t)xo[x];
        const uint32_t wa = pl + (uint32_t)(g * kMomPS + cl) * 8u, ca = pl + (uint32_t)cl * 8u;
        const double one = live ? 1.0 : 0.0;
        Ops a, b;
        if (wave == 0) write<0>(wa, ca, d, live, one, th);
        lds_barrier();                                           // panel 0 visible
        RMH_TS();
        fetch_q<0, 0>(base, a);
        fetch_q<0, 1>(base, a);
        fetch_q<0, 2>(base, a);
        fetch_q<0, 3>(base, a);
        if (wave == 1) write<1>(wa, ca, d, live, one, th);
#pragma nounroll
        for (int k = 0; k < 6; k += 2) {
            // the barrier before a round publishes the next panel AND says that every wave has
            // the current round's operands in registers (its reads complete before it arrives), so
            // their buffer may be refilled
            lds_barrier();
            RMH_TS();
            round<true, 1>(a, b, base, wave == k + 2, wa, ca, d, live, one, th);
            lds_barrier();
            RMH_TS();
            round<true, 0>(b, a, base, wave == k + 3, wa, ca, d, live, one, th);
        }
        lds_barrier();
        RMH_TS();
        round<true, 1>(a, b, base, false, wa, ca, d, live, one, th);
        lds_barrier();
        RMH_TS();
        round<false, 0>(b, a, base, false, wa, ca, d, live, one, th);
        RMH_TS();
#if RMH_MOM_ABL == 9
        if (blockIdx.x == 3 && lane == 0 && dbg_it == 7)
            printf("wave %d: wait %lld pro %lld | %lld %lld %lld %lld %lld %lld %lld %lld | total %lld\n", wave,
                   ts[1] - ts[0], ts[2] - ts[1], ts[3] - ts[2], ts[4] - ts[3], ts[5] - ts[4], ts[6] - ts[5],
                   ts[7] - ts[6], ts[8] - ts[7], ts[9] - ts[8], ts[10] - ts[9], ts[10] - ts[0]);
#endif
        // (the next iteration's first write, panel 0's buffer: last read for round 6, complete
        // before the barrier of round 7; its second write comes after a barrier of its own)
#pragma unroll
        for (int tt = 0; tt < TPW; ++tt) {
            if (po[tt] >= 0) {
                double *dst = partial_wg + po[tt] + lane;
                dst[0] = acc[tt].x;
                dst[64] = acc[tt].y;
                dst[128] = acc[tt].z;
                dst[192] = acc[tt].w;
            }
        }
    }
};

// N64: the 64-bit-resolution normal mode (one normal per stream word, rmh_device.hpp) with its longer table in LDS
template <int NF, int NW, bool ADAPT = false, bool PT = false, bool MOM = false, bool N64 = false>
__global__ void __launch_bounds__(64 * NW) k_run_rw_tail(GaussArgs m, RwArgs pr, ChainArgs ch, RunArgs run)
{
    // log's constants as opaque literals (rmh_device.hpp log_pos<OPQ>): hoisted out of the iteration loop they
    // are five more live register pairs -- spilled, and reloaded inside the loop, in these variants
    constexpr bool kOpaqueLog = MOM || (NF == 5 && !ADAPT && !PT) || (NF == 6 && (PT || ADAPT));
    constexpr int NT = NF + 1, S = 4 * NT, kFull = tri_tiles(NF) * 64; // NF full row tiles (NF = 6: 84 tile images)
    constexpr int kTD = tail_doubles(NF), KS = tail_ksteps(NF);   // KS k-steps carry coordinates
    using T = Tile<NT>;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double *Limg = lds, *Wimg = lds + kFull;
    double *Lt = Wimg + kFull, *Wt = Lt + kTD;
    double *mu = Wt + kTD;
    double *tab = mu + 4 * S;
    double *panels = tab + icdf_doubles(N64 ? 1 : 0);   // MOM: two 16-chain panels (MomFused)
    static_assert(!MOM || (NF == 6 && NW == 8 && !PT && !N64), "MomFused: d = 97..100, eight wavefronts, random walk (plain or adapted scale)");
    static_assert(!N64 || !PT, "the 64-bit normal mode of the tail kernels: plain and adapted-scale walks");
    for (int i = threadIdx.x; i < kFull; i += blockDim.x) {
        Limg[i] = pr.Ltiles[i];
        Wimg[i] = m.Wtiles[i];
    }
    for (int i = threadIdx.x; i < kTD; i += blockDim.x) {
        Lt[i] = pr.Ltail[i];
        Wt[i] = m.Wtail[i];
    }
    for (int i = threadIdx.x; i < 4 * S; i += blockDim.x) mu[i] = m.mu_pad[i];
    icdf_stage(tab, ch.icdf, N64 ? 1 : 0);
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cl = lane & 15, g = lane >> 4;
    // MOM: a workgroup takes run.mom_passes sets of 8 x 16 chains one after the other (the host sizes the
    // grid to one workgroup per CU): the 114 KB of tile images and tables are staged once per launch instead
    // of once per set -- an AdaptCov stretch is one or two iterations long on the reference's schedule, so
    // the staging was a third of a launch -- and there are half as many partial sums to add up afterwards
    const int npass = MOM ? run.mom_passes : 1;
    for (int pass = 0; pass < npass; ++pass) {
    int64_t blk = ((int64_t)pass * gridDim.x + blockIdx.x) * (blockDim.x >> 6) + wave;
    const bool active = blk < ch.nblocks;
    if (!active) blk = ch.nblocks - 1;
    const int64_t c = blk * 16 + cl;
    const int64_t slot = blk * 64 + lane;

    // the rows of k-steps KS .. S-1 of the state are padding (d <= 16 NF + 4): constants here, not three register pairs
    // carried through the loop for a store of the zeros they were loaded as (round 2 spilled two of
    // them); the stream state and the log posterior through raw buffer instructions (one 32-bit lane
    // offset each instead of five 64-bit addresses held from entry to exit)
    double th[S];
    T::load_state(ch.theta, ch.ld, c, g, th);
    th[KS] = th[KS + 1] = th[KS + 2] = 0.0;
    double lp = buf_load(rsrc(ch.logpost, ch.ld), (uint32_t)c * 8u, 0u);
    Xo rng = xo_load_buf(ch.rng, ch.nslots, slot);
    PtLane pt;
    if (PT) { // MultiGaussianDist: log prior 0, so the state's parts are (0, logL) = (0, lp at beta = 1)
        pt.setup(run, cl, c);
        double y0[S];
#pragma unroll
        for (int s = 0; s < S; ++s) y0[s] = th[s] - mu[4 * s + g];
        double parts0[2] = {0.0, gauss_loglik<NF>(Wimg, Wt, m, lane, g, cl, y0)};
        pt.sL = parts0[1];
        lp = pt.tempered(parts0);
    }
    Keep keep(run);
    keep.record<NT, KS>(run, ch, 0, c, g, active, th, lp);
    AdaptScale ad{0.0, 0.0, 1.0};
    if (ADAPT) ad.load(ch, c);
    MomFused mom;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    double *mom_wg = nullptr;
    const bool mom_live = active && c < run.mom_N;
    if constexpr (MOM) {
        mom.setup(wave_u, mom_live);
        mom_wg = run.mom_partial + (int64_t)blockIdx.x * (MomFused::NTRI * 256);
    }

    for (int64_t t = 0; t < run.T; ++t) {
        asm volatile("" ::: "memory");
        if (PT) pt.select(run, rng, cl);
        double xi[S];
        // d = 16 NF + 1 .. 16 NF + 4: 2 NF + 1 words per sub-stream, as compile-time constants
        if constexpr (N64) {
            __builtin_assume(ch.d > 16 * NF && ch.d <= 16 * NF + 4);   // only the last k-step's coordinates can lie beyond d
            T::normals64(rng, KS, ch.d, g, tab, xi);
        } else T::template normals_fixed<2 * NF + 1, 16 * NF + 4, (MOM || NF < 4) ? 2 : 3>(rng, ch.d, g, tab, xi);
        // theta' = theta + scale * L xi: row tiles 0..5 on the matrix core, rows 96..99 on the VALU
        // (ADAPT: the chain's own scale times the unscaled image's product, as in k_run_rw)
        double4_t acc[NF];
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            acc[i].x = ADAPT ? 0.0 : th[4 * i + 0];
            acc[i].y = ADAPT ? 0.0 : th[4 * i + 1];
            acc[i].z = ADAPT ? 0.0 : th[4 * i + 2];
            acc[i].w = ADAPT ? 0.0 : th[4 * i + 3];
        }
        TailAcc<NF> tl;
        tl.load(Lt, g, cl, 0);
        tail_prop_groups<NF, 0>(Limg, Lt, tl, xi, acc, lane, g, cl);
        tl.template group<NF>(xi);        // k-step 4 NF
        double thp[S];
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            thp[4 * i + 0] = acc[i].x;
            thp[4 * i + 1] = acc[i].y;
            thp[4 * i + 2] = acc[i].z;
            thp[4 * i + 3] = acc[i].w;
        }
        if (ADAPT) {
#pragma unroll
            for (int s = 0; s < 4 * NF; ++s) thp[s] = th[s] + ad.scale * thp[s];
            thp[4 * NF] = th[4 * NF] + ad.scale * tl.finish();
        } else {
            thp[4 * NF] = th[4 * NF] + tl.finish();
        }
        thp[KS] = thp[KS + 1] = thp[KS + 2] = 0.0;
        // log posterior: |W (theta' - mu)|^2, two row tiles in flight, then the four tail rows
        double y[S];
#pragma unroll
        for (int s = 0; s < S; ++s) y[s] = thp[s] - mu[4 * s + g];
        double parts[2] = {0.0, gauss_loglik<NF>(Wimg, Wt, m, lane, g, cl, y)};
        const double lpp = PT ? pt.tempered(parts) : combine_post(0.0, parts[1]);
        if constexpr (MOM) mom.begin(mom_wg, pass == 0 && t == 0, lane);   // (no stores in flight here: the record step comes last)
        const double u = T::accept_uniform(rng, g, cl);
        // (MOM: the logarithm's constants as literals at their use -- hoisted to kernel entry they are
        // spilled there, and a scratch reload in the loop waits for every store in flight)
        bool acc_ok = mh_accept<kOpaqueLog>(lpp, lp, 0.0, u);
        if (PT) acc_ok = pt.within(acc_ok);
#pragma unroll
        for (int s = 0; s < S; ++s) th[s] = acc_ok ? thp[s] : th[s];
        lp = acc_ok ? lpp : lp;
        int code = acc_ok ? 1 : 0;
        if (PT) code = pt.swap(lane, acc_ok, parts, th, lp);
        if (ADAPT) ad.adapt(acc_ok, ch.adapt_target);
        // adapt() after the step (sampler.py:88 -> adaptive.py:73-76), pooled over the chains.
        // BEFORE the record: vmcnt counts loads and stores in one queue, so any wait on a load in
        // the moment phase (the accumulators, a spilled address) would first drain the ~27 history
        // stores of the record step (about 1100 cycles each time, measured with clock64)
        if constexpr (MOM) {
#if RMH_MOM_ABL == 9
            mom.dbg_it = (int)t;
#endif
            mom.step(panels, mom_wg, wave_u, lane, ch.d, mom_live, th);
        }
        // (the accept byte through a buffer store at the state rows' lane offset / 8: a flat store keeps
        // a 64-bit address per lane alive through the loop -- spilled in the MOM variant, and a scratch
        // reload here waits for every store in flight)
        if (run.h_accept && active && g == 0)
            buf_store_u8(run.h_accept + t * ch.ld, ch.ld, (uint32_t)(g * ch.ld + c), (uint8_t)code);
        keep.record<NT, KS>(run, ch, t + 1, c, g, active, th, lp);
    }
    if (active) {
        T::store_state(ch.theta, ch.ld, c, g, th);
        if (g == 0) buf_store(rsrc(ch.logpost, ch.ld), (uint32_t)c * 8u, 0u, lp);
        if (ADAPT && g == 0) ad.store(ch, c);
        xo_store_buf(ch.rng, ch.nslots, slot, rng);
    }
    }   // pass
}

// ------------------------------------------------------------ HMC / MALA ----
// VanillaHMC.propose + leapfrog with M = None.  The gradient at the current
// state is cached in ch.grad (same point, same function => same bits as the
// reference's re-evaluation, hamiltonian.py:27), and the last gradient call of
// the trajectory (:40) is fused with the log_posterior call of sampler.py:82.
// Current state and cached gradient live in global memory between iterations.
// MASS: VanillaHMC with a mass matrix M (hamiltonian.py:71-74, 80-81, 29, 36, 85-87):
// p = chM xi;  q += eps M^-1 p with M^-1 p = V^T (V p), V = chM^-1;  K = |V p|^2 / 2.
// The three triangular tile images are read straight from global memory (L2).
// PT: parallel tempering with HMC within-chain moves (see k_run_rw; Gaussian, mixture, logistic).  As in the reference the
// proposal integrates with the gradient callable it was given -- the BASE model's, whatever
// the rung (samplers/ptsampler.py:82 hands every rung the same proposal) -- and only the accept
// test sees the tempered posterior.  A swap exchanges the two replicas' states and cached
// gradients in global memory.
// N64 (models with a compile-time stream length only, M::kTailNF > 0): the draw of the 64-bit normal mode in place of
// normals_fixed -- as a template parameter so that the 32-bit kernels keep their register allocation
// KSC > 0: the dimension's k-step count ceil(d/4) as a compile-time constant (KSC = 4 NT: d = 16 NT exactly, then the
// word count and d itself too) -- no wave-uniform test in front of the last k-steps of every triangular product
template <class M, int NW, bool ADAPT, bool MASS, bool PT = false, bool N64 = false, int KSC = 0>
__global__ void __launch_bounds__(64 * NW) k_run_hmc(typename M::Args m_in, HmcArgs pr, ChainArgs ch_in, RunArgs run)
{
    constexpr int NT = M::T::S / 4, S = M::T::S;
    using T = Tile<NT>;
    typename M::Args m = m_in;
    ChainArgs ch = ch_in;
    if constexpr (KSC > 0) {
        ch.ks = KSC;
        m.ks = KSC;
        if constexpr (KSC == S) {
            ch.d = 4 * S;
            ch.M = S / 2;
        }
    }
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double *tab = lds + M::lds_doubles(true);
    M::setup(lds, m, true);
    icdf_stage(tab, ch.icdf, ch.nmode);
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cl = lane & 15, g = lane >> 4;
    int64_t blk = (int64_t)blockIdx.x * (blockDim.x >> 6) + wave;
    const bool active = blk < ch.nblocks;
    if (!active) blk = ch.nblocks - 1;
    const int64_t c = blk * 16 + cl;
    const int64_t slot = blk * 64 + lane;

    // momentum parking column: ch.mom is (dp, ld + 128); the idle waves of the
    // last workgroup (which shadow its last block) get columns of their own
    const int64_t cm = active ? c : ch.ld + wave * 16 + cl;
    // Between uses the chain's scalars live in memory too (not in registers that would be
    // spilled across the evaluations): the log posterior in ch.logpost (read before the accept
    // test, written on accept; PT keeps the tempered value in a register), the kinetic energy
    // of the draw in four spare rows of the momentum parking array.
    // (descriptors are rebuilt from the kernel arguments at every use: held, they cost scalar
    // registers across the evaluations, and the kernel runs at the scalar register cap too)
    constexpr bool LPMEM = !PT;
    auto rs_lp = [&]() { return rsrc(ch.logpost, ch.ld); };
    auto rs_k0 = [&]() { return rsrc(ch.mom + (int64_t)(4 * S) * (ch.ld + 128), 4 * (ch.ld + 128)); };
    const uint32_t vo_lp = (uint32_t)c * 8u, vo_k0 = (uint32_t)(g * (ch.ld + 128) + cm) * 8u;
    double lp = ch.logpost[c];
    Xo rng = xo_load_buf(ch.rng, ch.nslots, slot);
    double eps = pr.eps, heps = 0.5 * pr.eps;
    Keep keep(run);
    AdaptScale ad{0.0, 0.0, 1.0};
    if (ADAPT) ad.load(ch, c);
    int rung = 0, ebase = 0;
    bool pt_valid = false;
    double beta = 1.0, sP = 0.0, sL = 0.0;
    if (PT) {
        const int K = run.pt_K;
        rung = cl % K;
        ebase = cl - rung;
        pt_valid = cl < (16 / K) * K;
        beta = run.pt_beta[c];
        double q0[S], g0[S], lp0, parts0[2];
        T::load_state(ch.theta, ch.ld, c, g, q0);
        M::template eval<true, true>(lds, m, true, lane, q0, lp0, g0, parts0);
        sP = parts0[0];
        sL = parts0[1];
        lp = combine_post(sP, beta * sL);   // Sampler.__init__ of the rung, sampler.py:42
    }

    for (int64_t t = 0; t <= run.T; ++t) {
        asm volatile("" ::: "memory"); // no LICM of the LDS tile reads (see k_run_rw)
        double p[S], q[S], gr[S], lpp = 0.0, parts[2];
        bool pt_init = false, pt_partner = false;
        double vswap = 0.0;
        if (PT && t < run.T) { // selector / swap words, then the sequential rule of ptsampler.py:104-109
            const double uw = u01(xo_next(rng));
            const double usel = __shfl(uw, cl);
            vswap = __shfl(uw, 16 + cl);
            const bool wants = pt_valid && rung < run.pt_K - 1 && !(usel > run.pt_pswap);
            const unsigned wm = (unsigned)(__ballot(wants) & 0xFFFFu) >> ebase;
            bool prev = false;
            for (int j = 0; j < run.pt_K; ++j) {
                const bool in = ((wm >> j) & 1u) && !prev;
                if (j == rung) {
                    pt_init = in;
                    pt_partner = prev;
                }
                prev = in;
            }
        }
        if (t < run.T) {
            // d = 16 NF + 1 .. 16 NF + 4 (the model's compile-time case): 2 NF + 1 words per sub-stream,
            // no wave-uniform word and k-step predicates (rmh_mfma.hpp)
            if constexpr (M::kTailNF > 0 && N64) {
                __builtin_assume(ch.d > 16 * M::kTailNF && ch.d <= 16 * M::kTailNF + 4);
                T::normals64(rng, 4 * M::kTailNF + 1, ch.d, g, tab, p);
            } else if constexpr (M::kTailNF > 0) T::template normals_fixed<2 * M::kTailNF + 1, 16 * M::kTailNF + 4>(rng, ch.d, g, tab, p);
            else T::normals_mode(ch.nmode, rng, ch.M, ch.ks, ch.d, g, tab, p);
        }
        T::load_state(ch.theta, ch.ld, c, g, q);
        if (LPMEM) lp = buf_load(rs_lp(), vo_lp, 0u);
        keep.record<NT>(run, ch, t, c, g, active, q, lp);
        if (t == run.T) break;
        if (ADAPT) { // AdaptScaleHMC.propose: eps = scale * eps0 (hamiltonian.py:101-102)
            eps = ad.scale * pr.eps;
            heps = 0.5 * eps;
        }
        double k0 = 0.0;
#pragma unroll
        for (int s = 0; s < S; ++s) k0 += p[s] * p[s];   // K of the N(0,I) draw (= |chM^-1 p0|^2)
        // k0 is used after the trajectory: without this the compiler sinks the sum there and
        // keeps the S normals alive (in scratch) across every evaluation
        asm volatile("" : "+v"(k0));
        buf_store(rs_k0(), vo_k0, 0u, k0);
        if (MASS) tri_apply<NT, false>(pr.chM, lane, p, ch.ks); // p0 = chM xi, :80-81
        T::load_state(ch.grad, ch.ld, c, g, gr);
#pragma unroll
        for (int s = 0; s < S; ++s) p[s] = p[s] + heps * gr[s];  // hamiltonian.py:27
        if (MASS) {
            double psc[S];
#pragma unroll
            for (int s = 0; s < S; ++s) psc[s] = p[s];
            tri_apply<NT, false>(pr.V, lane, psc, ch.ks);       // M^-1 p = V^T V p, :29
            tri_apply<NT, true>(pr.VT, lane, psc, ch.ks);
#pragma unroll
            for (int s = 0; s < S; ++s) q[s] = q[s] + (eps * pr.minv) * psc[s];
        } else {
#pragma unroll
            for (int s = 0; s < S; ++s) q[s] = q[s] + eps * p[s]; // :30
        }
        // one evaluation site for all Nsteps gradient calls: the last one is
        // also the log_posterior call of sampler.py:82 and ends with the half kick
        for (int step = 1; step <= pr.nsteps; ++step) {
            // the momentum is dead during the evaluation: park it in global memory
            // (L2) so that the evaluation has the register file to itself
            T::store_state(ch.mom, ch.ld + 128, cm, g, p);
            asm volatile("" ::: "memory");
            const bool last = step == pr.nsteps;
            // the trajectory's inner gradients (hamiltonian.py:35) need no density value
            if (last) M::template eval<true, true>(lds, m, true, lane, q, lpp, gr, parts);
            else M::template eval<true, false>(lds, m, true, lane, q, lpp, gr, parts);
            T::load_state(ch.mom, ch.ld + 128, cm, g, p);
            const double ek = last ? heps : eps; // :35 / :40
#pragma unroll
            for (int s = 0; s < S; ++s) p[s] = p[s] + ek * gr[s];
            if (!last) {
                if (MASS) {
                    double psc[S];
#pragma unroll
                    for (int s = 0; s < S; ++s) psc[s] = p[s];
                    tri_apply<NT, false>(pr.V, lane, psc, ch.ks);   // :36
                    tri_apply<NT, true>(pr.VT, lane, psc, ch.ks);
#pragma unroll
                    for (int s = 0; s < S; ++s) q[s] = q[s] + (eps * pr.minv) * psc[s];
                } else {
#pragma unroll
                    for (int s = 0; s < S; ++s) q[s] = q[s] + eps * p[s]; // :37
                }
            }
        }
        if (MASS) tri_apply<NT, false>(pr.V, lane, p, ch.ks);   // p = solve(chM, p), :87
        double k1 = 0.0;
#pragma unroll
        for (int s = 0; s < S; ++s) k1 += p[s] * p[s];
        asm volatile("" ::: "memory");
        k0 = buf_load(rs_k0(), vo_k0, 0u);
        if (LPMEM) lp = buf_load(rs_lp(), vo_lp, 0u);
        const double logq = 0.5 * (sum4(k1) - sum4(k0)); // :89
        const double u = T::accept_uniform(rng, g, cl);
        if (PT) lpp = combine_post(parts[0], beta * parts[1]); // TemperedModel, ptsampler.py:35-39
        bool acc_ok = mh_accept<true>(lpp, lp, logq, u);
        if (PT) acc_ok = acc_ok && pt_valid && !pt_init && !pt_partner;
        if (acc_ok) {
            if (active) {
                T::store_state(ch.theta, ch.ld, c, g, q);
                T::store_state(ch.grad, ch.ld, c, g, gr);
                if (LPMEM && g == 0) buf_store(rs_lp(), vo_lp, 0u, lpp);
            }
            lp = lpp;
        }
        int code = acc_ok ? 1 : 0;
        if (PT) {
            sP = acc_ok ? parts[0] : sP;
            sL = acc_ok ? parts[1] : sL;
            const bool swapper = pt_init || pt_partner;
            const int src = pt_init ? lane + 1 : pt_partner ? lane - 1 : lane;
            const double oP = __shfl(sP, src), oL = __shfl(sL, src), ob = __shfl(beta, src);
            const double olp = __shfl(lp, src), ov = __shfl(vswap, src);
            const double cross_me = combine_post(oP, beta * oL);
            const double cross_ot = combine_post(sP, ob * sL);
            const double delta = (cross_ot + cross_me) - (pt_init ? lp + olp : olp + lp);
            const bool sw_ok = swapper && mh_accept(delta, 0.0, 0.0, pt_partner ? ov : vswap);
            if (__any(sw_ok)) { // exchange state and cached gradient of the two replicas
                asm volatile("" ::: "memory");
                T::load_state(ch.theta, ch.ld, c, g, q);
                T::load_state(ch.grad, ch.ld, c, g, gr);
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    const double oq = __shfl(q[s], src), og = __shfl(gr[s], src);
                    q[s] = sw_ok ? oq : q[s];
                    gr[s] = sw_ok ? og : gr[s];
                }
                if (active) {
                    T::store_state(ch.theta, ch.ld, c, g, q);
                    T::store_state(ch.grad, ch.ld, c, g, gr);
                }
                asm volatile("" ::: "memory");
            }
            sP = sw_ok ? oP : sP;
            sL = sw_ok ? oL : sL;
            lp = sw_ok ? cross_me : lp;
            code = swapper ? (sw_ok ? 3 : 2) : code;
        }
        if (ADAPT) ad.adapt(acc_ok, ch.adapt_target);
        // (the chain index from the log-posterior offset: kept as a register of its own it is the one value
        // that does not fit beside the evaluation and goes to scratch)
        if (run.h_accept && active && g == 0) {
            uint32_t cc = vo_lp;
            asm volatile("v_lshrrev_b32 %0, 3, %0" : "+v"(cc));   // opaque: the compiler would fold it back to c
            buf_store_u8(run.h_accept + t * ch.ld, ch.ld, cc, (uint8_t)code);
        }
    }
    if (active) {
        if (!LPMEM && g == 0) ch.logpost[c] = lp;
        if (ADAPT && g == 0) ad.store(ch, c);
        xo_store_buf(ch.rng, ch.nslots, slot, rng);
    }
}

// ------------------------------------------------- standalone entry kernels --
// Model.log_posterior / gradient for the states in ch.theta
template <class M, bool GRAD, int NW>
__global__ void __launch_bounds__(64 * NW) k_eval(typename M::Args m, ChainArgs ch, double *lp_out, double *grad_out,
                                                 double *lprior_out, double *llik_out)
{
    constexpr int NT = M::T::S / 4, S = M::T::S;
    using T = Tile<NT>;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    M::setup(lds, m, GRAD);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cl = lane & 15, g = lane >> 4;
    int64_t blk = (int64_t)blockIdx.x * (blockDim.x >> 6) + wave;
    const bool active = blk < ch.nblocks;
    if (!active) blk = ch.nblocks - 1;
    const int64_t c = blk * 16 + cl;
    double th[S], gr[S], lp, parts[2];
    T::load_state(ch.theta, ch.ld, c, g, th);
    M::template eval<GRAD>(lds, m, GRAD, lane, th, lp, gr, parts);
    if (active) {
        if (g == 0 && lp_out) lp_out[c] = lp;
        if (g == 0 && lprior_out) lprior_out[c] = parts[0];
        if (g == 0 && llik_out) llik_out[c] = parts[1];
        if (GRAD) T::store_state(grad_out, ch.ld, c, g, gr);
    }
}

// MetropolisRandomWalk.propose (randomwalk.py:21-26) or pCN.propose (:89-101) for the states
// in ch.theta (advances ch.rng); logq_out is written for pCN only (0 for the random walk)
template <int NT, int NW, bool PCN>
__global__ void __launch_bounds__(64 * NW) k_propose_rw(RwArgs pr, ChainArgs ch, double *theta_p, double *logq_out)
{
    constexpr int S = 4 * NT;
    using T = Tile<NT>;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double *tab = lds + tri_tiles(NT) * 64;
    for (int i = threadIdx.x; i < tri_tiles(NT) * 64; i += blockDim.x) lds[i] = pr.Ltiles[i];
    icdf_stage(tab, ch.icdf, ch.nmode);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cl = lane & 15, g = lane >> 4;
    const int64_t blk = (int64_t)blockIdx.x * (blockDim.x >> 6) + wave;
    if (blk >= ch.nblocks) return;
    const int64_t c = blk * 16 + cl, slot = blk * 64 + lane;
    double th[S], xi[S];
    T::load_state(ch.theta, ch.ld, c, g, th);
    Xo rng = xo_load(ch.rng, ch.nslots, slot);
    T::normals_mode(ch.nmode, rng, ch.M, ch.ks, ch.d, g, tab, xi);
    double4_t acc[NT];
    if (PCN) {
        double thp[S], ufw[S], urv[S];
#pragma unroll
        for (int s = 0; s < S; ++s) thp[s] = pr.rho * th[s];
        T::pack(thp, acc);
        T::mul_lower(lds + lane, xi, acc, ch.ks);   // image holds rho_c * L
        T::unpack(acc, thp);
#pragma unroll
        for (int s = 0; s < S; ++s) {
            ufw[s] = thp[s] - pr.rho * th[s];
            urv[s] = th[s] - pr.rho * thp[s];
        }
        tri_apply<NT, false>(pr.Vtiles, lane, ufw, ch.ks);
        tri_apply<NT, false>(pr.Vtiles, lane, urv, ch.ks);
        double sf = 0.0, sr = 0.0;
#pragma unroll
        for (int s = 0; s < S; ++s) {
            sf += ufw[s] * ufw[s];
            sr += urv[s] * urv[s];
        }
        const double logq = -0.5 * (sum4(sf) - sum4(sr));
        T::store_state(theta_p, ch.ld, c, g, thp);
        if (g == 0) logq_out[c] = logq;
    } else {
        T::pack(th, acc);
        T::mul_lower(lds + lane, xi, acc, ch.ks);
        T::unpack(acc, th);
        T::store_state(theta_p, ch.ld, c, g, th);
    }
    xo_store(ch.rng, ch.nslots, slot, rng);
}

// VanillaHMC.propose for the states in ch.theta (advances ch.rng); MASS: with a mass matrix
// (hamiltonian.py:80-81, 29, 36, 85-87), tile images read from global memory
template <class M, int NW, bool MASS>
__global__ void __launch_bounds__(64 * NW) k_propose_hmc(typename M::Args m, HmcArgs pr, ChainArgs ch, double *theta_p,
                              double *logq_out)
{
    constexpr int NT = M::T::S / 4, S = M::T::S;
    using T = Tile<NT>;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double *tab = lds + M::lds_doubles(true);
    M::setup(lds, m, true);
    icdf_stage(tab, ch.icdf, ch.nmode);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cl = lane & 15, g = lane >> 4;
    int64_t blk = (int64_t)blockIdx.x * (blockDim.x >> 6) + wave;
    const bool active = blk < ch.nblocks;
    if (!active) blk = ch.nblocks - 1;
    const int64_t c = blk * 16 + cl, slot = blk * 64 + lane;
    double q[S], p[S], gr[S], lpp, parts[2];
    T::load_state(ch.theta, ch.ld, c, g, q);
    Xo rng = xo_load(ch.rng, ch.nslots, slot);
    T::normals_mode(ch.nmode, rng, ch.M, ch.ks, ch.d, g, tab, p);
    double k0 = 0.0;
#pragma unroll
    for (int s = 0; s < S; ++s) k0 += p[s] * p[s];
    if (MASS) tri_apply<NT, false>(pr.chM, lane, p, ch.ks);
    const double eps = pr.eps, heps = 0.5 * pr.eps;
    auto drift = [&]() {
        if (MASS) {
            double psc[S];
#pragma unroll
            for (int s = 0; s < S; ++s) psc[s] = p[s];
            tri_apply<NT, false>(pr.V, lane, psc, ch.ks);
            tri_apply<NT, true>(pr.VT, lane, psc, ch.ks);
#pragma unroll
            for (int s = 0; s < S; ++s) q[s] = q[s] + (eps * pr.minv) * psc[s];
        } else {
#pragma unroll
            for (int s = 0; s < S; ++s) q[s] = q[s] + eps * p[s];
        }
    };
    M::template eval<true>(lds, m, true, lane, q, lpp, gr, parts);
#pragma unroll
    for (int s = 0; s < S; ++s) p[s] = p[s] + heps * gr[s];
    drift();
    for (int step = 1; step < pr.nsteps; ++step) {
        asm volatile("" ::: "memory");
        M::template eval<true>(lds, m, true, lane, q, lpp, gr, parts);
#pragma unroll
        for (int s = 0; s < S; ++s) p[s] = p[s] + eps * gr[s];
        drift();
    }
    M::template eval<true>(lds, m, true, lane, q, lpp, gr, parts);
#pragma unroll
    for (int s = 0; s < S; ++s) p[s] = p[s] + heps * gr[s];
    if (MASS) tri_apply<NT, false>(pr.V, lane, p, ch.ks);
    double k1 = 0.0;
#pragma unroll
    for (int s = 0; s < S; ++s) k1 += p[s] * p[s];
    const double logq = 0.5 * (sum4(k1) - sum4(k0));
    if (active) {
        T::store_state(theta_p, ch.ld, c, g, q);
        if (g == 0) logq_out[c] = logq;
        xo_store(ch.rng, ch.nslots, slot, rng);
    }
}

} // namespace rmh
