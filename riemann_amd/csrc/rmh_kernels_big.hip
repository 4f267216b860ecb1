// rmh_kernels_big.hip -- the Gaussian / random-walk path for 112 < d <= 1024.
//
// Same arithmetic as k_run_rw<GaussModel> (rmh_run.hpp): theta' = theta + scale L xi
// (riemann/proposals/randomwalk.py:21-26), logL = -(|W(theta'-mu)|^2 + d log 2pi +
// logdetC)/2 (riemann/models/gaussian.py:46-52), accept (samplers/sampler.py:83-87) -- but
// a chain's state no longer fits the registers of one wavefront.  Here a WORKGROUP of four
// wavefronts owns CB blocks of 16 chains (CB = 4, 2 or 1 as d grows):
//   * the B operand of both triangular products (xi, then theta' - mu) lives in LDS in the
//     B-operand layout, Y[block][s][lane] = coordinate 4s + (lane>>4) of chain lane&15:
//     128 d bytes per block, which with the inverse-CDF table bounds CB d at 1024;
//   * output tile i (rows 16i..16i+15) belongs to wave i mod 4 for ALL blocks: the wave keeps
//     its tiles (<= 16) of the candidates theta' in registers and uses every image tile it loads for
//     CB MFMAs -- the tile images of L and W = Lt^-1 stream from L2 (3.3 MB at d = 896) and
//     that stream, not the matrix core, is the bound of this kernel;
//   * wave j (j < CB) holds the random streams of block j, draws its normals (the stream
//     is serial per sub-stream) and takes its accept decisions;
//   * the chain state lives in global memory between iterations; rows are written for all
//     16 chains of a block at once (accepted or not), so every store is a whole 128-byte line.
#include "rmh_internal.hpp"
#include "rmh_mfma.hpp"

namespace rmh {

namespace {

// The image tiles come from L2 (~800 cycles away) and a wave has its SIMD to itself, so the
// loads are software-pipelined: k-steps go in chunks of kBigChunk, the next chunk (of this
// tile or of the wave's next tile) is in flight while the current one feeds the MFMAs.
// Raw buffer loads (descriptor in SGPRs, tile offset scalar, lane offset one VGPR): flat
// addressing would cost a 64-bit VGPR pair per load in flight.
constexpr int kBigChunk = 16;

RMH_DEV int big_send(int i, int ks) { return (4 * i + 4 < ks) ? 4 * i + 4 : ks; } // k-steps >= ks carry zeros

RMH_DEV void big_fetch(__amdgpu_buffer_rsrc_t img, uint32_t voff, int i, int s0, int ks, double (&buf)[kBigChunk])
{
    const int send = big_send(i, ks);
    const uint32_t base = (uint32_t)(2 * i * (i + 1)) * 512u;
#pragma unroll
    for (int k = 0; k < kBigChunk; ++k) {
        const int s = (s0 + k < send) ? s0 + k : send - 1; // clamped: the MFMA is skipped
        buf[k] = buf_load(img, voff, base + (uint32_t)s * 512u);
    }
}

// one output tile for CB chain blocks: a[j] += sum_{s < send} img(i, s) Y[j][s].  On entry `buf`
// holds the tile's first chunk; on exit the first chunk of tile `inext` (when inext < NT).
template <int CB>
RMH_DEV void big_tile(__amdgpu_buffer_rsrc_t img, const double *Ylane, int ystride, uint32_t voff, int i,
                      int inext, int NT, int ks, double4_t (&a)[CB], double (&buf)[kBigChunk])
{
    const int send = big_send(i, ks);
    for (int s0 = 0; s0 < send; s0 += kBigChunk) {
        double nxt[kBigChunk];
        const bool more = s0 + kBigChunk < send;
        if (more) big_fetch(img, voff, i, s0 + kBigChunk, ks, nxt);
        else if (inext < NT) big_fetch(img, voff, inext, 0, ks, nxt);
#pragma unroll
        for (int k = 0; k < kBigChunk; ++k) {
            if (s0 + k < send) {
#pragma unroll
                for (int j = 0; j < CB; ++j) a[j] = mfma(buf[k], Ylane[j * ystride + (s0 + k) * 64], a[j]);
            }
        }
        if (more || inext < NT) {
#pragma unroll
            for (int k = 0; k < kBigChunk; ++k) buf[k] = nxt[k];
        }
    }
}

// acc[tt][j] (tile i = w + 4 tt, block j) += sum_s img(i, s) Y[j][s]
template <int NTW, int CB>
RMH_DEV void big_mul_lower(const double *img, const double *Y, int ystride, int lane, int w, int NT, int ks,
                           double4_t (&acc)[NTW][CB])
{
    const __amdgpu_buffer_rsrc_t r = rsrc(img, (int64_t)tri_tiles(NT) * 64);
    const uint32_t voff = (uint32_t)lane * 8u;
    double buf[kBigChunk];
    if (w < NT) big_fetch(r, voff, w, 0, ks, buf);
#pragma unroll
    for (int tt = 0; tt < NTW; ++tt) {
        const int i = w + 4 * tt;
        if (i < NT) big_tile<CB>(r, Y + lane, ystride, voff, i, i + 4, NT, ks, acc[tt], buf);
        asm volatile("" ::: "memory"); // one tile at a time: no hoisting of the later tiles' loads
    }
}

// uu[j] = sum over this wave's tiles of |img(i, .) Y[j]|^2 (per lane; the caller adds the lane groups)
template <int CB>
RMH_DEV void big_norm2(const double *img, const double *Y, int ystride, int lane, int w, int NT, int ks,
                       double (&uu)[CB])
{
    const __amdgpu_buffer_rsrc_t r = rsrc(img, (int64_t)tri_tiles(NT) * 64);
    const uint32_t voff = (uint32_t)lane * 8u;
    double buf[kBigChunk];
    if (w < NT) big_fetch(r, voff, w, 0, ks, buf);
#pragma unroll
    for (int j = 0; j < CB; ++j) uu[j] = 0.0;
    for (int i = w; i < NT; i += 4) {
        double4_t u[CB];
#pragma unroll
        for (int j = 0; j < CB; ++j) u[j] = double4_t{0.0, 0.0, 0.0, 0.0};
        big_tile<CB>(r, Y + lane, ystride, voff, i, i + 4, NT, ks, u, buf);
#pragma unroll
        for (int j = 0; j < CB; ++j) uu[j] += u[j].x * u[j].x + u[j].y * u[j].y + u[j].z * u[j].z + u[j].w * u[j].w;
    }
}

// rows 16i + 4r + g (r = 0..3) of chain c in a (rows, ld) parameter-major array, through raw
// buffer instructions: descriptor in SGPRs, voff = (g ld + c) 8 one VGPR per chain block, the
// row offset a scalar -- flat addressing makes the compiler keep a 64-bit address per
// (tile, block, row) live across the iteration loop (~300 VGPRs)
RMH_DEV double4_t big_load_tile(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t ld8, int i)
{
    double4_t v;
    v.x = buf_load(r, voff, ld8 * (uint32_t)(16 * i));
    v.y = buf_load(r, voff, ld8 * (uint32_t)(16 * i + 4));
    v.z = buf_load(r, voff, ld8 * (uint32_t)(16 * i + 8));
    v.w = buf_load(r, voff, ld8 * (uint32_t)(16 * i + 12));
    return v;
}

// rows >= `rows` are not stored (history entries hold 4 ceil(d/4) rows)
RMH_DEV void big_store_tile(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t ld8, int g, int i, int rows,
                            const double4_t &v)
{
    const int r0 = 16 * i + g;
    if (r0 < rows) buf_store(r, voff, ld8 * (uint32_t)(16 * i), v.x);
    if (r0 + 4 < rows) buf_store(r, voff, ld8 * (uint32_t)(16 * i + 4), v.y);
    if (r0 + 8 < rows) buf_store(r, voff, ld8 * (uint32_t)(16 * i + 8), v.z);
    if (r0 + 12 < rows) buf_store(r, voff, ld8 * (uint32_t)(16 * i + 12), v.w);
}

constexpr int big_red_doubles(int CB) { return CB * 4 * 16 + CB * 16; } // |u|^2 partials, accept flags

} // namespace

// T iterations of Sampler.sample for every chain; theta (dp, ld) and logpost (ld) in global
template <int NTW, int CB>
__global__ void __launch_bounds__(256) k_big_rw(GaussArgs m, RwArgs pr, ChainArgs ch, RunArgs run, int NT)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int ystride = NT * 256;
    double *Y = lds;                                // CB x (4 NT k-steps x 64 lanes)
    double *tab = lds + (size_t)CB * ystride;       // inverse-CDF table
    double *red = tab + kIcdfDoubles;               // [CB][4][16] partial |u|^2
    double *flag = red + CB * 64;                   // [CB][16] accept flags
    icdf_stage(tab, ch.icdf);
    __syncthreads();

    // (the wave index as a scalar: every tile / row offset derived from it is then scalar arithmetic)
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int cl = lane & 15, g = lane >> 4;
    // chain blocks of this workgroup; blocks beyond the end repeat the last one and store nothing
    int64_t cj[CB];
    uint32_t vj[CB]; // byte offset of (row g, chain) in a (rows, ld) array
    bool act[CB];
    const uint32_t ld8_0 = (uint32_t)ch.ld * 8u;
    uint32_t ld8 = ld8_0;
#pragma unroll
    for (int j = 0; j < CB; ++j) {
        int64_t b = (int64_t)blockIdx.x * CB + j;
        act[j] = b < ch.nblocks;
        if (!act[j]) b = ch.nblocks - 1;
        cj[j] = b * 16 + cl;
        vj[j] = (uint32_t)((int64_t)g * ch.ld + cj[j]) * 8u;
    }
    const __amdgpu_buffer_rsrc_t rth = rsrc(ch.theta, (int64_t)16 * NT * ch.ld);
    // wave j < CB owns the streams, the log posterior and the decisions of block j
    const int jw = w < CB ? w : 0;
    int64_t cown = cj[0];
    bool aown = act[0];
#pragma unroll
    for (int j = 1; j < CB; ++j)
        if (jw == j) {
            cown = cj[j];
            aown = act[j];
        }
    const int64_t slot = (cown >> 4) * 64 + lane;
    double lp = ch.logpost[cown];
    Xo rng = xo_load(ch.rng, ch.nslots, slot);

    // history bookkeeping of Sampler.run (sampler.py:49-54), as Keep in rmh_run.hpp
    int64_t knext = run.nburn, kentry = 0;
    auto due = [&](int64_t e, __amdgpu_buffer_rsrc_t &dst) {
        if (e != knext) return false;
        knext += run.nthin;
        const int64_t k = kentry++;
        if (!run.h_theta) return false;
        dst = rsrc(run.h_theta + k * (int64_t)(4 * ch.ks) * ch.ld, (int64_t)(4 * ch.ks) * ch.ld);
        if (w < CB && aown && g == 0) run.h_logpost[k * ch.ld + cown] = lp;
        return true;
    };
    {
        __amdgpu_buffer_rsrc_t dst = rth;
        if (due(0, dst))
            for (int i = w; i < NT; i += 4) {
#pragma unroll
                for (int j = 0; j < CB; ++j)
                    if (act[j]) big_store_tile(dst, vj[j], ld8, g, i, 4 * ch.ks, big_load_tile(rth, vj[j], ld8, i));
            }
    }

    for (int64_t t = 0; t < run.T; ++t) {
        asm volatile("" ::: "memory"); // nothing of an iteration is hoisted out of the loop
        // (nor its row offsets: hoisted, the ~100 offsets ld8 * row of a wave's tiles are spilled before the loop
        // and reloaded one by one inside it -- every reload a vmcnt(0) drain of the state stores in flight)
        ld8 = ld8_0;
        asm volatile("" : "+s"(ld8));
        // 1. waves 0..CB-1: the iteration's normals of their block (two per word, rmh-rng v4)
        if (w < CB) {
            double *Yw = Y + jw * ystride;
            for (int mm = 0; mm < ch.M; ++mm) { // Y[s] is coordinate 4s + g: halves of word mm -> s = 2mm, 2mm+1
                const uint64_t wd = xo_next(rng);
                const uint32_t hw[2] = {(uint32_t)(wd >> 32), (uint32_t)wd};
            double zw[2];
            normals_lockstep<2>(hw, tab, zw);
            const double zh = zw[0], zl = zw[1];
                Yw[(2 * mm) * 64 + lane] = (8 * mm + g < ch.d) ? zh : 0.0;
                if (2 * mm + 1 < 4 * NT) Yw[(2 * mm + 1) * 64 + lane] = (8 * mm + 4 + g < ch.d) ? zl : 0.0;
            }
        }
        __syncthreads();
        // 2. theta' = theta + (scale L) xi; register r of tile i is row 16i + 4r + g of chain cl
        double4_t thp[NTW][CB];
#pragma unroll
        for (int tt = 0; tt < NTW; ++tt) {
            const int i = w + 4 * tt;
#pragma unroll
            for (int j = 0; j < CB; ++j)
                thp[tt][j] = (i < NT) ? big_load_tile(rth, vj[j], ld8, i) : double4_t{0.0, 0.0, 0.0, 0.0};
        }
        big_mul_lower<NTW, CB>(pr.Ltiles, Y, ystride, lane, w, NT, ch.ks, thp);
        __syncthreads(); // everyone is done with xi
        // 3. Y <- theta' - mu in the B-operand layout (k-step 4i + r, same lane)
#pragma unroll
        for (int tt = 0; tt < NTW; ++tt) {
            const int i = w + 4 * tt;
            if (i < NT) {
                const double *mu = m.mu_pad + 16 * i + g;
                const double m0 = mu[0], m1 = mu[4], m2 = mu[8], m3 = mu[12];
#pragma unroll
                for (int j = 0; j < CB; ++j) {
                    double *Yj = Y + j * ystride + lane;
                    Yj[(4 * i + 0) * 64] = thp[tt][j].x - m0;
                    Yj[(4 * i + 1) * 64] = thp[tt][j].y - m1;
                    Yj[(4 * i + 2) * 64] = thp[tt][j].z - m2;
                    Yj[(4 * i + 3) * 64] = thp[tt][j].w - m3;
                }
            }
        }
        __syncthreads();
        // 4. |W (theta' - mu)|^2
        double uu[CB];
        big_norm2<CB>(m.Wtiles, Y, ystride, lane, w, NT, ch.ks, uu);
#pragma unroll
        for (int j = 0; j < CB; ++j) {
            const double s4 = sum4(uu[j]);
            if (g == 0) red[(j * 4 + w) * 16 + cl] = s4;
        }
        __syncthreads();
        // 5. wave j decides for block j (sampler.py:83-84) and publishes the flags
        if (w < CB) {
            const double *rj = red + jw * 64;
            const double tot = ((rj[cl] + rj[16 + cl]) + rj[32 + cl]) + rj[48 + cl];
            const double logL = -0.5 * ((tot + m.dlog2pi) + m.logdetC);
            const double lpp = combine_post(0.0, logL);
            const double uacc = Tile<1>::accept_uniform(rng, g, cl);
            const bool ok = mh_accept(lpp, lp, 0.0, uacc);
            lp = ok ? lpp : lp;
            if (g == 0) {
                flag[jw * 16 + cl] = ok ? 1.0 : 0.0;
                if (run.h_accept && aown) run.h_accept[t * ch.ld + cown] = ok ? 1 : 0;
            }
        }
        __syncthreads();
        // 6. new state = accepted ? theta' : theta, written for all 16 chains of a block (whole
        //    lines), and recorded when this state is kept
        __amdgpu_buffer_rsrc_t hdst = rth;
        const bool keep = due(t + 1, hdst);
#pragma unroll
        for (int tt = 0; tt < NTW; ++tt) {
            const int i = w + 4 * tt;
            if (i < NT) {
#pragma unroll
                for (int j = 0; j < CB; ++j) {
                    if (!act[j]) continue;
                    const bool ok = flag[j * 16 + cl] != 0.0;
                    const double4_t old = big_load_tile(rth, vj[j], ld8, i);
                    double4_t v;
                    v.x = ok ? thp[tt][j].x : old.x;
                    v.y = ok ? thp[tt][j].y : old.y;
                    v.z = ok ? thp[tt][j].z : old.z;
                    v.w = ok ? thp[tt][j].w : old.w;
                    big_store_tile(rth, vj[j], ld8, g, i, 16 * NT, v);
                    if (keep) big_store_tile(hdst, vj[j], ld8, g, i, 4 * ch.ks, v);
                }
            }
            asm volatile("" ::: "memory"); // tile by tile: the old rows are not all loaded up front
        }
        // red[], flag[] and Y are rewritten only after the next iteration's barriers
    }
    if (w < CB && aown) {
        if (g == 0) ch.logpost[cown] = lp;
        xo_store(ch.rng, ch.nslots, slot, rng);
    }
}

// Model.log_prior / log_likelihood / log_posterior for the states in ch.theta
__global__ void __launch_bounds__(256) k_big_eval(GaussArgs m, ChainArgs ch, int NT, double *lp_out,
                                                  double *lprior_out, double *llik_out)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double *Y = lds, *red = lds + (size_t)NT * 256;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int cl = lane & 15, g = lane >> 4;
    const int64_t c = (int64_t)blockIdx.x * 16 + cl;
    const __amdgpu_buffer_rsrc_t rth = rsrc(ch.theta, (int64_t)16 * NT * ch.ld);
    const uint32_t voff = (uint32_t)((int64_t)g * ch.ld + c) * 8u, ld8 = (uint32_t)ch.ld * 8u;
    for (int i = w; i < NT; i += 4) {
        const double4_t v = big_load_tile(rth, voff, ld8, i);
        const double *mu = m.mu_pad + 16 * i + g;
        Y[(4 * i + 0) * 64 + lane] = v.x - mu[0];
        Y[(4 * i + 1) * 64 + lane] = v.y - mu[4];
        Y[(4 * i + 2) * 64 + lane] = v.z - mu[8];
        Y[(4 * i + 3) * 64 + lane] = v.w - mu[12];
    }
    __syncthreads();
    double uu[1];
    big_norm2<1>(m.Wtiles, Y, NT * 256, lane, w, NT, ch.ks, uu);
    const double s4 = sum4(uu[0]);
    if (g == 0) red[w * 16 + cl] = s4;
    __syncthreads();
    if (w == 0 && g == 0) {
        const double tot = ((red[cl] + red[16 + cl]) + red[32 + cl]) + red[48 + cl];
        const double logL = -0.5 * ((tot + m.dlog2pi) + m.logdetC);
        if (lp_out) lp_out[c] = combine_post(0.0, logL);
        if (lprior_out) lprior_out[c] = 0.0;
        if (llik_out) llik_out[c] = logL;
    }
}

template <class K>
static hipError_t big_attr(K kernel, size_t lds)
{
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    if (lds > 48 * 1024)
        return hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    return hipSuccess;
}

template <int NTW, int CB>
static hipError_t big_rw_launch(int NT, const GaussArgs &m, const RwArgs &p, const ChainArgs &c,
                                const RunArgs &r, hipStream_t st)
{
    const size_t lds = sizeof(double) * ((size_t)CB * NT * 256 + kIcdfDoubles + big_red_doubles(CB));
    hipError_t e = big_attr(k_big_rw<NTW, CB>, lds);
    if (e != hipSuccess) return e;
    const unsigned grid = (unsigned)((c.nblocks + CB - 1) / CB);
    hipLaunchKernelGGL((k_big_rw<NTW, CB>), dim3(grid), dim3(256), lds, st, m, p, c, r, NT);
    return hipGetLastError();
}

// chain blocks per workgroup: as many as the LDS holds (each image tile loaded from L2 then
// feeds that many MFMAs) -- 4 up to d = 256 (one block per wave at most: wave j owns block j's
// streams), 2 up to d = 512, 1 up to d = 1024
hipError_t launch_big_gauss_rw(int NT, const GaussArgs &m, const RwArgs &p, const ChainArgs &c,
                               const RunArgs &r, hipStream_t st)
{
    if (NT <= 8) return big_rw_launch<2, 4>(NT, m, p, c, r, st);
    if (NT <= 16) return big_rw_launch<4, 4>(NT, m, p, c, r, st);
    if (NT <= 32) return big_rw_launch<8, 2>(NT, m, p, c, r, st);
    if (NT <= 64) return big_rw_launch<16, 1>(NT, m, p, c, r, st);
    return hipErrorInvalidValue;
}

hipError_t launch_big_gauss_eval(int NT, const GaussArgs &m, const ChainArgs &c, double *lp_out,
                                 double *lprior_out, double *llik_out, hipStream_t st)
{
    const size_t lds = sizeof(double) * ((size_t)NT * 256 + 64);
    hipError_t e = big_attr(k_big_eval, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_big_eval, dim3((unsigned)c.nblocks), dim3(256), lds, st, m, c, NT, lp_out,
                       lprior_out, llik_out);
    return hipGetLastError();
}

} // namespace rmh
