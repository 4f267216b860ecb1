// rmh_mfma.hpp -- the 16-chain wavefront tile shared by the MFMA kernels.
//
// One wavefront owns 16 chains.  Lane l = 16*g + cl holds chain cl (of the
// wave's 16) and the parameters p = 4*s + g, s = 0..S-1 (S = 4*NT, d padded to
// 16*NT): register s of the lane is parameter 4s+g.  That single layout is at
// once
//   * the B operand of v_mfma_f64_16x16x4_f64 for k-step s
//     (B[k = lane>>4][n = lane&15], one f64 per lane), and
//   * its C/D layout (col = lane&15, row = (lane>>4) + 4*reg): register r of
//     output tile i is row 16i + 4r + g, i.e. s = 4i + r,
// so the result of one triangular contraction feeds the next one without any
// lane movement, and a chain's state never leaves registers inside a launch.
//
// Triangular operands (the Cholesky factor of the proposal, W = L^-1 of the
// target) live in LDS as "tile images": tile (i, s) holds the 16x4 block
// rows 16i..16i+15, columns 4s..4s+3 in A-operand lane order
// (lane l <- M[16i + (l&15)][4s + (l>>4)]), 512 B per tile, so every A fetch is
// one conflict-free ds_read_b64 at base + 8*lane + imm.
#pragma once
#include "rmh_device.hpp"

namespace rmh {

RMH_DEV double4_t mfma(double a, double b, double4_t c)
{
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

// raw buffer access: descriptor in SGPRs, 32-bit lane offset, scalar row offset
typedef unsigned v2u_t __attribute__((ext_vector_type(2)));

RMH_DEV __amdgpu_buffer_rsrc_t rsrc(const void *base, int64_t doubles)
{
    const int64_t bytes = doubles * 8;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0,
                                             bytes > 0xFFFFFFFFLL ? (int)0xFFFFFFFF : (int)bytes,
                                             0x00020000);
}

RMH_DEV double buf_load(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff)
{
    const v2u_t v = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
    return __longlong_as_double((long long)(((uint64_t)v.y << 32) | v.x));
}

RMH_DEV void buf_store(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff, double x)
{
    const uint64_t b = (uint64_t)__double_as_longlong(x);
    v2u_t v;
    v.x = (unsigned)b;
    v.y = (unsigned)(b >> 32);
    __builtin_amdgcn_raw_buffer_store_b64(v, r, voff, soff, 0);
}

// xoshiro state through raw buffer instructions: one descriptor and ONE 32-bit lane offset for the
// four words (flat addressing keeps four 64-bit addresses alive from the load at kernel entry to
// the store at its end -- in the HMC kernels those were spilled to scratch)
RMH_DEV Xo xo_load_buf(const uint64_t *rng, int64_t nslots, int64_t slot)
{
    const __amdgpu_buffer_rsrc_t r = rsrc(rng, 4 * nslots);
    const uint32_t voff = (uint32_t)slot * 8u, row = (uint32_t)nslots * 8u;
    Xo x;
    x.s0 = (uint64_t)__double_as_longlong(buf_load(r, voff, 0u));
    x.s1 = (uint64_t)__double_as_longlong(buf_load(r, voff, row));
    x.s2 = (uint64_t)__double_as_longlong(buf_load(r, voff, 2u * row));
    x.s3 = (uint64_t)__double_as_longlong(buf_load(r, voff, 3u * row));
    return x;
}

RMH_DEV void xo_store_buf(uint64_t *rng, int64_t nslots, int64_t slot, const Xo &x)
{
    const __amdgpu_buffer_rsrc_t r = rsrc(rng, 4 * nslots);
    const uint32_t voff = (uint32_t)slot * 8u, row = (uint32_t)nslots * 8u;
    buf_store(r, voff, 0u, __longlong_as_double((long long)x.s0));
    buf_store(r, voff, row, __longlong_as_double((long long)x.s1));
    buf_store(r, voff, 2u * row, __longlong_as_double((long long)x.s2));
    buf_store(r, voff, 3u * row, __longlong_as_double((long long)x.s3));
}

// one byte per lane through a raw buffer store (the accept flags)
RMH_DEV void buf_store_u8(uint8_t *base, int64_t bytes, uint32_t voff, uint8_t x)
{
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(
        base, 0, bytes > 0xFFFFFFFFLL ? (int)0xFFFFFFFF : (int)bytes, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b8(x, r, voff, 0, 0);
}

// ---- triangular products with the tile image streamed from global memory (L2) ----------------
// pCN's (rho_c L)^-1 and the mass-matrix images of HMC do not fit LDS next to the model's own
// images, so those products read their A operands from L2, ~800 cycles away, on a SIMD the wave has
// to itself.  TriSeq is the order in which a product consumes the (tile row i, k-step s) slices
// (s-major: consecutive MFMAs go to different accumulators) with each slice's index in the image;
// tri_stream keeps kStreamChunk slices in flight ahead of the MFMAs through raw buffer loads
// (descriptor in SGPRs, one lane-offset VGPR, the slice offset an immediate/scalar) and feeds
// every loaded slice to NV right-hand sides.  (Before round 3 these products issued their loads
// two k-steps at a time behind a scheduling fence: latency-bound, mass-matrix HMC ran at 10x the
// identity-mass time.)
constexpr int kStreamChunk = 16;

template <int NT, bool UPPER>
struct TriSeq {
    static constexpr int S = 4 * NT, N = 2 * NT * (NT + 1);
    int i[N], s[N], idx[N];
    constexpr TriSeq() : i{}, s{}, idx{}
    {
        int n = 0;
        for (int ss = 0; ss < S; ++ss)
            for (int ii = (UPPER ? 0 : ss / 4); ii < (UPPER ? ss / 4 + 1 : NT); ++ii) {
                i[n] = ii;
                s[n] = ss;
                idx[n] = UPPER ? ii * S - 2 * ii * (ii - 1) + (ss - 4 * ii) : 2 * ii * (ii + 1) + ss;
                ++n;
            }
    }
};

template <int NT, bool UPPER, int C>
RMH_DEV void tri_stream_fetch(__amdgpu_buffer_rsrc_t img, uint32_t voff, double (&buf)[kStreamChunk])
{
    constexpr TriSeq<NT, UPPER> seq{};
#pragma unroll
    for (int k = 0; k < kStreamChunk; ++k) {
        constexpr int N = TriSeq<NT, UPPER>::N;
        const int n = C * kStreamChunk + k < N ? C * kStreamChunk + k : N - 1;
        buf[k] = buf_load(img, voff, (uint32_t)seq.idx[n] * 512u);
    }
}

template <int NT, bool UPPER, int NV, int C>
RMH_DEV void tri_stream_chunk(__amdgpu_buffer_rsrc_t img, uint32_t voff, const double (&B)[NV][4 * NT],
                              double4_t (&acc)[NV][NT], int ks, double (&cur)[kStreamChunk])
{
    constexpr TriSeq<NT, UPPER> seq{};
    constexpr int N = TriSeq<NT, UPPER>::N, S = 4 * NT;
    constexpr int NCH = (N + kStreamChunk - 1) / kStreamChunk;
    double nxt[kStreamChunk];
    if constexpr (C + 1 < NCH) tri_stream_fetch<NT, UPPER, C + 1>(img, voff, nxt);
#pragma unroll
    for (int k = 0; k < kStreamChunk; ++k) {
        const int n = C * kStreamChunk + k;
        if (n < N) {
            const int i = seq.i[n], s = seq.s[n];
            if (s < S - 4 || s < ks) { // k-steps >= ks = ceil(d/4) carry zeros (Tile::mul_lower)
#pragma unroll
                for (int v = 0; v < NV; ++v) acc[v][i] = mfma(cur[k], B[v][s], acc[v][i]);
            }
        }
    }
    if constexpr (C + 1 < NCH) tri_stream_chunk<NT, UPPER, NV, C + 1>(img, voff, B, acc, ks, nxt);
}

// acc[v] += T B[v] for the triangular tile image T at `img` (lower, or upper when UPPER)
template <int NT, bool UPPER, int NV>
RMH_DEV void tri_stream(const double *img, int lane, const double (&B)[NV][4 * NT], double4_t (&acc)[NV][NT], int ks)
{
    const __amdgpu_buffer_rsrc_t r = rsrc(img, (int64_t)TriSeq<NT, UPPER>::N * 64);
    const uint32_t voff = (uint32_t)lane * 8u;
    double buf[kStreamChunk];
    tri_stream_fetch<NT, UPPER, 0>(r, voff, buf);
    tri_stream_chunk<NT, UPPER, NV, 0>(r, voff, B, acc, ks, buf);
}

// one right-hand side: acc += T B
template <int NT, bool UPPER>
RMH_DEV void tri_stream1(const double *img, int lane, const double (&B)[4 * NT], double4_t (&acc)[NT], int ks)
{
    tri_stream<NT, UPPER, 1>(img, lane, reinterpret_cast<const double (&)[1][4 * NT]>(B),
                             reinterpret_cast<double4_t (&)[1][NT]>(acc), ks);
}

template <int NT>
struct Tile {
    static constexpr int S = 4 * NT;

    // index of tile (i, s) in a lower-triangular image (s <= 4i+3)
    static constexpr int lo(int i, int s) { return 2 * i * (i + 1) + s; }
    // index of tile (i, s) in an upper-triangular image (s >= 4i)
    static constexpr int up(int i, int s) { return i * S - 2 * i * (i - 1) + (s - 4 * i); }

    // acc[i] += sum_{s <= 4i+3} A_lo(i,s) * B[s]     (A = lower-triangular image)
    // k-steps s >= ks = ceil(d/4) carry B[s] = 0 (padding coordinates) and are
    // skipped: only the last four can be, a wave-uniform test each.
    // (img_lane points into LDS; images in global memory go through tri_stream)
    static RMH_DEV void mul_lower(const double *img_lane, const double (&B)[S], double4_t (&acc)[NT],
                                  int ks)
    {
#pragma unroll
        for (int s = 0; s < S; ++s) {
            if (s < S - 4 || s < ks) {
#pragma unroll
                for (int i = s / 4; i < NT; ++i)
                    acc[i] = mfma(img_lane[lo(i, s) * 64], B[s], acc[i]);
            }
        }
    }

    // acc[i] += sum_{s >= 4i} A_up(i,s) * B[s]       (A = upper-triangular image)
    static RMH_DEV void mul_upper(const double *img_lane, const double (&B)[S], double4_t (&acc)[NT],
                                  int ks)
    {
#pragma unroll
        for (int s = 0; s < S; ++s) {
            if (s < S - 4 || s < ks) {
#pragma unroll
                for (int i = 0; i <= s / 4; ++i)
                    acc[i] = mfma(img_lane[up(i, s) * 64], B[s], acc[i]);
            }
        }
    }

    static RMH_DEV void unpack(const double4_t (&acc)[NT], double (&v)[S])
    {
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            v[4 * i + 0] = acc[i].x;
            v[4 * i + 1] = acc[i].y;
            v[4 * i + 2] = acc[i].z;
            v[4 * i + 3] = acc[i].w;
        }
    }

    static RMH_DEV void pack(const double (&v)[S], double4_t (&acc)[NT])
    {
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            acc[i].x = v[4 * i + 0];
            acc[i].y = v[4 * i + 1];
            acc[i].z = v[4 * i + 2];
            acc[i].w = v[4 * i + 3];
        }
    }

    // S normals for this lane: Mw = ceil(d/8) words of the lane's sub-stream, two normals per
    // word -- the high half of word m is coordinate 8m+g (register 2m), the low half 8m+4+g
    // (register 2m+1); zero beyond d (rmh-rng v4).  tab = ICDF table in LDS.
    // Groups of 4 normals (2 words): the xoshiro steps are a serial chain, the four table
    // evaluations of a group are branch-free so that their LDS latencies and Horner chains
    // overlap (a single evaluation is latency-bound, ~400 cycles).
    // (A step-major variant -- all table reads of one Horner step for the whole group, then
    // its fmas, one coefficient pair live per normal -- was measured with the v2 stream: cfg2
    // 38.5 vs 38.1 ms, cfg5 23.7 vs 22.4 ms; not kept.)
    static RMH_DEV void normals(Xo &rng, int Mw, int d, int g, const double *tab, double (&xi)[S])
    {
        constexpr int GRP = 4;
#pragma unroll
        for (int m0 = 0; m0 < S; m0 += GRP) {
            if (m0 < 2 * Mw) {
                uint64_t w[GRP / 2];
#pragma unroll
                for (int k = 0; k < GRP / 2; ++k) {
                    w[k] = 0;
                    if (m0 / 2 + k < Mw) w[k] = xo_next(rng);
                }
                // four normals in lockstep; the small tiles run many waves per SIMD on few
                // registers and take them in pairs (d = 16: 3.62 ms per 1000 iterations against 3.80)
                constexpr int LS = (S <= 8) ? 2 : GRP;
#pragma unroll
                for (int k0 = 0; k0 < GRP; k0 += LS) {
                    uint32_t h[LS];
                    double z[LS];
#pragma unroll
                    for (int k = 0; k < LS; ++k)
                        h[k] = ((k0 + k) & 1) ? (uint32_t)w[(k0 + k) / 2] : (uint32_t)(w[(k0 + k) / 2] >> 32);
                    normals_lockstep<LS>(h, tab, z);
#pragma unroll
                    for (int k = 0; k < LS; ++k) xi[m0 + k0 + k] = (4 * (m0 + k0 + k) + g < d) ? z[k] : 0.0;
                }
            } else {
#pragma unroll
                for (int k = 0; k < GRP; ++k) xi[m0 + k] = 0.0;
            }
        }
    }

    // The 64-bit-resolution normal mode (rmh_device.hpp): ks = ceil(d/4) words of the lane's sub-stream, word
    // s is coordinate 4s + g (register s); zero beyond d.  Pairs in lockstep: a normal holds six
    // 16-byte coefficient pairs.
    static RMH_DEV void normals64(Xo &rng, int ks, int d, int g, const double *tab, double (&xi)[S])
    {
        if constexpr (S <= 8) { // the small tiles live on few registers and many waves per SIMD: one at a time
#pragma unroll
            for (int s = 0; s < S; ++s) {
                xi[s] = 0.0;
                if (s < ks) {
                    const uint64_t w[1] = {xo_next(rng)};
                    double z[1];
                    normals64_lockstep<1>(w, tab, z);
                    xi[s] = (4 * s + g < d) ? z[0] : 0.0;
                }
            }
        } else {
#pragma unroll
            for (int s0 = 0; s0 < S; s0 += 2) {
                if (s0 < ks) {
                    uint64_t w[2];
                    double z[2];
                    w[0] = xo_next(rng);
                    w[1] = 0;
                    if (s0 + 1 < ks) w[1] = xo_next(rng);
                    normals64_lockstep<2>(w, tab, z);
                    xi[s0] = (4 * s0 + g < d) ? z[0] : 0.0;
                    xi[s0 + 1] = (s0 + 1 < ks && 4 * (s0 + 1) + g < d) ? z[1] : 0.0;
                } else {
                    xi[s0] = xi[s0 + 1] = 0.0;
                }
            }
        }
    }

    // the kernel's normals in the chains' mode (nmode: ChainArgs::nmode, wave-uniform)
    static RMH_DEV void normals_mode(int nmode, Xo &rng, int Mw, int ks, int d, int g, const double *tab,
                                     double (&xi)[S])
    {
        if (nmode) normals64(rng, ks, d, g, tab, xi);
        else normals(rng, Mw, d, g, tab, xi);
    }

    // The same stream for a dimension known at compile time to need exactly MW words per
    // sub-stream and to be <= DMAX (8 (MW-1) < d <= min(8 MW, DMAX)): no wave-uniform word and
    // k-step predicates (in the generic version they live in spilled SGPRs: ~85 v_readlane and
    // two v_cndmask per normal and iteration for d = 100), a select only on the last two k-steps,
    // and no evaluation at all for a k-step whose coordinates are all >= DMAX (its half word is
    // still drawn with its word).
    template <int MW, int DMAX, int WPB = 2>
    static RMH_DEV void normals_fixed(Xo &rng, int d, int g, const double *tab, double (&xi)[S])
    {
        static_assert(2 * MW <= S && DMAX <= 8 * MW && DMAX > 8 * (MW - 1), "word count of the dimension");
#pragma unroll
        for (int s = 0; s < S; ++s) xi[s] = 0.0;
        fixed_block<MW, DMAX, WPB, 0>(rng, d, g, tab, xi);
    }

    // Words M0 .. M0+WPB-1 of the stream (WPB = 2: four normals) in one basic block, evaluated
    // in lockstep (normals_lockstep), then the following blocks.  Straight-line code over all
    // words lets the scheduler batch the table reads of all 25 normals and spill their
    // coefficients (105 scratch round trips per iteration, 210 ms); an opaque always-taken scalar
    // branch costs two SALU instructions.  Normals per block, cfg2: 1 (evaluated one after the
    // other) 32.6 ms, 2 in lockstep 32.0 ms, 4 in lockstep 31.8 ms (44 coefficient registers per
    // normal pair: kernels at four waves per SIMD take WPB = 1).  With the v4 stream (14 coefficient
    // registers per normal) the cfg2 kernel takes WPB = 3, six normals per block; eight spill.
    template <int MW, int DMAX, int WPB, int M0>
    static RMH_DEV void fixed_block(Xo &rng, int d, int g, const double *tab, double (&xi)[S])
    {
        if constexpr (M0 < MW) {
            constexpr int NWD = (MW - M0 < WPB) ? MW - M0 : WPB;       // words of this block
            constexpr int S0 = 2 * M0;                                                  // its first k-step
            constexpr int KS = (DMAX + 3) / 4;                  // k-steps that hold coordinates < DMAX
            constexpr int NS = (KS - S0 < 2 * NWD) ? KS - S0 : 2 * NWD;                 // normals to evaluate
            constexpr bool LAST = S0 + NS > 2 * MW - 2;         // k-steps 2MW-2.. may hold coordinates >= d
            int go;
            asm volatile("s_mov_b32 %0, 1" : "=s"(go));
            if (go) {
                uint32_t h[2 * NWD];
#pragma unroll
                for (int q = 0; q < NWD; ++q) {
                    const uint64_t w = xo_next(rng);
                    h[2 * q] = (uint32_t)(w >> 32);
                    h[2 * q + 1] = (uint32_t)w;
                }
                uint32_t hs[NS];
                double z[NS];
#pragma unroll
                for (int i = 0; i < NS; ++i) hs[i] = h[i];
                normals_lockstep<NS>(hs, tab, z);
#pragma unroll
                for (int i = 0; i < NS; ++i)
                    xi[S0 + i] = (!LAST || S0 + i < 2 * MW - 2 || 4 * (S0 + i) + g < d) ? z[i] : 0.0;
            }
            fixed_block<MW, DMAX, WPB, M0 + WPB>(rng, d, g, tab, xi);
        }
    }

    // the accept uniform: one word of sub-stream 0 (lanes g == 0), broadcast to
    // the chain's four lanes
    static RMH_DEV double accept_uniform(Xo &rng, int g, int cl)
    {
        Xo tmp = rng;
        const uint64_t r = xo_next(tmp);
        if (g == 0) rng = tmp;
        return bcast_row0(u01(r));
    }

    // (16*NT, ld) parameter-major state <-> registers.  State arrays are padded
    // to 16*NT rows (rows >= d hold zeros), so all S accesses are unconditional.
    // Raw buffer instructions: the descriptor (wave-uniform base) sits in SGPRs,
    // the row offset 4s*ld*8 is a scalar and the lane part (g*ld + c)*8 is ONE
    // 32-bit VGPR -- flat/global addressing would cost a 64-bit VGPR pair per row
    // (hoisted out of the iteration loop: 56 registers for d = 100).
    static RMH_DEV void load_state(const double *theta, int64_t ld, int64_t c, int g,
                                   double (&th)[S])
    {
        const __amdgpu_buffer_rsrc_t r = rsrc(theta, (int64_t)(4 * S) * ld);
        const uint32_t voff = (uint32_t)(g * ld + c) * 8u, row = (uint32_t)ld * 32u;
#pragma unroll
        for (int s = 0; s < S; ++s) th[s] = buf_load(r, voff, row * (uint32_t)s);
    }

    static RMH_DEV void store_state(double *theta, int64_t ld, int64_t c, int g,
                                    const double (&th)[S])
    {
        const __amdgpu_buffer_rsrc_t r = rsrc(theta, (int64_t)(4 * S) * ld);
        const uint32_t voff = (uint32_t)(g * ld + c) * 8u, row = (uint32_t)ld * 32u;
#pragma unroll
        for (int s = 0; s < S; ++s) buf_store(r, voff, row * (uint32_t)s, th[s]);
    }

    // history entries hold 4*ks rows (d rounded up to 4): rows 4s..4s+3 are
    // written iff s < ks, a wave-uniform test
    // KS > 0: ks is that compile-time constant (no test, no branch per row)
    template <int KS = 0>
    static RMH_DEV void store_rows(double *dst, int64_t ld, int64_t c, int g, int ks,
                                   const double (&th)[S])
    {
        const __amdgpu_buffer_rsrc_t r = rsrc(dst, (int64_t)(4 * ks) * ld);
        const uint32_t voff = (uint32_t)(g * ld + c) * 8u, row = (uint32_t)ld * 32u;
#pragma unroll
        for (int s = 0; s < S; ++s)
            if (KS > 0 ? s < KS : s < ks) buf_store(r, voff, row * (uint32_t)s, th[s]);
    }
};

} // namespace rmh
