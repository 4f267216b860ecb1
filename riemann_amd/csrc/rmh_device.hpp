// rmh_device.hpp -- device-side random stream and elementary functions (gfx950).
//
// "rmh-rng v4" (Random stream): chain c owns four xoshiro256++
// sub-streams g = 0..3 (stream id 4c+g).  In the MFMA kernels the four lanes
// {cl, cl+16, cl+32, cl+48} of a wavefront hold one chain and lane group g owns
// sub-stream g and the coordinates j = g (mod 4) -- exactly the B-operand
// layout of v_mfma_f64_16x16x4_f64, so normals are born in the register they
// are consumed from.  The inverse-CDF polynomial and the accept test's log are
// defined operation by operation (explicit fma; the file is compiled with
// -ffp-contract=off) so that the stream is bit-identical to the CPU oracle's
// restatement of the same spec.
//
// Replaces the process-global numpy generator used by the reference at
// riemann/proposals/randomwalk.py:25, riemann/proposals/hamiltonian.py:79 and
// riemann/samplers/sampler.py:84.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rmh {

typedef double double4_t __attribute__((ext_vector_type(4)));

#define RMH_DEV __device__ __forceinline__

constexpr uint64_t kGamma = 0x9E3779B97F4A7C15ULL;

RMH_DEV uint64_t mix64(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

struct Xo {
    uint64_t s0, s1, s2, s3;
};

RMH_DEV uint64_t rotl64(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }

RMH_DEV void xo_init(Xo &x, uint64_t seed, uint64_t sid)
{
    uint64_t v = mix64(mix64(seed + kGamma) + sid);
    v += kGamma; x.s0 = mix64(v);
    v += kGamma; x.s1 = mix64(v);
    v += kGamma; x.s2 = mix64(v);
    v += kGamma; x.s3 = mix64(v);
}

RMH_DEV uint64_t xo_next(Xo &x)
{
    const uint64_t r = rotl64(x.s0 + x.s3, 23) + x.s0;
    const uint64_t t = x.s1 << 17;
    x.s2 ^= x.s0;
    x.s3 ^= x.s1;
    x.s1 ^= x.s2;
    x.s0 ^= x.s3;
    x.s2 ^= t;
    x.s3 = rotl64(x.s3, 45);
    return r;
}

// the next word of the stream on the lanes where `own` holds (the stream advances there only);
// an unspecified word elsewhere
RMH_DEV uint64_t xo_next_if(Xo &x, bool own)
{
    Xo tmp = x;
    const uint64_t r = xo_next(tmp);
    if (own) x = tmp;
    return r;
}

// rng state in global memory: word w of stream (chain c, group g) lives at
// rng[w * 4*Npad + (c/16)*64 + g*16 + c%16]  (one coalesced 512-B row per wave)
RMH_DEV int64_t rng_slot(int64_t c, int g) { return (c >> 4) * 64 + g * 16 + (c & 15); }

RMH_DEV Xo xo_load(const uint64_t *rng, int64_t nslots, int64_t slot)
{
    Xo x;
    x.s0 = rng[slot];
    x.s1 = rng[nslots + slot];
    x.s2 = rng[2 * nslots + slot];
    x.s3 = rng[3 * nslots + slot];
    return x;
}

RMH_DEV void xo_store(uint64_t *rng, int64_t nslots, int64_t slot, const Xo &x)
{
    rng[slot] = x.s0;
    rng[nslots + slot] = x.s1;
    rng[2 * nslots + slot] = x.s2;
    rng[3 * nslots + slot] = x.s3;
}

RMH_DEV double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }

// log core: x positive, normal, finite; the result is log(x) + koff*ln2 with the
// exponent offset applied BEFORE the final expression (as the oracle does for
// subnormal inputs: k = -54 after scaling by 2^54).
// OPQ: the polynomial's constants are made opaque at their use.  As plain constants the compiler
// hoists their VGPR copies (v_fmac wants its addend in a register) to kernel entry; in the HMC
// kernels, whose stage loop runs at the register cap, that put five of them in scratch.
// (the constant enters as two 32-bit literals of v_mov_b32 in a volatile asm: nothing about it is
// loop-invariant code the compiler could hoist -- an opaque copy of a constant still is)
#define RMH_LGC(x) lgc<OPQ, (uint32_t)(__builtin_bit_cast(uint64_t, (double)(x)) >> 32), \
                       (uint32_t)__builtin_bit_cast(uint64_t, (double)(x))>(x)
template <bool OPQ, uint32_t HI, uint32_t LO>
RMH_DEV double lgc(double x)
{
    if (!OPQ) return x;
    uint32_t lo, hi;
    asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %3" : "=&v"(lo), "=v"(hi) : "n"(LO), "n"(HI));
    return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
}

template <bool OPQ = false>
RMH_DEV double log_core(double x, int koff)
{
    constexpr double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
    const double Lg1 = RMH_LGC(6.666666666666735130e-01), Lg2 = RMH_LGC(3.999999999940941908e-01),
                 Lg3 = RMH_LGC(2.857142874366239149e-01), Lg4 = RMH_LGC(2.222219843214978396e-01),
                 Lg5 = RMH_LGC(1.818357216161805012e-01), Lg6 = RMH_LGC(1.531383769920937332e-01),
                 Lg7 = RMH_LGC(1.479819860511658591e-01);
    const uint64_t ix = (uint64_t)__double_as_longlong(x);
    int k = (int)(ix >> 52) - 1023 + koff;
    const uint64_t mant = ix & 0x000FFFFFFFFFFFFFULL;
    const bool hi = mant >= 0x0006A09E667F3BCDULL;
    const double m = __longlong_as_double(
        (long long)(mant | (hi ? 0x3FE0000000000000ULL : 0x3FF0000000000000ULL)));
    k += hi ? 1 : 0;
    const double f = m - 1.0;
    const double s = f / (2.0 + f);
    const double z = s * s;
    const double w = z * z;
    const double t1 = w * fma_(w, fma_(w, Lg6, Lg4), Lg2);
    const double t2 = z * fma_(w, fma_(w, fma_(w, Lg7, Lg5), Lg3), Lg1);
    const double R = t2 + t1;
    const double hfsq = 0.5 * f * f;
    const double dk = (double)k;
    return dk * ln2_hi - ((hfsq - fma_(s, hfsq + R, dk * ln2_lo)) - f);
}

// log of a positive, normal, finite double (all the stream needs): no specials.
template <bool OPQ = false>
RMH_DEV double log_pos(double x) { return log_core<OPQ>(x, 0); }

// log with IEEE/numpy specials: nan->nan, <0 -> nan, 0 -> -inf, inf -> inf,
// subnormals rescaled by 2^54.
RMH_DEV double log_full(double x)
{
    const double inf = __builtin_huge_val();
    const bool ok = (x > 0.0) && (x < inf);
    const bool sub = x < 2.2250738585072014e-308;
    const double xs = sub ? x * 0x1p54 : x;
    double r = log_core(ok ? xs : 1.0, sub ? -54 : 0);
    if (!ok) {
        if (x != x) r = x;
        else if (x < 0.0) r = __builtin_nan("");
        else if (x == 0.0) r = -inf;
        else r = x; // +inf
    }
    return r;
}

// log_full with the specials off the hot path: positive normal finite arguments (all but
// measure zero of what the samplers feed it) take log_core only; anything else sends the WAVE
// through log_full once (a wave-uniform, rarely taken branch instead of a dozen selects per call).
RMH_DEV double log_fast(double x)
{
    const bool ok = (x >= 2.2250738585072014e-308) && (x < __builtin_huge_val());
    double r = log_core(ok ? x : 1.0, 0);
    if (__builtin_expect(__any(!ok), 0)) r = ok ? r : log_full(x);
    return r;
}

// three logarithms side by side (straight-line code: the three dependency chains interleave), one
// rare-path test for all of them
RMH_DEV void log_fast3(double a, double b, double c, double &la, double &lb, double &lc)
{
    const double tiny = 2.2250738585072014e-308, inf = __builtin_huge_val();
    const bool oa = (a >= tiny) && (a < inf), ob = (b >= tiny) && (b < inf), oc = (c >= tiny) && (c < inf);
    la = log_core(oa ? a : 1.0, 0);
    lb = log_core(ob ? b : 1.0, 0);
    lc = log_core(oc ? c : 1.0, 0);
    if (__builtin_expect(__any(!(oa && ob && oc)), 0)) {
        la = oa ? la : log_full(a);
        lb = ob ? lb : log_full(b);
        lc = oc ? lc : log_full(c);
    }
}

// ---- link function of the logistic model (not part of the stream spec: the
// oracle evaluates the same quantities with libm, tolerance-checked) -------------
// Table (tools/gen_logistic_table.py): |eta| in [0, 37.5) is cut into 600 cells of width
// 1/16, each with two degree-5 polynomials in t = 2 frac(16 |eta|) - 1:
//   f = log1p(exp(-|eta|)),  s = sigmoid(-|eta|);  worst relative error 5.0e-14.
// Per (f, s) pair 10 fma, six ds_read_b128 (one 96-byte cell) and ~12 other operations.  On this
// chip every VALU cycle is taken from the f64 MFMA pipe, so the link function is priced in
// instructions: round 1 evaluated degree-9 polynomials on cells of width 1/4 (2.1e-16; 18 fma
// and ten reads per pair, 20 coefficient registers per value in flight -- that register
// pressure, not the arithmetic, was what spilled the 8-wave kernel); a closed form (Cody-Waite
// exp + v_rcp_f64 with two Newton steps + fdlibm log, ~63 f64 operations) was slower than
// either.  Beyond 37.5 both functions are < 6e-17 and are taken at the table edge.
// tab = table in LDS (57.6 KB).
constexpr int kLogiNI = 600, kLogiNC = 6, kLogiCell = 2 * kLogiNC, kLogiDoubles = kLogiNI * kLogiCell;

// N values in lockstep: all cells' reads first, then the Horner chains alternating (as
// normals_lockstep).  Outputs in the form the residual and the density term take them:
//   qs = sigmoid(eta) - 1/2 = copysign(q(|eta|), eta)           (y - sigmoid = (y - 1/2) - qs)
//   hv = 16 (log(1 + exp(eta)) - eta/2) = |eta16|/2 + 16 f(|eta|)  (16 (y eta - log(1+exp eta)) = (y - 1/2) eta16 - hv)
// The argument is eta16 = 16 eta (the model stores 16 X): the cell index is its integer part as it
// comes, and the table holds 16 f.  All scalings are powers of two: the same bits as unscaled.
// VALUE = false: qs only (gradient-only evaluations inside an HMC trajectory: half the cell).
template <int N, bool VALUE>
RMH_DEV void logistic_lockstep(const double (&eta)[N], const double *tab, double (&hv)[N], double (&qs)[N])
{
    typedef double d2_t __attribute__((ext_vector_type(2)));
    double t[N];
    const d2_t *c[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        // min(|eta16|, edge) as ONE instruction (a NaN gives the edge, or cell 0 if it was signalling: in
        // range either way); written in C++ the compiler adds a v_max_f64 only to quiet signalling NaNs
        double u;
        asm("v_min_f64 %0, |%1|, %2" : "=v"(u) : "v"(eta[i]), "s"(599.9999999));
        const int idx = (int)u;
        t[i] = __builtin_amdgcn_fract(u);              // the cells' polynomials are in frac(|eta16|) itself
        // 96-byte cells: v_mul_u32_u24 (full rate; a plain 32-bit multiply is a quarter-rate op)
        c[i] = reinterpret_cast<const d2_t *>(__builtin_assume_aligned(
            reinterpret_cast<const char *>(tab) + __umul24((unsigned)idx, (unsigned)(kLogiCell * 8)), 16));
    }
    d2_t cf[N][3], cs[N][3];
#pragma unroll
    for (int k = 2; k >= 0; --k)
#pragma unroll
        for (int i = 0; i < N; ++i) {
            cs[i][k] = c[i][3 + k];
            if (VALUE) cf[i][k] = c[i][k];
        }
    double f[N], sv[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        sv[i] = cs[i][2].y;
        if (VALUE) f[i] = cf[i][2].y;
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
        sv[i] = fma_(sv[i], t[i], cs[i][2].x);
        if (VALUE) f[i] = fma_(f[i], t[i], cf[i][2].x);
    }
#pragma unroll
    for (int k = 1; k >= 0; --k) {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            sv[i] = fma_(sv[i], t[i], cs[i][k].y);
            if (VALUE) f[i] = fma_(f[i], t[i], cf[i][k].y);
        }
#pragma unroll
        for (int i = 0; i < N; ++i) {
            sv[i] = fma_(sv[i], t[i], cs[i][k].x);
            if (VALUE) f[i] = fma_(f[i], t[i], cf[i][k].x);
        }
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
        if (VALUE) hv[i] = fma_(0.5, __builtin_fabs(eta[i]), f[i]);
        qs[i] = __builtin_copysign(sv[i], eta[i]);          // one v_bfi_b32 on the high word
    }
}


// t <- fma(row-broadcast of lane N of c, x, t): the DPP row broadcast of v_fmac_f64
// (row_newbcast: lane N of the own 16-lane row).  The DPP source must not have been written by
// the VALU in the two preceding wait states; the compiler does not see inside the asm,
// tools/check_dpp_hazards.py checks the build (riemann_amd/build.py runs it on every build).
template <int N>
RMH_DEV void fmac_bcast(double &t, double c, double x)
{
    static_assert(N >= 0 && N < 16, "lane of the 16-lane row");
    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(t) : "v"(c), "v"(x), "n"(N));
}

// ---- two standard normals per 64-bit word: table-driven inverse CDF (rmh-rng v4)
// Each 32-bit half h of a stream word (high half first) gives one normal:
//   sign = h & 1;  q = h | 1;  r = clz32(q) (<= 31);  N = (q << r) | 2^31
//   cell = 8 r + N[30:28];  m = N[27:1]  (t = m 2^-26 - 1 in [-1, 1))
//   z = Horner_fma(cell polynomial of degree 6 in m, coefficients with t's affine map folded in), sign by xor
// i.e. the tail probability p = 2^(-2-r) (1 + (cell + (t+1)/2)/8) the half word encodes, through
// the normal quantile, on a lattice of 2^-27 in p within a cell, out to |z| = 6.4.  The cell
// polynomial reproduces the quantile to 2.3e-12 (tools/gen_icdf_table.py) -- three orders below the
// spacing of the lattice it is evaluated on (~1e-9 in z), so the draws are the lattice's quantiles for
// every statistical purpose; v3 carried degree 10 (2.1e-16), four more fma and two more 16-byte table
// reads per normal that bought nothing the 32-bit input could express (cfg2 +4 %, cfg5 +8 %).
// No division, square root or trigonometry: 7 fma + ~12 integer operations per normal and HALF a
// xoshiro step (v2 spent a whole 64-bit word per normal).  On this chip f64 MFMA and VALU share the
// FP64 units, so every VALU cycle counts.
// The table (rows 0..31 of the generator's 56) is staged in LDS with an 80-byte cell stride: three
// ds_read_b128 and one ds_read_b64 per normal.  20-dword cells put the eight cells of a row on
// disjoint 4-bank windows, and the 160-dword rows shift by 32 banks, so the cells of rows r and r+1
// (3/4 of all draws) are disjoint too.
constexpr int kIcdfRows = 32, kIcdfSub = 8, kIcdfNC = 7, kIcdfStride = 10;
constexpr int kIcdfRowStride = kIcdfSub * kIcdfStride;            // doubles
constexpr int kIcdfDoubles = kIcdfRows * kIcdfRowStride;          // 2560 = 20 480 B

// The 64-bit-resolution normal mode (Sampler(..., normals="f64"); ChainArgs::nmode = 1): ONE normal per
// stream word,
//   sign = x & 1;  q = x | 1;  r = min(clz64(q), 55);  N = (q << r) | 2^63
//   cell = 8 r + N[62:60];  v = 1 + N[59:8] 2^-52;  t = 2 v - 3;  z = Horner_fma(degree 10 in t)
// 52 bits inside a cell, p down to 2^-58 (|z| <= 8.6), the map good to 2.1e-16 -- the resolution of the
// float64 normals numpy hands the reference (randomwalk.py:25).  Its table (56 rows, 11 coefficients in
// a 12-double cell, rows 2 doubles longer than their cells so that rows r and r+1 sit on disjoint banks:
// 43 904 B) takes the place of the 32-bit one in LDS; word s of sub-stream g is coordinate 4s+g.
// Served by the generic kernels (k_run_rw, k_run_hmc, the stand-alone proposals, rmh_rng_draw); the
// launchers route around the kernels whose stream length is a compile-time fact.
constexpr int kIcdf64Rows = 56, kIcdf64NC = 11, kIcdf64Stride = 12;
constexpr int kIcdf64RowStride = kIcdfSub * kIcdf64Stride + 2;    // doubles
constexpr int kIcdf64Doubles = kIcdf64Rows * kIcdf64RowStride;    // 5488 = 43 904 B

constexpr __host__ __device__ int icdf_doubles(int nmode) { return nmode ? kIcdf64Doubles : kIcdfDoubles; }

RMH_DEV void icdf_stage(double *lds_tab, const double *glb_tab, int nmode = 0)
{
    const int n = icdf_doubles(nmode);
    for (int i = threadIdx.x; i < n; i += blockDim.x) lds_tab[i] = glb_tab[i];
}

// N normals of the 64-bit mode in lockstep (see normals_lockstep below)
template <int N>
RMH_DEV void normals64_lockstep(const uint64_t (&w)[N], const double *tab, double (&z)[N])
{
    typedef double d2_t __attribute__((ext_vector_type(2)));
    double t[N];
    const d2_t *c[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const uint64_t q = w[i] | 1ULL;
        int r = __builtin_clzll(q);
        r = r > kIcdf64Rows - 1 ? kIcdf64Rows - 1 : r;
        const uint64_t Nn = (q << r) | 0x8000000000000000ULL;
        const int j = (int)(Nn >> 60) & 7;
        const uint64_t mant = (Nn >> 8) & 0x000FFFFFFFFFFFFFULL;
        const double v = __longlong_as_double((long long)(0x3FF0000000000000ULL | mant));
        t[i] = fma_(2.0, v, -3.0);
        c[i] = static_cast<const d2_t *>(__builtin_assume_aligned(tab + r * kIcdf64RowStride + j * kIcdf64Stride, 16));
    }
    d2_t a[N][6];
#pragma unroll
    for (int k = 5; k >= 0; --k)
#pragma unroll
        for (int i = 0; i < N; ++i) a[i][k] = c[i][k];
#pragma unroll
    for (int i = 0; i < N; ++i) z[i] = a[i][5].x;
#pragma unroll
    for (int k = 4; k >= 0; --k) {
#pragma unroll
        for (int i = 0; i < N; ++i) z[i] = fma_(z[i], t[i], a[i][k].y);
#pragma unroll
        for (int i = 0; i < N; ++i) z[i] = fma_(z[i], t[i], a[i][k].x);
    }
#pragma unroll
    for (int i = 0; i < N; ++i)
        z[i] = __longlong_as_double(__double_as_longlong(z[i]) ^ ((long long)(w[i] & 1ULL) << 63));
}

// N normals in lockstep (the halves of one or two stream words): all cells' table reads first,
// then the N Horner chains alternating.  Evaluated one after the other (what the compiler makes
// of separate normal_from_half calls) every normal exposes its LDS latency and six dependent
// fma latencies.  Same operations per normal, in the same order: the same bits.
template <int N>
RMH_DEV void normals_lockstep(const uint32_t (&h)[N], const double *tab, double (&z)[N])
{
    typedef double d2_t __attribute__((ext_vector_type(2)));
    double t[N];
    const d2_t *c[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const uint32_t q = h[i] | 1u;
        const int r = __builtin_clz(q);
        const uint32_t Nn = q << r;
        const int j = (int)(Nn >> 28) & 7;
        // the cells' polynomials are in the 27-bit integer m27 itself (t = m27 2^-26 - 1 folded into the
        // coefficients by the generator): no fma to form the argument
        t[i] = (double)((Nn >> 1) & 0x7FFFFFFu);
        c[i] = static_cast<const d2_t *>(__builtin_assume_aligned(tab + r * kIcdfRowStride + j * kIcdfStride, 16));
    }
    // (the leading coefficient as ONE double: read as half of a 16-byte pair its unused upper
    // registers get reused at once, and the next read then has to wait for this one)
#pragma unroll
    for (int i = 0; i < N; ++i) z[i] = reinterpret_cast<const double *>(c[i])[kIcdfNC - 1];
    constexpr int NPAIR = (kIcdfNC - 1) / 2;
    d2_t a[N][NPAIR];
#pragma unroll
    for (int k = NPAIR - 1; k >= 0; --k)
#pragma unroll
        for (int i = 0; i < N; ++i) a[i][k] = c[i][k];
#pragma unroll
    for (int k = NPAIR - 1; k >= 0; --k) {
#pragma unroll
        for (int i = 0; i < N; ++i) z[i] = fma_(z[i], t[i], a[i][k].y);
#pragma unroll
        for (int i = 0; i < N; ++i) z[i] = fma_(z[i], t[i], a[i][k].x);
    }
#pragma unroll
    for (int i = 0; i < N; ++i)
        z[i] = __longlong_as_double(__double_as_longlong(z[i]) ^ ((long long)(h[i] & 1u) << 63));
}

RMH_DEV double normal_from_half(uint32_t h, const double *tab)
{
    const uint32_t hh[1] = {h};
    double z[1];
    normals_lockstep<1>(hh, tab, z);
    return z[0];
}

constexpr double k2m53 = 0x1p-53;

RMH_DEV double u01(uint64_t r) { return (double)(r >> 11) * k2m53; }

// A chain's four lanes {cl, cl+16, cl+32, cl+48} sit in the four 16-lane rows of the wavefront.
// Exchanges between rows use the gfx950 row swaps v_permlane16_swap / v_permlane32_swap: plain
// VALU instructions, where __shfl (ds_bpermute) makes an LDS round trip whose latency the two
// waves of a SIMD cannot cover (cfg2: 38 bpermutes per iteration, 35.2 -> see DESIGN.md section 5).
struct RowPair {
    double a, b;
};
// rows of a = {x0, y0, x2, y2}, rows of b = {x1, y1, x3, y3}   (x_r = row r of x)
RMH_DEV RowPair swap16(double x, double y)
{
    const uint64_t ux = (uint64_t)__double_as_longlong(x), uy = (uint64_t)__double_as_longlong(y);
    const auto lo = __builtin_amdgcn_permlane16_swap((uint32_t)ux, (uint32_t)uy, false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap((uint32_t)(ux >> 32), (uint32_t)(uy >> 32), false, false);
    return {__longlong_as_double((long long)((uint64_t)lo[0] | ((uint64_t)hi[0] << 32))),
            __longlong_as_double((long long)((uint64_t)lo[1] | ((uint64_t)hi[1] << 32)))};
}
// rows of a = {x0, x1, y0, y1}, rows of b = {x2, x3, y2, y3}
RMH_DEV RowPair swap32(double x, double y)
{
    const uint64_t ux = (uint64_t)__double_as_longlong(x), uy = (uint64_t)__double_as_longlong(y);
    const auto lo = __builtin_amdgcn_permlane32_swap((uint32_t)ux, (uint32_t)uy, false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap((uint32_t)(ux >> 32), (uint32_t)(uy >> 32), false, false);
    return {__longlong_as_double((long long)((uint64_t)lo[0] | ((uint64_t)hi[0] << 32))),
            __longlong_as_double((long long)((uint64_t)lo[1] | ((uint64_t)hi[1] << 32)))};
}

// sum over the chain's four lanes: (x0 + x1) + (x2 + x3), the same bits on all four
RMH_DEV double sum4(double x)
{
    const RowPair p = swap16(x, x);
    const double s = p.a + p.b;
    const RowPair q = swap32(s, s);
    return q.a + q.b;
}

// row g receives the sum of t_g over the chain's four lanes (a reduce-scatter of four values:
// three swaps instead of four sum4); same bits as sum4(t_g)
RMH_DEV double sum4_scatter(double t0, double t1, double t2, double t3)
{
    const RowPair p = swap16(t0, t1), q = swap16(t2, t3);
    const RowPair r = swap32(p.a + p.b, q.a + q.b);
    return r.a + r.b;
}

// all-gather over the chain's four lanes: o[r] = the value row r holds, on all four lanes
RMH_DEV void gather4(double x, double (&o)[4])
{
    const RowPair p = swap16(x, x);            // a = {x0, x0, x2, x2}, b = {x1, x1, x3, x3}
    const RowPair e = swap32(p.a, p.a), f = swap32(p.b, p.b);
    o[0] = e.a;
    o[1] = f.a;
    o[2] = e.b;
    o[3] = f.b;
}

// the value of row 0 (lane cl) on all four lanes of the chain
RMH_DEV double bcast_row0(double x)
{
    const RowPair p = swap16(x, x);
    return swap32(p.a, p.a).a;
}

// Model.log_posterior, riemann/models/model.py:49-55
RMH_DEV double combine_post(double logP, double logL)
{
    const double inf = __builtin_huge_val();
    const bool bad = (logP != logP) || (logL != logL) || logP == inf || logP == -inf ||
                     logL == inf || logL == -inf;
    return bad ? -inf : logP + logL;
}

// Sampler.sample, riemann/samplers/sampler.py:83-84, with PYTHON min(0, D):
// D if D < 0 else 0, so a NaN ratio accepts.
template <bool OPQ = false>
RMH_DEV bool mh_accept(double lp_prop, double lp_old, double logq, double u)
{
    const double delta = lp_prop - lp_old - logq;
    const double mh = (delta < 0.0) ? delta : 0.0;
    const double lu = (u > 0.0) ? log_pos<OPQ>(u) : -__builtin_huge_val();
    return lu < mh;
}

} // namespace rmh
