/*
 * rmh_oracle.c -- CPU ORACLE (test infrastructure, NOT the product).
 *
 * A plain-C restatement of the rscalzo/riemann Metropolis-Hastings hot path
 *   Sampler.run() -> Sampler.sample() -> Proposal.propose() -> Model.log_posterior()
 * extended from one chain to N independent chains.  Only tests/, the smoke
 * check in __graft_entry__.py and bench.py's `cpu_baseline` leg may load this
 * library; nothing under riemann_amd/ links, imports or calls it.
 *
 * Parity status: PINNED.  the .npz files under tests/golden/ were produced by importing the
 * reference itself (tests/golden/make_golden.py) with this file's random
 * stream injected through np.random.normal / np.random.uniform;
 * tests/test_oracle_golden.py checks this restatement against those files.
 * The logistic-regression Model has no reference counterpart (SURVEY.md
 * section 8a-9): its arithmetic is pinned by the spec in DESIGN.md and a
 * long-double cross-check only; the MALA/HMC/accept logic around it is pinned
 * by driving it through the reference's own Sampler + VanillaHMC.
 *
 * Every function cites the reference lines it restates (paths relative to
 * /root/reference).  Floating-point contraction must be OFF when compiling
 * (-ffp-contract=off): the random-number spec below is defined operation by
 * operation (explicit fma where the spec says fma) so that the HIP kernels
 * reproduce the same doubles bit for bit.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------- */
/* 1. "rmh-rng v4": 4 xoshiro256++ sub-streams per chain                       */
/* ------------------------------------------------------------------------- */
/*
 * The reference draws from numpy's process-global MT19937:
 *   d normals   proposals/randomwalk.py:25, proposals/hamiltonian.py:79
 *   1 uniform   samplers/sampler.py:84
 * in that order, every iteration.  A global generator cannot be shared by 10^5
 * chains, so the build defines its own stream (SURVEY.md 8b "Threading", 8c
 * "RNG parity mechanism") and injects it into the reference when the golden
 * vectors are made:
 *
 *   chain c (GLOBAL chain id) owns sub-streams g = 0..3, stream id 4c+g;
 *   every iteration each sub-stream draws M = ceil(d/8) 64-bit words, two
 *   standard normals per word (table-driven inverse CDF of each 32-bit half,
 *   below); the high half of word m of sub-stream g is coordinate j = 8m+g, the
 *   low half j = 8m+4+g (dropped when j >= d); afterwards sub-stream 0 draws one
 *   more word for the accept uniform.
 *
 * Four sub-streams because the HIP kernels put a chain on the four lanes
 * {c, c+16, c+32, c+48} of a wavefront (the B-operand layout of
 * v_mfma_f64_16x16x4_f64), each lane owning the coordinates j = g (mod 4).
 */
typedef uint64_t u64;

#define ORC_GAMMA 0x9E3779B97F4A7C15ULL

static inline u64 orc_mix64(u64 z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

static inline u64 orc_rotl(u64 x, int k) { return (x << k) | (x >> (64 - k)); }

/* state words of stream `sid` under `seed` */
static void orc_stream_init(u64 seed, u64 sid, u64 s[4])
{
    u64 x = orc_mix64(orc_mix64(seed + ORC_GAMMA) + sid);
    for (int i = 0; i < 4; i++) {
        x += ORC_GAMMA;
        s[i] = orc_mix64(x);
    }
}

/* xoshiro256++ (Blackman & Vigna 2019) */
static inline u64 orc_next(u64 s[4])
{
    const u64 r = orc_rotl(s[0] + s[3], 23) + s[0];
    const u64 t = s[1] << 17;
    s[2] ^= s[0];
    s[3] ^= s[1];
    s[1] ^= s[2];
    s[0] ^= s[3];
    s[2] ^= t;
    s[3] = orc_rotl(s[3], 45);
    return r;
}

static inline double orc_from_bits(u64 b) { double d; memcpy(&d, &b, 8); return d; }
static inline u64 orc_to_bits(double d) { u64 b; memcpy(&b, &d, 8); return b; }

/* natural log, defined operation by operation (fdlibm-style reduction,
 * k*ln2 + log1p(f) with s = f/(2+f)); < 1 ulp.  Specials follow IEEE/numpy:
 * log(<0)=nan, log(0)=-inf, log(inf)=inf, log(nan)=nan. */
double orc_log(double x)
{
    static const double
        ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
        Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01,
        Lg3 = 2.857142874366239149e-01, Lg4 = 2.222219843214978396e-01,
        Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
        Lg7 = 1.479819860511658591e-01;
    if (x != x) return x;
    if (x < 0.0) return NAN;
    if (x == 0.0) return -INFINITY;
    if (x == INFINITY) return x;
    int k = 0;
    u64 ix = orc_to_bits(x);
    if (ix < 0x0010000000000000ULL) { /* subnormal */
        x *= 0x1p54;
        k = -54;
        ix = orc_to_bits(x);
    }
    k += (int)(ix >> 52) - 1023;
    u64 mant = ix & 0x000FFFFFFFFFFFFFULL;
    double m;
    if (mant >= 0x0006A09E667F3BCDULL) { /* m in [sqrt(2)/2, 1) */
        m = orc_from_bits(mant | 0x3FE0000000000000ULL);
        k += 1;
    } else {                             /* m in [1, sqrt(2)) */
        m = orc_from_bits(mant | 0x3FF0000000000000ULL);
    }
    const double f = m - 1.0;
    const double s = f / (2.0 + f);
    const double z = s * s;
    const double w = z * z;
    const double t1 = w * fma(w, fma(w, Lg6, Lg4), Lg2);
    const double t2 = z * fma(w, fma(w, fma(w, Lg7, Lg5), Lg3), Lg1);
    const double R = t2 + t1;
    const double hfsq = 0.5 * f * f;
    const double dk = (double)k;
    return dk * ln2_hi - ((hfsq - fma(s, hfsq + R, dk * ln2_lo)) - f);
}

/* ---- two standard normals per 64-bit word: table-driven inverse CDF ("rmh-rng v4") ----
 * (tools/gen_icdf_table.py documents the construction of the table; this file's own copy is
 * oracle/rmh_icdf_table.h; rows 0..31 of it are used)
 * Each 32-bit half h of a stream word (high half first) gives one normal:
 *   sign = h & 1;  q = h | 1;  r = min(clz32(q), 31) = clz32(q);  N = (q << r) | 2^31
 *   cell = 8 r + N[30:28];  m = N[27:1]  (t = m 2^-26 - 1 in [-1, 1))
 *   z = Horner_fma(cell polynomial in m, t's affine map folded into the coefficients), negated when sign is set
 * i.e. the tail probability p = 2^(-2-r) (1 + (cell + (t+1)/2)/8) read off the half word,
 * mapped through the normal quantile by the cell's degree-6 polynomial (2.3e-12 as a map: three
 * orders below the spacing of the lattice it is evaluated on), on a lattice of 2^-27 in p
 * within a cell (coarser in the last rows, which have few bits left) and out to |z| = 6.4 (p >= 2^-33) -- what a 32-bit generator offers.
 * (v2 spent a whole 64-bit word per normal, v3 a degree-10 polynomial per normal; on MI355X the
 * stream is the largest VALU cost of the random-walk kernels and f64 MFMA shares the VALU's FP64
 * units: DESIGN.md section 3.)   */
#include "rmh_icdf_table.h"
#define ORC_ICDF_RMAX 31

double orc_normal_from_half(uint32_t h)
{
    const uint32_t q = h | 1u;
    int r = __builtin_clz(q);
    if (r > ORC_ICDF_RMAX) r = ORC_ICDF_RMAX;
    const uint32_t N = (q << r) | 0x80000000u;
    const int j = (int)(N >> 28) & 7;
    const u64 m27 = (N >> 1) & 0x7FFFFFFu;
    const double t = (double)m27; /* the table's polynomials are in m27 itself (t = m27 2^-26 - 1 folded in) */
    const double *c = rmh_icdf_table + (size_t)(r * RMH_ICDF_SUB + j) * RMH_ICDF_NC;
    double z = c[RMH_ICDF_NC - 1];
    for (int k = RMH_ICDF_NC - 2; k >= 0; k--) z = fma(z, t, c[k]);
    return (h & 1u) ? -z : z;
}

#define ORC_2M53 0x1p-53

/* ---- the 64-bit-resolution normal mode (Sampler(..., normals="f64")): ONE normal per word ----
 *   sign = x & 1;  q = x | 1;  r = min(clz64(q), 55);  N = (q << r) | 2^63
 *   cell = 8 r + N[62:60];  v = 1 + N[59:8] 2^-52;  t = 2 v - 3 in [-1, 1)
 *   z = Horner_fma(cell polynomial of degree 10 in t), negated when sign is set
 * the tail probability p = 2^(-2-r) (1 + (cell + (t+1)/2)/8) on a lattice of 2^-57-r.. in p (52 bits inside a
 * cell while the word has them), out to p = 2^-58 (|z| = 8.6), through a map good to 2.1e-16
 * (oracle/rmh_icdf64_table.h, tools/gen_icdf_table.py --f64): what numpy's float64 normals offer the
 * reference (proposals/randomwalk.py:25, hamiltonian.py:79).  Word s of sub-stream g is coordinate 4s+g;
 * every sub-stream draws ceil(d/4) words per vector of normals. */
#include "rmh_icdf64_table.h"
static int g_normal_mode = 0; /* 0: two normals per word (v4, above); 1: one normal per word */

void orc_set_normal_mode(int mode) { g_normal_mode = mode ? 1 : 0; }
int orc_get_normal_mode(void) { return g_normal_mode; }

double orc_normal_from_word(u64 x)
{
    const u64 q = x | 1ULL;
    int r = __builtin_clzll(q);
    if (r > RMH_ICDF64_ROWS - 1) r = RMH_ICDF64_ROWS - 1;
    const u64 N = (q << r) | 0x8000000000000000ULL;
    const int j = (int)(N >> 60) & 7;
    const u64 mant = (N >> 8) & 0x000FFFFFFFFFFFFFULL;
    const double v = 1.0 + (double)mant * 0x1p-52; /* exact: mant < 2^52 */
    const double t = fma(2.0, v, -3.0);
    const double *c = rmh_icdf64_table + (size_t)(r * RMH_ICDF64_SUB + j) * RMH_ICDF64_NC;
    double z = c[RMH_ICDF64_NC - 1];
    for (int k = RMH_ICDF64_NC - 2; k >= 0; k--) z = fma(z, t, c[k]);
    return (x & 1ULL) ? -z : z;
}

/* d normals for one chain; st = 16 words [g][4].  Every sub-stream draws M = ceil(d/8)
 * words; the high half of word m of sub-stream g is coordinate 8m+g, the low half 8m+4+g.
 * (64-bit mode: ceil(d/4) words, word s of sub-stream g is coordinate 4s+g) */
void orc_rng_normals(u64 *st, int d, double *xi)
{
    if (g_normal_mode) {
        const int ks = (d + 3) / 4;
        for (int g = 0; g < 4; g++)
            for (int s = 0; s < ks; s++) {
                const double z = orc_normal_from_word(orc_next(st + 4 * g));
                if (4 * s + g < d) xi[4 * s + g] = z;
            }
        return;
    }
    const int M = (d + 7) / 8;
    for (int g = 0; g < 4; g++)
        for (int m = 0; m < M; m++) {
            const u64 w = orc_next(st + 4 * g);
            const double zh = orc_normal_from_half((uint32_t)(w >> 32));
            const double zl = orc_normal_from_half((uint32_t)w);
            if (8 * m + g < d) xi[8 * m + g] = zh;
            if (8 * m + 4 + g < d) xi[8 * m + 4 + g] = zl;
        }
}

/* U[0,1) from sub-stream 0 */
double orc_rng_uniform(u64 *st) { return (double)(orc_next(st) >> 11) * ORC_2M53; }

/* one word from each sub-stream of one chain, as U[0,1) */
void orc_rng_words(u64 *st, double *u)
{
    for (int g = 0; g < 4; g++) u[g] = (double)(orc_next(st + 4 * g) >> 11) * ORC_2M53;
}

void orc_rng_seed(u64 seed, u64 chain0, int64_t N, u64 *state)
{
    for (int64_t c = 0; c < N; c++)
        for (int g = 0; g < 4; g++)
            orc_stream_init(seed, 4 * (chain0 + (u64)c) + (u64)g, state + 16 * c + 4 * g);
}

/* the per-iteration stream as the reference consumes it: xi (T,N,d), u (T,N) */
void orc_rng_stream(u64 *state, int64_t N, int d, int T, double *xi, double *u)
{
    for (int t = 0; t < T; t++)
        for (int64_t c = 0; c < N; c++) {
            orc_rng_normals(state + 16 * c, d, xi + ((int64_t)t * N + c) * d);
            u[(int64_t)t * N + c] = orc_rng_uniform(state + 16 * c);
        }
}

/* the PT sweep's draws for S slots (section 5): usel, vswp, ua (T,S), xi (T,S,d) */
void orc_rng_pt_stream(u64 *state, int64_t S, int d, int T, double *usel, double *vswp,
                       double *xi, double *ua)
{
    for (int t = 0; t < T; t++)
        for (int64_t c = 0; c < S; c++) {
            u64 *st = state + 16 * c;
            usel[(int64_t)t * S + c] = (double)(orc_next(st) >> 11) * ORC_2M53;
            vswp[(int64_t)t * S + c] = (double)(orc_next(st + 4) >> 11) * ORC_2M53;
            (void)orc_next(st + 8);
            (void)orc_next(st + 12);
            orc_rng_normals(st, d, xi + ((int64_t)t * S + c) * d);
            ua[(int64_t)t * S + c] = orc_rng_uniform(st);
        }
}

/* ------------------------------------------------------------------------- */
/* 2. Models                                                                   */
/* ------------------------------------------------------------------------- */
enum { ORC_GAUSS = 1, ORC_LOGREG = 2, ORC_CHANGEPOINT = 3, ORC_MIXTURE = 4 };

typedef struct orc_model {
    int kind, d;
    /* gaussian: models/gaussian.py:27-44 */
    double *mu, *L, logdetC;
    /* logreg (new plugin) */
    int64_t n;
    double *X, *y, tau2;
    /* changepoint: models/changepoint.py:86-104 */
    int k; /* number of changepoints; pieces = k+1 */
    double *cx, *cy, xmin, xmax, lamb, alpha, beta;
    /* mixture of Gaussians: models/benchmarks.py:31-60 */
    int ncomp;
    double *wts, logP0;
    struct orc_model **comps;
} orc_model;

static double *orc_dup(const double *p, size_t n)
{
    double *q = (double *)malloc(n * sizeof(double));
    memcpy(q, p, n * sizeof(double));
    return q;
}

/* models/gaussian.py:27-44: caller passes L = cholesky(C) (lower, row-major) */
orc_model *orc_model_gaussian(const double *mu, const double *L, int d)
{
    orc_model *m = (orc_model *)calloc(1, sizeof(*m));
    m->kind = ORC_GAUSS;
    m->d = d;
    m->mu = orc_dup(mu, d);
    m->L = orc_dup(L, (size_t)d * d);
    double s = 0;
    for (int i = 0; i < d; i++) s += log(L[i * d + i]);
    m->logdetC = 2 * s; /* gaussian.py:43 */
    return m;
}

orc_model *orc_model_logreg(const double *X, const double *y, int64_t n, int d, double tau2)
{
    orc_model *m = (orc_model *)calloc(1, sizeof(*m));
    m->kind = ORC_LOGREG;
    m->d = d;
    m->n = n;
    m->X = orc_dup(X, (size_t)n * d);
    m->y = orc_dup(y, n);
    m->tau2 = tau2;
    return m;
}

orc_model *orc_model_changepoint(const double *x, const double *y, int n, int k,
                                 double xmin, double xmax, double lamb,
                                 double alpha, double beta)
{
    orc_model *m = (orc_model *)calloc(1, sizeof(*m));
    m->kind = ORC_CHANGEPOINT;
    m->k = k;
    m->d = 2 * k + 2;
    m->n = n;
    m->cx = orc_dup(x, n);
    m->cy = orc_dup(y, n);
    m->xmin = xmin; m->xmax = xmax; m->lamb = lamb; m->alpha = alpha; m->beta = beta;
    return m;
}

/* benchmarks.MixtureDist(components) with MultiGaussianDist components,
 * models/benchmarks.py:31-42.  `wts` are the NORMALISED weights (:40) and logP0 the
 * constant log_prior, log(sum(wts * exp(0))) (:54-62), both formed by the caller
 * with numpy exactly as the reference does. */
orc_model *orc_model_mixture(const double *wts, const double *mu, const double *L, int ncomp,
                             int d, double logP0)
{
    orc_model *m = (orc_model *)calloc(1, sizeof(*m));
    m->kind = ORC_MIXTURE;
    m->d = d;
    m->ncomp = ncomp;
    m->wts = orc_dup(wts, ncomp);
    m->logP0 = logP0;
    m->comps = (orc_model **)calloc(ncomp, sizeof(orc_model *));
    for (int k = 0; k < ncomp; k++)
        m->comps[k] = orc_model_gaussian(mu + (size_t)k * d, L + (size_t)k * d * d, d);
    return m;
}

void orc_model_free(orc_model *m)
{
    if (!m) return;
    for (int k = 0; k < m->ncomp; k++) orc_model_free(m->comps[k]);
    free(m->comps); free(m->wts);
    free(m->mu); free(m->L); free(m->X); free(m->y); free(m->cx); free(m->cy);
    free(m);
}

int orc_model_dim(const orc_model *m) { return m->d; }

/* u = solve(L, theta-mu): models/gaussian.py:49-51 (np.linalg.solve on the
 * lower-triangular factor; restated as forward substitution) */
static void orc_gauss_u(const orc_model *m, const double *th, double *u)
{
    const int d = m->d;
    for (int i = 0; i < d; i++) {
        double acc = th[i] - m->mu[i];
        for (int j = 0; j < i; j++) acc -= m->L[i * d + j] * u[j];
        u[i] = acc / m->L[i * d + i];
    }
}

/* numpy searchsorted(a, v) side='left' (changepoint.py:181) */
static inline int orc_searchsorted(const double *a, int n, double v)
{
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = lo + ((hi - lo) >> 1);
        if (a[mid] < v) lo = mid + 1; else hi = mid;
    }
    return lo;
}

/* log_prior of one chain */
static double orc_log_prior1(const orc_model *m, const double *th)
{
    switch (m->kind) {
    case ORC_GAUSS:
        return 0.0; /* gaussian.py:46-47 */
    case ORC_MIXTURE:
        return m->logP0; /* benchmarks.py:54-62: the components' log_prior is 0 */
    case ORC_LOGREG: {
        double s = 0;
        for (int j = 0; j < m->d; j++) s += th[j] * th[j];
        return -s / (2.0 * m->tau2);
    }
    case ORC_CHANGEPOINT: { /* changepoint.py:128-160 */
        const int kc = m->k, kp = kc + 1; /* k = len(cpv) = number of pieces */
        const double *cpx = th, *cpv = th + kc, sig = th[2 * kc + 1];
        const double logP_k = kp * log(m->lamb) - lgamma((double)kp) - m->lamb;
        double logP_v = 0; /* log_gamma_dist, changepoint.py:18-19 */
        for (int j = 0; j < kp; j++)
            logP_v += m->alpha * log(m->beta) + (m->alpha - 1) * log(cpv[j])
                      - m->beta * cpv[j] - lgamma(m->alpha);
        const double Lx = m->xmax - m->xmin;
        double sumlog = 0, prev = m->xmin;
        for (int j = 0; j <= kc; j++) {
            const double cur = (j < kc) ? cpx[j] : m->xmax;
            sumlog += log(cur - prev);
            prev = cur;
        }
        const double logP_s = lgamma(2.0 * kp + 1) + sumlog - kp * log(Lx);
        double logP_sig2 = log(1.0 / (sig * sig));
        if (sig < 0) logP_sig2 = NAN;
        double logP = logP_k + logP_v + logP_s + logP_sig2;
        if (logP != logP) logP = -INFINITY;
        return logP;
    }
    }
    return NAN;
}

static inline double orc_log1pexp(double e) /* log(1+exp(e)) */
{
    return e > 0 ? e + log1p(exp(-e)) : log1p(exp(e));
}

static double orc_log_lik1(const orc_model *m, const double *th, double *scratch)
{
    switch (m->kind) {
    case ORC_MIXTURE: { /* benchmarks.py:44-52: log(sum(w * exp(logL_k))), no log-sum-exp */
        double acc = 0;
        for (int k = 0; k < m->ncomp; k++)
            acc += m->wts[k] * exp(orc_log_lik1(m->comps[k], th, scratch));
        return log(acc);
    }
    case ORC_GAUSS: { /* gaussian.py:49-52 */
        const int d = m->d;
        orc_gauss_u(m, th, scratch);
        double uu = 0;
        for (int i = 0; i < d; i++) uu += scratch[i] * scratch[i];
        return -0.5 * (uu + d * log(2 * M_PI) + m->logdetC);
    }
    case ORC_LOGREG: {
        double ll = 0;
        for (int64_t i = 0; i < m->n; i++) {
            const double *xr = m->X + i * m->d;
            double eta = 0;
            for (int j = 0; j < m->d; j++) eta += xr[j] * th[j];
            ll += m->y[i] * eta - orc_log1pexp(eta);
        }
        return ll;
    }
    case ORC_CHANGEPOINT: { /* changepoint.py:106-126, predict :175-181 */
        const int kc = m->k, n = (int)m->n;
        const double *cpx = th, *cpv = th + kc, sig = th[2 * kc + 1];
        const double sig2 = sig * sig;
        double ss = 0;
        for (int i = 0; i < n; i++) {
            const double r = cpv[orc_searchsorted(cpx, kc, m->cx[i])] - m->cy[i];
            ss += (r * r) / sig2;
        }
        double logL = -0.5 * (ss + n * log(sig2) + n * log(2 * M_PI));
        if (logL != logL) logL = -INFINITY;
        return logL;
    }
    }
    return NAN;
}

/* Model.log_posterior, models/model.py:43-55 */
static double orc_log_post1(const orc_model *m, const double *th, double *scratch)
{
    const double lp = orc_log_prior1(m, th), ll = orc_log_lik1(m, th, scratch);
    if (isinf(lp) || isnan(lp) || isinf(ll) || isnan(ll)) return -INFINITY;
    return lp + ll;
}

/* gradient of log posterior of one chain.
 * gaussian: grad_log_likelihood, gaussian.py:54-58 (log_prior = 0). */
static void orc_grad1(const orc_model *m, const double *th, double *g, double *scratch)
{
    const int d = m->d;
    switch (m->kind) {
    case ORC_GAUSS: {
        double *u = scratch;
        orc_gauss_u(m, th, u);
        for (int i = d - 1; i >= 0; i--) { /* v = solve(L.T, u) */
            double acc = u[i];
            for (int j = i + 1; j < d; j++) acc -= m->L[j * d + i] * g[j];
            g[i] = acc / m->L[i * d + i];
        }
        for (int i = 0; i < d; i++) g[i] = -g[i];
        break;
    }
    case ORC_LOGREG: {
        for (int j = 0; j < d; j++) g[j] = 0;
        for (int64_t i = 0; i < m->n; i++) {
            const double *xr = m->X + i * d;
            double eta = 0;
            for (int j = 0; j < d; j++) eta += xr[j] * th[j];
            const double sg = eta >= 0 ? 1.0 / (1.0 + exp(-eta)) : exp(eta) / (1.0 + exp(eta));
            const double r = m->y[i] - sg;
            for (int j = 0; j < d; j++) g[j] += xr[j] * r;
        }
        for (int j = 0; j < d; j++) g[j] -= th[j] / m->tau2;
        break;
    }
    case ORC_MIXTURE: { /* d/dtheta log sum_k w_k N_k = sum_k w_k N_k grad(log N_k) / sum_k w_k N_k;
                         * NOT in the reference (benchmarks.MixtureDist has no gradient): the build's
                         * plugin addition, like the logistic model */
        double den = 0;
        double *gk = (double *)malloc(sizeof(double) * 2 * d), *sk = gk + d;
        for (int j = 0; j < d; j++) g[j] = 0;
        for (int k = 0; k < m->ncomp; k++) {
            const double e = m->wts[k] * exp(orc_log_lik1(m->comps[k], th, sk));
            orc_grad1(m->comps[k], th, gk, sk);
            for (int j = 0; j < d; j++) g[j] += e * gk[j];
            den += e;
        }
        for (int j = 0; j < d; j++) g[j] /= den;
        free(gk);
        break;
    }
    default:
        for (int j = 0; j < d; j++) g[j] = NAN; /* no gradient (changepoint.py:108-113) */
    }
}

void orc_log_prior(const orc_model *m, const double *theta, int64_t N, double *out)
{
    for (int64_t c = 0; c < N; c++) out[c] = orc_log_prior1(m, theta + c * m->d);
}

void orc_log_likelihood(const orc_model *m, const double *theta, int64_t N, double *out)
{
    double *scratch = (double *)malloc(sizeof(double) * m->d);
    for (int64_t c = 0; c < N; c++) out[c] = orc_log_lik1(m, theta + c * m->d, scratch);
    free(scratch);
}

void orc_log_posterior(const orc_model *m, const double *theta, int64_t N, double *out)
{
    double *scratch = (double *)malloc(sizeof(double) * m->d);
    for (int64_t c = 0; c < N; c++) out[c] = orc_log_post1(m, theta + c * m->d, scratch);
    free(scratch);
}

void orc_grad_log_posterior(const orc_model *m, const double *theta, int64_t N, double *out)
{
    double *scratch = (double *)malloc(sizeof(double) * m->d);
    for (int64_t c = 0; c < N; c++) orc_grad1(m, theta + c * m->d, out + c * m->d, scratch);
    free(scratch);
}

/* integer path of changepoint.predict (changepoint.py:181): piece index per datum */
void orc_changepoint_indices(const orc_model *m, const double *theta, int64_t N, int32_t *idx)
{
    for (int64_t c = 0; c < N; c++)
        for (int i = 0; i < (int)m->n; i++)
            idx[c * m->n + i] = orc_searchsorted(theta + c * m->d, m->k, m->cx[i]);
}

/* ------------------------------------------------------------------------- */
/* 3. Proposals                                                                */
/* ------------------------------------------------------------------------- */
enum { ORC_RW = 1, ORC_HMC = 2, ORC_PCN = 3 };

typedef struct orc_proposal {
    int kind, d;
    double *L, scale; /* MetropolisRandomWalk: randomwalk.py:17-19 */
    double eps;       /* VanillaHMC: hamiltonian.py:60-74 (M=None only) */
    int nsteps;
    double adapt_target; /* > 0: AdaptScaleProposal mix-in, proposals/adaptive.py:11-35 */
    double *chM;         /* cholesky(M) when a mass matrix is given, hamiltonian.py:71-74 */
    double minv;         /* M^-1 = minv (chM chM^T)^-1; 0 reads as 1.  AdaptCovHMC sets M = C and
                          * chM = L with C = d^0.4 L L^T (hamiltonian.py:117-119 with adaptive.py:101-102) */
    double rho, rho_c;   /* pCN, randomwalk.py:83-87 */
    /* block-mixture random walk, ChangepointRegression1DProp (riemann/test_changepoint.py:18-56) */
    int nblk, blk_end[3];
    double blk_p[3];
} orc_proposal;

orc_proposal *orc_proposal_rw(const double *L, int d, double scale)
{
    orc_proposal *p = (orc_proposal *)calloc(1, sizeof(*p));
    p->kind = ORC_RW; p->d = d; p->L = orc_dup(L, (size_t)d * d); p->scale = scale;
    return p;
}

/* ChangepointRegression1DProp.propose at fixed k (riemann/test_changepoint.py:43-56): the
 * coordinates are cut into nblk <= 3 consecutive blocks (cpx, cpv, sig) of the given sizes,
 * each with its own MetropolisRandomWalk (:36-41; L must be block diagonal); CHAINED
 * independent uniforms pick the block: block b iff u_b < pcum[b] and no earlier block was
 * picked (:47-55).  If none is picked the reference makes a trans-dimensional move (:56-71),
 * which does not exist at fixed k: the proposal is then theta itself.
 * Stream: after the iteration's normals every sub-stream draws one more word; sub-stream
 * 0's is the accept uniform as always, those of sub-streams 1, 2, 3 are u_0, u_1, u_2.  The
 * picked block uses the normals of its own coordinates. */
void orc_proposal_set_blocks(orc_proposal *p, int nblk, const int *sizes, const double *pcum)
{
    p->nblk = nblk;
    int end = 0;
    for (int b = 0; b < nblk; b++) {
        end += sizes[b];
        p->blk_end[b] = end;
        p->blk_p[b] = pcum[b];
    }
}

/* pCN(C, rho): proposals/randomwalk.py:78-101.  L = cholesky(C). */
orc_proposal *orc_proposal_pcn(const double *L, int d, double rho)
{
    orc_proposal *p = (orc_proposal *)calloc(1, sizeof(*p));
    p->kind = ORC_PCN; p->d = d; p->L = orc_dup(L, (size_t)d * d); p->scale = 1.0;
    p->rho = rho; p->rho_c = sqrt(1 - rho * rho);
    return p;
}

orc_proposal *orc_proposal_hmc(double eps, int nsteps, int d)
{
    orc_proposal *p = (orc_proposal *)calloc(1, sizeof(*p));
    p->kind = ORC_HMC; p->d = d; p->eps = eps; p->nsteps = nsteps;
    return p;
}

/* VanillaHMC(..., M): chM = cholesky(M), lower, (d,d) row-major (hamiltonian.py:74) */
void orc_proposal_set_mass(orc_proposal *p, const double *chM)
{
    free(p->chM);
    p->chM = chM ? orc_dup(chM, (size_t)p->d * p->d) : NULL;
}

/* AdaptScaleRandomWalk (randomwalk.py:29-38, target 0.25) / AdaptScaleHMC
 * (hamiltonian.py:94-104, target 0.75): enable the scale adaptation */
void orc_proposal_set_adapt(orc_proposal *p, double target) { p->adapt_target = target; }

void orc_proposal_free(orc_proposal *p)
{
    if (!p) return;
    free(p->L);
    free(p->chM);
    free(p);
}

/* Proposal.propose for one chain given its normals xi: returns logqratio.
 * RW : randomwalk.py:21-26.   HMC: hamiltonian.py:76-91 + leapfrog :13-52,
 * gradient callable = the model's grad of the log posterior. */
static double orc_propose1(const orc_proposal *p, const orc_model *m, const double *th,
                           const double *xi, double *thp, double *work, double chain_scale)
{
    const int d = p->d;
    /* non-adaptive: the proposal's own .scale (randomwalk.py:18) / eps;
     * adaptive: the chain's adapted scale, and eps = scale * eps0 (hamiltonian.py:101-102) */
    const double scale = p->adapt_target > 0 ? chain_scale : p->scale;
    if (p->kind == ORC_RW) {
        for (int i = 0; i < d; i++) {
            double acc = 0; /* np.dot(L, xi) */
            for (int j = 0; j < d; j++) acc += p->L[i * d + j] * xi[j];
            thp[i] = th[i] + scale * acc;
        }
        return 0.0;
    }
    if (p->kind == ORC_PCN) { /* randomwalk.py:89-101 */
        double *ufw = work, *urv = work + d;
        for (int i = 0; i < d; i++) {
            double acc = 0;
            for (int j = 0; j < d; j++) acc += p->L[i * d + j] * xi[j];
            thp[i] = p->rho * th[i] + p->rho_c * acc;
        }
        double sf = 0, sr = 0;
        for (int i = 0; i < d; i++) { /* u = solve(L * rho_c, dtheta), forward substitution */
            double af = thp[i] - p->rho * th[i], ar = th[i] - p->rho * thp[i];
            for (int j = 0; j < i; j++) {
                af -= p->L[i * d + j] * p->rho_c * ufw[j];
                ar -= p->L[i * d + j] * p->rho_c * urv[j];
            }
            ufw[i] = af / (p->L[i * d + i] * p->rho_c);
            urv[i] = ar / (p->L[i * d + i] * p->rho_c);
            sf += ufw[i] * ufw[i];
            sr += urv[i] * urv[i];
        }
        return -0.5 * (sf - sr);
    }
    double *mom = work, *g = work + d, *scr = work + 2 * d;
    const double eps = p->adapt_target > 0 ? chain_scale * p->eps : p->eps;
    const double *C = p->chM;
    double *psc = (double *)malloc(sizeof(double) * 3 * d), *p0 = psc + d, *tmp = psc + 2 * d;
    /* p = normal(d); with a mass matrix p = chM p (hamiltonian.py:79-81) */
    for (int i = 0; i < d; i++) {
        if (C) {
            double acc = 0;
            for (int j = 0; j <= i; j++) acc += C[i * d + j] * xi[j];
            p0[i] = acc;
        } else p0[i] = xi[i];
    }
#define ORC_MINV(src, dst)                                                            \
    do { /* dst = solve(M, src) = chM^-T chM^-1 src (hamiltonian.py:29,36) */           \
        if (C) {                                                                      \
            for (int i = 0; i < d; i++) {                                             \
                double acc = (src)[i];                                                \
                for (int j = 0; j < i; j++) acc -= C[i * d + j] * tmp[j];             \
                tmp[i] = acc / C[i * d + i];                                          \
            }                                                                         \
            for (int i = d - 1; i >= 0; i--) {                                        \
                double acc = tmp[i];                                                  \
                for (int j = i + 1; j < d; j++) acc -= C[j * d + i] * (dst)[j];       \
                (dst)[i] = acc / C[i * d + i];                                        \
            }                                                                         \
            if (p->minv != 0.0)                                                       \
                for (int i = 0; i < d; i++) (dst)[i] *= p->minv;                      \
        } else                                                                        \
            for (int i = 0; i < d; i++) (dst)[i] = (src)[i];                          \
    } while (0)
    orc_grad1(m, th, g, scr);
    for (int i = 0; i < d; i++) mom[i] = p0[i] + 0.5 * eps * g[i]; /* :27 */
    ORC_MINV(mom, psc);
    for (int i = 0; i < d; i++) thp[i] = th[i] + eps * psc[i];     /* :29-30 */
    for (int s = 0; s < p->nsteps - 1; s++) {                      /* :34-38 */
        orc_grad1(m, thp, g, scr);
        for (int i = 0; i < d; i++) mom[i] = mom[i] + eps * g[i];
        ORC_MINV(mom, psc);
        for (int i = 0; i < d; i++) thp[i] = thp[i] + eps * psc[i];
    }
    orc_grad1(m, thp, g, scr);
    for (int i = 0; i < d; i++) mom[i] = mom[i] + 0.5 * eps * g[i]; /* :40 */
    double k1 = 0, k0 = 0;
    if (C) { /* p = solve(chM, p) for both momenta (:85-87) */
        for (int pass = 0; pass < 2; pass++) {
            const double *src = pass ? mom : p0;
            double kk = 0;
            for (int i = 0; i < d; i++) {
                double acc = src[i];
                for (int j = 0; j < i; j++) acc -= C[i * d + j] * tmp[j];
                tmp[i] = acc / C[i * d + i];
                kk += tmp[i] * tmp[i];
            }
            if (pass) k1 = kk; else k0 = kk;
        }
    } else
        for (int i = 0; i < d; i++) { k1 += mom[i] * mom[i]; k0 += xi[i] * xi[i]; }
#undef ORC_MINV
    free(psc);
    return 0.5 * (k1 - k0); /* :89 */
}

/* standalone propose over N chains, consuming the chains' streams */
void orc_propose(const orc_proposal *p, const orc_model *m, const double *theta, int64_t N,
                 u64 *rng, double *theta_p, double *logq)
{
    const int d = p->d;
    double *xi = (double *)malloc(sizeof(double) * d * 4);
    for (int64_t c = 0; c < N; c++) {
        orc_rng_normals(rng + 16 * c, d, xi);
        logq[c] = orc_propose1(p, m, theta + c * d, xi, theta_p + c * d, xi + d, 1.0);
    }
    free(xi);
}

/* ------------------------------------------------------------------------- */
/* 4. Sampler.run over N chains: samplers/sampler.py:44-90                     */
/* ------------------------------------------------------------------------- */
/*
 * theta (N,d), logpost (N), rng (N,16) are updated in place.
 * hist_theta ((T+1),N,d) / hist_logpost ((T+1),N): entry 0 is the state on
 * entry (sampler.py:49-50 keeps the last state as element 0), entries 1..T the
 * states appended by sample() (:89); either may be NULL.  accept (T,N) uint8.
 * Accept rule, sampler.py:83-84: mhratio = min(0, D) with PYTHON min, i.e.
 * D if D < 0 else 0 (so NaN -> 0 -> accept); accept iff log(U) < mhratio.
 */
/* adapt (N,3) or NULL: per chain [Nsamples, Naccepts, scale] of AdaptScaleProposal
 * (adaptive.py:18-23), updated in place by adapt() after every step (sampler.py:88). */
int orc_run_adapt(const orc_model *m, const orc_proposal *p, int64_t N, double *theta,
                  double *logpost, u64 *rng, int T, double *hist_theta, double *hist_logpost,
                  uint8_t *accept, int nthreads, double *adapt, double *aux);

int orc_run(const orc_model *m, const orc_proposal *p, int64_t N, double *theta,
            double *logpost, u64 *rng, int T, double *hist_theta, double *hist_logpost,
            uint8_t *accept, int nthreads)
{
    return orc_run_adapt(m, p, N, theta, logpost, rng, T, hist_theta, hist_logpost, accept,
                         nthreads, NULL, NULL);
}

/* aux (N) or NULL: AdaptScalepCN's per-chain rho (randomwalk.py:104-119): before every
 * proposal rho <- tanh(rho / scale) (:118), a recursion on the chain's own rho; rho_c keeps
 * its initial value.  NaN entries start from the proposal's rho. */
int orc_run_adapt(const orc_model *m, const orc_proposal *p, int64_t N, double *theta,
                  double *logpost, u64 *rng, int T, double *hist_theta, double *hist_logpost,
                  uint8_t *accept, int nthreads, double *adapt, double *aux)
{
    const int d = m->d;
    if (p->d != d) return -1;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel
    {
        double *work = (double *)malloc(sizeof(double) * d * 6);
        double *xi = work, *thp = work + d, *w2 = work + 2 * d;
#pragma omp for schedule(static)
        for (int64_t c = 0; c < N; c++) {
            double *th = theta + c * d;
            u64 *st = rng + 16 * c;
            if (hist_theta) memcpy(hist_theta + c * d, th, sizeof(double) * d);
            if (hist_logpost) hist_logpost[c] = logpost[c];
            for (int t = 0; t < T; t++) {
                orc_rng_normals(st, d, xi);
                if (p->nblk > 0) { /* block pick, test_changepoint.py:47-55 */
                    int pick = -1;
                    for (int b = 0; b < 3; b++) {
                        const double ub = (double)(orc_next(st + 4 * (b + 1)) >> 11) * ORC_2M53;
                        if (pick < 0 && b < p->nblk && ub < p->blk_p[b]) pick = b;
                    }
                    const int lo = pick <= 0 ? 0 : p->blk_end[pick - 1];
                    const int hi = pick < 0 ? 0 : p->blk_end[pick];
                    for (int j = 0; j < d; j++)
                        if (j < lo || j >= hi) xi[j] = 0.0;
                }
                orc_proposal q = *p;
                if (aux && adapt && p->kind == ORC_PCN && p->adapt_target > 0) {
                    const double r0 = (aux[c] != aux[c]) ? p->rho : aux[c];
                    aux[c] = q.rho = tanh(r0 / adapt[3 * c + 2]);
                }
                const double logq = orc_propose1(&q, m, th, xi, thp, w2, adapt ? adapt[3 * c + 2] : 1.0);
                const double lp_prop = orc_log_post1(m, thp, w2);
                const double delta = lp_prop - logpost[c] - logq;
                const double mh = (delta < 0) ? delta : 0.0;
                const double u = orc_rng_uniform(st);
                const int acc = orc_log(u) < mh;
                if (acc) {
                    memcpy(th, thp, sizeof(double) * d);
                    logpost[c] = lp_prop;
                }
                if (adapt && p->adapt_target > 0) { /* AdaptScaleProposal.adapt, adaptive.py:25-35 */
                    double *a = adapt + 3 * c;
                    /* the first call compares with last_theta = None: always "moved" */
                    const int moved = (a[0] == 0.0) ? 1 : acc;
                    a[0] += 1;
                    a[1] += moved;
                    const double rate = a[1] / a[0], ratio = exp(1.0 / a[0]);
                    if (rate > p->adapt_target) a[2] *= ratio; else a[2] /= ratio;
                }
                if (accept) accept[(int64_t)t * N + c] = (uint8_t)acc;
                if (hist_theta)
                    memcpy(hist_theta + ((int64_t)(t + 1) * N + c) * d, th, sizeof(double) * d);
                if (hist_logpost) hist_logpost[(int64_t)(t + 1) * N + c] = logpost[c];
            }
        }
        free(work);
    }
    return 0;
}

/* ------------------------------------------------------------------------- */
/* 4b. AdaptCovRandomWalk (Haario et al. 2001): proposals/adaptive.py:38-103,    */
/*     proposals/randomwalk.py:41-56, POOLED over the N chains                  */
/* ------------------------------------------------------------------------- */
/*
 * The reference adapts the proposal of ONE chain from that chain's history:
 * adapt(theta) (sampler.py:88) after every step accumulates S += 1, SX += theta,
 * SX2 += theta theta^T (adaptive.py:73-76) and, when n = S is a perfect square > 2
 * (:81-82), sets
 * -- in fact whenever the float test np.sqrt(n)**2 == n passes, see the run loop --
 *   Cs = (SX2 - SX SX^T / n) / (n - 1)                                     :83
 *   smooth_adapt:  C = (n Cs + t_adapt C0) / (n + t_adapt)                 :86
 *   else        :  C = C0 if n < t_adapt                (:90 stores it in a dead
 *                      variable `_cov`; this restatement does what it says)
 *                  C = Cs + 1e-12 mean(diag C0) I  otherwise              :92-93
 *   marginalize :  C = diag(diag C)                                        :96-97
 *   C /= d^0.4;  L = cholesky(C) / d^0.2                                   :101-102
 * With N chains the build keeps ONE proposal for all of them and pools the
 * accumulators over chains: after iteration n of every chain the
 * sample count is N n, the update schedule stays "n a perfect square > 2", and
 * `n` in the three formulas above is the pooled count N n.  For N = 1 this is the
 * reference's algorithm (np.float, removed from numpy, read as float; :81).
 *
 * st: [n, SX (d), SX2 (d*d), C (d*d), L (d*d), minv] updated in place (minv: HMC variant only,
 * 0 until the first update); on the first call
 * n = 0, sums 0, C = C0, L = cholesky(C0) (adaptive.py:59-67).  p supplies .scale.
 * Returns 0, -1 on a dimension mismatch, -2 if a covariance is not positive definite
 * (the reference raises LinAlgError there).
 */
static int orc_cholesky(const double *C, int d, double *L)
{
    memset(L, 0, sizeof(double) * d * d);
    for (int j = 0; j < d; j++) {
        double s = C[j * d + j];
        for (int k = 0; k < j; k++) s -= L[j * d + k] * L[j * d + k];
        if (!(s > 0.0)) return -2;
        const double ljj = sqrt(s);
        L[j * d + j] = ljj;
        for (int i = j + 1; i < d; i++) {
            double t = C[i * d + j];
            for (int k = 0; k < j; k++) t -= L[i * d + k] * L[j * d + k];
            L[i * d + j] = t / ljj;
        }
    }
    return 0;
}

int orc_adaptcov_update(double *st, const double *C0, int d, double ntot, double t_adapt,
                        int marginalize, int smooth)
{
    double *SX = st + 1, *SX2 = SX + d, *C = SX2 + (size_t)d * d, *L = C + (size_t)d * d;
    double tr = 0;
    for (int i = 0; i < d; i++) tr += C0[i * d + i];
    const double creg = 1e-12 * (tr / d);
    for (int i = 0; i < d; i++)
        for (int j = 0; j < d; j++) {
            const double cs = (SX2[i * d + j] - SX[i] * SX[j] / ntot) / (ntot - 1);
            double c;
            if (smooth) c = (ntot * cs + t_adapt * C0[i * d + j]) / (ntot + t_adapt);
            else if (ntot < t_adapt) c = C0[i * d + j];
            else c = cs + (i == j ? creg : 0.0);
            if (marginalize && i != j) c = 0.0;
            C[i * d + j] = c / pow((double)d, 0.4);
        }
    const int rc = orc_cholesky(C, d, L);
    if (rc) return rc;
    const double f = pow((double)d, 0.2);
    for (int i = 0; i < d * d; i++) L[i] /= f;
    return 0;
}

int orc_run_adaptcov(const orc_model *m, const orc_proposal *p, int64_t N, double *theta,
                     double *logpost, u64 *rng, int T, double *st, const double *C0,
                     double t_adapt, int marginalize, int smooth, int exact_squares,
                     double *hist_theta, double *hist_logpost, uint8_t *accept, double *adapt)
{
    /* adapt (N,3) non-NULL with p->adapt_target > 0: AdaptScaleCovRandomWalk
     * (randomwalk.py:62-75): the scale adaptation of adaptive.py:25-35 per chain, then the
     * covariance adaptation, after every step (:73-75) */
    const int d = m->d;
    if (p->d != d || (p->kind != ORC_RW && p->kind != ORC_HMC)) return -1;
    double *SX = st + 1, *SX2 = SX + d, *L = SX2 + 2 * (size_t)d * d;
    double *work = (double *)malloc(sizeof(double) * d * 6);
    double *xi = work, *thp = work + d, *w2 = work + 2 * d;
    orc_proposal q = *p;
    q.nblk = 0;
    if (p->kind == ORC_RW) q.L = L;
    else { /* AdaptCovHMC.propose, hamiltonian.py:117-119: M, chM = C, L */
        q.chM = L;
        q.minv = st[0] > 0 && st[1 + d + 3 * (size_t)d * d] != 0.0 ? st[1 + d + 3 * (size_t)d * d] : 0.0;
    }
    if (hist_theta) memcpy(hist_theta, theta, sizeof(double) * N * d);
    if (hist_logpost) memcpy(hist_logpost, logpost, sizeof(double) * N);
    int rc = 0;
    for (int t = 0; t < T && !rc; t++) {
        for (int64_t c = 0; c < N; c++) {
            double *th = theta + c * d;
            u64 *sc = rng + 16 * c;
            orc_rng_normals(sc, d, xi);
            const double logq = orc_propose1(&q, m, th, xi, thp, w2, adapt ? adapt[3 * c + 2] : 1.0);
            const double lp_prop = orc_log_post1(m, thp, w2);
            const double delta = lp_prop - logpost[c] - logq;
            const double mh = (delta < 0) ? delta : 0.0;
            const int acc = orc_log(orc_rng_uniform(sc)) < mh;
            if (acc) {
                memcpy(th, thp, sizeof(double) * d);
                logpost[c] = lp_prop;
            }
            if (adapt && p->adapt_target > 0) { /* adaptive.py:25-35, as in orc_run_adapt */
                double *a = adapt + 3 * c;
                const int moved = (a[0] == 0.0) ? 1 : acc;
                a[0] += 1;
                a[1] += moved;
                const double rate = a[1] / a[0], ratio = exp(1.0 / a[0]);
                if (rate > p->adapt_target) a[2] *= ratio; else a[2] /= ratio;
            }
            if (accept) accept[(int64_t)t * N + c] = (uint8_t)acc;
            for (int i = 0; i < d; i++) { /* adapt(), adaptive.py:73-76 */
                SX[i] += th[i];
                for (int j = 0; j < d; j++) SX2[i * d + j] += th[i] * th[j];
            }
        }
        st[0] += 1;
        if (hist_theta) memcpy(hist_theta + (int64_t)(t + 1) * N * d, theta, sizeof(double) * N * d);
        if (hist_logpost) memcpy(hist_logpost + (int64_t)(t + 1) * N, logpost, sizeof(double) * N);
        /* :81-82 `np.sqrt(n)**2 == n and n > 2`: meant as "n a perfect square", but in
         * double arithmetic fl(fl(sqrt n)^2) == n also holds for about half of the other n
         * (11, 14, 17, 21, 22, ...).  The reference's behaviour is the float test, restated
         * literally; the documented intent is available as the exact_squares variant. */
        const double n = st[0], r = sqrt(n);
        const int due = exact_squares ? (floor(r + 0.5) * floor(r + 0.5) == n) : (r * r == n);
        if (due && n > 2) {
            rc = orc_adaptcov_update(st, C0, d, n * (double)N, t_adapt, marginalize, smooth);
            if (p->kind == ORC_HMC) { /* from now on C = d^0.4 L L^T */
                st[1 + d + 3 * (size_t)d * d] = pow((double)d, -0.4);
                q.minv = st[1 + d + 3 * (size_t)d * d];
            }
        }
    }
    free(work);
    return rc;
}

/* ------------------------------------------------------------------------- */
/* 5. PTSampler.run: samplers/ptsampler.py:42-127                               */
/* ------------------------------------------------------------------------- */
/*
 * Replicas live in SLOTS grouped in tiles of 16 (one wavefront of the HIP kernel):
 * with a ladder of K temperatures a tile holds EPW = 16/K ensembles, ensemble e of
 * the tile in slots e*K .. e*K+K-1 (rung i = temperature betas[i]); slots >= EPW*K
 * of a tile are idle and untouched.  theta (S,d), rng (S,16), S a multiple of 16.
 *
 * Each replica is a Sampler over TemperedModel(model, beta) (ptsampler.py:11-39:
 * log_likelihood * beta, log_prior unchanged), so its log posterior is
 * Model.log_posterior (model.py:43-55) of (logP, beta * logL).
 *
 * Stream use per sweep and slot (rmh-rng extension): first ONE word from each
 * of the slot's four sub-streams -- sub-stream 0's is the selector uniform `u` of
 * ptsampler.py:103, sub-stream 1's the swap uniform of :119, the other two are
 * unused -- then the within-chain draw of Sampler.sample (d normals, accept
 * uniform), consumed whether or not the rung makes a within-chain move.
 *
 * Sweep, ptsampler.py:100-125: rungs in order; a rung already swapped is skipped
 * (:104-106); rung i makes a within-chain MH step iff u > Pswap or i == K-1 (:107-109),
 * else proposes to swap with rung i+1 (:110-125), accepted iff
 *   U < min(1, exp(D)),  D = (logpost_ji + logpost_ij) - (logpost_ii + logpost_jj)
 * restated as log(U) < min(0, D) with Python's min (NaN -> accept), the form the
 * within-chain rule already has (sampler.py:83-84).
 * Every replica appends exactly one state per sweep (sample() or _add_state()).
 * code (T,S) uint8: bit 0 = the replica's state changed hands (accepted move or
 * swap), bit 1 = the replica took part in a swap proposal.
 * hist_theta ((T+1),S,d), hist_logpost ((T+1),S) as in orc_run; either may be NULL.
 */
static double orc_tempered(double lp, double ll, double beta)
{
    const double tl = ll * beta; /* ptsampler.py:35-36 */
    if (isinf(lp) || isnan(lp) || isinf(tl) || isnan(tl)) return -INFINITY; /* model.py:49-52 */
    return lp + tl;
}

int orc_run_pt(const orc_model *m, const orc_proposal *p, int64_t S, double *theta,
               double *logpost, u64 *rng, int T, const double *betas, int K, double pswap,
               double *hist_theta, double *hist_logpost, uint8_t *code)
{
    const int d = m->d;
    if (p->d != d || (p->kind != ORC_RW && p->kind != ORC_HMC && p->kind != ORC_PCN) || K < 1 || K > 16 || (S & 15)) return -1;
    const int EPW = 16 / K;
    double *work = (double *)malloc(sizeof(double) * d * 8);
    double *xi = work, *thp = work + d, *w2 = work + 2 * d;
    double *lP = (double *)malloc(sizeof(double) * S), *lL = (double *)malloc(sizeof(double) * S);
    for (int64_t c = 0; c < S; c++) { /* Sampler.__init__ of every rung, sampler.py:42 */
        const int cl = (int)(c & 15);
        lP[c] = orc_log_prior1(m, theta + c * d);
        lL[c] = orc_log_lik1(m, theta + c * d, w2);
        if (cl < EPW * K) logpost[c] = orc_tempered(lP[c], lL[c], betas[cl % K]);
        if (hist_theta) memcpy(hist_theta + c * d, theta + c * d, sizeof(double) * d);
        if (hist_logpost) hist_logpost[c] = logpost[c];
    }
    for (int t = 0; t < T; t++) {
        for (int64_t e0 = 0; e0 < S; e0 += 16)
            for (int e = 0; e < EPW; e++) {
                const int64_t base = e0 + (int64_t)e * K;
                int swapped_next = 0;
                for (int i = 0; i < K; i++) {
                    const int64_t c = base + i;
                    u64 *st = rng + 16 * c;
                    const double usel = (double)(orc_next(st) >> 11) * ORC_2M53;
                    const double vswp = (double)(orc_next(st + 4) >> 11) * ORC_2M53;
                    (void)orc_next(st + 8);
                    (void)orc_next(st + 12);
                    orc_rng_normals(st, d, xi);
                    const double ua = orc_rng_uniform(st);
                    uint8_t cd = 0;
                    if (swapped_next) { /* :104-106; its state was appended by rung i-1 */
                        swapped_next = 0;
                        continue;
                    }
                    double *th = theta + c * d;
                    if (usel > pswap || i == K - 1) { /* within-chain step, Sampler.sample */
                        /* the proposal is the SAME object for every rung (ptsampler.py:82): an HMC
                         * proposal integrates with the base model's gradient whatever the rung */
                        const double logq = orc_propose1(p, m, th, xi, thp, w2, 1.0);
                        const double pP = orc_log_prior1(m, thp), pL = orc_log_lik1(m, thp, w2);
                        const double lpp = orc_tempered(pP, pL, betas[i]);
                        const double delta = lpp - logpost[c] - logq;
                        const double mh = (delta < 0) ? delta : 0.0;
                        if (orc_log(ua) < mh) {
                            memcpy(th, thp, sizeof(double) * d);
                            logpost[c] = lpp; lP[c] = pP; lL[c] = pL;
                            cd = 1;
                        }
                        if (code) code[(int64_t)t * S + c] = cd;
                    } else { /* swap proposal with the next rung, :110-125 */
                        const int64_t j = c + 1;
                        const double lij = orc_tempered(lP[j], lL[j], betas[i]);     /* model_i(theta_j) */
                        const double lji = orc_tempered(lP[c], lL[c], betas[i + 1]); /* model_j(theta_i) */
                        const double delta = (lji + lij) - (logpost[c] + logpost[j]);
                        const double mh = (delta < 0) ? delta : 0.0;
                        cd = 2;
                        if (orc_log(vswp) < mh) {
                            double *tj = theta + j * d;
                            for (int q = 0; q < d; q++) { const double x = th[q]; th[q] = tj[q]; tj[q] = x; }
                            double x = lP[c]; lP[c] = lP[j]; lP[j] = x;
                            x = lL[c]; lL[c] = lL[j]; lL[j] = x;
                            logpost[c] = lij; logpost[j] = lji;
                            cd = 3;
                        }
                        if (code) { code[(int64_t)t * S + c] = cd; code[(int64_t)t * S + j] = cd; }
                        swapped_next = 1;
                    }
                }
            }
        if (hist_theta) memcpy(hist_theta + (int64_t)(t + 1) * S * d, theta, sizeof(double) * S * d);
        if (hist_logpost) memcpy(hist_logpost + (int64_t)(t + 1) * S, logpost, sizeof(double) * S);
    }
    free(work); free(lP); free(lL);
    return 0;
}

/* ------------------------------------------------------------------------- */
/* 6. Trans-dimensional changepoint sampling                    */
/* ------------------------------------------------------------------------- */
/*
 * The reference's changepoint script samples over the NUMBER of changepoints:
 * ChangepointRegression1DProp.propose (riemann/test_changepoint.py:43-73) picks a block
 * random walk (cpx / cpv / sig, :47-55) or else a birth / death move (:56-71) built on
 * ChangepointRegression1D.add_changepoint / subtract_changepoint
 * (riemann/models/changepoint.py:193-240) with the height map of :48-69; the model's
 * log_prior / log_likelihood (:106-160) take any number of changepoints.
 *
 * State of one chain: a PADDED vector of 2 KM + 3 doubles,
 *   [cpx (KM, +inf beyond k), cpv (KM+1, 1.0 beyond k+1), sig, k],
 * KM = capacity (the model's kmax).  A birth at k = KM cannot be represented and is
 * rejected (the reference's prior does not bound k; with Poisson(lamb) pieces such states
 * have negligible mass).
 *
 * Stream per iteration (rmh-rng extension): the normals of the 2 KM + 2 padded
 * coordinates (a block of k entries uses the first k normals of its padded block), then
 * TWO more words from every sub-stream:
 *   first : sub-stream 0 accept uniform, 1..3 the chained block selectors U1, U2, U3
 *   second: sub-stream 0 U4 (birth if k = 0 or U4 > 0.5, :58), 1 the new location
 *           s = xmin + (xmax - xmin) U (:59), 2 the height split u = 0.5 + (-0.1 + 0.2 U) /
 *           sqrt(n) (:60), 3 the changepoint to remove, floor(U k) (:66).
 * Block covariances (:36-41): C_cpx = 0.01 (xmax-xmin)/(k+1), C_cpv = 0.01 hscale^2 / n,
 * C_sig = 0.01 hscale, each times the identity; the proposal is theta + sqrt(C) xi.
 * logqratio of a birth / death = log |det J| of the height map, which the reference gets
 * from autograd; analytically log(h / (u (1-u))) for (h, u) -> (h1, h2) and its negative
 * for the inverse.
 * move (T,N) uint8: 0 cpx, 1 cpv, 2 sig, 3 birth, 4 death, 5 birth refused at capacity.
 */
static double orc_cprj_post(const orc_model *m, int KM, const double *pad, double *scratch)
{
    const int k = (int)pad[2 * KM + 2];
    orc_model mm = *m;
    mm.kind = ORC_CHANGEPOINT;
    mm.k = k;
    mm.d = 2 * k + 2;
    double *c = scratch; /* compact [cpx k, cpv k+1, sig] */
    for (int j = 0; j < k; j++) c[j] = pad[j];
    for (int j = 0; j <= k; j++) c[k + j] = pad[KM + j];
    c[2 * k + 1] = pad[2 * KM + 1];
    return orc_log_post1(&mm, c, scratch + 2 * KM + 4);
}

void orc_cprj_log_posterior(const orc_model *m, int KM, const double *theta, int64_t N, double *out)
{
    double *scratch = (double *)malloc(sizeof(double) * (4 * KM + 16));
    for (int64_t c = 0; c < N; c++) out[c] = orc_cprj_post(m, KM, theta + c * (2 * KM + 3), scratch);
    free(scratch);
}

int orc_run_cprj(const orc_model *m, int KM, double hscale, const double *pcum, int64_t N,
                 double *theta, double *logpost, u64 *rng, int T, double *hist_theta,
                 double *hist_logpost, uint8_t *accept, uint8_t *move)
{
    const int D = 2 * KM + 3, Dn = 2 * KM + 2, n = (int)m->n;
    double *xi = (double *)malloc(sizeof(double) * (Dn + 4)), *thp = (double *)malloc(sizeof(double) * D);
    double *scratch = (double *)malloc(sizeof(double) * (4 * KM + 16));
    const double sv = sqrt(0.01 * hscale * hscale / n), ss = sqrt(0.01 * hscale);
    for (int64_t c = 0; c < N; c++) {
        if (hist_theta) memcpy(hist_theta + c * D, theta + c * D, sizeof(double) * D);
        if (hist_logpost) hist_logpost[c] = logpost[c];
    }
    for (int t = 0; t < T; t++)
        for (int64_t c = 0; c < N; c++) {
            double *th = theta + c * D;
            u64 *st = rng + 16 * c;
            orc_rng_normals(st, Dn, xi);
            double w1[4], w2[4];
            orc_rng_words(st, w1);
            orc_rng_words(st, w2);
            const int k = (int)th[2 * KM + 2];
            memcpy(thp, th, sizeof(double) * D);
            double logq = 0.0;
            int mv;
            if (w1[1] < pcum[0]) { /* :47-49 */
                mv = 0;
                const double sx = sqrt(0.01 * (m->xmax - m->xmin) / (k + 1));
                for (int j = 0; j < k; j++) thp[j] = th[j] + sx * xi[j];
            } else if (w1[2] < pcum[1]) { /* :50-52 */
                mv = 1;
                for (int j = 0; j <= k; j++) thp[KM + j] = th[KM + j] + sv * xi[KM + j];
            } else if (w1[3] < pcum[2]) { /* :53-55 */
                mv = 2;
                thp[2 * KM + 1] = th[2 * KM + 1] + ss * xi[2 * KM + 1];
            } else if (k == 0 || w2[0] > 0.5) { /* birth, :58-63 and changepoint.py:193-218 */
                mv = 3;
                const double s = m->xmin + (m->xmax - m->xmin) * w2[1];
                const double u = 0.5 + (-0.1 + 0.2 * w2[2]) / sqrt((double)n);
                if (k == KM) mv = 5;
                else {
                    int nn = 0;
                    while (nn < k && th[nn] < s) nn++; /* searchsorted(cpx, s) */
                    const double h = th[KM + nn], ufac = sqrt((1 - u) / u);
                    for (int j = k; j > nn; j--) thp[j] = th[j - 1];
                    thp[nn] = s;
                    for (int j = k + 1; j > nn + 1; j--) thp[KM + j] = th[KM + j - 1];
                    thp[KM + nn] = h / ufac;
                    thp[KM + nn + 1] = h * ufac;
                    thp[2 * KM + 2] = k + 1;
                    logq = log(fabs(h / (u * (1 - u))));
                }
            } else { /* death, :65-71 and changepoint.py:220-240 */
                mv = 4;
                int nn = (int)(w2[3] * k);
                if (nn >= k) nn = k - 1;
                const double h1 = th[KM + nn], h2 = th[KM + nn + 1];
                const double h = sqrt(h1 * h2), u = 1.0 / (1 + h2 / h1);
                for (int j = nn; j < k - 1; j++) thp[j] = th[j + 1];
                thp[k - 1] = INFINITY;
                thp[KM + nn] = h;
                for (int j = nn + 1; j < k; j++) thp[KM + j] = th[KM + j + 1];
                thp[KM + k] = 1.0;
                thp[2 * KM + 2] = k - 1;
                logq = log(fabs(u * (1 - u) / h));
            }
            const double lpp = (mv == 5) ? -INFINITY : orc_cprj_post(m, KM, thp, scratch);
            const double delta = lpp - logpost[c] - logq;
            const double mh = (delta < 0) ? delta : 0.0;
            const int acc = orc_log(w1[0]) < mh;
            if (acc) {
                memcpy(th, thp, sizeof(double) * D);
                logpost[c] = lpp;
            }
            if (accept) accept[(int64_t)t * N + c] = (uint8_t)acc;
            if (move) move[(int64_t)t * N + c] = (uint8_t)mv;
            if (hist_theta) memcpy(hist_theta + ((int64_t)(t + 1) * N + c) * D, th, sizeof(double) * D);
            if (hist_logpost) hist_logpost[(int64_t)(t + 1) * N + c] = logpost[c];
        }
    free(xi); free(thp); free(scratch);
    return 0;
}

int orc_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
