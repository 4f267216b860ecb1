/*
 * riemann_hip.h -- C ABI of libriemann_hip.so, the MI355X (gfx950) back end of
 * the many-chain Metropolis-Hastings hot path
 *     Sampler.run() -> Proposal.propose() -> Model.log_posterior()
 * of rscalzo/riemann.  Plain pointers and sizes only; called from pure Python
 * through ctypes (riemann_amd/_lib.py) -- the binding a maintainer of the
 * reference would add is shown in INTEGRATION.md.
 *
 * Each entry point cites the reference interface it replaces (paths relative
 * to the reference repository root).
 *
 * Conventions
 *  - every function returns 0 on success, non-zero on failure;
 *    rmh_last_error() then describes the failure (thread-local string).
 *    RMH_ERR_PARAM marks argument/shape errors, which the Python layer raises
 *    as ParameterError (riemann/sampling_errors.py:23-27, raise sites
 *    proposals/randomwalk.py:23-24, models/gaussian.py:35-39).
 *  - host arrays are caller-owned, C-contiguous, borrowed for the call:
 *    theta is (N, d) float64 row-major as in the reference; device memory is
 *    library-owned behind the opaque handles and released by the *_free calls.
 *  - non-finite log densities are data, not errors: they follow
 *    Model.log_posterior (models/model.py:49-52: nan or +-inf -> -inf) and the
 *    accept rule of Sampler.sample (samplers/sampler.py:83-84).
 *  - one device per process (rmh_init); handles are not thread-safe.
 *  - random numbers: the "rmh-rng v4" stream, keyed by
 *    (seed, GLOBAL chain id); it replaces the process-global np.random used at
 *    proposals/randomwalk.py:25, proposals/hamiltonian.py:79, samplers/sampler.py:84.
 */
#ifndef RIEMANN_HIP_H
#define RIEMANN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RMH_OK 0
#define RMH_ERR_PARAM 1   /* bad argument / shape: ParameterError */
#define RMH_ERR_HIP 2     /* HIP runtime failure: RuntimeError */
#define RMH_ERR_STATE 3   /* call out of sequence (e.g. empty history) */

typedef struct rmh_model *rmh_model_t;
typedef struct rmh_proposal *rmh_proposal_t;
typedef struct rmh_state *rmh_state_t;
typedef struct rmh_history *rmh_history_t;

/* ---- context ------------------------------------------------------------ */
int rmh_init(int device);                 /* hipSetDevice + stream; idempotent */
int rmh_device_count(int *count);
int rmh_device_name(char *buf, int buflen);
int rmh_synchronize(void);                /* drain the library's stream */
const char *rmh_last_error(void);
/* duration in ms (HIP events on the library's stream) of the kernels of the
 * last rmh_run call, and their number */
int rmh_last_run_ms(double *ms, int *launches);

/* ---- Model: riemann/models/model.py:4-64 -------------------------------- */
/* MultiGaussianDist(mu, C), models/gaussian.py:27-44.  L = cholesky(C), lower,
 * (d,d) row-major, as gaussian.py:42 caches it. */
int rmh_model_gaussian_create(const double *mu, const double *L, int d, rmh_model_t *out);
/* Logistic regression with N(0, tau2 I) prior -- a new plugin against the
 * Model ABC (not in the reference; spec in DESIGN.md).  X (n,d) row-major, y (n). */
int rmh_model_logreg_create(const double *X, const double *y, int64_t n, int d,
                            double tau2, rmh_model_t *out);
/* ChangepointRegression1D at fixed k, models/changepoint.py:86-181.
 * theta packs (cpx[k], cpv[k+1], sig): d = 2k+2. */
int rmh_model_changepoint_create(const double *x, const double *y, int n, int k,
                                 double xmin, double xmax, double lamb,
                                 double alpha, double beta, rmh_model_t *out);
/* benchmarks.MixtureDist(components) with MultiGaussianDist components,
 * models/benchmarks.py:31-62 (targets benchmark_gauss1d_mix2, benchmark_gauss2d_mix2_cross,
 * :75-85).  weights (ncomp) NORMALISED as at :40, mu (ncomp,d), L (ncomp,d,d) the components'
 * Cholesky factors, log_prior the constant log(sum(w exp(0))) of :54-62.  The gradient
 * (rmh_grad_log_posterior, HMC proposals) is an addition: the reference class has none. */
int rmh_model_mixture_create(const double *weights, const double *mu, const double *L, int ncomp,
                             int d, double log_prior, rmh_model_t *out);
/* ChangepointRegression1D with a VARIABLE number of changepoints k <= kmax (the model's own
 * log_prior / log_likelihood take any k, changepoint.py:106-160).  theta is the padded vector
 * [cpx (kmax, +inf beyond k), cpv (kmax+1, 1.0 beyond k+1), sig, k]: d = 2 kmax + 3.
 * Sampled by rmh_proposal_cprj_create (trans-dimensional moves); no gradient. */
int rmh_model_changepoint_rj_create(const double *x, const double *y, int n, int kmax, double xmin,
                                    double xmax, double lamb, double alpha, double beta,
                                    rmh_model_t *out);
int rmh_model_dim(rmh_model_t m, int *d);
int rmh_model_free(rmh_model_t m);

/* Model.log_prior / log_likelihood / log_posterior (models/model.py:27-55) and
 * the gradient contract of hamiltonian.py:60-70, for N states at once.
 * theta (N,d) -> out (N) or (N,d).  Host in, host out. */
int rmh_log_prior(rmh_model_t m, const double *theta, int64_t N, double *out);
int rmh_log_likelihood(rmh_model_t m, const double *theta, int64_t N, double *out);
int rmh_log_posterior(rmh_model_t m, const double *theta, int64_t N, double *out);
int rmh_grad_log_posterior(rmh_model_t m, const double *theta, int64_t N, double *out);

/* ---- Proposal: riemann/proposals/proposal.py:1-26 ----------------------- */
/* MetropolisRandomWalk(C), proposals/randomwalk.py:17-26.  L = cholesky(C). */
int rmh_proposal_rw_create(const double *L, int d, double scale, rmh_proposal_t *out);
/* ChangepointRegression1DProp.propose at fixed k, riemann/test_changepoint.py:18-56: the d
 * coordinates are cut into nblk <= 3 consecutive blocks (cpx, cpv, sig: sizes k, k+1, 1), each
 * with its own MetropolisRandomWalk (:36-41; L must be block diagonal); chained independent
 * uniforms pick the block: block b iff u_b < pcum[b] and no earlier block was picked
 * (:47-55, pcum = P_cumprop_cpx, _cpv, _sig).  No block picked = the reference's
 * trans-dimensional branch (:56-71), which does not exist at fixed k: theta' = theta.
 * nblk = 0 returns to the full-vector random walk.  (Changepoint model, fused run only.) */
int rmh_proposal_rw_set_blocks(rmh_proposal_t p, int nblk, const int *sizes, const double *pcum);
/* ChangepointRegression1DProp(model, hscale) WITH its trans-dimensional branch,
 * riemann/test_changepoint.py:18-73: chained uniforms pick the cpx / cpv / sig block random
 * walk (pcum = P_cumprop_cpx, _cpv, _sig; covariances :36-41) and otherwise a birth (k = 0 or
 * U > 0.5) or a death move on ChangepointRegression1D.add_changepoint / subtract_changepoint
 * (models/changepoint.py:193-240) whose logqratio is log|det J| of the height map (:48-78;
 * autograd in the reference, analytic here).  d = 2 kmax + 3 of the variable-k model; a
 * birth at k = kmax is refused.  In the history the accept byte carries bit 0 = accepted and
 * bits 1..3 = the move (0 cpx, 1 cpv, 2 sig, 3 birth, 4 death, 5 birth refused).  Fused run
 * only.  Stream use: DESIGN.md. */
int rmh_proposal_cprj_create(double hscale, const double *pcum, int d, rmh_proposal_t *out);
int rmh_proposal_cprj_set_probs(rmh_proposal_t p, const double *pcum);
/* pCN(C, rho), proposals/randomwalk.py:78-101.  L = cholesky(C).  (Gaussian, mixture and
 * logistic models.) */
int rmh_proposal_pcn_create(const double *L, int d, double rho, rmh_proposal_t *out);
/* VanillaHMC(eps, Nsteps, gradlogpost, M=None), proposals/hamiltonian.py:60-91;
 * the gradient is the device Model's.  Nsteps = 1 is MALA. */
int rmh_proposal_hmc_create(double eps, int nsteps, int d, rmh_proposal_t *out);
/* VanillaHMC(..., M): mass matrix, hamiltonian.py:71-74.  chM = cholesky(M), lower,
 * (d,d) row-major; NULL removes it. */
int rmh_proposal_hmc_set_mass(rmh_proposal_t p, const double *chM);
int rmh_proposal_set_scale(rmh_proposal_t p, double scale); /* .scale, randomwalk.py:18 */
int rmh_proposal_set_eps(rmh_proposal_t p, double eps);     /* .eps, hamiltonian.py:69 */
/* AdaptScaleProposal mix-in, proposals/adaptive.py:11-35: AdaptScaleRandomWalk
 * (randomwalk.py:29-38, target 0.25) and AdaptScaleHMC (hamiltonian.py:94-104,
 * target 0.75, eps = scale * eps0).  target > 0 switches the proposal to its
 * adaptive variant (on a pCN proposal: AdaptScalepCN, randomwalk.py:104-119, target 0.25,
 * Gaussian model); the per-chain state [Nsamples, Naccepts, scale] lives with the
 * chains (rmh_state_get_adapt / rmh_state_set_adapt, (N,3) float64). */
int rmh_proposal_set_adapt(rmh_proposal_t p, double target_accept_rate);
/* AdaptCovProposal mix-in (Haario et al. 2001), proposals/adaptive.py:38-103, as
 * AdaptCovRandomWalk (randomwalk.py:41-56): the proposal covariance follows the covariance of
 * the chain history.  The reference adapts one chain's proposal from that chain; here ONE
 * proposal serves all chains of the run and the accumulators S, SX, SX2 of adapt()
 * (:73-76) are pooled over chains (and over ranks when rmh_comm_init was called: one
 * all-reduce per update) -- for one chain this is the reference's algorithm.  Updates
 * happen after the n-th step of every chain when `np.sqrt(n)**2 == n and n > 2` (:81-82)
 * evaluates true in double arithmetic, which besides the perfect squares it is meant to
 * select also holds for about half of the other n; exact_squares != 0 restricts updates to
 * perfect squares.  C0 (d,d) initial covariance (NULL switches the adaptation off); t_adapt,
 * marginalize, smooth_adapt as at :45-58.  `C = C0 if n < t_adapt` does what :90 says
 * (the reference stores it in an unused variable).  d <= 111.  Random-walk proposals (plain,
 * or with rmh_proposal_set_adapt: AdaptScaleCovRandomWalk, randomwalk.py:62-75) on the
 * Gaussian, mixture and logistic models; HMC proposals (AdaptCovHMC, hamiltonian.py:107-120:
 * the adapted C, L become the mass matrix M, chM) on the Gaussian and logistic models.
 * rmh_run then issues one launch per stretch between updates. */
int rmh_proposal_set_adapt_cov(rmh_proposal_t p, const double *C0, double t_adapt, int marginalize,
                               int smooth_adapt, int exact_squares);
/* adaptation state for checkpoints: [n, SX (d), SX2 (d,d), C (d,d), L (d,d), minv] (this rank's
 * sums; C, L the current proposal covariance and factor, self.C / self.L of :99-102);
 * `count` = samples in the sums (chains x n); minv: HMC variant, M^-1 = minv (L L^T)^-1, else 0 */
int rmh_proposal_adapt_cov_get(rmh_proposal_t p, double *st);
int rmh_proposal_adapt_cov_set(rmh_proposal_t p, const double *st, double count);
int rmh_proposal_free(rmh_proposal_t p);

/* Proposal.propose (proposal.py:10-17) for N states: theta (N,d) ->
 * theta_p (N,d), logqratio (N).  rng (N,16) uint64 is advanced in place. */
int rmh_propose(rmh_model_t m, rmh_proposal_t p, const double *theta, int64_t N,
                uint64_t *rng, double *theta_p, double *logq);
/* the same with the normals of a given mode (rmh_state_set_normal_mode; 0 = rmh_propose) */
int rmh_propose_mode(rmh_model_t m, rmh_proposal_t p, const double *theta, int64_t N,
                     uint64_t *rng, int normal_mode, double *theta_p, double *logq);
/* The draws of T iterations of Sampler.sample for N chains from the host copy of their stream
 * state ((N,16) uint64, advanced in place): per iteration the d standard normals of
 * proposals/randomwalk.py:25 / proposals/hamiltonian.py:79 (xi, (T,N,d); d = 0: none, xi may be
 * null) and the accept uniform of samplers/sampler.py:84 (u, (T,N); null: the normals only and
 * the uniform's word is NOT drawn), in the order and with the bits the fused kernels use.  For the mixed-plugin path: a Python Model or Proposal next to
 * device ones runs sampler.py:72-90 on the host with THIS stream. */
int rmh_rng_draw(uint64_t *rng, int64_t N, int d, int T, double *xi, double *u);
/* the same in a given normal mode (0: as rmh_rng_draw; 1: the 64-bit-resolution mode of
 * rmh_state_set_normal_mode, ceil(d/4) words per sub-stream and vector) */
int rmh_rng_draw_mode(uint64_t *rng, int64_t N, int d, int T, int normal_mode, double *xi, double *u);


/* ---- chain state (device resident; survives between run() calls as the
 *      history lists do, samplers/sampler.py:49-50) ------------------------ */
/* N chains with global ids chain0 .. chain0+N-1 and streams seeded from `seed`. */
int rmh_state_create(rmh_model_t m, int64_t N, uint64_t seed, uint64_t chain0,
                     rmh_state_t *out);
/* Sampler.__init__, sampler.py:34-42: store theta0 and evaluate log_posterior. */
int rmh_state_set_theta(rmh_state_t s, const double *theta);
int rmh_state_get(rmh_state_t s, double *theta, double *logpost); /* current_state(), :56-60 */
int rmh_state_get_rng(rmh_state_t s, uint64_t *rng);   /* (N,16) */
int rmh_state_set_rng(rmh_state_t s, const uint64_t *rng);
/* Resolution of the chains' standard normals (np.random.normal of proposals/randomwalk.py:25 and
 * proposals/hamiltonian.py:79 draws float64 values).  mode 0 (default): two normals per 64-bit stream
 * word, each the quantile of a 32-bit half word (a lattice of 2^-27 in p inside a cell, |z| <= 6.4);
 * mode 1: ONE normal per word -- 52 bits inside a cell, p down to 2^-58 (|z| <= 8.6), the quantile map
 * good to 2.1e-16 -- at twice the stream words and a longer table evaluation; served by every kernel
 * family except the specialised changepoint kernels (routed to the generic one), d > 112 and the
 * variable-k / block-mixture proposals (refused). */
int rmh_state_set_normal_mode(rmh_state_t s, int mode);
int rmh_state_get_normal_mode(rmh_state_t s, int *mode);
int rmh_state_get_adapt(rmh_state_t s, double *adapt);       /* (N,3) */
int rmh_state_set_adapt(rmh_state_t s, const double *adapt);
/* the auxiliary per-chain value of an adaptive proposal, (N) float64: AdaptScalepCN's
 * recursive rho (randomwalk.py:118); NaN = not started */
int rmh_state_get_adapt_aux(rmh_state_t s, double *aux);
int rmh_state_set_adapt_aux(rmh_state_t s, const double *aux);
int rmh_state_device_ptrs(rmh_state_t s, void **theta_dn, void **logpost,
                          int64_t *ld);                 /* theta is (d, ld) on device */
int rmh_state_free(rmh_state_t s);

/* ---- history (device resident) ------------------------------------------ */
/* room for `entries` kept states and `iters` accept flags per chain */
int rmh_history_create(rmh_state_t s, int64_t entries, int64_t iters, rmh_history_t *out);
int rmh_history_entries(rmh_history_t h, int64_t *entries, int64_t *iters);
/* Pinned (page-locked) host memory for history downloads: rmh_history_download into such a
 * buffer overlaps the device-side transposes with DMA transfers at the PCIe rate; into pageable
 * memory the runtime stages every chunk.  The reference keeps its history in Python lists
 * (samplers/sampler.py:41, 69-70); this is the hand-over of the (n, N, d) array that replaces them. */
int rmh_host_alloc(void **ptr, uint64_t bytes);
int rmh_host_free(void *ptr);
/* entries [e0,e1) -> thetas (e1-e0, N, d), logpost (e1-e0, N); either may be NULL */
int rmh_history_download(rmh_history_t h, int64_t e0, int64_t e1, double *thetas,
                         double *logpost);
/* Reductions of entries [e0, e1) on the device, for diagnostics at sizes where the history itself
 * should not travel (cfg2: 53 GB of history, 105 MB of reductions): mean_dn, var_dn (d, N)
 * parameter-major -- per chain and parameter the mean and the biased variance -- and acf
 * (nlags, d) -- per parameter the autocorrelation function sum_t (x_t - m)(x_{t+k} - m) /
 * sum_t (x_t - m)^2 averaged over the chains, lags 0 .. nlags-1 (nlags = 32, 64, ... 320): the
 * input of the integrated autocorrelation time the reference's example prints
 * (examples/test_randomwalk.py:42).  Any output may be null. */
int rmh_history_reduce(rmh_history_t h, int64_t e0, int64_t e1, int nlags, double *mean_dn,
                       double *var_dn, double *acf);

/* The same reductions with the autocorrelation sums left as integers: acf_sums (nlags, d) =
 * sum over the chains of round(rho_chain(k) * RMH_ACF_SCALE).  Integer sums add up exactly, so the
 * sums of the ranks of a sharded job (riemann_amd.distributed.pool_history: one all-reduce) equal
 * the single-process sums bit for bit; acf = acf_sums / (N * RMH_ACF_SCALE). */
#define RMH_ACF_SCALE 1099511627776.0 /* 2^40 */
int rmh_history_reduce_sums(rmh_history_t h, int64_t e0, int64_t e1, int nlags, double *mean_dn,
                            double *var_dn, int64_t *acf_sums);
/* Pooled first and second moments of entries [e0, e1) over all chains, formed on the device:
 * moments = [count, sum theta (d), sum theta theta^T (d, d)] (the layout of rmh_allreduce_moments);
 * d <= 111.  Pooled mean and covariance of a recorded run without the history leaving the GPU. */
int rmh_history_moments(rmh_history_t h, int64_t e0, int64_t e1, double *moments);
/* accept flags of iterations [t0,t1) -> (t1-t0, N) uint8 */
int rmh_history_accept(rmh_history_t h, int64_t t0, int64_t t1, uint8_t *accept);
int rmh_history_device_ptrs(rmh_history_t h, void **thetas_edn, void **logpost_en,
                            void **accept_tn, int64_t *ld);
int rmh_history_free(rmh_history_t h);

/* ---- Sampler.run(Nsamples, Nburn, Nthin): samplers/sampler.py:44-54 ------
 * Runs T = Nsamples fused iterations of Sampler.sample (sampler.py:72-90) for
 * every chain of `s`.  The T+1 states (state on entry, then one per iteration)
 * are numbered e = 0..T; e is kept iff e >= nburn and (e-nburn) % nthin == 0
 * (the slice [Nburn::Nthin] of sampler.py:53-54) and written to history entry
 * (e-nburn)/nthin when `h` is not NULL.  As in the reference the chain's
 * current state afterwards is the LAST KEPT entry. */
int rmh_run(rmh_model_t m, rmh_proposal_t p, rmh_state_t s, int64_t T, int64_t nburn,
            int64_t nthin, rmh_history_t h);

/* ---- PTSampler.run: samplers/ptsampler.py:42-127 ----------------------------
 * Parallel tempering over a ladder betas[0..K-1] (1 <= K <= 16).  The chains of `s`
 * are SLOTS grouped in tiles of 16 (one wavefront): a tile holds 16/K ensembles,
 * ensemble e of the tile in slots e*K .. e*K+K-1, rung i sampling
 * TemperedModel(model, betas[i]) (:11-39); slots >= K*(16/K) of a tile are idle.
 * One sweep = PTSampler.sample (:92-127): per rung one selector uniform, then either
 * the within-chain MH step of Sampler.sample or a swap proposal with the next rung;
 * a swap exchanges the two replicas' registers inside the wavefront.  Every replica
 * records one state per sweep; T, nburn, nthin, h as in rmh_run (the reference
 * ignores Nburn/Nthin here, :84; pass 0, 1 for its behaviour).  Accept codes in the
 * history: bit 0 = the replica's state changed (accepted move or swap), bit 1 = the
 * replica took part in a swap proposal.  The state's log posteriors are the TEMPERED
 * ones.  Within-chain proposals: MetropolisRandomWalk and pCN (all four models), VanillaHMC / MALA
 * with or without a mass matrix (Gaussian, mixture, logistic; trajectories with the base model's gradient
 * on every rung, as the reference's shared proposal object gives).  Adaptive proposals are
 * refused (that shared object would couple the rungs of a sweep sequentially).
 * Stream use per sweep: DESIGN.md, "Parallel tempering". */
int rmh_run_pt(rmh_model_t m, rmh_proposal_t p, rmh_state_t s, const double *betas, int K,
               double pswap, int64_t T, int64_t nburn, int64_t nthin, rmh_history_t h);

/* ---- pooling the chains of several GPUs (no reference counterpart:
 *      riemann/README.md:21-23 only states the aspiration) ---------------------
 * One process per GPU.  Rank 0 calls rmh_comm_unique_id and ships the 128 bytes to
 * the other ranks (any channel); every rank then calls rmh_comm_init.  RCCL is
 * dlopen'ed on first use.  Both collectives run on the library's stream, outside
 * the sampling loop. */
int rmh_comm_unique_id(uint8_t *id128);
int rmh_comm_init(const uint8_t *id128, int nranks, int rank);
int rmh_comm_destroy(void);
/* ranks of the communicator and this process's rank (0, 0 when there is none) */
int rmh_comm_info(int *nranks, int *rank);
/* every rank has drained its stream and reached the call (one 8-byte all-reduce) */
int rmh_comm_barrier(void);
/* element-wise reduction over the ranks of a small host buffer, in place on every rank:
 * op 0 = sum, 1 = max, 2 = min (bench.py: max-over-ranks of the timed region) */
int rmh_comm_allreduce_host(double *buf, int64_t n, int op);
/* chains held by every rank (counts[nranks], may be null) and their sum */
int rmh_comm_pooled_chains(rmh_state_t s, int64_t *total, int64_t *counts);
/* all-gather of the ranks' current states: pooled (sum of the ranks' N, d), rank-major.
 * Shards may differ in size (every rank contributes a zero-padded block of the largest
 * shard to ONE ncclAllGather); size the output with rmh_comm_pooled_chains. */
int rmh_allgather_samples(rmh_state_t s, double *pooled);
/* duration (HIP events on the library's stream) and bytes contributed per rank of the
 * last rmh_allgather_samples */
int rmh_comm_last_ms(double *ms, int64_t *bytes_per_rank);
int rmh_allreduce_moments(rmh_state_t s, double *moments);

/* All-gather of recorded samples: entries e0, e0 + stride, ... < e1 of every rank's history ->
 * pooled (n_sel, sum_r N_r, d) on every rank (chain blocks in rank order, ragged shards padded to
 * the largest for one ncclAllGather per entry).  The thinned-sample counterpart of
 * rmh_allgather_samples; all ranks must pass the same e0, e1, stride. */
int rmh_allgather_history(rmh_history_t h, int64_t e0, int64_t e1, int64_t stride, double *pooled);
/* ---- test hooks ---------------------------------------------------------- */
/* the stream as the reference consumes it: xi (T,N,d), u (T,N) for chains
 * chain0..chain0+N-1, generated on the device */
int rmh_debug_rng_stream(uint64_t seed, uint64_t chain0, int64_t N, int d, int T,
                         double *xi, double *u);
int rmh_debug_rng_stream_mode(uint64_t seed, uint64_t chain0, int64_t N, int d, int T, int normal_mode,
                              double *xi, double *u);
/* D = A(16x4) B(4x16) + C(16x16) through one v_mfma_f64_16x16x4_f64, row-major in/out */
int rmh_debug_mfma(const double *A, const double *B, const double *C, double *D);

/* Kernel selection for A/B measurements and cross-checks (tests, tools/): a bit mask that
 * turns specialised kernels off or forces them.  Not part of the reference's interface; the
 * library never reads the environment for this.  Returns the previous mask through *old
 * (may be null).  Default 0 = the library picks. */
#define RMH_POLICY_NO_TAIL        0x01  /* d = 16 q + 1..4: generic padded row tile instead of the VALU tail rows */
#define RMH_POLICY_CP_NO_SS       0x02  /* changepoint: O(n) likelihood loop instead of sufficient statistics */
#define RMH_POLICY_CP_SS64        0x04  /* changepoint: one-lane-per-chain kernels at any chain count */
#define RMH_POLICY_CP_SS16        0x08  /* changepoint: four-lanes-per-chain kernels at any chain count */
#define RMH_POLICY_CP_NO_BUCKETS  0x10  /* changepoint: plain binary searches, no bucket table */
#define RMH_POLICY_CPRJ_NO_SS     0x20  /* trans-dimensional changepoint: O(n k) likelihood loop */
#define RMH_POLICY_ACOV_HOST      0x40  /* AdaptCov: covariance update on the host */
#define RMH_POLICY_ACOV_UNFUSED   0x80  /* AdaptCov: scratch history + k_moments instead of in-kernel moments */
int rmh_debug_set_kernel_policy(int mask, int *old);

#ifdef __cplusplus
}
#endif
#endif /* RIEMANN_HIP_H */
