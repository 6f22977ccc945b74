/*
 * tinyda_amd.h  --  C-ABI of the MI355X many-chain MH / DA / MLDA engine (libtinyda_hip.so).
 *
 * The reference (mikkelbue/tinyDA) has NO native boundary: its hot path is the Python loop
 *   tda.sample()            tinyDA/sampler.py:21-292
 *     Chain.sample          tinyDA/chain.py:78-129       (one Python iteration per MH step per chain)
 *       Proposal.*          tinyDA/proposal.py:171-512   (GaussianRandomWalk / CrankNicolson / AdaptiveMetropolis)
 *       Posterior.create_link  tinyDA/posterior.py:78-110
 *       GaussianLogLike     tinyDA/distributions.py:203-329
 *       RecursiveSampleMoments tinyDA/utils.py:104-124
 * This header is the boundary a maintainer would bind with ctypes from sampler.py (see
 * INTEGRATION.md): plain pointers and sizes, no torch / numpy types.  The Python mirror of the
 * reference API (tinyda_amd/) is a thin layer over exactly these entry points.
 *
 * Conventions
 *   - every function returns TDA_OK (0) or a negative tda_status; tda_last_error() gives the message
 *     (thread local);
 *   - the caller owns every buffer it passes; the engine owns its internal device state;
 *   - data pointers may be HOST or DEVICE memory (detected with hipPointerGetAttributes) unless a
 *     parameter says "host";  all matrices are row-major fp64;
 *   - one engine per GPU; calls on one engine are not thread-safe; different engines may be driven
 *     from different threads / processes (one process per GPU under torch.distributed);
 *   - chains are rows of device matrices: chain c of this engine has global id chain_offset + c, and
 *     the global id (not the GPU count) keys its random stream, so results do not depend on sharding.
 *
 * RNG stream (part of the contract; restated independently in oracle/tinyda_oracle.py)
 *   Philox4x32-10, key = (seed & 0xffffffff, seed >> 32), counter = (block, step, global chain, stream).
 *   u53(a, b) = ((a >> 5) * 2^26 + (b >> 6)) * 2^-53.
 *   stream 0, block b of step t : x0..x3 -> u1 = u53(x0,x1), u2 = u53(x2,x3),
 *        r = sqrt(-2 ln(1 - u1));  z[2b] = r cos(2 pi u2), z[2b+1] = r sin(2 pi u2)   (proposal normals)
 *   stream 1, block = level    : u = u53(x0,x1)                                       (accept uniform)
 *   stream 2, block b          : as stream 0, step = 0                                (theta0 ~ prior)
 *   stream 3, block 0, step = fine iteration : promoted index = (x0 * L) >> 32        (DA randomize_subchain_length)
 *   DREAM(Z) (proposal.py:811-852; round 4 layout, restated in tests/test_gpu_dreamz.py::_philox_dreamz_variates):
 *   stream 4, step field = step : block i < delta : archive rows r1 = (x0 M) >> 32, r2 = (x1 (M - 1)) >> 32, r2 += (r2 >= r1);
 *        block delta : crossover index by inverse cdf of u53(x0,x1) over pCR, forced index = (x2 dim) >> 32
 *   stream 5, block = delta + 1 + j, step field = step : crossover uniform of parameter j = u53(x0,x1), e-uniform = u53(x2,x3)
 *   stream 7, block = delta + 1 + j, step field = step >> 1 : eps normals of parameter j by the Box-Muller map of stream 0 in
 *        double precision; z0 serves the even step 2Q, z1 the odd step 2Q + 1
 *        (the reference draws both in double precision, proposal.py:846-847; round 3 used 32-bit uniforms and single-precision
 *        normals for a < 5 % gain of one kernel and is gone: checkpoint blobs of that release are refused)
 *   For levels >= 1 the accept uniform of that level's step n is stream 1, block = level, step = n.
 *   "step" is the count of base-level proposals made so far on the chain (proposal.t in
 *   tinyDA/proposal.py:223,229).
 */
#ifndef TINYDA_AMD_H
#define TINYDA_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct tda_engine tda_engine;

typedef enum tda_status {
  TDA_OK = 0,
  TDA_ERR_INVALID = -1,     /* bad argument / shape */
  TDA_ERR_HIP = -2,         /* a HIP runtime call failed */
  TDA_ERR_STATE = -3,       /* call order (e.g. run before init) */
  TDA_ERR_UNSUPPORTED = -4, /* configuration outside what the device engine lowers */
  TDA_ERR_NUMERIC = -5,     /* e.g. covariance not positive definite */
  TDA_ERR_CALLBACK = -6     /* a host forward-model callback reported failure */
} tda_status;

/* GaussianLogLike factory outcome, tinyDA/distributions.py:237-243 */
typedef enum tda_noise_kind {
  TDA_NOISE_ISO = 0,
  TDA_NOISE_DIAG = 1,
  TDA_NOISE_DENSE = 2,
  TDA_NOISE_ADAPTIVE = 3 /* AdaptiveGaussianLogLike (distributions.py:332-449): dense covariance + per-chain bias */
} tda_noise_kind;

/* adaptive_error_model of sample() (sampler.py:82-87) */
typedef enum tda_error_model {
  TDA_AEM_NONE = 0,
  TDA_AEM_STATE_INDEPENDENT = 1,
  TDA_AEM_STATE_DEPENDENT = 2,
  /* extension (the reference keeps a full m x m bias covariance per chain and re-inverts it every level step, which does
   * not scale): the state-independent model with DIAGONAL tracker covariances and a diagonal noise covariance -- the same
   * recursion element by element, corrected likelihood -1/2 sum_o (F_o + b_o - y_o)^2 / (sigma_o^2 + s_o^2); O(m) memory
   * and work per chain, any output dimension.  Levels are set with TDA_NOISE_ISO / TDA_NOISE_DIAG (their variances are
   * Sigma_e); Delayed Acceptance and MLDA; linear, callback and source-defined levels. */
  TDA_AEM_STATE_INDEPENDENT_DIAGONAL = 3
} tda_error_model;

/* tinyDA/proposal.py: GaussianRandomWalk :132, CrankNicolson :261, AdaptiveMetropolis :372, DREAMZ :608 / DREAM :1627 */
typedef enum tda_proposal_kind {
  TDA_PROP_GRW = 0, TDA_PROP_PCN = 1, TDA_PROP_AM = 2, TDA_PROP_DREAMZ = 3,
  TDA_PROP_INDEPENDENCE = 4, /* IndependenceSampler with a Gaussian q (proposal.py:65-129) */
  TDA_PROP_OWCN = 5,         /* OperatorWeightedCrankNicolson with fixed operators (proposal.py:515-605) */
  TDA_PROP_MALA = 6          /* MALA with the exact gradient of a linear-Gaussian posterior (proposal.py:861-1005): scaling = sigma */
} tda_proposal_kind;

typedef struct tda_config {
  uint32_t struct_size;  /* sizeof(tda_config) */
  int32_t device;        /* HIP device ordinal */
  int64_t n_chains;      /* chains held by this engine (rows of the state matrix) */
  int64_t chain_offset;  /* global id of local chain 0 */
  int32_t dim;           /* parameter dimension d: 1..64; 0.5: 65..128 for single-level chains and hierarchies of two to four levels (linear models, tda_engine_set_level_source and tda_engine_set_level_callback; error models: the dense one)
                          * with linear models, source-defined models or host callbacks, isotropic / diagonal noise, any prior the engine knows and TDA_PROP_GRW / TDA_PROP_PCN / TDA_PROP_AM --
                          * anything else at more than 64 parameters is refused by tda_engine_init with TDA_ERR_UNSUPPORTED */
  int32_t n_levels;      /* 1 = MH (sampler.py:213), 2 = Delayed Acceptance (:231), 3..6 = MLDA (:260); 0.5: 5 and 6 for the engine's own models without error model,
                          * dense observation covariance or DREAM(Z), dim <= 64 (0.4: at most 4) */
  uint64_t seed;
  void* stream;          /* hipStream_t to run on, or NULL for an engine-owned stream */
  int32_t block_steps;   /* max MH steps fused per launch group (0 = default 128) */
  int32_t reserved;
} tda_config;

/* Proposal constructor arguments, same meaning and defaults as the reference constructors
 * (proposal.py:171, :302, :416). */
typedef struct tda_proposal_params {
  uint32_t struct_size;
  int32_t kind;      /* tda_proposal_kind */
  double scaling;    /* GRW: scaling (default 1); pCN: beta (default 0.1); AM: ignored (1, proposal.py:462) */
  int32_t adaptive;  /* global scaling adaptation towards 0.24 acceptance (proposal.py:228-245) */
  int32_t period;    /* adaptation period (default 100) */
  double gamma;      /* adaptivity coefficient (default 1.01) */
  const double* C;   /* HOST d x d: GRW covariance / AM initial covariance C0; NULL for pCN (prior cov) */
  double sd;         /* AM scaling; <= 0 means min(1, 2.4^2/d) (proposal.py:465-468) */
  double epsilon;    /* AM regulariser (default 1e-6) */
  int32_t t0;        /* AM: first adapt() count at which C may be swapped (default 0) */
  int32_t block_moments; /* AM, extension (default 0).  0: the running covariance follows RecursiveSampleMoments.update
                          * (utils.py:113-124) operation for operation, so results do not depend on how a run is cut
                          * into run() calls and match the reference's traces to its own rounding.  1: one rank-S
                          * update per block of steps on the matrix cores -- algebraically identical, free of the
                          * reference form's cancellation, ~8x cheaper; deviates from the reference recursion by that
                          * recursion's rounding error (1e-11 .. 1e-8 relative in Sigma on ill-conditioned chains). */
  const double* q_mean; /* INDEPENDENCE: HOST [d] mean of q = N(q_mean, C); proposals are q_mean + chol(C) z, accepted with
                         * exp(post' - post + log q(theta) - log q(theta')); no adaptation.  NULL otherwise. */
} tda_proposal_params;

/* DREAMZ / DREAM constructor arguments (proposal.py:663-742, :1627-1641). */
typedef struct tda_dreamz_params {
  uint32_t struct_size;
  int32_t M0;        /* rows of the initial archive */
  int32_t delta;     /* archive pairs per jump (1..4) */
  int32_t nCR;       /* crossover values (1..8) */
  double b;          /* e ~ U(-b, b) */
  double b_star;     /* eps ~ N(0, b_star) */
  int32_t adaptive;  /* adapt global scaling and crossover probabilities every `period` */
  int32_t period;
  double gamma;
  int32_t shared;    /* 0: DREAMZ, one archive per chain; 1: DREAM, one archive for all chains (ray.py:365-384),
                        synchronised at block boundaries instead of racily on every proposal */
  int32_t sync_every; /* shared: steps between archive synchronisations (<= block_steps; 0 = block_steps) */
  int64_t capacity;  /* archive rows to reserve (M0 + steps [* chains when shared]) */
} tda_dreamz_params;

/* Per-step records of one run() call; any pointer may be NULL (that record is then not produced).
 * tda_engine_run takes an ARRAY of n_levels of these, coarsest level first.  Level k < n_levels-1 gets one
 * record per *local* step of that level (what sampler.py:421-427 / :535-538 return after compress(is_coarse /
 * is_local)): n_iterations * prod(subchain_lengths[k:]) rows; the finest level gets n_iterations rows.
 * Record r (0-based within this call) of chain c:
 *   params  [(r * n_chains + c) * dim + j]       chain state after the accept/reject decision
 *   stats   [(r * n_chains + c) * 3 + {0,1,2}]   log-prior, log-likelihood, log-posterior (link.py:41-48)
 *   accepted[ r * n_chains + c]                  1 if the proposal of that step was accepted */
typedef struct tda_outputs {
  uint32_t struct_size;
  uint32_t rows;     /* capacity of every non-NULL buffer below, in records (rows of n_chains entries).  run() returns
                      * TDA_ERR_INVALID before launching anything when the call would produce more records for this level
                      * than `rows` (it never writes past a caller's buffer); 0 is only valid when all three pointers are NULL */
  double* params;
  double* stats;
  uint8_t* accepted;
} tda_outputs;

/* Wall-clock-free device timing of the last run() (HIP events on the engine's stream). */
typedef struct tda_profile {
  uint32_t struct_size;
  uint32_t n_launch_propose, n_launch_steps, n_launch_adapt;
  double ms_propose, ms_steps, ms_adapt; /* summed kernel durations */
  double ms_total;                        /* first launch -> last completion */
  /* ABI 0.4 (a caller passing the 48-byte 0.3 struct gets the fields above only): the error-model refresh of host-sequenced
   * hierarchies (level decision + tracker update + re-inversion + update_link), which 0.3 counted under `adapt` */
  uint32_t n_launch_aem, reserved0;
  double ms_aem;
} tda_profile;

const char* tda_last_error(void);
/* "tinyda_amd <abi>.<minor> (...)".  ABI history: 0.1 round 1; 0.2 tda_outputs.reserved became `rows` (a caller built against 0.1
 * that passes 0 with non-NULL buffers is refused with TDA_ERR_INVALID); 0.3 tda_release_cached_memory, tda_engine_set_record_thinning,
 * tda_engine_set_progress / get_progress, tda_engine_detach_proposal_state + tda_proposal_snapshot_*,
 * tda_engine_set_proposal_spectrum; 0.4 tda_profile grew n_launch_aem / ms_aem (struct_size 48 is still accepted), checkpoint blobs
 * carry the ABI / RNG-contract version and older blobs are refused; 0.5 no entry point added or changed: tda_config.dim up to 128
 * (single-level chains and two-level hierarchies, see tda_config), the dense error model kept as the Cholesky factor instead of its triangular inverse (same
 * results through tda_engine_get_error_model; checkpoint blobs are format 4 -- six levels' counters -- and older blobs are refused). */
const char* tda_version(void);

/* Released engines park their large device buffers in a per-process pool (TINYDA_POOL_GB, default 8 GiB) so that the next
 * engine does not pay hipMalloc / hipFree again (tda.sample() creates one engine per call).  Returns the bytes handed back
 * to the driver. */
int64_t tda_release_cached_memory(void);

int tda_engine_create(const tda_config* cfg, tda_engine** out);
void tda_engine_destroy(tda_engine* e);

/* Prior = scipy.stats.multivariate_normal(mean, cov) (posterior.py:92; normalised logpdf). HOST pointers. */
int tda_engine_set_prior(tda_engine* e, const double* mean, const double* cov);

/* Level `level` posterior: forward model F(theta) = A theta + b (A is m x d, b may be NULL), data y,
 * Gaussian noise of the given kind: ISO noise[0] = variance; DIAG noise[m] = diagonal;
 * DENSE noise[m*m] = covariance (distributions.py:246-329). HOST pointers. */
int tda_engine_set_level(tda_engine* e, int level, int m, const double* A, const double* b,
                         const double* data, int noise_kind, const double* noise);

int tda_engine_set_proposal(tda_engine* e, const tda_proposal_params* p);
/* DREAMZ / DREAM (proposal.py:608-852, :1627-1656).  Single-level chains; DREAMZ (shared = 0) also as the base proposal of a
 * Delayed Acceptance / MLDA hierarchy (n_levels > 1; the configuration of the reference's MLDA notebook): every chain keeps its
 * own archive, which grows by the chain's level-0 state after every base step; the hierarchy is then sequenced by the host
 * (linear, callback and source-defined levels; iso / diag noise; diagonal prior).  capacity >= M0 + number of base steps. */
int tda_engine_set_proposal_dreamz(tda_engine* e, const tda_dreamz_params* p);
/* OperatorWeightedCrankNicolson (proposal.py:515-605; kind TDA_PROP_OWCN set by tda_engine_set_proposal with C = NULL, adaptive = 0):
 * theta' = state_operator theta + noise_operator N(0, C_prior), acceptance on the likelihood ratio like pCN.  Both operators
 * [dim][dim] row-major, what the reference computes in setup_proposal (:576-580: real(sqrtm(I - scaling B)), real(sqrtm(scaling B))).
 * Single-level chains with a linear forward model.  Call after set_proposal, before init. */
int tda_engine_set_proposal_operators(tda_engine* e, const double* state_operator, const double* noise_operator);
/* OperatorWeightedCrankNicolson with PER-CHAIN operators -- what adaptive=True needs (proposal.py:582-590: the operators are
 * recomputed from the chain's own scaling every period).  For a symmetric B = V diag(lambda) V^T they are functions of the
 * spectrum: sqrtm(I - s B) = V diag(sqrt(1 - s lambda)) V^T, sqrtm(s B) = V diag(sqrt(s lambda)) V^T (real parts).  V [dim][dim]
 * row-major with the eigenvectors in its columns, lambda [dim]; HOST pointers; after set_proposal(kind = TDA_PROP_OWCN, scaling,
 * adaptive, gamma, period), instead of set_proposal_operators.  Single level, linear forward model. */
int tda_engine_set_proposal_spectrum(tda_engine* e, const double* V, const double* lambda);

/* Initial archive Z (DREAMZ.setup_proposal, proposal.py:744-788): [n_chains][M0][dim] (per chain) or [M0][dim]
 * (shared).  NULL = draw the rows from the prior with RNG stream 2. Call after set_proposal_dreamz, before init. */
int tda_engine_set_archive(tda_engine* e, const double* Z0);

/* Forward model of the reference's Rosenbrock example (examples/MALA Rosenbrock.ipynb) generalised to d
 * parameters: F(theta) = [ sum_i (a - theta_i)^2 + b (theta_{i+1} - theta_i^2)^2 ], one observation `data`,
 * isotropic variance `noise_var`. */
int tda_engine_set_level_rosenbrock(tda_engine* e, int level, double a, double b, double data, double noise_var);

/* Multi-level schedule (sampler.py:260-264; chain.py:231-232): lengths[k], k = 0..n_levels-2, is the number of
 * level-k steps per step of level k+1.  randomize != 0 selects DAChain's randomize_subchain_length
 * (chain.py:310-321, 525-527; two levels only, needs lengths[0] > 1).  HOST pointer. */
int tda_engine_set_subchains(tda_engine* e, const int32_t* lengths, int randomize);

/* Adaptive error model (chain.py:268-305, 485-523; :643-678, 739-765): every level below the finest must have been set
 * with TDA_NOISE_ADAPTIVE (noise = m x m covariance), all levels share m <= 256 (dense model; 0.4: 128; the diagonal one: any m).  State-dependent
 * is two-level only. */
int tda_engine_set_error_model(tda_engine* e, int kind);
/* Error-model state of adaptive level `level` (HOST, any may be NULL): bias [n_chains][m], cov_inverse [n_chains][m][m]
 * (the diagonal model fills the diagonal of each matrix). */
int tda_engine_get_error_model(tda_engine* e, int level, double* bias, double* cov_inverse);

/* Start the chains: theta0 is n_chains x dim, or NULL to draw theta0 ~ prior from RNG stream 2
 * (sampler.py:209).  Evaluates the initial links (chain.py:70) and sets up the proposal
 * (chain.py:74-76, proposal.py:492-500). */
int tda_engine_init(tda_engine* e, const double* theta0);

/* Current state of every chain on the finest level: theta [n_chains*dim], stats [n_chains*3]; either may be NULL. */
int tda_engine_get_current(tda_engine* e, double* theta, double* stats);
/* Same for an explicit level. */
int tda_engine_get_level_state(tda_engine* e, int level, double* theta, double* stats);

/* Parity mode: consume caller-supplied variates instead of Philox.  z is [n_steps][n_chains][dim]
 * standard normals (mapped through chol(C) like tests/golden/gen_golden.py), u is [n_steps][n_chains]
 * uniforms; step index counts from the engine's current step.  NULL/0 switches replay off. */
int tda_engine_set_replay(tda_engine* e, const double* z, const double* u, int64_t n_steps);

/* Parity mode for levels >= 1: u is [n_steps_of_that_level][n_chains] (NaN where the reference drew none).
 * level == -1 sets the DA promoted index instead: values in [-L, -1] per fine iteration (chain.py:525-527). */
int tda_engine_set_replay_level(tda_engine* e, int level, const double* u, int64_t n_steps);

/* Parity mode for DREAMZ: everything DREAMZ.make_proposal draws, per step and chain, as recorded from the
 * reference (tests/golden/gen_golden.py g6_*): r [T][N][delta][2] archive rows, mcr [T][N] crossover index,
 * sub_u [T][N][d] subspace uniforms, forced [T][N] index used when the subspace is empty, e_u [T][N][d] uniforms
 * mapped to (-b, b), eps_n [T][N][d] standard normals, u [T][N] accept uniforms. */
int tda_engine_set_replay_dreamz(tda_engine* e, const int32_t* r, const int32_t* mcr, const double* sub_u,
                                 const int32_t* forced, const double* e_u, const double* eps_n, const double* u,
                                 int64_t n_steps);

/* Export mode: the engine writes the variates it generated into z / u (same layout as replay),
 * so the CPU oracle can be driven with the identical stream.  NULL/0 switches export off. */
int tda_engine_set_export(tda_engine* e, double* z, double* u, int64_t n_steps);

/* Advance every chain by n_iterations steps of the FINEST level (Chain.sample chain.py:95-125, DAChain.sample
 * :342-402, MLDAChain.sample :697-765); `out` points to n_levels tda_outputs (may be NULL).  Asynchronous on the
 * engine's stream when all outputs are device pointers; tda_engine_sync() waits. */
int tda_engine_run(tda_engine* e, int64_t n_iterations, const tda_outputs* out);
int tda_engine_sync(tda_engine* e);

/* Record thinning (single-level GRW / pCN / AM / MALA runs): of the iterations a run() advances, only those with
 * (t + 1) % thin == 0 -- t = iterations the engine had taken before that one, so the phase carries over split runs and
 * checkpoints -- reach the caller's buffers, packed: a run of n iterations from t produces (t + n) / thin - t / thin records
 * (integer divisions), and tda_outputs.rows is checked against that.  Adaptation still sees every iteration.  thin = 1: off.
 * (The reference keeps every Link in a Python list; SURVEY section 5 asks for a thinned mode for long histories.) */
int tda_engine_set_record_thinning(tda_engine* e, int32_t thin);

/* Progress without host synchronisation (the reference's tqdm bar, chain.py:96-99 / :343-351): when enabled, the end of every
 * block of a run() leaves the number of iterations completed so far (finest-level iterations for DA / MLDA) and the mean accept
 * flag of that block (-1 when the driver does not have it) in page-locked memory; get_progress only reads that memory -- it may
 * be called from another host thread while run() or sync() is blocking. */
int tda_engine_set_progress(tda_engine* e, int enable);
int tda_engine_get_progress(tda_engine* e, int64_t* iterations_done, int64_t* iterations_queued, double* accept_rate);

/* Proposal state, HOST pointers, any may be NULL: scaling[n_chains], C[n_chains*d*d] (covariance in use),
 * am_mu[n_chains*d], am_sigma[n_chains*d*d] (RecursiveSampleMoments), counters[2] = {t, k}. */
int tda_engine_get_proposal_state(tda_engine* e, double* scaling, double* C, double* am_mu,
                                  double* am_sigma, int64_t* counters);

/* DREAMZ state, HOST pointers, any may be NULL: pCR [n_chains][nCR], archive_rows[1]. */
/* The proposal state of an engine that has finished its work, taken over WITHOUT a copy (tda.sample() returns it lazily: the
 * 4096 per-chain covariances of BASELINE config 2 are 134 MB that most callers never look at).  detach moves the engine's
 * proposal buffers into the snapshot; the engine can only be destroyed afterwards.  snapshot_read takes the same host output
 * pointers (any of them NULL) as tda_engine_get_proposal_state. */
typedef struct tda_proposal_snapshot tda_proposal_snapshot;
int tda_engine_detach_proposal_state(tda_engine* e, tda_proposal_snapshot** out);
int tda_proposal_snapshot_read(tda_proposal_snapshot* s, double* scaling, double* C, double* am_mu, double* am_sigma, int64_t* counters);
void tda_proposal_snapshot_destroy(tda_proposal_snapshot* s);

int tda_engine_get_dreamz_state(tda_engine* e, double* pCR, int64_t* archive_rows);

/* Shared-archive exchange for one process per GPU (DREAM over RCCL): after run() with auto-append off, take the
 * rows this engine produced since the last exchange ([steps][n_chains][dim], device or host pointer; a device buffer is
 * filled asynchronously on the engine's stream, so an exchange can be queued behind it without a host wait) ... */
int tda_engine_archive_take(tda_engine* e, double* rows, int64_t* n_steps);
/* ... and append rows (any number, [n_rows][dim], canonical order = step-major, global chain id minor). */
int tda_engine_archive_append(tda_engine* e, const double* rows, int64_t n_rows);
int tda_engine_set_archive_auto_append(tda_engine* e, int on);

/* Distributed shared archive (extension; DREAM over several GPUs without replicating the archive).  The reference appends
 * every chain's state to one archive at every step (ray.py:365-384); replicating it costs every GPU the other ranks' rows over
 * xGMI at every step (2 MiB per GPU and step at 8192 chains x 32 parameters).  Here every rank keeps only the rows of its own
 * chains -- segment = [M0 shared initial rows][step][n_chains][dim padded] -- and a proposal reads the 2 delta rows it needs
 * from the owner's segment in place (peer-mapped memory): a quarter of the traffic, none of it on the critical path of a step.
 *   tda_engine_archive_ipc_handle   64-byte IPC handle of this engine's segment, to be sent to the other ranks
 *   tda_engine_archive_pointer      device address of the segment (what engines of ONE process hand to each other)
 *   tda_engine_set_archive_peers    once, right after init (chain count a multiple of 16, the same on every rank): `handles`
 *                                   = n_ranks x 64 bytes (entry my_rank ignored) or, for engines of one process, `pointers`
 *                                   = the segments' device addresses (the other may be NULL)
 *   tda_engine_run                  then covers at most one exchange interval (sync_every steps, not across an adaptation
 *                                   boundary) per call; its rows stay invisible until
 *   tda_engine_archive_publish      (publishes the OLDEST unpublished block) called after EVERY rank has finished that block (a
 *                                   barrier / the collective below).  At most two blocks may be unpublished: a block may run
 *                                   while its predecessor's collective is in flight (rows of block b visible from b + 2).
 *   tda_engine_archive_local_sums   [2][dim] host: column sums / sums of squares of this rank's visible rows not yet in the
 *                                   archive sums; the ranks add these up (an all-gather of 2 dim doubles, which also is the
 *                                   barrier) and hand the total to publish, which then runs a pending crossover adaptation.
 * Results equal the replicated archive's up to the rounding of those sums (they are added per rank, not per row). */
int tda_engine_archive_ipc_handle(tda_engine* e, void* handle);
int tda_engine_archive_pointer(tda_engine* e, void** pointer);
int tda_engine_set_archive_peers(tda_engine* e, int n_ranks, int my_rank, const void* handles, const double* const* pointers);
int tda_engine_archive_local_sums(tda_engine* e, double* sums);
int tda_engine_archive_publish(tda_engine* e, const double* sums_total);

/* Pooled AdaptiveMetropolis (extension; tinyDA's AM is strictly per chain, proposal.py:492-500): sums over the rows
 * of a record buffer, out = [n_rows, sum x (dim), sum x x^T (dim*dim)], `rows` a DEVICE pointer to [n_rows][dim],
 * `out` device or host.  One process per GPU all-reduces `out` over RCCL and hands the pooled covariance back with
 * tda_engine_set_proposal_covariance (GaussianRandomWalk engines; takes effect at the next run()). */
int tda_engine_reduce_moments(tda_engine* e, const double* rows, int64_t n_rows, double* out);
int tda_engine_set_proposal_covariance(tda_engine* e, const double* C);

/* Checkpoint / resume (the reference has none: sample() cannot continue a previous run).  The blob holds every chain's
 * state at every level, the proposal state (scaling, factors, running moments, archives, error-model trackers) and the
 * step / RNG counters; restoring it into an engine that was configured and init()-ed identically continues the run
 * bit for bit.  With a distributed archive (tda_engine_set_archive_peers) the blob holds this rank's segment and the position
 * of the publish protocol; the restoring process sets the peers up again before tda_engine_set_state.  HOST pointers. */
int64_t tda_engine_state_size(tda_engine* e);
int tda_engine_get_state(tda_engine* e, void* blob, int64_t bytes);
int tda_engine_set_state(tda_engine* e, const void* blob, int64_t bytes);

/* JointPrior of independent scalar components (distributions.py:8-100) instead of a multivariate normal: kind[j] = 0 is
 * scipy.stats.norm(loc[j], scale[j]), kind[j] = 1 scipy.stats.uniform(loc[j], scale[j]) (density 1/scale on
 * [loc, loc + scale], log-density -inf outside: such proposals are rejected).  HOST arrays [dim].  Single-level chains,
 * GRW / AM, iso / diag noise (also with tda_engine_set_level_source); explicit initial parameters. */
int tda_engine_set_prior_joint(tda_engine* e, const int32_t* kind, const double* loc, const double* scale);

/* Forward model given as HIP source (extension; the reference calls a Python callable per chain and step,
 * posterior.py:95-96).  The source must define
 *     __device__ double tda_forward(const double* theta, int dim, int o);     // output o of F(theta), o in [0, m)
 * and is compiled at run time (hiprtc, gfx950) into a fused step kernel: one wave per chain, the lanes stride over the
 * outputs.  data: HOST [m]; noise: ISO (noise[0] = variance), DIAG (HOST [m]) or DENSE (HOST [m][m], m <= 2048); below the finest level of a hierarchy also
 * TDA_NOISE_ADAPTIVE (HOST [m][m] covariance, m <= 128) for the adaptive error model.  Diagonal prior.  Single-level chains
 * (GRW / pCN / AM fused; DREAM(Z) step by step), and any level of a Delayed Acceptance / MLDA hierarchy: there the engine
 * sequences the levels from the host and a level step is propose -> tda_user_eval (compiled with the model) -> accept on
 * the stream; hierarchies may mix source-defined, callback and linear (iso / diag / adaptive noise) levels.
 * A source that does not compile returns TDA_ERR_INVALID with the compiler log in tda_last_error(). */
int tda_engine_set_level_source(tda_engine* e, int level, const char* source, int32_t m, const double* data,
                                int32_t noise_kind, const double* noise);

/* Forward model behind a batched host callback (extension; the reference calls a Python callable once per chain and step,
 * posterior.py:95-96, and umbridge.py:56-80 does the same over HTTP).  Once per step the engine hands the callback ALL
 * chains' proposals, theta: [n_chains][dim], and takes all model outputs back, F: [n_chains][m]; both are page-locked HOST
 * buffers owned by the engine and valid during the call only.  Proposals, log-densities, the accept test, adaptation and the
 * records stay on the device.  The callback returns 0, or non-zero to abort the run (tda_engine_init / tda_engine_run then
 * return TDA_ERR_CALLBACK).  It is called on the thread that calls init / run; it must not call into the same engine.
 * data: HOST [m]; noise: ISO (noise[0] = variance), DIAG (HOST [m]), DENSE (HOST [m][m], m <= 2048) or, below the finest level of a hierarchy,
 * TDA_NOISE_ADAPTIVE (HOST [m][m], m <= 128).  Diagonal prior (also tda_engine_set_prior_joint).  Single-level chains with
 * GRW / pCN / AM / DREAM(Z), and any level of a Delayed Acceptance / MLDA hierarchy (one call per level step for all chains;
 * error model of both kinds, adaptive scaling, randomised DA subchains as for linear levels). */
typedef int (*tda_forward_batch_fn)(void* user, const double* theta, double* F, int64_t n_chains, int32_t dim, int32_t m);
int tda_engine_set_level_callback(tda_engine* e, int level, tda_forward_batch_fn fn, void* user, int32_t m,
                                  const double* data, int32_t noise_kind, const double* noise);

/* Convergence diagnostics of a device-resident history (the reference hands its chains to ArviZ, diagnostics.py:6-111):
 * rank-normalised split bulk ESS and R-hat (Vehtari et al. 2021) of every parameter.  params: DEVICE
 * [n_steps][n_chains][dim] (layout of tda_outputs.params), the first `burnin` steps are dropped; ess, rhat: HOST [dim].
 * `stream` may be NULL. */
int tda_diag_ess_rhat(int device, void* stream, const double* params, int64_t n_steps, int64_t n_chains, int32_t dim,
                      int64_t burnin, double* ess, double* rhat);

/* Per-chain error flags (bit 0: Cholesky of an adapted covariance failed, previous factor kept). HOST. */
int tda_engine_get_flags(tda_engine* e, int32_t* flags);

/* Evaluate log-prior / log-likelihood of arbitrary states with the level-evaluation kernel only
 * (Posterior.create_link without the chain): theta [n][dim] -> stats [n][3].  n <= n_chains. */
int tda_engine_evaluate(tda_engine* e, int level, const double* theta, int64_t n, double* stats);

/* Device RNG probe: fills z [n_chains][dim] and u [n_chains] for the given step from streams 0 / 1. HOST. */
int tda_engine_rng_probe(tda_engine* e, int64_t step, double* z, double* u);

/* Raw generator probe: one Philox4x32-10 block computed ON THE DEVICE (device >= 0) or by the same source compiled for the
 * host (device = -1): out[4] = philox4x32_10(counter[4], key[2]).  Ties the engine's integer stream to the published
 * algorithm: tests compare it with the Random123 known-answer vectors (kat_vectors, philox4x32 10 rounds).  HOST pointers. */
int tda_rng_philox(int device, const uint32_t* counter, const uint32_t* key, uint32_t* out);

int tda_engine_set_profiling(tda_engine* e, int enable);
int tda_engine_get_profile(tda_engine* e, tda_profile* p);

#ifdef __cplusplus
}
#endif
#endif /* TINYDA_AMD_H */
