/*
 * adaprox_hip.h — C-ABI of the MI355X-native adaptive proximal-gradient hot path.
 *
 * Drop-in boundary for AdaProx.jl (pylat/adaptive-proximal-algorithms).  The
 * reference has no FFI of its own: its boundary is Julia multiple dispatch on
 * `eval_with_pullback(f, x)` / `prox(g, v, gamma)` / `A*x` / `A'*y`
 * (src/AdaProx.jl:11-16, src/counting.jl:36-82) and the keyword-argument solver
 * entry points (src/AdaProx.jl:50, :312-322, :367-381, :418, :457).  Each entry
 * point below names the reference interface it replaces.  Julia reaches these
 * with `ccall` (see INTEGRATION.md and julia/AdaProxHIP.jl); tests and bench.py
 * reach them with Python `ctypes`.
 *
 * Conventions
 *  - every function returns an int32 status (APX_OK = 0); apx_last_error() gives
 *    the message of the last failure on the calling thread.  Nothing throws
 *    across the boundary.
 *  - host pointers are plain caller-owned buffers, valid for the duration of the
 *    call only.  Device memory lives behind the opaque handles.
 *  - matrices are column-major (Julia layout).  dtype selects fp64 / fp32 storage.
 *  - a context is bound to one device and one HIP stream and is NOT thread-safe
 *    (the reference is single-threaded too: src/counting.jl:4-14); different contexts may be
 *    driven from different threads.
 *  - the destroy / free entry points are safe to call from finalizers that run while the process
 *    exits (Julia runs them at exit): once exit() has reached this library's handlers they return
 *    APX_OK without touching the HIP runtime, which may be gone by then.
 */
#ifndef ADAPROX_HIP_H
#define ADAPROX_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define APX_ABI_VERSION 5   /* 2: apx_problem_desc.comm, apx_comm_*, APX_ERR_COMM
                               3: apx_ctx_one_pass, apx_ctx_set_option / get_option, APX_KSTAT_ONE_PASS + apx_ctx_kernel_name,
                                  APX_FLAG_TEAM_FALLBACK, apx_last_incident, apx_problem_team_waits ; apx_ctx_fused_small no
                                  longer switches the one-pass kernels ; no environment variables are read any more
                               4: apx_problem_team_wait_slots (additive: everything of version 3 is unchanged)
                               5: one resident copy of A — APX_LAYOUT_ROWS, apx_matrix_create_layout / upload_layout,
                                  apx_matrix_drop_primary, apx_matrix_layout ; APX_OPT_FAULT_SETUP ; apx_ctx_device_clock (additive) */

/* ---- status codes -------------------------------------------------------- */
enum {
  APX_OK = 0,
  APX_ERR_INVALID = 1,      /* bad argument (Julia: `error(...)` / `@assert`)         */
  APX_ERR_HIP = 2,          /* HIP runtime failure (no device, OOM, launch failure)   */
  APX_ERR_UNSUPPORTED = 3,  /* combination not built (e.g. callback f + hipGraph)     */
  APX_ERR_DOMAIN = 4,       /* Julia would have thrown DomainError (sqrt of negative) */
  APX_ERR_CALLBACK = 5,     /* a user callback returned non-zero                      */
  APX_ERR_COMM = 6          /* RCCL reported an error (apx_comm_*, or a collective of the loop) */
};

/* warning bits reported in apx_result.flags (the loop keeps going, like the reference) */
enum {
  APX_FLAG_TINY_GAMMA = 1,  /* backtracking gamma < 1e-12: `@error` logged, src/AdaProx.jl:40-42 */
  APX_FLAG_NAN_GAMMA = 2,   /* the rule produced a NaN step (Julia `min` propagates NaN)         */
  APX_FLAG_DOMAIN = 4,      /* sqrt of a negative number inside a rule (Julia: DomainError)      */
  APX_FLAG_TEAM_FALLBACK = 8 /* (no reference counterpart) a hand-off of the one-pass TEAM kernels failed during this call: the
                                result returned is that of a second run of the same call, from the same starting point, on the
                                two GEMV passes ; what the failure looked like: apx_last_incident()                            */
};

enum { APX_F64 = 0, APX_F32 = 1 };

/* smooth term f — `eval_with_pullback(f, x)` implementations of the experiments */
enum {
  APX_F_ZERO = 0,           /* experiments/least_absolute_deviation/runme.jl:18-21 */
  APX_F_LEAST_SQUARES = 1,  /* LinearLeastSquares, experiments/lasso/runme.jl:16-27 */
  APX_F_LOGISTIC = 2,       /* LogisticLoss, experiments/sparse_logreg/runme.jl:18-39 (bias = last coord) */
  APX_F_CALLBACK = 3,       /* any user f: host callbacks (the f callable convention)  */
  APX_F_QUADRATIC = 4       /* Quadratic(Q, q): 0.5 x'Qx + q'x, experiments/dual_svm/runme.jl:19-28 (A_f = Q, b = q) */
};

/* nonsmooth terms g / h — ProximalOperators.jl / ProximalCore.jl objects */
enum {
  APX_G_ZERO = 0,           /* ProximalCore.Zero      : prox = identity                */
  APX_G_L1 = 1,             /* NormL1(lambda)         : soft-threshold                 */
  APX_G_BOX = 2,            /* IndBox(lb, ub) scalars : clamp                          */
  APX_G_BOXV = 3,           /* IndBox(lb[], ub[])     : per-coordinate clamp           */
  APX_G_INDZERO = 4,        /* ProximalCore.IndZero   : prox = 0                       */
  APX_G_L2 = 5              /* NormL2(lambda)         : block soft-threshold           */
};

/* step-size rules — src/AdaProx.jl:208-308 */
enum {
  APX_RULE_FIXED = 0,       /* FixedStepsize(gamma, t)            :208-215 */
  APX_RULE_MM = 1,          /* MalitskyMishchenkoRule(gamma, t)   :217-230 */
  APX_RULE_OURS = 2,        /* OurRule(gamma,t,norm_A,delta,Theta):232-273 */
  APX_RULE_OURS_PLUS = 3    /* OurRulePlus(gamma,xi,nu,r)         :277-308 */
};

typedef struct apx_ctx apx_ctx;
typedef struct apx_matrix apx_matrix;
typedef struct apx_problem apx_problem;
typedef struct apx_solver apx_solver;
typedef struct apx_comm apx_comm;       /* an RCCL communicator bound to one context (row-sharded multi-GPU path) */

/* A proximable term: kind + parameters.  `shift` (length = vector length, host,
 * may be NULL) wraps the term in Translate(., shift): x -> g(x + shift)
 * (ProximalOperators.Translate; LAD uses Translate(NormL1(), -y),
 * experiments/least_absolute_deviation/runme.jl:41). */
typedef struct {
  int32_t kind;          /* APX_G_*                                   */
  int32_t _pad;
  double lambda;         /* NormL1 / NormL2 weight                    */
  double lb, ub;         /* IndBox scalar bounds                      */
  const void* lb_vec;    /* IndBox vector bounds (host, dtype of problem), APX_G_BOXV */
  const void* ub_vec;
  const void* shift;     /* Translate shift b (host, dtype of problem) or NULL */
} apx_prox_desc;

/* host callbacks for a user-defined smooth term (APX_F_CALLBACK):
 *   eval(user, x, n, &f_x)   — `f_x, pb = eval_with_pullback(f, x)`; must retain what pb needs
 *   pullback(user, grad, n)  — `pb()`: gradient at the x of the latest eval
 * x / grad are host buffers of the problem's dtype.  Return 0 on success. */
typedef int32_t (*apx_eval_fn)(void* user, const void* x, int64_t n, double* f_x);
typedef int32_t (*apx_pullback_fn)(void* user, void* grad, int64_t n);

/* all-reduce hook for the row-sharded multi-GPU path: sums `count` elements of
 * dtype `dtype` at DEVICE pointer `buf` across ranks, enqueued on `stream`
 * (hipStream_t).  The product path is apx_problem_desc.comm (RCCL inside the library, below);
 * this callback is the TEST transport (gloo on CPU / two ranks sharing one GPU, where RCCL refuses).
 * NULL = single device or comm. */
typedef int32_t (*apx_allreduce_fn)(void* user, void* buf, int64_t count, int32_t dtype, void* stream);

typedef struct {
  int32_t f_kind;          /* APX_F_*                                                        */
  int32_t dtype;           /* APX_F64 / APX_F32 — storage of all vectors of this problem     */
  int64_t n;               /* primal dimension (logistic: columns of X + 1 for the bias)     */
  int64_t m;               /* dual dimension = rows of A_lin (0 when there is no A_lin)      */
  const apx_matrix* A_f;   /* matrix of the smooth term (A of least squares, X of logistic)  */
  const void* b;           /* host: b (least squares) or labels y in {0,1} (logistic); length rows(A_f) */
  int64_t n_total_rows;    /* logistic: N of the mean (global rows when row-sharded); 0 => rows(A_f)    */
  apx_eval_fn eval_cb;     /* APX_F_CALLBACK                                                 */
  apx_pullback_fn pullback_cb;
  void* cb_user;
  apx_prox_desc g;         /* g : prox_{gamma g}                                              */
  apx_prox_desc h;         /* h : the loop uses prox of convex_conjugate(h), src/AdaProx.jl:325 */
  const apx_matrix* A_lin; /* linear operator A of h(Ax); NULL <=> `A = 0` (adaptive_proxgrad, :419) */
  apx_allreduce_fn allreduce;  /* row-sharded A_f: all-reduce of [A_k' r_k ; partial f]        */
  void* allreduce_user;
  int32_t shard_mask;      /* with a hook: bit 0 = rows of A_f are sharded, bit 1 = rows of A_lin are sharded
                              (y, A_x and the dual residual become shards; A'y and ||dual_res||^2 are all-reduced);
                              0 means bit 0                                                                       */
  int32_t _pad2;
  apx_comm* comm;          /* row-sharded problem: the communicator whose ranks hold the other row blocks; the loop
                              enqueues ncclAllReduce on the context's stream where the callback above would be called
                              (shard_mask as above).  The problem keeps it alive: apx_comm_destroy on a communicator
                              that still has problems is deferred to the last of them.  NULL = single device / callback. */
} apx_problem_desc;

typedef struct {
  int32_t kind;            /* APX_RULE_*                      */
  int32_t _pad;
  double gamma;            /* all rules                        */
  double t;                /* Fixed / MM / Ours                */
  double norm_A;           /* Ours                             */
  double delta;            /* Ours                             */
  double Theta;            /* Ours                             */
  double xi, nu, r;        /* OursPlus                         */
} apx_rule;

/* One `@logmsg Record` row (src/AdaProx.jl:56, :351).  The *_evals fields of the
 * reference record are functions of `it` and are rebuilt host-side. */
typedef struct {
  int64_t it;
  double gamma;
  double sigma;            /* NaN for backtracking_proxgrad (no sigma field at :56) */
  double norm_res;
  double objective;        /* f_x + g(x) + h(A_x)  (:351)   or  f_z + g_z (:56)     */
  double f_x;
  int64_t f_evals;         /* cumulative eval_with_pullback calls at this record     */
  int64_t grad_evals;      /* cumulative pullback calls at this record               */
} apx_record;

typedef struct {
  int64_t numit;           /* the `it` / `numit` the reference returns                */
  int32_t converged;       /* 1: returned through `norm_res <= tol`, 0: maxit         */
  int32_t flags;           /* APX_FLAG_* warning bits                                 */
  double gamma, sigma, norm_res, objective;   /* last iteration                       */
  /* Counting semantics (src/counting.jl:16-23) */
  int64_t f_evals, grad_f_evals, prox_g_evals, prox_h_evals, A_evals, At_evals;
  int64_t n_records;       /* rows written to the caller's record buffer              */
} apx_result;

typedef struct {
  apx_record* records;     /* caller buffer for per-iteration records, or NULL        */
  int64_t record_cap;      /* rows available; the first record_cap iterations are kept */
  int32_t use_graph;       /* 1: replay the iteration as a captured hipGraph (device-only
                              iterations: no callback f, no all-reduce hook); bit-identical
                              results, fewer launch gaps on launch-bound (small) problems;
                              2: the library's choice (currently eager: replay measured no
                              faster on MI355X / ROCm 7.2, tools/small_bench.py); 0: eager */
  int32_t poll_every;      /* convergence flag is read back every this many iterations (0 = default 16) */
} apx_solve_opts;

/* kernel timing collected with HIP events on the context's stream (bench.py roofline) */
typedef struct {
  int64_t launches;
  double total_ms;
  double bytes;            /* algorithmic bytes moved by those launches (m*n*sizeof(T) each) */
} apx_kernel_stat;

/* timing slots: the two GEMV passes, and the ONE-pass launches (row kernel + its closing kernel: f and grad f, or A x, the
 * dual step and A' y, from one read of A — csrc/slab.hip, csrc/slab_pd.hip, csrc/team_rows.h) */
enum { APX_KSTAT_GEMV_N = 0, APX_KSTAT_GEMV_T = 1, APX_KSTAT_ONE_PASS = 2, APX_KSTAT_COUNT = 3 };

/* ---- library / context ----------------------------------------------------- */
int32_t apx_abi_version(void);
const char* apx_last_error(void);
/* stream: an existing hipStream_t to adopt (e.g. torch's current stream) or NULL to create one */
int32_t apx_ctx_create(int32_t device, void* stream, apx_ctx** out);
int32_t apx_ctx_destroy(apx_ctx* ctx);
int32_t apx_ctx_sync(apx_ctx* ctx);
int32_t apx_ctx_stream(apx_ctx* ctx, void** stream_out);
/* enable/disable per-launch HIP-event timing of the GEMV kernels; read / reset the totals */
int32_t apx_ctx_profile(apx_ctx* ctx, int32_t enable);
int32_t apx_ctx_kernel_stats(apx_ctx* ctx, apx_kernel_stat* stats /* [APX_KSTAT_COUNT] */, int32_t reset);
/* the kernels last booked under a timing slot, as the library launched them — e.g.
 * "k_team_rows<least_squares, 16 members, 5 vectors, 4 slots, 2 rows of slack> + k_slab_close" ; "" before the first launch */
const char* apx_ctx_kernel_name(apx_ctx* ctx, int32_t slot);

/* tuning hooks (no reference counterpart): pick one of the compiled GEMV variants and the
 * target grid size; negative / zero arguments leave a field unchanged.  Defaults are the
 * measured-best configuration on MI355X. */
int32_t apx_ctx_set_gemv_config(apx_ctx* ctx, int32_t n_variant, int32_t t_variant, int64_t n_blocks,
                                int64_t t_blocks);
/* Opt-in (default off): `A*x` on a dense column-major A streams only the columns with x_j != 0 —
 * what reference BLAS dgemv does too (`IF (X(JX).NE.ZERO)`), and what a lasso iterate invites
 * after the soft-threshold.  Same result up to summation order; not used by bench.py's headline. */
int32_t apx_ctx_exploit_sparsity(apx_ctx* ctx, int32_t enable);
/* Default ON: execution policy for the reference's own small experiment sizes (experiments/lasso/runme.jl:191-195,
 * least_absolute_deviation/runme.jl:128-160, dual_svm/runme.jl:140-180) — no all-reduce hook, dense matrices:
 *   - n <= 4096, m <= 4096 (m <= 8192 with m*n <= 4 M): every iteration of src/AdaProx.jl:334-362 as two kernels in
 *     which each workgroup redundantly does the O(m + n) work (csrc/small.hip: proximal gradient with
 *     LinearLeastSquares / LogisticLoss ; csrc/small_pd.hip: primal-dual with f = Zero or Quadratic);
 *   - tall, skinny A (n <= 124, f = Zero): a thread per row and a one-block closing kernel (csrc/skinny_pd.hip);
 *   - tall, skinny A in the proximal-gradient loop (n <= 124, m >= 2048, LinearLeastSquares / LogisticLoss):
 *     a row kernel whose waves split the columns and a one-block closing kernel (csrc/skinny_pg.hip);
 *   - the linesearch methods as state machines in device memory, no read-back per trial: backtracking_proxgrad,
 *     backtracking_nesterov (csrc/small_bt.hip), adaptive_linesearch_primal_dual, malitsky_pock on tall-skinny A
 *     (csrc/skinny_ls.hip, csrc/skinny_mp.hip); fixed_nesterov and agraal likewise (csrc/small_mom.hip); all four
 *     proximal-gradient comparison methods also on tall, skinny A (csrc/skinny_bt.hip, csrc/skinny_mom.hip);
 *     malitsky_pock with f = Quadratic and at most 64 rows of A — the dual SVM — in csrc/small_mpq.hip.
 *   - a problem built on a SPARSE matrix that is small as a dense one (m x n within the limits above, <= 256 MB:
 *     the three LIBSVM datasets of experiments/sparse_logreg/runme.jl:177-204) runs on a dense copy of it, which the
 *     matrix keeps (products through apx_gemv_n/t stay on the compressed storage).
 * Same arithmetic, different summation order than the generic path; 0 switches back to the generic kernels and
 * the host-driven loops.  (ABI 2 also switched the one-pass kernels of large matrices with this call: they have their own
 * switch now, apx_ctx_one_pass.) */
int32_t apx_ctx_fused_small(apx_ctx* ctx, int32_t enable);
/* Default 1: execution policy for LARGE dense matrices (>= apx_ctx_one_pass_threshold bytes, rows of 128 ... 180 224 entries):
 * ONE pass over A per iteration instead of the two GEMV passes of the reference's cost model (experiments/lasso/runme.jl:179-180).
 *   - LinearLeastSquares / LogisticLoss: f and grad f from one read of A (csrc/slab.hip) ; the primal-dual loop with a
 *     separable h: A x, the dual step and A' y from one read (csrc/slab_pd.hip).  The kernels read the transposed copy of A,
 *     which a matrix of up to 16 GB is given when a solver is created on it (apx_ctx_auto_transpose).
 *   - rows short enough for one workgroup's registers (Float64 <= 11 264 entries, Float32 <= 22 528) need nothing else.
 *   - rows beyond that (BASELINE config C2's 100 000, C5's 20 000 ; Float64 up to 180 224 entries, Float32 up to 225 280 — the
 *     primal-dual loop alike since round 5) are split over TEAMS of 2 / 4 / 8 / 16 workgroups that
 *     hand their partial dot products over through the XCD's L2 (csrc/team_rows.h).  The teams need the part's 8 XCDs x 32 CUs
 *     to themselves: all 256 workgroups of a launch resident together.  That the kernel FITS (one workgroup per CU) is asked of
 *     the runtime when the problem is set up ; that the CUs are FREE cannot be known in advance, so every hand-off wait is
 *     bounded (APX_OPT_TEAM_WAIT_MS, default 1000 ms, and 4096 polls), a failure is sticky (later launches return at once),
 *     leaves a diagnostic record (in the error message, and in apx_last_incident()) and takes the PROBLEM off the team
 *     kernels for good.  The one-shot entry points (apx_adaptive_primal_dual and its wrappers, apx_backtracking_proxgrad,
 *     apx_backtracking_nesterov, apx_fixed_nesterov, apx_agraal) then run the same call again on the two GEMV passes and set
 *     APX_FLAG_TEAM_FALLBACK ; apx_eval re-evaluates ; the stepping interface returns APX_ERR_HIP (its solver's iterate is
 *     invalid: destroy it — a new solver on the same problem takes the two passes).  A row-sharded problem behaves the same on
 *     every rank: the error word is summed over the ranks wherever it is read, so all ranks see a failure of any of them at the
 *     same iteration and fall back together.  Team launches cannot be replayed as a hipGraph (APX_ERR_UNSUPPORTED).
 *   mode 0: off — the two GEMV passes ; 1: on ; 2: on, but never the team kernels (a GPU shared with other work). */
int32_t apx_ctx_one_pass(apx_ctx* ctx, int32_t mode);
/* Options by key (no reference counterpart).  A/B switches of measured alternatives and the hooks the tests use — the library
 * reads NO environment variables. */
enum {
  APX_OPT_MULTI_ROWS_MASK = 1,   /* default 15: short rows (Float64 <= 5632 entries, Float32 <= 11 264) on k_multi_rows — bit 0
                                    least squares, 1 logistic, 2 the primal-dual loop, 3 Float32 data at all ; a cleared bit
                                    runs the first-generation row shapes (A/B)                                               */
  APX_OPT_SLAB_GENERATION = 2,   /* default 0 = by measurement ; 1 / 2 force the first / second generation of the
                                    single-workgroup kernel for rows of 5121 ... 10 752 entries (A/B)                       */
  APX_OPT_TEAM_WAIT_MS = 3,      /* default 1000: wall-clock bound of a team hand-off wait                                  */
  APX_OPT_TEAM_COOPERATIVE = 4,  /* default 0 ; 1: team kernels through hipLaunchCooperativeKernel (the runtime refuses a grid
                                    that cannot be co-resident ; measured no faster and not needed: DESIGN.md section 9) ; not
                                    honoured on a row-sharded problem (a refusal is one rank's, the collectives are all ranks') */
  APX_OPT_TEAM_FAULT_GRID = 5,   /* TEST hook, default 0: launch this many of a team launch's 256 workgroups, so that teams
                                    miss members (tests/test_gpu_team_fault.py) ; 257 ... 512 (with APX_OPT_TEAM_COOPERATIVE
                                    only): a grid that cannot be co-resident, which the runtime refuses to launch          */
  APX_OPT_GEMV_N_VARIANT = 6, APX_OPT_GEMV_T_VARIANT = 7, APX_OPT_GEMV_N_BLOCKS = 8, APX_OPT_GEMV_T_BLOCKS = 9,
                                 /* = apx_ctx_set_gemv_config, one field at a time (tools/gemv_sweep.py)                    */
  APX_OPT_FUSED_TAIL = 10,       /* default 1: the residual / step-size reductions and the rule ride in the closing kernel of a
                                    one-pass launch (two kernels between consecutive passes over A instead of four) ; 0: the
                                    separate kernels (A/B ; bit-identical results)                                          */
  APX_OPT_MULTI_WAVES = 11,      /* default 0 = by fit ; 10 / 11: compute waves a row (k_multi_rows) or a member's slice of it
                                    (k_team_rows) is dealt to (A/B)                                                          */
  APX_OPT_TEAM_MEMBERS = 12,     /* default 0 = by the row length ; 2 / 4 / 8 / 16: workgroups per team where the row allows (A/B) */
  APX_OPT_TEAM_SLOTS = 13,       /* default 0 ; 52 / 62 / 53 / 63: row slots x 10 + rows of slack, teams of four with four vectors
                                    per thread in the primal-dual loop only (C5's length as teams of four: A/B) ; 42 / 31: teams of
                                    two in the primal-dual loop with five (six) vectors keep the four-slot shape / take the
                                    three-slot one (A/B of round 4's default)                                                */
  APX_OPT_ROW_DEAL = 14,         /* default -1: the rows of A are dealt to the workgroups (teams) of a one-pass launch in CHUNKS of
                                    consecutive rows (about 4 MB), the chunks cyclically — the chip streams one window of the
                                    matrix at a time, whatever the allocation ; 0: one slab of consecutive rows each (round 2) ;
                                    a power of two: that many rows per chunk (A/B: profiles/r03_row_deal.md)                */
  APX_OPT_TEAM_INTERLEAVE = 15   /* default 1: the members of a team of FOUR OR MORE take a row's 10 KB pieces in turn (with every
                                    load instruction a team covers one contiguous stretch of the row) ; 2: teams of two as well ;
                                    0: each member its contiguous P-th of the row (round 2 ; A/B: profiles/r03_ab_interleave.jsonl) */
  ,
  APX_OPT_TEAM_PD_DEFER = 16     /* default 1: the communication wave of the primal-dual team kernels keeps a row's terms of the
                                    residual norm and of h(A x), and member 0's stores of y+ and A x, in the lane of the row's
                                    number and works them off sixteen rows at a time, lane-parallel — off the path between two
                                    hand-offs ; 0: row by row behind each publication (round 3 ; A/B)                          */
  ,
  APX_OPT_FAULT_SETUP = 17       /* TEST hook, default 0 ; bit 0: the set-up of the one-pass primal-dual kernels fails as if an
                                    allocation had (on ONE rank of a row shard: the ranks must still agree on the two passes —
                                    tests/two_rank_pd_worker.py)                                                              */
  ,
  APX_OPT_SPARSE_WINDOWS = 18    /* default 2: the round-5 forms of the sparse products, decided when a sparse matrix is uploaded on this
                                    context.  1: (a) an orientation whose gathered operand (x for A x: n entries ; r for A' r: m entries)
                                    exceeds 3 MB is stored WINDOW-MAJOR and its product runs window by window of the operand (2 MB: half
                                    an XCD's L2 ; tools/gather_probe.hip: 2.0 against 1.1 TB/s of the entries' bytes) ; (b) the
                                    STREAMING form: blocks of rows whose entries are loaded as one run, products through LDS, eight lanes
                                    per row.  2: besides, LDS TILES where they pay (1 M entries or more, 4096 or more per tile of 8192
                                    rows x 64 KB of operand): the operand's window staged in LDS, every gather an LDS read (the probe:
                                    5.4 - 5.9 TB/s), the rows' sums in LDS across the windows ; an orientation that has tiles keeps its
                                    plain rows next to them.  3: TEST hook — tiles on every matrix.  0: round 1's lanes-per-row kernel
                                    over whole rows (A/B).                                                                          */
  ,
  APX_OPT_ONE_PASS_MIN_ROWS = 19 /* TEST hook, default 4096: the fewest rows a matrix needs for the one-pass kernels (below, a launch's
                                    fixed costs outweigh the second pass) ; 256 ... 4096 — the parity tests of the widest rows run on
                                    matrices of ~1000 rows, so that the CPU oracle beside them takes a second instead of five       */
};
int32_t apx_ctx_set_option(apx_ctx* ctx, int32_t key, int64_t value);
int32_t apx_ctx_get_option(apx_ctx* ctx, int32_t key, int64_t* value);
/* the diagnostic text of the last team-kernel failure on the calling thread — also when the call recovered by itself
 * (APX_FLAG_TEAM_FALLBACK) and apx_last_error() was never set ; "" if there was none */
const char* apx_last_incident(void);
/* Tuning hook: the matrix size in bytes from which the one-pass kernels (csrc/slab.hip, csrc/slab_pd.hip) replace the two
 * GEMV passes.  Default 192 MB — the measured break-even on MI355X; the parity tests lower it to reach the kernels
 * with small matrices. */
int32_t apx_ctx_one_pass_threshold(apx_ctx* ctx, int64_t min_bytes);
/* Default ON: a dense matrix of up to 16 GB that a one-pass-eligible problem is evaluated on is given its transposed
 * copy on the spot (as by apx_matrix_build_transpose: twice the memory of A, held until apx_matrix_free /
 * apx_matrix_drop_transpose / an in-place edit).  0: the library never allocates it by itself — the one-pass kernels
 * then run only where the caller built the copy. */
int32_t apx_ctx_auto_transpose(apx_ctx* ctx, int32_t enable);
int32_t apx_gemv_variant_count(int32_t transposed);
const char* apx_gemv_variant_name(int32_t transposed, int32_t i);

/* ---- RCCL communicators: the row-sharded multi-GPU path ---------------------------------
 * No reference counterpart (the reference is one Julia process on CPU).  A = [A_1; ...; A_P] by rows, rank k holds
 * A_k and b_k; x, the prox and the step-size state are replicated; what crosses ranks is the gradient the pullback
 * returns — `A' * res` of experiments/lasso/runme.jl:23, consumed at src/AdaProx.jl:336 — plus the two partial f
 * scalars in the same message: ONE ncclAllReduce(sum, n + 2 elements) per gradient, enqueued by the native loop on
 * its own stream (with a sharded A_lin also A'y and the dual residual, src/AdaProx.jl:344-358).  The caller of
 * experiments/lasso/runme.jl:141-149 only swaps its matrix for a row block and passes the communicator.
 *
 *  - one process per GPU: rank 0 obtains an id (ncclGetUniqueId), the launcher (torch.distributed.run, MPI, a shared
 *    file) hands its APX_COMM_ID_BYTES bytes to every rank, each rank calls apx_comm_init_rank (ncclCommInitRank) on
 *    its own context;
 *  - one process, several GPUs (a Julia session): one context per device, apx_comm_init_all (ncclCommInitAll), then
 *    apx_adaptive_primal_dual_group, which runs every device's loop on its own host thread inside the call. */
#define APX_COMM_ID_BYTES 128
int32_t apx_comm_unique_id(void* id_out /* [APX_COMM_ID_BYTES] */);
int32_t apx_comm_init_rank(apx_ctx* ctx, int32_t nranks, int32_t rank, const void* id, apx_comm** out);
int32_t apx_comm_init_all(apx_ctx* const* ctxs, int32_t ndev, apx_comm** comms_out /* [ndev] */);
/* ncclCommCount / ncclCommUserRank of the live communicator */
int32_t apx_comm_info(const apx_comm* comm, int32_t* nranks, int32_t* rank);
/* sum of a host fp64 vector over the ranks, in place (set-up code: power iteration for Lf, norms; not per iteration) */
int32_t apx_comm_allreduce_host(apx_comm* comm, double* buf, int64_t count);
int32_t apx_comm_destroy(apx_comm* comm);
/* path of the librccl the communicators run on — bound at the first apx_comm_* call, not when this library is loaded:
 * the copy already in the process if there is one, else the one next to the libamdhip64 in use (csrc/comm.hip);
 * "" when it could not be loaded (every apx_comm_* call then fails with APX_ERR_COMM) */
const char* apx_comm_library(void);
/* adaptive_primal_dual (src/AdaProx.jl:312-364) on ndev row shards held by ONE process: ps[i] lives on device i's context
 * and carries comm i of apx_comm_init_all; x0 is replicated, y0s[i] / y_outs[i] are shard i of y (NULL arrays when there
 * is no linear operator); x_out, res and opts->records are rank 0's (all ranks take identical steps).  With ndev = 1 and
 * no communicator it is apx_adaptive_primal_dual. */
/* (A failure on ONE device before its loop's first collective leaves the others waiting in theirs: arguments are checked on
 * the calling thread first, so this can only be a device-side failure such as an allocation.) */
int32_t apx_adaptive_primal_dual_group(int32_t ndev, apx_problem* const* ps, const apx_rule* rule, const void* x0,
                                       const void* const* y0s, double tol, int64_t maxit, const apx_solve_opts* opts,
                                       void* x_out, void* const* y_outs, apx_result* res);

/* ---- dense matrices (the `A` / `X` the experiments build) ------------------ */
int32_t apx_matrix_create(apx_ctx* ctx, int64_t m, int64_t n, int32_t dtype, apx_matrix** out); /* zero-filled */
/* column-major host matrix with leading dimension lda (elements) -> device */
int32_t apx_matrix_upload(apx_ctx* ctx, const void* host, int64_t m, int64_t n, int64_t lda,
                          int32_t dtype, apx_matrix** out);
/* sparse design matrix — the `SparseMatrixCSC` the experiments load from LIBSVM files
 * (experiments/libsvm.jl:3-61; used as `f.X * w`, `f.X' * r` at experiments/sparse_logreg/runme.jl:24,31).
 * layout 0: CSR (ptr has m+1 entries, idx = column indices); layout 1: CSC (ptr has n+1 entries, idx = row
 * indices — Julia's colptr/rowval minus 1).  0-based, int64 pointers, int32 indices.  Usable wherever a
 * dense matrix is (A_f, A_lin, apx_gemv_n/t, apx_matrix_sumsq); element-wise helpers are dense-only. */
int32_t apx_matrix_upload_sparse(apx_ctx* ctx, int64_t m, int64_t n, int64_t nnz, int32_t layout, const int64_t* ptr,
                                 const int32_t* idx, const void* vals, int32_t dtype, apx_matrix** out);
int32_t apx_matrix_download(const apx_matrix* A, void* host, int64_t lda);
/* Storage of a dense matrix (no reference counterpart: a Julia `Matrix` is column-major, and so is every `host` pointer here).
 *   APX_LAYOUT_COLS  column-major in HBM as on the host — what apx_matrix_create / apx_matrix_upload give.  The one-pass
 *                    kernels (apx_ctx_one_pass) read a second, ROW-contiguous copy next to it (apx_matrix_build_transpose, or
 *                    built for matrices up to 16 GB when a solver first needs it): two copies resident for a loop that reads one.
 *   APX_LAYOUT_ROWS  ONLY the row-contiguous storage (rows-only): one resident copy — C2's 50 000 x 100 000 fp64 matrix is
 *                    40 GB instead of 80, and a single GPU holds a matrix of ~280 GB instead of ~140.  Everything works on it:
 *                    the one-pass kernels as before (bit-identical iterates), `A*x` / `A'*y` of the two-pass fallback and of
 *                    apx_gemv_n/t by the same kernel pair with the two dimensions' roles swapped, the element-wise helpers and
 *                    upload / download (transposing through a 256 MB staging buffer).  The small-size kernel families
 *                    (apx_ctx_fused_small) read column-major storage and do not apply: this layout is for matrices that fill HBM.
 * apx_matrix_drop_primary turns a column-major matrix into a rows-only one (building the row copy first if it is not there:
 * both are resident for that moment ; APX_ERR_INVALID while problems created on the matrix exist — a solver of theirs may be
 * reading the column-major storage) ; apx_matrix_layout reports bit 0 = column-major storage present, bit 1 = row storage
 * present (0 for a sparse matrix). */
enum { APX_LAYOUT_COLS = 0, APX_LAYOUT_ROWS = 1 };
int32_t apx_matrix_create_layout(apx_ctx* ctx, int64_t m, int64_t n, int32_t dtype, int32_t layout, apx_matrix** out);
int32_t apx_matrix_upload_layout(apx_ctx* ctx, const void* host, int64_t m, int64_t n, int64_t lda, int32_t dtype, int32_t layout,
                                 apx_matrix** out);
int32_t apx_matrix_drop_primary(apx_matrix* A);
int32_t apx_matrix_layout(const apx_matrix* A, int32_t* mask);
/* Optional dual layout (no reference counterpart): keeps a second, transposed copy of a dense A in
 * HBM so that `A*x` streams whole rows contiguously, like `A'*r` streams whole columns (+3-5 % on
 * the A*x pass for 2x the matrix memory: 80 GB of the 288 GB at 50 000 x 100 000 fp64).  Any later
 * in-place edit of A (fill / scale) drops the copy.  (A rows-only matrix: nothing to build, APX_OK.) */
int32_t apx_matrix_build_transpose(apx_matrix* A);
/* releases that second copy (A itself is untouched); problems on A fall back to the two GEMV passes.  APX_ERR_INVALID on a
 * rows-only matrix (there it is the matrix). */
int32_t apx_matrix_drop_transpose(apx_matrix* A);
int32_t apx_matrix_free(apx_matrix* A);
int32_t apx_matrix_shape(const apx_matrix* A, int64_t* m, int64_t* n, int32_t* dtype);
/* element (i,j) = lo + (hi-lo)*u(seed, j*m_total + row_offset + i); counter-based, so a
 * row shard of a global m_total x n matrix is generated independently on each GPU
 * (stands in for `rand(m, n) .* 2 .- 1`, experiments/lasso/runme.jl:49) */
int32_t apx_matrix_fill_uniform(apx_matrix* A, uint64_t seed, double lo, double hi,
                                int64_t row_offset, int64_t m_total);
/* A[:, j] *= alpha[j]   (`C * diagm(0 => alpha)`, experiments/lasso/runme.jl:68) */
int32_t apx_matrix_scale_columns(apx_matrix* A, const double* alpha);
/* A[:, j] = value       (`hcat(X, ones(m,1))`, experiments/least_absolute_deviation/runme.jl:42) */
int32_t apx_matrix_fill_column(apx_matrix* A, int64_t j, double value);
/* sum of squares of all entries: norm(A)^2, Frobenius (runme.jl:46; Appendix A.1) */
int32_t apx_matrix_sumsq(const apx_matrix* A, double* out);
/* `A*x` and `A'*y` (src/counting.jl:68-82): host vectors in, host vectors out */
int32_t apx_gemv_n(const apx_matrix* A, const void* x, void* y);
int32_t apx_gemv_t(const apx_matrix* A, const void* r, void* g);

/* ---- problem = (f, g, h, A) ------------------------------------------------ */
int32_t apx_problem_create(apx_ctx* ctx, const apx_problem_desc* desc, apx_problem** out);
int32_t apx_problem_destroy(apx_problem* p);
/* `f_x, pb = eval_with_pullback(f, x)` (src/AdaProx.jl:11): keeps the residual on device */
int32_t apx_eval(apx_problem* p, const void* x, double* f_x);
/* `pb()`: gradient at the x of the latest apx_eval (one GEMV-T on the saved residual) */
int32_t apx_pullback(apx_problem* p, void* grad);
/* Statistics of the team hand-offs of this problem's one-pass kernels since it was set up: out[0] = hand-offs that were
 * not complete at the first look, out[1] = the most polls one of them took, out[2] = the longest one in nanoseconds ;
 * out[3] = team launches ; all 0 for a problem that runs no team kernels.  (tools/soak.py: how close normal operation comes to
 * the bound of APX_OPT_TEAM_WAIT_MS.) */
int32_t apx_problem_team_waits(apx_problem* p, int64_t* out /* [4] */);
/* The same statistics per WORKGROUP SLOT of the team launches, with where and when each slot saw its longest wait (ABI 4): up to
 * `cap` records of the slots that waited at all, *count = how many there are.  Many slots with their longest wait at the same
 * launch and clock: the whole queue was not running (every polling wave sees the gap) ; only the partners of one member: that
 * member's workgroup was late.  tools/soak.py prints the clusters ; DESIGN.md section 9. */
typedef struct apx_team_wait_slot {
  int64_t waits;          /* hand-offs that were not complete at the first look                              */
  int64_t long_waits;     /* ... of them, longer than 100 us                                                 */
  int64_t max_polls;      /* the most polls one of them took                                                  */
  int64_t max_wait_ns;    /* the longest one                                                                  */
  int64_t at_launch;      /* the problem's team launch in which the longest one happened (0-based)            */
  int64_t at_step;        /* ... the row of the team's deal it waited for                                     */
  int64_t at_clock_ns;    /* ... the device's wall clock (s_memrealtime) when it ended                        */
  int32_t block;          /* the slot: blockIdx.x                                                             */
  int32_t team, member;   /* what that workgroup was in that launch (teams form at run time, by XCD)          */
  int32_t xcc;            /* ... and the XCD it ran on                                                        */
  int32_t kind;           /* 0: f's one-pass kernel, 1: the primal-dual loop's                                */
  int32_t _pad;
} apx_team_wait_slot;
int32_t apx_problem_team_wait_slots(apx_problem* p, apx_team_wait_slot* out, int32_t cap, int32_t* count);
/* The device's wall clock (s_memrealtime, in ns — what apx_team_wait_slot.at_clock_ns is counted in) and the host's
 * CLOCK_REALTIME (ns since the epoch) for the same instant, to within the latency of a one-thread kernel and a stream
 * synchronisation (the host time is the midpoint of the two host readings around the device's ; *uncertainty_ns half their
 * distance): lines the moments a team kernel recorded up against what the host did (tools/pause_probe.py).  ABI 5. */
int32_t apx_ctx_device_clock(apx_ctx* ctx, int64_t* device_ns, int64_t* host_realtime_ns, int64_t* uncertainty_ns);
/* `y, g_y = prox(g, v, gamma)` (ProximalCore.prox); conj != 0 applies convex_conjugate(desc) */
int32_t apx_prox(apx_ctx* ctx, const apx_prox_desc* desc, int32_t conj, int32_t dtype,
                 const void* v, int64_t n, double gamma, void* y, double* g_y);

/* `g(x)` — the callable value of a proximable term, used by the logged objective
 * `f_x + g(x) + h(A_x)` (src/AdaProx.jl:351) */
int32_t apx_prox_value(apx_ctx* ctx, const apx_prox_desc* desc, int32_t dtype, const void* x, int64_t n,
                       double* value);

/* ---- solvers --------------------------------------------------------------- */
/* adaptive_primal_dual(x, y; f, g, h, A, rule, tol, maxit)  — src/AdaProx.jl:312-364 */
int32_t apx_adaptive_primal_dual(apx_problem* p, const apx_rule* rule, const void* x0, const void* y0,
                                 double tol, int64_t maxit, const apx_solve_opts* opts,
                                 void* x_out, void* y_out, apx_result* res);
/* adaptive_proxgrad(x; f, g, rule, tol, maxit)              — src/AdaProx.jl:418-421 */
int32_t apx_adaptive_proxgrad(apx_problem* p, const apx_rule* rule, const void* x0,
                              double tol, int64_t maxit, const apx_solve_opts* opts,
                              void* x_out, apx_result* res);
/* fixed_proxgrad(x; f, g, gamma, tol, maxit)                — src/AdaProx.jl:457-459 */
int32_t apx_fixed_proxgrad(apx_problem* p, double gamma, const void* x0,
                           double tol, int64_t maxit, const apx_solve_opts* opts,
                           void* x_out, apx_result* res);
/* condat_vu(x, y; f, g, h, A, Lf, norm_A, tol, maxit), default-parameter branch — src/AdaProx.jl:367-416.
 * norm_A < 0 => computed as the Frobenius norm of A_lin (`norm(A)`, :403). */
int32_t apx_condat_vu(apx_problem* p, double Lf, double norm_A, const void* x0, const void* y0,
                      double tol, int64_t maxit, const apx_solve_opts* opts,
                      void* x_out, void* y_out, apx_result* res);
/* backtracking_proxgrad(x0; f, g, gamma0, xi, shrink, tol, maxit) — src/AdaProx.jl:34-64 */
int32_t apx_backtracking_proxgrad(apx_problem* p, const void* x0, double gamma0, double xi, double shrink,
                                  double tol, int64_t maxit, const apx_solve_opts* opts,
                                  void* x_out, apx_result* res);

/* adaptive_linesearch_primal_dual(x, y; f, g, h, A, gamma, eta, t, delta, Theta, r, R, tol, maxit)
 * "AdaPDM+" — src/AdaProx.jl:463-550.  gamma <= 0 means `nothing` (=> 1/(2 Theta t eta), :484-486).
 * max_linesearch bounds the reference's unbounded inner `while true` (:516); 0 => 10000.
 * The `@assert`s of :481-488 come back as APX_ERR_INVALID. */
typedef struct {
  double gamma, eta, t, delta, Theta, r, R;
  int64_t max_linesearch;
} apx_linesearch_params;
int32_t apx_adaptive_linesearch_primal_dual(apx_problem* p, const apx_linesearch_params* prm, const void* x0,
                                            const void* y0, double tol, int64_t maxit, const apx_solve_opts* opts,
                                            void* x_out, void* y_out, apx_result* res);

/* ---- comparison baselines of the experiments -------------------- */
/* backtracking_nesterov(x0; f, g, gamma0, shrink, tol, maxit) — src/AdaProx.jl:66-84 */
int32_t apx_backtracking_nesterov(apx_problem* p, const void* x0, double gamma0, double shrink, double tol,
                                  int64_t maxit, const apx_solve_opts* opts, void* x_out, apx_result* res);
/* fixed_nesterov(x0; f, g, Lf | gamma, muf, mug, theta, tol, maxit) — src/AdaProx.jl:91-142.
 * Pass gamma (= 1/Lf when the caller has Lf); theta < 0 means `nothing`. */
int32_t apx_fixed_nesterov(apx_problem* p, const void* x0, double gamma, double muf, double mug, double theta,
                           double tol, int64_t maxit, const apx_solve_opts* opts, void* x_out, apx_result* res);
/* agraal(x1; f, g, x0, gamma0, gamma_max, phi, tol, maxit) — src/AdaProx.jl:150-192.
 * x0 is required (the host draws x1 + randn for `nothing`); gamma0 <= 0 means `nothing`. */
int32_t apx_agraal(apx_problem* p, const void* x1, const void* x0, double gamma0, double gamma_max, double phi,
                   double tol, int64_t maxit, const apx_solve_opts* opts, void* x_out, apx_result* res);
/* malitsky_pock(x, y; f, g, h, A, sigma, t, tol, maxit) — src/AdaProx.jl:555-629 */
int32_t apx_malitsky_pock(apx_problem* p, const void* x0, const void* y0, double sigma, double t, double tol,
                          int64_t maxit, const apx_solve_opts* opts, void* x_out, void* y_out, apx_result* res);

/* ---- stepping interface (bench.py times exactly K iterations with data resident) ---- */
/* runs the pre-loop of adaptive_primal_dual (src/AdaProx.jl:324-332) and leaves the state on device */
int32_t apx_solver_create(apx_problem* p, const apx_rule* rule, const void* x0, const void* y0,
                          double tol, const apx_solve_opts* opts, apx_solver** out);
/* enqueues `iters` iterations of :334-362 on the stream; does not wait unless sync != 0 */
int32_t apx_solver_step(apx_solver* s, int64_t iters, int32_t sync);
int32_t apx_solver_result(apx_solver* s, void* x_out, void* y_out, apx_result* res);
int32_t apx_solver_destroy(apx_solver* s);

#ifdef __cplusplus
}
#endif
#endif /* ADAPROX_HIP_H */
