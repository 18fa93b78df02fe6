// solver_internal.h — problem / solver objects and the eval / pullback machinery shared by
// solver.hip (the reference's generic loops) and baselines.hip (its comparison methods).
#pragma once
#include <cstddef>
#include <vector>

#include "apx_internal.h"
#include "vecops.h"
#include "team_dev.h"

namespace apx {
struct SlabState;          // (slab.hip: f and grad f in one pass over A)
void slab_destroy(SlabState* st);
// what the generic loop hands to an evaluation of f so that the closing kernel of a one-pass launch can take the end of the
// iteration along (csrc/slab.hip SlabTail): the solver's n-vectors, g, the record sink
struct SlabTailReq {
  const void *x, *xprev, *gprev, *v;
  ProxDev g;
  apx_record* recs;
  int64_t rec_cap;
  // filled in by the evaluation that took the request: where the closing kernel left the wave partials of the reductions and
  // the snapshot of the control block — the step kernel's inputs (csrc/vecops.hip k_step_rule)
  const double* wred = nullptr;
  const Ctrl* snap = nullptr;
  int32_t ldw = 0;
};
}

struct apx_problem {
  apx_ctx* ctx = nullptr;
  int32_t f_kind = APX_F_ZERO, dtype = APX_F64;
  int64_t n = 0, m = 0, n_pad = 0, m_pad = 0;
  const apx_matrix* A_f = nullptr;
  const apx_matrix* A_lin = nullptr;
  const apx_matrix* held[2] = {nullptr, nullptr};      // the caller's handles (A_f / A_lin above may be their dense views): apx_matrix::users
  apx_eval_fn eval_cb = nullptr;
  apx_pullback_fn pb_cb = nullptr;
  void* cb_user = nullptr;
  apx_allreduce_fn allreduce = nullptr;   // the collective of a row-sharded problem: a host callback (tests), or
  void* ar_user = nullptr;
  apx_comm* comm = nullptr;               // an RCCL communicator the loop enqueues ncclAllReduce on (comm.hip)
  bool sharded() const { return allreduce != nullptr || comm != nullptr; }
  bool f_sharded = false, lin_sharded = false;   // which operand's rows the all-reduce hook sums over
  double n_mean = 0.0;
  apx::DBuf b, res, grad, fsc_buf, xtmp, ctrl_buf;
  apx::DBuf g_lb, g_ub, g_shift, h_lb, h_ub, h_shift;
  apx::ProxDev gd{}, hd{};
  std::vector<char> hbuf;
  bool have_eval = false;
  bool fsc_reduced = false;
  apx_solver* active = nullptr;
  // f and grad f in one pass over A (slab.hip): set up on the first evaluation of an eligible problem ; grad_ready: the
  // evaluation at hand already left the gradient in `grad`, the pullback has nothing to do
  apx::SlabState* slab = nullptr;
  bool slab_checked = false, grad_ready = false;
  // team_failed: a team hand-off failed on this problem (the GPU is shared with other work) ; teams_off: it runs on the two
  // GEMV passes from now on.  The one-shot entry points set the second when they see the first and run their call again
  // (with_team_fallback below) ; the stepping interface returns the error and sets it for the next solver.
  bool teams_off = false, team_failed = false;
  bool teams_dead() const { return teams_off || team_failed; }     // (no team launch on this problem any more)
  // the end of a proximal-gradient iteration riding in the closing kernel of the one-pass launch (csrc/slab.hip SlabTail): the
  // generic loop asks for it around its eval_f ; tail_done: that evaluation took it — k_reduce_n / k_rule are not launched
  apx::SlabTailReq* tail_req = nullptr;
  bool tail_done = false;
  ~apx_problem() { apx::slab_destroy(slab); }

  apx::Ctrl* ctrl() const { return (apx::Ctrl*)ctrl_buf.p; }
  const int* done_ptr() const { return (const int*)((const char*)ctrl_buf.p + offsetof(apx::Ctrl, done)); }
  bool fsc_packed = false;   // fp64 + all-reduce: the f scalars live at grad[fsc_off .. +1]
  int64_t fsc_off() const { return apx::round_up(n, apx::kPad); }   // >= n and >= A_f->n_pad: never a gradient slot
  double* fsc() const { return fsc_packed ? (double*)grad.p + fsc_off() : (double*)fsc_buf.p; }
  size_t es() const { return apx::dtype_size(dtype); }
};

namespace apx {

// ---- small.hip: the two-kernel iteration of small dense problems ---------------------------
constexpr int kSmallMax = 4096;    // largest n: x sits in static LDS and in per-thread register chunks
constexpr int kSmallMaxM = 8192;   // largest m: the m-vector (r or y) sits in dynamic LDS; measured break-even with the
                                   // generic kernels (beyond it the redundant O(m) elementwise work costs more than it saves)

struct SmallDev {
  const void* A;    // m_pad x ncols, column-major
  const void* AT;   // ncols_pad x m, column-major (= A')
  int64_t ld, ldT;
  int32_t m, ncols, n;   // n = ncols (+1: logistic bias)
  int32_t w_n, w_t;      // waves per dot product of A x / A' r
  const void* b;
  void *X, *XP, *GP, *V;      // state
  void *XN, *VN, *GN;         // staging (written by k_small_a, rotated in by k_small_b)
  double *axp, *grp;          // A x [m] and A' r [ncols], fp64
  Ctrl *ca, *cb;              // A: written by k_small_b ; B: written by k_small_a (the closing state)
  ProxDev g;
  apx_record* recs;
  int64_t rec_cap;
};

struct SmallState {
  DBuf xn, vn, gn, axp, grp, at, cb;
  SmallDev dev{};
};

// ---- small_pd.hip: the same for the primal-dual loop (f = Zero, dense A) ----------------------
struct PdDev {
  const void* A;    // m_pad x n, column-major
  const void* AT;   // n_pad x m (= A')
  int64_t ld, ldT;
  int32_t m, n, w_n, w_t;
  int32_t fk, _pad;          // APX_F_ZERO or APX_F_QUADRATIC
  const void* QT;            // Quadratic: Q' (n_pad x n) ; q
  int64_t ldQ;
  const void* qvec;
  const double *qx_prev, *qx_cur;   // Q x, rotating like A x
  double* qx_next;
  const void *x_prev, *x_cur;
  void* x_next;
  const void* v_cur;
  void* v_next;
  const void* y_cur;
  void* y_next;
  const double *ax_prev, *ax_cur;
  double* ax_next;
  const double* aty_cur;
  double* aty_next;
  const Ctrl* c_cur;
  Ctrl* c_next;
  ProxDev g, h;
  apx_record* recs;
  int64_t rec_cap;
};

struct SmallPdState {
  DBuf xb[3], axb[3], qxb[3];         // prev / cur / next, roles rotate with the iteration index
  DBuf vb[2], yb[2], atyb[2], cb[2];  // cur / next
  DBuf at, qt;
  PdDev base{};                       // the loop-invariant fields
};

// ---- skinny_pd.hip: primal-dual loop for tall, skinny dense A (n <= 120, f = Zero) --------------
constexpr int kSkinnyMaxN = 124;     // n + 4 scalar partial columns <= 128 (a5a: 123 features + the bias)

struct SkinnyDev {
  const void* A;          // m_pad x n, column-major
  int64_t ld;
  int32_t m, n, npow, _pad;
  int64_t ldpart;
  void *X, *XP, *V;       // x, x_prev, v  (n, written by k_skinny_close only)
  double* aty;            // A'y (n)
  void* Y;                // y (m), updated in place by the row kernel
  double *ax, *axn, *wbuf;   // A_x of the previous iteration (m) ; l2 scratch: A_x and w of this one
  double* part;           // [blocks][ldpart]: n partials of A'y+, ||dual_res||^2, h(A_x), ||w/sigma + shift||^2
  Ctrl *c, *cs;           // control block ; staging written by block 0 of the row kernel
  ProxDev g, h;
  apx_record* recs;
  int64_t rec_cap;
};

struct SkinnyState {
  DBuf x, xp, v, aty, y, ax, axn, wbuf, part, cs;
  SkinnyDev dev{};
  int B = 0;              // row blocks
};

// ---- skinny_pg.hip: proximal-gradient loop for tall, skinny dense A (LinearLeastSquares / LogisticLoss) --------
struct SkinnyPgDev {       // what the closing kernel of skinny_pg.hip needs (the row kernel: skinny_rows_dev.h)
  int32_t ncols, n, npow, _pad;
  int64_t ldpart;
  void *X, *XP, *GP, *V;  // the solver's n-vectors, written by k_spg_close only
  const double* part;     // [blocks][ldpart]: ncols partials of A' res, ||res||^2 or the loss sum, sum(res)
  Ctrl* c;
  ProxDev g;
  apx_record* recs;
  int64_t rec_cap;
};

struct SkinnyPgState;      // (skinny_pg.hip: holds the row plan)
void skinny_pg_destroy(SkinnyPgState* st);
struct SlabPdState;        // (slab_pd.hip: A x, the dual step and A' y in one pass over A)
void slab_pd_destroy(SlabPdState* st);

// ---- small_mom.hip: fixed Nesterov and aGRAAL as device-resident loops (small dense problems) ---------
enum { MOM_AGRAAL = 0, MOM_NESTEROV = 1 };

struct MomCtrl {
  double tol, n_mean;
  int64_t maxit, rec_cap;
  // aGRAAL (:172-178)
  double gamma, theta, rho, phi, gamma_max;
  // Nesterov (:108-128)
  double q, mu, muf, mug;
  double bias[3];                   // logistic: mean(probs - y) belonging to gradient buffer 0 / 1 / 2
  double norm_res, g_raw, f_new, objective;
  int64_t it;                       // finished iterations
  int32_t need_gamma0, stop_after, done, flags;
  int32_t frozen, phase;            // frozen: nothing left to do ; phase (skinny_mom.hip): what the point in XT is
};

}  // namespace apx

struct apx_solver {
  apx_problem* p = nullptr;
  apx_rule rule{};
  double tol = 0.0;
  int32_t poll_every = 16;
  apx::DBuf x, xprev, gprev, v, aty, y, ax, axprev, recs;
  // f = Zero with a linear operator: the step kernel leaves the next iteration's reduction partials here (vecops.hip k_step_red)
  apx::DBuf red_own;
  bool red_ready = false;
  int64_t rec_cap = 0;
  apx_record* host_recs = nullptr;
  int64_t iters = 0;        // iterations enqueued so far
  bool stopped = false;     // the host has seen done == 1
  bool poisoned = false;    // a team hand-off failed under this solver: its iterate is invalid, every later call says so
  bool use_graph = false;
  hipGraph_t graph = nullptr;
  hipGraphExec_t graph_exec = nullptr;
  apx::SmallState* small = nullptr;   // non-null: the iteration runs as small.hip's two kernels
  apx::SmallPdState* small_pd = nullptr;   // non-null: ... as small_pd.hip's two kernels
  apx::SkinnyState* skinny = nullptr;      // non-null: ... as skinny_pd.hip's row + closing kernels
  apx::SkinnyPgState* skinny_pg = nullptr; // non-null: ... as skinny_pg.hip's row + closing kernels
  apx::SlabPdState* slab_pd = nullptr;     // non-null: ... and A x, the dual step, A' y from slab_pd.hip
  ~apx_solver() {
    delete small;
    delete small_pd;
    delete skinny;
    apx::skinny_pg_destroy(skinny_pg);
    apx::slab_pd_destroy(slab_pd);
  }
};

namespace apx {

// A one-shot entry point whose run ended in a failed team hand-off runs again, from the same starting point, with the
// problem taken off the team kernels — and says so (APX_FLAG_TEAM_FALLBACK, apx_last_incident()).  On a row shard every rank
// sees the failure at the same point (slab_check sums the error word over the ranks), so all of them come through here
// together and their collectives stay matched.
template <class F>
inline int32_t with_team_fallback(apx_problem* p, apx_result* res, F&& run) {
  int32_t rc = run();
  if (rc == APX_ERR_HIP && p != nullptr && p->team_failed && !p->teams_off) {
    p->teams_off = true;
    rc = run();
    if (rc == APX_OK && res != nullptr) res->flags |= APX_FLAG_TEAM_FALLBACK;
  }
  return rc;
}

// A team launch the runtime refused (a cooperative launch — APX_OPT_TEAM_COOPERATIVE — whose grid cannot be co-resident): nothing
// ran, the launch and sequence counters stayed where they were (team_launch), and the problem is taken off the team kernels
// like after a failed hand-off, so the one-shot entry points run again on the two passes instead of asking again.
inline int32_t team_launch_refused(apx_problem* p, const char* which) {
  const hipError_t e = hipGetLastError();        // (clears the sticky launch error)
  p->team_failed = true;
  const std::string what = std::string(which) + ": the launch was refused (" + (e != hipSuccess ? hipGetErrorString(e) : "no HIP error on record") +
                           ") — a cooperative launch whose grid cannot be co-resident on this device. Nothing ran ; the problem now runs on "
                           "the two GEMV passes (apx_ctx_one_pass(ctx, 2) keeps every new problem off the team kernels).";
  note_incident(what);
  return fail(APX_ERR_HIP, what);
}

// the problem's all-reduce (its communicator, or the callback) on the loop's stream: sums `count` elements over the ranks in place
int32_t problem_allreduce(apx_problem* p, void* buf, int64_t count, int32_t dtype);
int32_t h2d(apx_ctx* ctx, void* dst, const void* src, size_t bytes);
int32_t d2h(apx_ctx* ctx, void* dst, const void* src, size_t bytes);
// `f_x, pb = eval_with_pullback(f, x)` on a device x (need_f: f_x final in Ctrl on return to the stream)
int32_t eval_f(apx_problem* p, const void* x_dev, bool need_f);
// `pb()`: gradient at the x of the latest eval, written to p->grad
int32_t pullback_f(apx_problem* p);
int32_t init_ctrl(apx_problem* p, const apx_rule* rule, double tol);
int32_t read_ctrl(apx_problem* p, Ctrl* out);
// A_x = A_lin * x  into ax (T, length m_pad)
int32_t mul_A_lin(apx_problem* p, const void* x_dev, void* ax);
// matrix.hip: dT (ldT x A->ld, zero-initialised by the caller) = A'
int32_t transpose_into(apx_ctx* ctx, const apx_matrix* A, void* dT, int64_t ldT);
// small.hip
bool small_eligible(const apx_problem* p);
int32_t small_setup(apx_solver* s);      // after the generic pre-loop of apx_solver_create
int32_t small_enqueue(apx_solver* s);    // one iteration
// small_pd.hip
bool small_pd_eligible(const apx_problem* p);
int32_t small_pd_setup(apx_solver* s, const void* x0, const void* y0);
int32_t small_pd_enqueue(apx_solver* s);
const void* small_pd_ctrl(const apx_solver* s);
// skinny_pg.hip
bool slab_pd_eligible(const apx_problem* p);
int32_t slab_pd_setup(apx_solver* s);
int32_t slab_pd_setup_voted(apx_solver* s);      // ... on a row shard: a failure is a vote, not a return
int32_t slab_pd_iteration(apx_solver* s, bool* done_here);
bool slab_pd_trial_ok(const apx_solver* s);
int32_t slab_pd_trial(apx_solver* s, const void* x, const void* y, const void* axprev, void* ynext, void* axnew, void* aty_next,
                      double sigma, double rho);
int32_t slab_pd_check(apx_solver* s);
int32_t slab_pd_team_waits(apx_solver* s, int64_t out[4]);
int32_t slab_pd_team_wait_slots(apx_solver* s, const TeamWait** slots);
bool slab_pd_uses_teams(const apx_solver* s);
bool slab_uses_teams(const apx_problem* p);   // (team kernels: host-side launch counters — no hipGraph replay)      // the team kernel's error word (a timed-out hand-off): APX_ERR_HIP
// slab.hip: true when f and grad f at x_dev were formed in one pass over A (f -> Ctrl, grad -> p->grad)
int32_t slab_eval_with_gradient(apx_problem* p, const void* x_dev, bool* done_here);
int32_t slab_check(apx_problem* p);        // the team kernel's error word (a timed-out hand-off): APX_ERR_HIP
int32_t slab_tail_after_allreduce(apx_problem* p, bool* done);
int32_t slab_team_waits(apx_problem* p, int64_t out[4]);
int32_t slab_team_wait_slots(apx_problem* p, const TeamWait** slots);
// sum of a scalar over the ranks of a row-sharded problem (the value itself otherwise): what the ranks agree on paths with
int32_t ranks_sum(apx_problem* p, double local, double* total);
bool skinny_pg_eligible(const apx_problem* p);
int32_t skinny_pg_setup(apx_solver* s);
int32_t skinny_pg_enqueue(apx_solver* s);
// skinny_pd.hip
bool skinny_pd_eligible(const apx_problem* p);
int32_t skinny_pd_setup(apx_solver* s, const void* x0, const void* y0);
int32_t skinny_pd_enqueue(apx_solver* s);
// small_mom.hip (the caller fills the method's parameters into c0)
bool skinny_mom_eligible(const apx_problem* p);    // skinny_mom.hip: the same on tall, skinny A (n <= 124)
int32_t skinny_mom_run(apx_problem* p, int method, const MomCtrl& c0, const void* x1, const void* x0, int64_t maxit,
                       const apx_solve_opts* opts, void* x_out, apx_result* res);
bool small_mom_eligible(const apx_problem* p);
int32_t small_mom_run(apx_problem* p, int method, const MomCtrl& c0, const void* x1, const void* x0, int64_t maxit,
                      const apx_solve_opts* opts, void* x_out, apx_result* res);
// small_bt.hip: backtracking_proxgrad with its linesearch as a device-side state machine (small dense problems)
bool skinny_bt_eligible(const apx_problem* p);     // skinny_bt.hip: the same on tall, skinny A (n <= 124)
int32_t skinny_bt_run(apx_problem* p, bool nesterov, const void* x0, double gamma0, double xi, double shrink, double tol,
                      int64_t maxit, const apx_solve_opts* opts, void* x_out, apx_result* res);
bool small_bt_eligible(const apx_problem* p);
int32_t small_bt_run(apx_problem* p, bool nesterov, const void* x0, double gamma0, double xi, double shrink, double tol,
                     int64_t maxit, const apx_solve_opts* opts, void* x_out, apx_result* res);
// skinny_mp.hip: Malitsky-Pock with its linesearch as a device-side state machine
bool small_mpq_eligible(const apx_problem* p);     // small_mpq.hip: Malitsky-Pock for f = Quadratic, few rows of A (dual SVM)
int32_t small_mpq_run(apx_problem* p, const void* x0, const void* y0, double sigma, double t, double tol, int64_t maxit,
                      const apx_solve_opts* opts, void* x_out, void* y_out, apx_result* res);
bool skinny_mp_eligible(const apx_problem* p);
int32_t skinny_mp_run(apx_problem* p, const void* x0, const void* y0, double sigma, double t, double tol, int64_t maxit,
                      const apx_solve_opts* opts, void* x_out, void* y_out, apx_result* res);
// skinny_ls.hip: AdaPDM+ with the linesearch as a device-side state machine
bool skinny_ls_eligible(const apx_problem* p);
int32_t skinny_ls_run(apx_problem* p, const apx_linesearch_params* prm, double gamma, const void* x0, const void* y0,
                      double tol, int64_t maxit, const apx_solve_opts* opts, void* x_out, void* y_out, apx_result* res);
void small_pd_result_ptrs(const apx_solver* s, const Ctrl& c, const void** x, const void** y);

}  // namespace apx
