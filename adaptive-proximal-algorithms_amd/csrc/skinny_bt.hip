// skinny_bt.hip — `backtracking_proxgrad` (src/AdaProx.jl:34-64) and `backtracking_nesterov` (:66-84) as DEVICE-
// RESIDENT loops for tall, skinny dense A (at most 124 unknowns, 2048 rows or more): the shape of all three
// datasets of the reference's logistic experiment, which runs these methods five times per dataset
// (experiments/sparse_logreg/runme.jl:66-124).  The host-driven forms read f_z back per linesearch trial (45 - 80 us
// per iteration); here one trial is one TICK of the two launches of skinny_pg.hip:
//
//   k_spg_rows  (skinny_rows_dev.h) at the point XT: block partials of A' res, of the loss, of sum(res)
//   k_sbt_close (one block): f and grad f at XT from the partials ; the test f_z > ub_z (:38, :26) ;
//               rejected: gamma *= shrink (:39) ; accepted: norm_res, record, termination (:55-59 / :73-77),
//               x = z, f_x = f_z, grad_x = the gradient this very pass produced (:60-61) — Nesterov: theta, the
//               extrapolated point (:78-80), whose evaluation (:81) is the next tick ; then the next trial point
//               z = prox_{gamma g}(x - gamma grad_x) (:35 / :43) with <grad_x, z - x>, ||z - x||^2, g(z) -> XT
//
// With n <= 124 the whole n side lives in one workgroup, so x, z and the gradients are plain n-vectors that the
// closing block copies (no buffer flipping), and the control block has one writer.  The host enqueues ticks in
// chunks and reads the counters between chunks.
#include <cstdlib>
#include <cstring>
#include <limits>
#include <memory>

#include "skinny_rows_dev.h"

namespace apx {

struct SbtCtrl {
  double xi, shrink, tol, n_mean;
  int64_t maxit, rec_cap;
  double gamma, gamma_try, f_x;
  double s_dot, s_ss, s_graw;       // <grad_x, z - x>, ||z - x||^2, raw g(z) of the trial in XT
  double norm_res, objective;
  int64_t it, trials_total;
  int32_t done, flags;
  int32_t nesterov;                 // backtracking_nesterov: trial ticks and extrapolation ticks alternate
  int32_t phase;                    // Nesterov: 0 = XT is a trial z, 1 = XT is the extrapolated x
  int32_t halt, _pad;               // done != 0 || it >= maxit: what the row kernel polls
  double theta, beta;               // :78-80
};

struct SbtDev {
  int32_t ncols, n, npow, _pad;
  int64_t ldpart;
  const double* part;
  void* XT;            // the point the row kernel evaluates: a trial z, or Nesterov's extrapolated x
  void* X;             // x the trials start from (Nesterov: the extrapolated x)
  void* Z;             // Nesterov: z of the last accepted trial (:71 z_prev = z)
  double* GX;          // grad f(X), entries rounded to T
  SbtCtrl* c;
  ProxDev g;
  apx_record* recs;
};

constexpr int kSbtBlock = 1024;

// MODE 0: pre-loop (:52 / :69): f(x0), grad f(x0), the first trial ; MODE 1: close a tick
template <typename T, int FK, int MODE>
__global__ __launch_bounds__(kSbtBlock) void k_sbt_close(SbtDev d, int B) {
  __shared__ SkinnyRed red[16];
  __shared__ SbtCtrl lc;
  __shared__ double fin[2];
  __shared__ double qs[2][8];
  __shared__ int act;                 // 0: rejected trial, 1: accepted (go on), 2: nothing more to form
  const int tid = threadIdx.x;
  T* __restrict__ XT = (T*)d.XT;
  T* __restrict__ X = (T*)d.X;
  T* __restrict__ Z = (T*)d.Z;
  // ---- every load of this kernel, then the branch ---------------------------------------------------
  const int halt = MODE == 1 ? d.c->halt : 0;
  const int nc = d.ncols + 2;
  SkinnyColLoads L;
  skinny_colsum_issue<kSbtBlock>(d.part, d.ldpart, nc, d.npow, B, L);
  const bool mine = tid < d.n;
  const T xt_t = mine ? XT[tid] : (T)0, x_t = mine ? X[tid] : (T)0, z_t = mine ? Z[tid] : (T)0;
  const double gx_t = mine ? d.GX[tid] : 0.0;
  if (tid == 0) lc = *d.c;
  if (halt != 0) return;
  const double tot = skinny_colsum_finish(d.part, d.ldpart, nc, B, L, red);
  if (tid == d.ncols) fin[0] = tot;
  if (tid == d.ncols + 1) fin[1] = tot;
  __syncthreads();
  // ---- f and grad f at XT ----------------------------------------------------------------------------
  double gj = 0.0;
  if (mine) {
    if (FK == APX_F_LOGISTIC && tid == d.n - 1) {
      gj = (double)(T)(fin[1] / lc.n_mean);                        // grad[end] = mean(probs - y)
    } else {
      const T gs = (T)tot;                                          // A' * res
      gj = FK == APX_F_LOGISTIC ? (double)(T)((double)gs / lc.n_mean) : (double)gs;
    }
  }
  if (tid == 0) {
    double f_pt;
    if constexpr (FK == APX_F_LEAST_SQUARES) {
      const double nr = sqrt(fin[0]);                               // 0.5 * norm(res)^2 : nrm2 THEN square
      f_pt = 0.5 * (nr * nr);
    } else {
      f_pt = -(fin[0] / lc.n_mean);
    }
    const bool nes = lc.nesterov != 0;
    int a = 1;
    if (MODE == 0 || (nes && lc.phase == 1)) {
      lc.f_x = f_pt;                                                // :52 / :69 / :81  f_x, grad_x = eval_with_gradient(f, x)
      lc.phase = 0;
    } else {
      const double gamma = lc.gamma_try;
      const double nrm = sqrt(lc.s_ss);
      const double ub_z = lc.f_x + lc.s_dot + 1.0 / (2.0 * gamma) * (nrm * nrm);        // :26
      lc.trials_total += 1;
      // (gamma == 0 or not finite: the reference would spin forever; the host-driven path stops there too)
      if (!(f_pt > ub_z) || gamma == 0.0 || !isfinite(gamma)) {                         // :38  accepted
        lc.gamma = gamma;
        lc.norm_res = nrm / gamma;                                                      // :55 / :73
        lc.objective = f_pt + gval_final(d.g, lc.s_graw);
        lc.it += 1;
        if (d.recs != nullptr && lc.it <= lc.rec_cap) {                                 // :56 / :74
          apx_record r;
          r.it = lc.it;
          r.gamma = gamma;
          r.sigma = NAN;
          r.norm_res = lc.norm_res;
          r.objective = lc.objective;
          r.f_x = f_pt;
          r.f_evals = 1 + lc.trials_total + (nes ? lc.it - 1 : 0);                      // (+ the extrapolated points)
          r.grad_evals = lc.it;
          d.recs[lc.it - 1] = r;
        }
        if (lc.norm_res <= lc.tol) lc.done = 1;                                         // :57-59 / :75-77
        if (nes) {
          const double theta_prev = lc.theta;                                           // :78
          lc.theta = (1.0 + sqrt(1.0 + 4.0 * (theta_prev * theta_prev))) / 2.0;         // :79
          lc.beta = (theta_prev - 1.0) / lc.theta;
          lc.gamma_try = gamma;                                                         // (no xi in backtracking_nesterov)
          lc.phase = 1;
          a = 2;                                                    // the next tick evaluates the extrapolated point
        } else {
          lc.f_x = f_pt;                                                                // :60
          lc.gamma_try = lc.xi * gamma;                                                 // :54 of the next iteration
        }
      } else {
        a = 0;
        lc.gamma_try = gamma * lc.shrink;                                               // :39
        if (lc.gamma_try < 1e-12) lc.flags |= APX_FLAG_TINY_GAMMA;                      // :40-42 (logs, continues)
      }
    }
    act = a;
  }
  __syncthreads();
  const int a = act;
  // ---- the n-vectors --------------------------------------------------------------------------------
  double bx = (double)x_t, bg = gx_t;                 // the point and gradient the next trial starts from
  if (a != 0) {                                       // XT was evaluated as (or accepted to be) the new base
    if (a == 2) {                                     // Nesterov, accepted: z_prev = z ; z = XT ; x = z + beta (z - z_prev)
      if (mine) {
        const double zc = (double)xt_t;
        const T xn = (T)(zc + lc.beta * (zc - (double)z_t));                            // :80
        Z[tid] = xt_t;
        X[tid] = xn;
        XT[tid] = xn;
      }
    } else {
      bx = (double)xt_t;
      bg = gj;
      if (mine) {
        X[tid] = xt_t;                                // :60  x = z  (pre-loop / extrapolation tick: X == XT already)
        d.GX[tid] = gj;                               // :61 / :81
        if (MODE == 0) Z[tid] = xt_t;                 // :67  z = x0
      }
    }
  }
  // ---- the next trial z = prox(x - gamma grad_x) (:35 / :43) unless the loop is over ------------------
  const bool form = a != 2 && lc.done == 0;
  if (tid < 128) {                                    // n <= 124: the first two waves hold it all
    double q[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (form && mine) {
      const double gamma = lc.gamma_try;
      const T vi = (T)(bx - gamma * bg);
      const T zi = (T)prox_elem_val<T>(d.g, (double)vi, gamma, tid, 0.0, &q[2]);
      const double dd = (double)zi - bx;
      q[0] = bg * dd;                                 // <grad_x, z - x>
      q[1] = dd * dd;                                 // ||z - x||^2
      XT[tid] = zi;
    }
    wave_sum_multi<8>(q);                             // lane l: sum l / 8
    if ((tid & 7) == 0) qs[tid >> 6][(tid & 63) >> 3] = q[0];
  }
  __syncthreads();
  if (tid == 0) {
    if (form) {
      lc.s_dot = qs[0][0] + qs[1][0];
      lc.s_ss = qs[0][1] + qs[1][1];
      lc.s_graw = qs[0][2] + qs[1][2];
    }
    lc.halt = (lc.done != 0 || lc.it >= lc.maxit) ? 1 : 0;
    *d.c = lc;
  }
}

// ---------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------
bool skinny_bt_eligible(const apx_problem* p) { return skinny_rows_eligible(p); }

template <typename T, int FK, int MODE>
static void sbt_tick_t(const SkinnyRowsPlan& pl, const SbtDev& d, hipStream_t q) {
  skinny_rows_launch<T, FK>(pl, d.XT, &d.c->halt, q);
  hipLaunchKernelGGL((k_sbt_close<T, FK, MODE>), dim3(1), dim3(kSbtBlock), 0, q, d, pl.B);
}
template <int MODE>
static void sbt_tick(const SkinnyRowsPlan& pl, const SbtDev& d, int32_t dtype, int fk, hipStream_t q) {
  const bool ls = fk == APX_F_LEAST_SQUARES;
  if (dtype == APX_F32) {
    if (ls) sbt_tick_t<float, APX_F_LEAST_SQUARES, MODE>(pl, d, q);
    else sbt_tick_t<float, APX_F_LOGISTIC, MODE>(pl, d, q);
  } else {
    if (ls) sbt_tick_t<double, APX_F_LEAST_SQUARES, MODE>(pl, d, q);
    else sbt_tick_t<double, APX_F_LOGISTIC, MODE>(pl, d, q);
  }
}

int32_t skinny_bt_run(apx_problem* p, bool nesterov, const void* x0, double gamma0, double xi, double shrink, double tol,
                      int64_t maxit, const apx_solve_opts* opts, void* x_out, apx_result* res) {
  apx_ctx* ctx = p->ctx;
  const size_t es = p->es();
  const size_t nb = (size_t)p->n_pad * es, nd = (size_t)p->n_pad * sizeof(double);
  DBuf bxt, bx, bz, bgx, bpart, bc, brecs;
  SkinnyRowsPlan pl;
  skinny_rows_plan(p->A_f, p->n, p->b.p, &pl);
  APX_TRY(bpart.alloc_zero((size_t)pl.B * pl.dev.ldpart * sizeof(double), ctx->stream));
  pl.dev.part = (double*)bpart.p;
  APX_TRY(bxt.alloc_zero(nb, ctx->stream));
  APX_TRY(bx.alloc_zero(nb, ctx->stream));
  APX_TRY(bz.alloc_zero(nb, ctx->stream));
  APX_TRY(bgx.alloc_zero(nd, ctx->stream));
  APX_TRY(bc.alloc_zero(sizeof(SbtCtrl), ctx->stream));
  apx_record* host_recs = opts ? opts->records : nullptr;
  const int64_t rec_cap = (opts && opts->records) ? opts->record_cap : 0;
  if (rec_cap > 0) APX_TRY(brecs.alloc_zero((size_t)rec_cap * sizeof(apx_record), ctx->stream));
  SbtDev d{};
  d.ncols = pl.dev.ncols;
  d.n = pl.dev.n;
  d.npow = pl.npow;
  d.ldpart = pl.dev.ldpart;
  d.part = pl.dev.part;
  d.XT = bxt.p;
  d.X = bx.p;
  d.Z = bz.p;
  d.GX = (double*)bgx.p;
  d.c = (SbtCtrl*)bc.p;
  d.g = p->gd;
  d.recs = (apx_record*)brecs.p;

  SbtCtrl c;
  memset(&c, 0, sizeof(c));
  c.xi = xi;
  c.shrink = shrink;
  c.tol = tol;
  c.n_mean = p->n_mean;
  c.maxit = maxit;
  c.rec_cap = rec_cap;
  c.gamma = gamma0;                                  // :51 / :67
  c.gamma_try = nesterov ? gamma0 : xi * gamma0;     // :54 / :72
  c.nesterov = nesterov ? 1 : 0;
  c.theta = 1.0;                                     // :68
  c.norm_res = c.objective = std::numeric_limits<double>::quiet_NaN();
  c.halt = maxit > 0 ? 0 : 1;
  APX_TRY(h2d(ctx, bc.p, &c, sizeof(c)));
  APX_TRY(h2d(ctx, bxt.p, x0, (size_t)p->n * es));
  APX_TRY(h2d(ctx, bx.p, x0, (size_t)p->n * es));
  APX_TRY(h2d(ctx, bz.p, x0, (size_t)p->n * es));
  hipStream_t q = ctx->stream;
  if (maxit > 0) sbt_tick<0>(pl, d, p->dtype, p->f_kind, q);   // :52 / :69  f_x, grad_x = eval_with_gradient(f, x)
  APX_HIP(hipGetLastError());
  while (maxit > 0) {                                // every iteration takes at least one tick (Nesterov: two)
    const int64_t missing = (maxit - c.it) * (nesterov ? 2 : 1);
    const int chunk = (int)std::min<int64_t>(64, std::max<int64_t>(missing, 1));
    for (int k = 0; k < chunk; ++k) sbt_tick<1>(pl, d, p->dtype, p->f_kind, q);
    APX_HIP(hipGetLastError());
    APX_TRY(d2h(ctx, &c, bc.p, sizeof(c)));
    if (c.done != 0 || c.it >= maxit) break;
  }

  APX_TRY(d2h(ctx, x_out, nesterov ? bz.p : bx.p, (size_t)p->n * es));   // z of the last accepted trial (x0 when there is none)
  const bool conv = c.done == 1;
  const int64_t k = c.it;
  if (host_recs != nullptr && rec_cap > 0)
    APX_TRY(d2h(ctx, host_recs, brecs.p, (size_t)std::min<int64_t>(k, rec_cap) * sizeof(apx_record)));
  if (res != nullptr) {
    memset(res, 0, sizeof(*res));
    res->numit = k;
    res->converged = conv ? 1 : 0;
    res->flags = c.flags;
    res->gamma = k > 0 ? c.gamma : gamma0;
    res->sigma = std::numeric_limits<double>::quiet_NaN();
    res->norm_res = c.norm_res;
    res->objective = c.objective;
    // evaluations after an accepted trial happen only if the loop went on (:60-61 / :80-81)
    const int64_t cont = conv ? k - 1 : k;
    res->f_evals = 1 + c.trials_total + (nesterov ? std::max<int64_t>(cont, 0) : 0);
    res->grad_f_evals = 1 + std::max<int64_t>(cont, 0);
    res->prox_g_evals = c.trials_total;
    res->prox_h_evals = -1;
    res->A_evals = -1;
    res->At_evals = -1;
    res->n_records = std::min<int64_t>(k, rec_cap);
  }
  return APX_OK;
}

}  // namespace apx
