// baselines.hip — the comparison methods every experiment of the reference runs next to
// AdaPGM / AdaPDM, on the same kernels:
//
//   backtracking_nesterov   src/AdaProx.jl:66-84
//   fixed_nesterov          src/AdaProx.jl:91-142
//   agraal                  src/AdaProx.jl:150-192
//   malitsky_pock           src/AdaProx.jl:555-629
//
// All four are host-driven: their step-size logic branches on reductions every iteration
// (linesearches, `min` of data-dependent terms), so the host reads 3-5 scalars back per step.
// The O(mn) work is still the two HIP GEMVs and every vector stays resident in HBM.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>

#include "solver_internal.h"

using namespace apx;

namespace {

constexpr double kNaN = std::numeric_limits<double>::quiet_NaN();

struct Scalars {     // pinned read-back of ctx->scal[0..n) and Ctrl.f_x
  apx_ctx* ctx;
  apx_problem* p;
  int32_t read(int n, bool with_f) {
    APX_HIP(hipMemcpyAsync(ctx->h_scal, ctx->scal, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    if (with_f)
      APX_HIP(hipMemcpyAsync(ctx->h_scal + 15, (char*)p->ctrl_buf.p + offsetof(Ctrl, f_x), sizeof(double),
                             hipMemcpyDeviceToHost, ctx->stream));
    APX_HIP(hipStreamSynchronize(ctx->stream));
    return APX_OK;
  }
  double operator[](int i) const { return ctx->h_scal[i]; }
  double f() const { return ctx->h_scal[15]; }
};

int32_t begin(apx_problem* p, int64_t maxit, bool need_lin) {
  APX_REQUIRE(p != nullptr, "problem is NULL");
  APX_REQUIRE(p->active == nullptr, "a solver is active on this problem");
  APX_REQUIRE(maxit >= 0, "maxit must be non-negative");
  if (need_lin) {
    APX_REQUIRE(p->A_lin != nullptr, "this method needs a linear operator A");
    // (a row-sharded A_lin, round 4: y, A_x and the dual residual are row shards — A' y, ||A_x - A_x_prev||^2 of a trial and the
    //  residual sums are all-reduced, so every rank takes the linesearch's decisions from identical numbers ; a separable h only:
    //  the l2 ball's norm over all ranks' rows in front of every dual step is not wired here)
    if (p->lin_sharded && p->hd.kind == APX_G_L2)
      return fail(APX_ERR_UNSUPPORTED, "malitsky_pock with a row-sharded linear operator (shard_mask bit 1) takes a separable h only: "
                                       "the l2 ball runs on adaptive_primal_dual / condat_vu");
  } else APX_REQUIRE(p->A_lin == nullptr, "this method takes no linear operator (A = 0)");
  APX_HIP(hipSetDevice(p->ctx->device));
  apx_rule r{};
  r.kind = APX_RULE_FIXED;
  r.gamma = 1.0;
  r.t = 1.0;
  return init_ctrl(p, &r, 0.0);
}

void put_record(const apx_solve_opts* opts, int64_t it, double gamma, double sigma, double norm_res, double objective,
                double f_x, int64_t f_evals, int64_t grad_evals) {
  if (opts == nullptr || opts->records == nullptr || it > opts->record_cap) return;
  apx_record& r = opts->records[it - 1];
  r.it = it;
  r.gamma = gamma;
  r.sigma = sigma;
  r.norm_res = norm_res;
  r.objective = objective;
  r.f_x = f_x;
  r.f_evals = f_evals;
  r.grad_evals = grad_evals;
}

void fill_result(apx_result* res, const apx_solve_opts* opts, int64_t it, bool conv, int32_t flags, double gamma,
                 double sigma, double norm_res, double objective, int64_t f_evals, int64_t grad_evals, int64_t prox_g,
                 int64_t prox_h, int64_t A_evals, int64_t At_evals) {
  if (res == nullptr) return;
  memset(res, 0, sizeof(*res));
  res->numit = it;
  res->converged = conv ? 1 : 0;
  res->flags = flags;
  res->gamma = gamma;
  res->sigma = sigma;
  res->norm_res = norm_res;
  res->objective = objective;
  res->f_evals = f_evals;
  res->grad_f_evals = grad_evals;
  res->prox_g_evals = prox_g;
  res->prox_h_evals = prox_h;
  res->A_evals = A_evals;
  res->At_evals = At_evals;
  res->n_records = (opts && opts->records) ? std::min<int64_t>(it, opts->record_cap) : 0;
}

}  // namespace

extern "C" {

// backtracking_nesterov(x0; f, g, gamma0, shrink, tol, maxit) — src/AdaProx.jl:66-84
static int32_t backtracking_nesterov_run(apx_problem* p, const void* x0, double gamma0, double shrink, double tol,
                                  int64_t maxit, const apx_solve_opts* opts, void* x_out, apx_result* res);
int32_t apx_backtracking_nesterov(apx_problem* p, const void* x0, double gamma0, double shrink, double tol,
                                  int64_t maxit, const apx_solve_opts* opts, void* x_out, apx_result* res) {
  // (a failed team hand-off of the one-pass kernels: the call runs again on the two passes — solver_internal.h with_team_fallback)
  return with_team_fallback(p, res, [&] { return backtracking_nesterov_run(p, x0, gamma0, shrink, tol, maxit, opts, x_out, res); });
}
static int32_t backtracking_nesterov_run(apx_problem* p, const void* x0, double gamma0, double shrink, double tol,
                                  int64_t maxit, const apx_solve_opts* opts, void* x_out, apx_result* res) {
  APX_TRY(begin(p, maxit, false));
  APX_REQUIRE(x0 != nullptr && x_out != nullptr, "apx_backtracking_nesterov: NULL argument");
  // small dense problems: the linesearch stays on the device (csrc/small_bt.hip)
  if (skinny_bt_eligible(p)) return skinny_bt_run(p, true, x0, gamma0, 1.0, shrink, tol, maxit, opts, x_out, res);
  if (small_bt_eligible(p)) return small_bt_run(p, true, x0, gamma0, 1.0, shrink, tol, maxit, opts, x_out, res);
  apx_ctx* ctx = p->ctx;
  const size_t es = p->es(), nb = (size_t)p->n_pad * es;
  // bgx: grad_x of :69 / :81, private to the loop (an evaluation at a trial point may leave ITS gradient in p->grad:
  // csrc/slab.hip forms f and grad f in one pass; the shrink trials of :38-46 still need grad f(x))
  DBuf bx, bz0, bz1, bz2, bgx;
  for (DBuf* b : {&bx, &bz0, &bz1, &bz2, &bgx}) APX_TRY(b->alloc_zero(nb, ctx->stream));
  void *x = bx.p, *z = bz0.p, *zprev = bz1.p, *znew = bz2.p;
  APX_TRY(h2d(ctx, x, x0, (size_t)p->n * es));
  APX_HIP(hipMemcpyAsync(z, x, nb, hipMemcpyDeviceToDevice, ctx->stream));             // :67  x, z = x0, x0
  Scalars sc{ctx, p};
  int64_t f_evals = 0, grad_evals = 0, prox_evals = 0;
  int32_t flags = 0;
  double gamma = gamma0, theta = 1.0;                                                   // :67-68
  APX_TRY(eval_f(p, x, true));                                                          // :69
  f_evals++;
  APX_TRY(sc.read(0, true));
  double f_x = sc.f();
  APX_TRY(pullback_f(p));
  grad_evals++;
  APX_HIP(hipMemcpyAsync(bgx.p, p->grad.p, nb, hipMemcpyDeviceToDevice, ctx->stream));
  int64_t it_done = maxit;
  bool conv = false;
  double norm_res = kNaN, objective = kNaN;
  for (int64_t it = 1; it <= maxit; ++it) {                                             // :70
    std::swap(z, zprev);   // z_prev = z (:71); the new z is written into `znew`, rotated below
    double f_z = 0, ub_z = 0, nrm = 0, g_z = 0;
    for (bool first = true;; first = false) {                                           // backtrack_stepsize :34-48
      if (!first) {
        gamma *= shrink;                                                                // :39
        if (gamma < 1e-12) flags |= APX_FLAG_TINY_GAMMA;                                // :40-42
      }
      APX_TRY(vo_bt_trial(ctx, p->dtype, x, bgx.p, znew, p->n, p->gd, gamma, ctx->scal));
      prox_evals++;
      APX_TRY(eval_f(p, znew, true));
      f_evals++;
      APX_TRY(sc.read(3, true));
      nrm = std::sqrt(sc[1]);
      ub_z = f_x + sc[0] + 1.0 / (2.0 * gamma) * (nrm * nrm);                           // :26
      g_z = vo_gval_final_host(p->gd, sc[2]);
      f_z = sc.f();
      if (!(f_z > ub_z) || gamma == 0.0 || !std::isfinite(gamma)) break;                // :38
    }
    norm_res = nrm / gamma;                                                             // :73
    objective = f_z + g_z;
    put_record(opts, it, gamma, kNaN, norm_res, objective, f_z, f_evals, grad_evals);   // :74
    if (norm_res <= tol) {                                                              // :75-77
      it_done = it;
      conv = true;
      std::swap(z, znew);
      break;
    }
    const double theta_prev = theta;                                                    // :78
    theta = (1.0 + std::sqrt(1.0 + 4.0 * (theta_prev * theta_prev))) / 2.0;             // :79
    // zprev currently holds the old z (swapped above); `z` holds the one before it and is free
    APX_TRY(vo_extrap(ctx, p->dtype, x, znew, zprev, (theta_prev - 1.0) / theta, p->n));   // :80
    void* freebuf = z;
    z = znew;
    znew = freebuf;
    APX_TRY(eval_f(p, x, true));                                                        // :81
    f_evals++;
    APX_TRY(sc.read(0, true));
    f_x = sc.f();
    APX_TRY(pullback_f(p));
    grad_evals++;
    APX_HIP(hipMemcpyAsync(bgx.p, p->grad.p, nb, hipMemcpyDeviceToDevice, ctx->stream));
  }
  if (maxit == 0) z = x;
  APX_TRY(d2h(ctx, x_out, z, (size_t)p->n * es));
  fill_result(res, opts, it_done, conv, flags, gamma, kNaN, norm_res, objective, f_evals, grad_evals, prox_evals, -1,
              -1, -1);
  return APX_OK;
}

// fixed_nesterov(x0; f, g, gamma, muf, mug, theta, tol, maxit) — src/AdaProx.jl:91-142
// (gamma is given; `Lf` of the reference is 1/gamma on the host side.  theta < 0 <=> `nothing`.)
static int32_t fixed_nesterov_run(apx_problem* p, const void* x0, double gamma, double muf, double mug, double theta,
                           double tol, int64_t maxit, const apx_solve_opts* opts, void* x_out, apx_result* res);
int32_t apx_fixed_nesterov(apx_problem* p, const void* x0, double gamma, double muf, double mug, double theta,
                           double tol, int64_t maxit, const apx_solve_opts* opts, void* x_out, apx_result* res) {
  // (a failed team hand-off of the one-pass kernels: the call runs again on the two passes — solver_internal.h with_team_fallback)
  return with_team_fallback(p, res, [&] { return fixed_nesterov_run(p, x0, gamma, muf, mug, theta, tol, maxit, opts, x_out, res); });
}
static int32_t fixed_nesterov_run(apx_problem* p, const void* x0, double gamma, double muf, double mug, double theta,
                           double tol, int64_t maxit, const apx_solve_opts* opts, void* x_out, apx_result* res) {
  APX_TRY(begin(p, maxit, false));
  APX_REQUIRE(x0 != nullptr && x_out != nullptr, "apx_fixed_nesterov: NULL argument");
  APX_REQUIRE(gamma > 0, "fixed_nesterov: gamma (or Lf) must be given");                // :104-107
  const double mu = muf + mug;                                                          // :108
  const double q = gamma * mu / (1.0 + gamma * mug);                                    // :109
  APX_REQUIRE(q < 1, "fixed_nesterov: assertion q < 1 failed");                         // :110
  if (theta < 0) theta = q > 0 ? 1.0 / std::sqrt(q) : 0.0;                              // :111-117
  APX_REQUIRE(theta >= 0 && theta <= 1.0 / std::sqrt(q), "fixed_nesterov: assertion 0 <= theta <= 1/sqrt(q) failed");
  const bool skinny = skinny_mom_eligible(p);
  if (skinny || small_mom_eligible(p)) {      // the loop stays on the device (csrc/skinny_mom.hip, csrc/small_mom.hip)
    MomCtrl c0;
    memset(&c0, 0, sizeof(c0));
    c0.tol = tol;
    c0.gamma = gamma;
    c0.theta = theta;
    c0.q = q;
    c0.mu = mu;
    c0.muf = muf;
    c0.mug = mug;
    return skinny ? skinny_mom_run(p, MOM_NESTEROV, c0, x0, x0, maxit, opts, x_out, res)
                  : small_mom_run(p, MOM_NESTEROV, c0, x0, x0, maxit, opts, x_out, res);
  }
  apx_ctx* ctx = p->ctx;
  const size_t es = p->es(), nb = (size_t)p->n_pad * es;
  DBuf b0, b1, b2, b3;
  for (DBuf* b : {&b0, &b1, &b2, &b3}) APX_TRY(b->alloc_zero(nb, ctx->stream));
  void *x = b0.p, *xprev = b1.p, *z = b2.p, *xnew = b3.p;
  APX_TRY(h2d(ctx, x, x0, (size_t)p->n * es));
  APX_HIP(hipMemcpyAsync(xprev, x, nb, hipMemcpyDeviceToDevice, ctx->stream));          // :119
  Scalars sc{ctx, p};
  const bool want_obj = opts != nullptr && opts->records != nullptr;
  int64_t f_evals = 0, grad_evals = 0, prox_evals = 0, it_done = maxit;
  bool conv = false;
  double norm_res = kNaN, objective = kNaN;
  for (int64_t it = 1; it <= maxit; ++it) {                                             // :120
    const double theta_prev = theta;                                                    // :121
    double beta;
    if (mu == 0) {                                                                      // :122-128
      theta = (1.0 + std::sqrt(1.0 + 4.0 * (theta_prev * theta_prev))) / 2.0;
      beta = (theta_prev - 1.0) / theta;
    } else {
      const double a = 1.0 - q * (theta_prev * theta_prev);
      theta = (a + std::sqrt(a * a + 4.0 * (theta_prev * theta_prev))) / 2.0;
      beta = (theta_prev - 1.0) * (1.0 + gamma * mug - theta * gamma * mu) / theta / (1.0 - gamma * muf);
    }
    APX_TRY(vo_extrap(ctx, p->dtype, z, x, xprev, beta, p->n));                         // :129
    APX_TRY(eval_f(p, z, false));                                                       // :130
    APX_TRY(pullback_f(p));
    f_evals++;
    grad_evals++;
    APX_TRY(vo_bt_trial(ctx, p->dtype, z, p->grad.p, xnew, p->n, p->gd, gamma, ctx->scal));   // :132
    prox_evals++;
    double f_new = kNaN;
    if (want_obj) APX_TRY(eval_f(p, xnew, true));     // objective = f(x) + g_x, lazily, inside without_counting (:134-136)
    APX_TRY(sc.read(3, want_obj));
    if (want_obj) f_new = sc.f();
    norm_res = std::sqrt(sc[1]) / gamma;                                                // :133
    objective = f_new + vo_gval_final_host(p->gd, sc[2]);
    put_record(opts, it, gamma, kNaN, norm_res, objective, f_new, f_evals, grad_evals);
    std::swap(xprev, x);                                                                // :131  x_prev = x
    std::swap(x, xnew);                                                                 //        x = prox(...)
    if (norm_res <= tol) {                                                              // :137-139
      it_done = it;
      conv = true;
      break;
    }
  }
  APX_TRY(slab_check(p));                       // (evaluations whose value nobody read: a failed team hand-off must not pass for an iterate)
  APX_TRY(d2h(ctx, x_out, x, (size_t)p->n * es));
  fill_result(res, opts, it_done, conv, 0, gamma, kNaN, norm_res, objective, f_evals, grad_evals, prox_evals, -1, -1, -1);
  return APX_OK;
}

// agraal(x1; f, g, x0, gamma0, gamma_max, phi, tol, maxit) — src/AdaProx.jl:150-192
// (x0 must be supplied: the host draws `x1 + randn` when the caller passes `nothing`; gamma0 <= 0 <=> `nothing`)
static int32_t agraal_run(apx_problem* p, const void* x1, const void* x0, double gamma0, double gamma_max, double phi,
                   double tol, int64_t maxit, const apx_solve_opts* opts, void* x_out, apx_result* res);
int32_t apx_agraal(apx_problem* p, const void* x1, const void* x0, double gamma0, double gamma_max, double phi,
                   double tol, int64_t maxit, const apx_solve_opts* opts, void* x_out, apx_result* res) {
  // (a failed team hand-off of the one-pass kernels: the call runs again on the two passes — solver_internal.h with_team_fallback)
  return with_team_fallback(p, res, [&] { return agraal_run(p, x1, x0, gamma0, gamma_max, phi, tol, maxit, opts, x_out, res); });
}
static int32_t agraal_run(apx_problem* p, const void* x1, const void* x0, double gamma0, double gamma_max, double phi,
                   double tol, int64_t maxit, const apx_solve_opts* opts, void* x_out, apx_result* res) {
  APX_TRY(begin(p, maxit, false));
  APX_REQUIRE(x1 != nullptr && x0 != nullptr && x_out != nullptr, "apx_agraal: NULL argument");
  const bool skinny = skinny_mom_eligible(p);
  if (skinny || small_mom_eligible(p)) {      // the loop stays on the device (csrc/skinny_mom.hip, csrc/small_mom.hip)
    MomCtrl c0;
    memset(&c0, 0, sizeof(c0));
    c0.tol = tol;
    c0.gamma = gamma0 > 0 ? gamma0 : 0.0;
    c0.need_gamma0 = gamma0 > 0 ? 0 : 1;                    // :168-170
    c0.theta = 1.0;                                          // :173
    c0.phi = phi;
    c0.rho = 1.0 / phi + 1.0 / (phi * phi);                  // :172
    c0.gamma_max = gamma_max;
    return skinny ? skinny_mom_run(p, MOM_AGRAAL, c0, x1, x0, maxit, opts, x_out, res)
                  : small_mom_run(p, MOM_AGRAAL, c0, x1, x0, maxit, opts, x_out, res);
  }
  apx_ctx* ctx = p->ctx;
  const size_t es = p->es(), nb = (size_t)p->n_pad * es;
  DBuf b0, b1, b2, b3, b4;
  for (DBuf* b : {&b0, &b1, &b2, &b3, &b4}) APX_TRY(b->alloc_zero(nb, ctx->stream));
  void *x = b0.p, *xprev = b1.p, *xbar = b2.p, *xnew = b3.p, *gprev = b4.p;
  APX_TRY(h2d(ctx, x, x1, (size_t)p->n * es));                                          // :165
  APX_TRY(h2d(ctx, xprev, x0, (size_t)p->n * es));
  APX_HIP(hipMemcpyAsync(xbar, x, nb, hipMemcpyDeviceToDevice, ctx->stream));
  Scalars sc{ctx, p};
  const bool want_obj = opts != nullptr && opts->records != nullptr;
  int64_t f_evals = 0, grad_evals = 0, prox_evals = 0, it_done = maxit;
  // :166-167 — gradient at x_prev first so that p->grad ends up holding grad_x
  APX_TRY(eval_f(p, xprev, false));
  APX_TRY(pullback_f(p));
  APX_HIP(hipMemcpyAsync(gprev, p->grad.p, nb, hipMemcpyDeviceToDevice, ctx->stream));
  APX_TRY(eval_f(p, x, false));
  APX_TRY(pullback_f(p));
  f_evals += 2;
  grad_evals += 2;
  double gamma = gamma0;
  if (!(gamma0 > 0)) {                                                                  // :168-170
    APX_TRY(vo_diff_sumsq(ctx, p->dtype, x, xprev, p->n, ctx->scal));
    APX_TRY(vo_diff_sumsq(ctx, p->dtype, p->grad.p, gprev, p->n, ctx->scal + 1));
    APX_TRY(sc.read(2, false));
    gamma = std::sqrt(sc[0]) / std::sqrt(sc[1]);
  }
  const double rho = 1.0 / phi + 1.0 / (phi * phi);                                     // :172
  double theta = 1.0;                                                                   // :173
  bool conv = false;
  double norm_res = kNaN, objective = kNaN;
  for (int64_t it = 1; it <= maxit; ++it) {                                             // :174
    APX_TRY(vo_diff_sumsq(ctx, p->dtype, x, xprev, p->n, ctx->scal));
    APX_TRY(vo_diff_sumsq(ctx, p->dtype, p->grad.p, gprev, p->n, ctx->scal + 1));
    APX_TRY(sc.read(2, false));
    const double ndx = std::sqrt(sc[0]), ndg = std::sqrt(sc[1]);
    const double C = (ndx * ndx) / (ndg * ndg);                                         // :175
    const double gamma_prev = gamma;                                                    // :176
    double gm = rho * gamma_prev;                                                       // :177  Julia min (NaN-propagating)
    const double t2 = phi * theta * C / (4.0 * gamma_prev);
    gm = (std::isnan(gm) || std::isnan(t2)) ? kNaN : std::min(gm, t2);
    gm = (std::isnan(gm) || std::isnan(gamma_max)) ? kNaN : std::min(gm, gamma_max);
    gamma = gm;
    theta = phi * gamma / gamma_prev;                                                   // :178
    APX_TRY(vo_xbar(ctx, p->dtype, xbar, x, phi, p->n));                                // :179
    APX_HIP(hipMemcpyAsync(gprev, p->grad.p, nb, hipMemcpyDeviceToDevice, ctx->stream));   // :180 grad_x_prev = grad_x
    APX_TRY(vo_bt_trial(ctx, p->dtype, xbar, gprev, xnew, p->n, p->gd, gamma, ctx->scal));   // :181
    prox_evals++;
    APX_TRY(vo_diff_sumsq(ctx, p->dtype, xnew, x, p->n, ctx->scal + 3));                 // :182  norm(x - x_prev)
    double f_new = kNaN;
    if (want_obj) APX_TRY(eval_f(p, xnew, true));                                       // :183-185 (lazy)
    APX_TRY(sc.read(4, want_obj));
    if (want_obj) f_new = sc.f();
    norm_res = std::sqrt(sc[3]) / gamma;
    objective = f_new + vo_gval_final_host(p->gd, sc[2]);
    put_record(opts, it, gamma, kNaN, norm_res, objective, f_new, f_evals, grad_evals);
    std::swap(xprev, x);                                                                // :180  x_prev = x
    std::swap(x, xnew);                                                                 // :181  x = prox(...)
    if (norm_res <= tol) {                                                              // :186-188
      it_done = it;
      conv = true;
      break;
    }
    APX_TRY(eval_f(p, x, false));                                                       // :189
    APX_TRY(pullback_f(p));
    f_evals++;
    grad_evals++;
  }
  APX_TRY(slab_check(p));                       // (evaluations whose value nobody read: a failed team hand-off must not pass for an iterate)
  APX_TRY(d2h(ctx, x_out, x, (size_t)p->n * es));
  fill_result(res, opts, it_done, conv, 0, gamma, kNaN, norm_res, objective, f_evals, grad_evals, prox_evals, -1, -1, -1);
  return APX_OK;
}

// malitsky_pock(x, y; f, g, h, A, sigma, t, tol, maxit) — src/AdaProx.jl:555-629
// Round 4: every linesearch trial's `A_x = A * x` (:561 / :574) takes the NEXT iteration's dual step along — w = y + sigma A_x,
// y = prox(h_conj, w, sigma), At_y = A' y (:601-603) with the trial's sigma, which is the one the next iteration starts from if
// the trial is accepted: y[i] needs row i of A only, so one launch of the one-pass primal-dual kernels in their trial mode
// (csrc/slab_pd.hip slab_pd_trial, rho = 0) leaves both from one read of A, and the GEMV-T of the dual step disappears from the
// loop (a rejected trial's dual step is simply not used).
static int32_t malitsky_pock_run(apx_problem* p, const void* x0, const void* y0, double sigma, double t, double tol,
                                 int64_t maxit, const apx_solve_opts* opts, void* x_out, void* y_out, apx_result* res);
int32_t apx_malitsky_pock(apx_problem* p, const void* x0, const void* y0, double sigma, double t, double tol,
                          int64_t maxit, const apx_solve_opts* opts, void* x_out, void* y_out, apx_result* res) {
  APX_TRY(begin(p, maxit, true));
  APX_REQUIRE(x0 != nullptr && y0 != nullptr && x_out != nullptr, "apx_malitsky_pock: NULL argument");
  return with_team_fallback(p, res, [&] { return malitsky_pock_run(p, x0, y0, sigma, t, tol, maxit, opts, x_out, y_out, res); });
}
static int32_t malitsky_pock_run(apx_problem* p, const void* x0, const void* y0, double sigma, double t, double tol,
                                 int64_t maxit, const apx_solve_opts* opts, void* x_out, void* y_out, apx_result* res) {
  // tall, skinny dense A with f = Zero (the reference's LAD / square-root-lasso data): device-resident linesearch
  if (skinny_mp_eligible(p)) return skinny_mp_run(p, x0, y0, sigma, t, tol, maxit, opts, x_out, y_out, res);
  if (small_mpq_eligible(p)) return small_mpq_run(p, x0, y0, sigma, t, tol, maxit, opts, x_out, y_out, res);
  apx_ctx* ctx = p->ctx;
  const size_t es = p->es(), nb = (size_t)p->n_pad * es, mb = (size_t)p->m_pad * es;
  DBuf bx, bxn, bv, bgp, baty, batyp, batyq, by, byn, byq, bax, baxn;
  for (DBuf* b : {&bx, &bxn, &bv, &bgp, &baty, &batyp, &batyq}) APX_TRY(b->alloc_zero(nb, ctx->stream));
  for (DBuf* b : {&by, &byn, &byq, &bax, &baxn}) APX_TRY(b->alloc_zero(mb, ctx->stream));
  void *x = bx.p, *xnew = bxn.p, *aty = baty.p, *atyprev = batyp.p, *y = by.p, *ynew = byn.p, *ax = bax.p,
       *axnew = baxn.p;
  void *yq = byq.p, *atyq = batyq.p;      // the NEXT iteration's dual step, formed with a trial (one pass over A)
  // the one-pass kernels of the primal-dual loop, when this problem is theirs (a shell of a solver carries their state)
  apx_solver shell;
  shell.p = p;
  if (slab_pd_eligible(p)) APX_TRY(slab_pd_setup_voted(&shell));      // (a shard: a failed set-up is this rank's vote)
  bool one_pass = slab_pd_trial_ok(&shell);
  const bool shard = p->lin_sharded;
  if (shard) {                            // (the ranks' iterates must come from the same arithmetic: all of them, or none)
    double cannot = one_pass ? 0.0 : 1.0;
    APX_TRY(ranks_sum(p, cannot, &cannot));
    one_pass = cannot == 0.0;
  }
  const bool teams = one_pass && slab_pd_uses_teams(&shell);
  bool have_next = false;                 // yq / atyq hold this iteration's dual step already (the last accepted trial left it)
  APX_TRY(h2d(ctx, x, x0, (size_t)p->n * es));
  APX_TRY(h2d(ctx, y, y0, (size_t)p->m * es));
  Scalars sc{ctx, p};
  int32_t flags = 0;
  int64_t f_evals = 0, grad_evals = 0, trials_total = 0, it_done = maxit;
  const double theta_outer = 1.0;                                                       // :595 (never updated in the reference)
  APX_TRY(mul_A_lin(p, x, ax));                                                         // :597
  APX_TRY(launch_gemv_t(ctx, p->A_lin, y, aty, nullptr));                               // :598
  if (shard) APX_TRY(problem_allreduce(p, aty, p->A_lin->n_pad, p->dtype));             // A' y = sum_k A_k' y_k
  bool conv = false;
  double gamma = kNaN, norm_res = kNaN, objective = kNaN;
  for (int64_t it = 1; it <= maxit; ++it) {                                             // :599
    std::swap(aty, atyprev);                                                            // :600  At_y_prev = At_y
    // :601-602  w = y + sigma*A_x ; y = prox(h_conj, w, sigma)   (rho = 0 => w = y + sigma*A_x exactly)
    if (have_next) {                                                                    // (it came with the accepted trial)
      std::swap(ynew, yq);
      std::swap(aty, atyq);
    } else {
      APX_TRY(vo_dual_trial(ctx, p->dtype, y, ax, ax, ynew, p->m, p->hd, sigma, 0.0, ctx->scal + 8));
      APX_TRY(launch_gemv_t(ctx, p->A_lin, ynew, aty, nullptr));                        // :603
      if (shard) APX_TRY(problem_allreduce(p, aty, p->A_lin->n_pad, p->dtype));
    }
    const double sigma_prev = sigma;                                                    // :605
    sigma = sigma * std::sqrt(1.0 + theta_outer);                                       // :606
    APX_TRY(eval_f(p, x, true));                                                        // :608
    APX_TRY(sc.read(0, true));
    const double f_x_prev = sc.f();
    APX_TRY(pullback_f(p));
    f_evals++;
    grad_evals++;
    APX_HIP(hipMemcpyAsync(bgp.p, p->grad.p, nb, hipMemcpyDeviceToDevice, ctx->stream));
    double f_x = 0.0, g_raw = 0.0;
    for (bool first = true;; first = false) {                                           // backtrack_stepsize_MP :555-579
      if (!first) {
        sigma /= 2;                                                                     // :565
        if (sigma < 1e-12) flags |= APX_FLAG_TINY_GAMMA;                                // :566-568
      }
      const double theta = sigma / sigma_prev;                                          // :556 / :569
      gamma = t * t * sigma;                                                            // :557 / :570
      APX_TRY(vo_mp_trial(ctx, p->dtype, x, aty, atyprev, bgp.p, bv.p, xnew, p->n, p->gd, theta, gamma, ctx->scal));
      if (one_pass) {
        // A_x = A * x (:561 / :574) and, from the same read of A, the dual step the next iteration starts with if this trial is
        // accepted (:601-603 with this trial's sigma ; rho = 0: A_x_prev does not enter) ; for the l2 ball y once more from the
        // stored A x, in the dual step's own arithmetic
        APX_TRY(slab_pd_trial(&shell, xnew, ynew, ax, yq, axnew, atyq, sigma, 0.0));
        if (p->hd.kind == APX_G_L2) APX_TRY(vo_dual_trial(ctx, p->dtype, ynew, axnew, axnew, yq, p->m, p->hd, sigma, 0.0, ctx->scal + 8));
      } else {
        APX_TRY(mul_A_lin(p, xnew, axnew));                                             // :561 / :574
      }
      APX_TRY(eval_f(p, xnew, true));                                                   // :562 / :575
      APX_TRY(vo_diff_sumsq(ctx, p->dtype, axnew, ax, p->m, ctx->scal + 3));
      if (shard) APX_TRY(problem_allreduce(p, ctx->scal + 3, 1, APX_F64));              // ||A_x - A_x_prev||^2 over the row shards
      APX_TRY(sc.read(4, true));
      trials_total++;
      f_evals++;
      f_x = sc.f();
      g_raw = sc[2];
      const double ndx = std::sqrt(sc[0]), ndax = std::sqrt(sc[3]);
      if (teams) APX_TRY(slab_pd_check(&shell));                                        // (the error word of the team hand-offs)
      const double lhs = gamma * sigma * (ndax * ndax) + 2.0 * gamma * (f_x - f_x_prev - sc[1]);   // :563 / :576
      if (!(lhs > 0.95 * (ndx * ndx)) || sigma == 0.0) break;                           // :564
    }
    have_next = one_pass;
    // (the accepted trial's dual step: its A' y over the row shards — only now, every rank accepts the same trial)
    if (have_next && shard) APX_TRY(problem_allreduce(p, atyq, p->A_lin->n_pad, p->dtype));
    APX_TRY(pullback_f(p));                                                             // :612  grad_x = pb()
    grad_evals++;
    APX_TRY(vo_primal_res(ctx, p->dtype, bv.p, xnew, p->grad.p, aty, p->n, gamma, p->gd, ctx->scal));       // :616
    APX_TRY(vo_mp_dualres(ctx, p->dtype, y, ax, ynew, axnew, p->m, sigma_prev, p->hd, ctx->scal + 2));      // :617
    if (shard) APX_TRY(problem_allreduce(p, ctx->scal + 2, 2, APX_F64));                // ||dual_res||^2, raw h(A_x)
    APX_TRY(sc.read(4, false));
    const double npr = std::sqrt(sc[0]), ndr = std::sqrt(sc[2]);
    norm_res = std::sqrt(npr * npr + ndr * ndr);                                        // :618
    objective = f_x + vo_gval_final_host(p->gd, sc[1]) + vo_gval_final_host(p->hd, sc[3]);
    (void)g_raw;
    put_record(opts, it, gamma, sigma, norm_res, objective, f_x, f_evals, grad_evals);  // :620-622
    std::swap(x, xnew);
    std::swap(ax, axnew);
    std::swap(y, ynew);                                                                 // :614  y_prev = y
    if (norm_res <= tol) {                                                              // :624-626
      it_done = it;
      conv = true;
      break;
    }
  }
  APX_TRY(d2h(ctx, x_out, x, (size_t)p->n * es));
  if (y_out != nullptr) APX_TRY(d2h(ctx, y_out, y, (size_t)p->m * es));
  const int64_t k = it_done;
  fill_result(res, opts, k, conv, flags, gamma, sigma, norm_res, objective, f_evals, grad_evals, trials_total, k,
              1 + trials_total, 1 + k);
  return APX_OK;
}

}  // extern "C"
