// solver.hip — problems (f, g, h, A) and the solver loops of the reference, driven from the
// host as stream-ordered launches with every loop scalar resident on the device.
//
//   adaptive_primal_dual   src/AdaProx.jl:312-364   (the generic loop; everything else wraps it)
//   adaptive_proxgrad      src/AdaProx.jl:418-421   (h = Zero, A = 0)
//   fixed_proxgrad         src/AdaProx.jl:457-459
//   condat_vu              src/AdaProx.jl:367-416   (default-parameter branch)
//   backtracking_proxgrad  src/AdaProx.jl:34-64     (host reads f_z / ub_z per trial: data-dependent branch)
//
// Termination: the reference tests `norm_res <= tol` every iteration.  Here the test runs
// on the device (k_rule / k_final set Ctrl.done) and every later kernel starts with
// `if (done) return`, so the state FREEZES at the converged iterate while the host is
// still enqueueing; the host only reads the flag back every `poll_every` iterations.
// The returned (x, y, it) are exactly those of the iteration whose test passed.
#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstring>
#include <limits>
#include <memory>

#include "solver_internal.h"

using namespace apx;

namespace apx {

int32_t h2d(apx_ctx* ctx, void* dst, const void* src, size_t bytes) {
  if (bytes == 0) return APX_OK;
  APX_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
  APX_HIP(hipStreamSynchronize(ctx->stream));   // src is pageable caller memory: do not outlive the call
  return APX_OK;
}
int32_t d2h(apx_ctx* ctx, void* dst, const void* src, size_t bytes) {
  if (bytes == 0) return APX_OK;
  APX_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
  APX_HIP(hipStreamSynchronize(ctx->stream));
  return APX_OK;
}

static int32_t make_prox_dev(apx_ctx* ctx, const apx_prox_desc& d, int32_t dtype, int64_t len, DBuf& lb, DBuf& ub,
                      DBuf& shift, ProxDev* out) {
  APX_REQUIRE(d.kind >= APX_G_ZERO && d.kind <= APX_G_L2, "unknown prox kind");
  const size_t es = dtype_size(dtype);
  ProxDev p{};
  p.kind = d.kind;
  p.lambda = d.lambda;
  p.lb = d.lb;
  p.ub = d.ub;
  if (d.kind == APX_G_BOXV) {
    APX_REQUIRE(d.lb_vec != nullptr && d.ub_vec != nullptr, "IndBox with vector bounds needs lb_vec and ub_vec");
    APX_TRY(lb.alloc_zero((size_t)round_up(len, kPad) * es, ctx->stream));
    APX_TRY(ub.alloc_zero((size_t)round_up(len, kPad) * es, ctx->stream));
    APX_TRY(h2d(ctx, lb.p, d.lb_vec, (size_t)len * es));
    APX_TRY(h2d(ctx, ub.p, d.ub_vec, (size_t)len * es));
    p.lb_vec = lb.p;
    p.ub_vec = ub.p;
  }
  if (d.shift != nullptr) {
    APX_TRY(shift.alloc_zero((size_t)round_up(len, kPad) * es, ctx->stream));
    APX_TRY(h2d(ctx, shift.p, d.shift, (size_t)len * es));
    p.shift = shift.p;
    p.has_shift = 1;
  }
  *out = p;
  return APX_OK;
}

static int32_t do_allreduce(apx_problem* p, void* buf, int64_t count, int32_t dtype) {
  if (p->comm != nullptr) return comm_allreduce(p->comm, buf, count, dtype);   // ncclAllReduce on the loop's stream (comm.hip)
  const int32_t rc = p->allreduce(p->ar_user, buf, count, dtype, (void*)p->ctx->stream);
  if (rc != 0) return fail(APX_ERR_CALLBACK, "all-reduce callback failed");
  return APX_OK;
}

int32_t problem_allreduce(apx_problem* p, void* buf, int64_t count, int32_t dtype) { return do_allreduce(p, buf, count, dtype); }

int32_t ranks_sum(apx_problem* p, double local, double* total) {
  *total = local;
  if (!p->sharded()) return APX_OK;
  apx_ctx* ctx = p->ctx;
  ctx->h_scal[56] = local;
  APX_HIP(hipMemcpyAsync(ctx->scal + 15, &ctx->h_scal[56], sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  APX_TRY(do_allreduce(p, ctx->scal + 15, 1, APX_F64));
  APX_HIP(hipMemcpyAsync(&ctx->h_scal[57], ctx->scal + 15, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  APX_HIP(hipStreamSynchronize(ctx->stream));
  *total = ctx->h_scal[57];
  return APX_OK;
}

// `f_x, pb = eval_with_pullback(f, x)` on a device x.  need_f: f_x must be final in Ctrl
// when this returns to the stream (backtracking); otherwise a row-sharded f may defer the
// scalar all-reduce to the pullback, where it rides along with the gradient.
int32_t eval_f(apx_problem* p, const void* x_dev, bool need_f) {
  apx_ctx* ctx = p->ctx;
  p->fsc_reduced = false;
  p->grad_ready = false;
  switch (p->f_kind) {
    case APX_F_ZERO:
      break;
    case APX_F_LEAST_SQUARES:
    case APX_F_LOGISTIC: {
      // rows short enough for a workgroup's registers: f AND grad f in one pass over A (csrc/slab.hip); the pullback
      // then has nothing left to do
      bool one_pass = false;
      APX_TRY(slab_eval_with_gradient(p, x_dev, &one_pass));
      if (one_pass) {
        if (p->f_sharded) {
          // the shard's A_k' r_k (the bias entry and the 1 / N of the logistic gradient included: both linear) and its two
          // loss sums, summed over the row shards — in fp64 ONE collective, the scalars riding behind the gradient
          if (p->fsc_packed) {
            APX_TRY(do_allreduce(p, p->grad.p, p->fsc_off() + 2, p->dtype));
          } else {
            APX_TRY(do_allreduce(p, p->grad.p, p->fsc_off(), p->dtype));
            APX_TRY(do_allreduce(p, p->fsc(), 2, APX_F64));
          }
          // (the generic loop's request for the end of the iteration: behind the collective on a row shard — k_slab_tail closes f
          //  and leaves the reductions' wave partials for k_step_rule, two kernels instead of four ; round 4)
          bool tail = false;
          APX_TRY(slab_tail_after_allreduce(p, &tail));
          if (!tail) APX_TRY(vo_f_final(ctx, p->f_kind, p->fsc(), p->ctrl()));
        }
        p->grad_ready = true;
        p->fsc_reduced = true;
        if (need_f) APX_TRY(slab_check(p));        // (the caller reads f now: a timed-out team hand-off must not pass for a value)
        break;
      }
      int splits = 0;
      APX_TRY(launch_gemv_n(ctx, p->A_f, x_dev, &splits, p->done_ptr()));
      const bool local_final = !p->f_sharded;
      APX_TRY(vo_finish_rows(ctx, p->dtype, p->f_kind, splits, p->A_f->ld, p->A_f->m, p->b.p, x_dev, p->n, p->res.p,
                             p->ctrl(), p->fsc(), local_final));
      if (!local_final && need_f) {
        APX_TRY(do_allreduce(p, p->fsc(), 2, APX_F64));
        APX_TRY(vo_f_final(ctx, p->f_kind, p->fsc(), p->ctrl()));
        p->fsc_reduced = true;
      }
      break;
    }
    case APX_F_QUADRATIC: {   // temp = Q*x (kept in `res`), f = 0.5*dot(x, temp) + dot(x, q)
      int splits = 0;
      APX_TRY(launch_gemv_n(ctx, p->A_f, x_dev, &splits, p->done_ptr()));
      APX_TRY(vo_finish_rows(ctx, p->dtype, p->f_kind, splits, p->A_f->ld, p->A_f->m, p->b.p, x_dev, p->n, p->res.p,
                             p->ctrl(), p->fsc(), true));
      break;
    }
    case APX_F_CALLBACK: {
      const size_t bytes = (size_t)p->n * p->es();
      p->hbuf.resize(bytes);
      APX_TRY(d2h(ctx, p->hbuf.data(), x_dev, bytes));
      double fx = 0.0;
      if (p->eval_cb(p->cb_user, p->hbuf.data(), p->n, &fx) != 0) return fail(APX_ERR_CALLBACK, "eval callback failed");
      ctx->h_scal[32] = fx;
      APX_HIP(hipMemcpyAsync((char*)p->ctrl_buf.p + offsetof(Ctrl, f_x), &ctx->h_scal[32], sizeof(double),
                             hipMemcpyHostToDevice, ctx->stream));
      APX_HIP(hipStreamSynchronize(ctx->stream));
      break;
    }
    default:
      return fail(APX_ERR_INVALID, "unknown f kind");
  }
  p->have_eval = true;
  return APX_OK;
}

// `pb()`: gradient at the x of the latest eval, written to p->grad
int32_t pullback_f(apx_problem* p) {
  apx_ctx* ctx = p->ctx;
  APX_REQUIRE(p->have_eval, "pullback before eval_with_pullback");
  switch (p->f_kind) {
    case APX_F_ZERO:      // `pb() = zero(x)`: the buffer was allocated zeroed and nothing of an f = Zero problem ever writes it
      break;
    case APX_F_LEAST_SQUARES:
    case APX_F_LOGISTIC: {
      if (p->grad_ready) break;                    // (formed together with f by csrc/slab.hip)
      APX_TRY(launch_gemv_t(ctx, p->A_f, p->res.p, p->grad.p, p->done_ptr()));
      if (p->f_sharded) {
        // [A_k' r_k] summed over the row shards (K12).  In fp64 the two partial f scalars sit
        // right behind the gradient (fsc_packed) and ride along in the SAME collective: one
        // latency-bound 0.8 MB all-reduce per iteration instead of two.
        if (p->fsc_packed && !p->fsc_reduced) {
          APX_TRY(do_allreduce(p, p->grad.p, p->fsc_off() + 2, p->dtype));
          APX_TRY(vo_f_final(ctx, p->f_kind, p->fsc(), p->ctrl()));
          p->fsc_reduced = true;
        } else {
          APX_TRY(do_allreduce(p, p->grad.p, p->A_f->n_pad, p->dtype));
          if (!p->fsc_reduced) {
            APX_TRY(do_allreduce(p, p->fsc(), 2, APX_F64));
            APX_TRY(vo_f_final(ctx, p->f_kind, p->fsc(), p->ctrl()));
            p->fsc_reduced = true;
          }
        }
      }
      if (p->f_kind == APX_F_LOGISTIC) APX_TRY(vo_logistic_grad(ctx, p->dtype, p->grad.p, p->n, p->ctrl()));
      break;
    }
    case APX_F_QUADRATIC:     // quadratic_pullback() = temp + f.q : no second pass over Q
      APX_TRY(vo_add(ctx, p->dtype, p->grad.p, p->res.p, p->b.p, p->n, p->done_ptr()));
      break;
    case APX_F_CALLBACK: {
      const size_t bytes = (size_t)p->n * p->es();
      p->hbuf.resize(bytes);
      if (p->pb_cb(p->cb_user, p->hbuf.data(), p->n) != 0) return fail(APX_ERR_CALLBACK, "pullback callback failed");
      APX_TRY(h2d(ctx, p->grad.p, p->hbuf.data(), bytes));
      break;
    }
  }
  return APX_OK;
}

int32_t init_ctrl(apx_problem* p, const apx_rule* rule, double tol) {
  Ctrl c{};
  c.tol = tol;
  c.n_mean = p->n_mean;
  c.rule_kind = rule->kind;
  c.r_gamma = rule->gamma;
  c.r_t = rule->t;
  c.r_normA = rule->norm_A;
  c.r_delta = rule->delta;
  c.r_Theta = rule->Theta;
  c.r_xi = rule->xi;
  c.r_nu = rule->nu;
  c.r_r = rule->r;
  switch (rule->kind) {
    case APX_RULE_FIXED:       // :213-215
      APX_REQUIRE(rule->gamma > 0, "FixedStepsize: gamma must be positive");
      c.gamma = rule->gamma;
      c.sigma = rule->gamma * (rule->t * rule->t);
      break;
    case APX_RULE_MM:          // :222-224
      APX_REQUIRE(rule->gamma > 0, "MalitskyMishchenkoRule: gamma must be positive");
      c.gamma = rule->gamma;
      c.sigma = rule->gamma * (rule->t * rule->t);
      c.st0 = rule->gamma;
      c.st1 = std::numeric_limits<double>::infinity();
      break;
    case APX_RULE_OURS: {      // :240-256
      double g = 0.0;
      if (rule->gamma > 0) g = rule->gamma;
      else if (rule->norm_A > 0) g = 1.0 / (2.0 * rule->Theta * rule->t * rule->norm_A);
      else return fail(APX_ERR_INVALID, "you must provide gamma > 0 if norm_A = 0");   // :246
      c.r_gamma = g;
      c.gamma = g;
      c.sigma = g * (rule->t * rule->t);
      c.st0 = g;
      c.st1 = g;
      break;
    }
    case APX_RULE_OURS_PLUS:   // :284-297
      if (!(rule->gamma > 0)) return fail(APX_ERR_INVALID, "you must provide gamma > 0");   // :288
      c.gamma = rule->gamma;
      c.sigma = rule->gamma;
      c.st0 = rule->gamma;
      c.st1 = rule->gamma;
      break;
    default:
      return fail(APX_ERR_INVALID, "unknown step-size rule");
  }
  c.gamma_prev = c.gamma;
  c.rho = 1.0;
  APX_TRY(h2d(p->ctx, p->ctrl_buf.p, &c, sizeof(Ctrl)));
  return APX_OK;
}

int32_t read_ctrl(apx_problem* p, Ctrl* out) { return d2h(p->ctx, out, p->ctrl_buf.p, sizeof(Ctrl)); }

int32_t mul_A_lin(apx_problem* p, const void* x_dev, void* ax) {
  int splits = 0;
  APX_TRY(launch_gemv_n(p->ctx, p->A_lin, x_dev, &splits, nullptr));
  return launch_sum_partials_rows(p->ctx, p->dtype, splits, p->m, p->m_pad, ax, nullptr);
}

}  // namespace apx

extern "C" {

int32_t apx_problem_create(apx_ctx* ctx, const apx_problem_desc* d, apx_problem** out) {
  APX_REQUIRE(ctx != nullptr && d != nullptr && out != nullptr, "apx_problem_create: NULL argument");
  *out = nullptr;
  APX_REQUIRE(d->dtype == APX_F64 || d->dtype == APX_F32, "apx_problem_create: bad dtype");
  APX_REQUIRE(d->n > 0, "apx_problem_create: n must be positive");
  APX_HIP(hipSetDevice(ctx->device));
  std::unique_ptr<apx_problem> p(new apx_problem());
  p->ctx = ctx;
  p->f_kind = d->f_kind;
  p->dtype = d->dtype;
  p->n = d->n;
  p->n_pad = round_up(d->n, kPad) + kPad;   // room for a logistic X with n-1 == k*32 columns
  p->A_f = d->A_f;
  p->A_lin = d->A_lin;
  APX_REQUIRE(d->comm == nullptr || d->allreduce == nullptr,
              "apx_problem_create: give a communicator (comm) or an all-reduce callback, not both");
  APX_REQUIRE(d->comm == nullptr || comm_ctx(d->comm) == ctx, "apx_problem_create: comm belongs to another context");
  const bool shards = d->allreduce != nullptr || d->comm != nullptr;
  if (!shards) {                        // small sparse matrices: the problem runs on their dense copies
    APX_TRY(small_dense_view(d->A_f, &p->A_f));
    APX_TRY(small_dense_view(d->A_lin, &p->A_lin));
  }
  p->allreduce = d->allreduce;
  p->ar_user = d->allreduce_user;
  p->comm = d->comm;
  // shard_mask: bit 0 = rows of A_f sharded, bit 1 = rows of A_lin sharded; 0 with a hook / comm = bit 0
  p->f_sharded = shards && (d->shard_mask == 0 || (d->shard_mask & 1) != 0);
  p->lin_sharded = shards && (d->shard_mask & 2) != 0;
  const size_t es = dtype_size(d->dtype);

  switch (d->f_kind) {
    case APX_F_ZERO:
      break;
    case APX_F_LEAST_SQUARES:
    case APX_F_LOGISTIC: {
      APX_REQUIRE(d->A_f != nullptr && d->b != nullptr, "f needs its matrix A_f and the vector b / labels y");
      APX_REQUIRE(d->A_f->ctx == ctx && d->A_f->dtype == d->dtype, "A_f: wrong context or dtype");
      const int64_t ncol = d->f_kind == APX_F_LOGISTIC ? d->n - 1 : d->n;
      APX_REQUIRE(d->A_f->n == ncol, d->f_kind == APX_F_LOGISTIC
                                         ? "LogisticLoss: n must be columns(X) + 1 (bias is the last coordinate)"
                                         : "LinearLeastSquares: n must equal columns(A)");
      APX_REQUIRE(p->A_f->n_pad <= p->n_pad, "internal: gradient buffer too small");
      APX_TRY(p->b.alloc_zero((size_t)p->A_f->ld * es, ctx->stream));
      APX_TRY(p->res.alloc_zero((size_t)p->A_f->ld * es, ctx->stream));
      APX_TRY(h2d(ctx, p->b.p, d->b, (size_t)p->A_f->m * es));
      p->n_mean = (double)(d->n_total_rows > 0 ? d->n_total_rows : p->A_f->m);
      break;
    }
    case APX_F_QUADRATIC: {   // Quadratic(Q, q), experiments/dual_svm/runme.jl:19-28
      APX_REQUIRE(d->A_f != nullptr && d->b != nullptr, "Quadratic needs Q (A_f) and q (b)");
      APX_REQUIRE(d->A_f->ctx == ctx && d->A_f->dtype == d->dtype, "Q: wrong context or dtype");
      APX_REQUIRE(p->A_f->m == d->n && d->A_f->n == d->n, "Quadratic: Q must be n x n");
      APX_REQUIRE(!p->f_sharded, "Quadratic f cannot be row-sharded by the library");
      APX_TRY(p->b.alloc_zero((size_t)p->A_f->ld * es, ctx->stream));
      APX_TRY(p->res.alloc_zero((size_t)p->A_f->ld * es, ctx->stream));
      APX_TRY(h2d(ctx, p->b.p, d->b, (size_t)d->n * es));
      break;
    }
    case APX_F_CALLBACK:
      APX_REQUIRE(d->eval_cb != nullptr && d->pullback_cb != nullptr, "callback f needs eval_cb and pullback_cb");
      APX_REQUIRE(!p->f_sharded, "callback f cannot be row-sharded by the library");
      p->eval_cb = d->eval_cb;
      p->pb_cb = d->pullback_cb;
      p->cb_user = d->cb_user;
      break;
    default:
      return fail(APX_ERR_INVALID, "apx_problem_create: unknown f kind");
  }

  if (d->A_lin != nullptr) {
    APX_REQUIRE(d->A_lin->ctx == ctx && d->A_lin->dtype == d->dtype, "A_lin: wrong context or dtype");
    APX_REQUIRE(d->A_lin->n == d->n, "A_lin: columns must equal n");
    APX_REQUIRE(d->m == 0 || d->m == p->A_lin->m, "m must equal rows(A_lin)");
    p->m = p->A_lin->m;
    p->m_pad = p->A_lin->ld;
  } else {
    // `A = 0`, `h = Zero()` — adaptive_proxgrad (src/AdaProx.jl:419)
    APX_REQUIRE(d->h.kind == APX_G_ZERO && d->h.shift == nullptr, "h must be Zero when there is no linear operator");
    p->m = 0;
    p->m_pad = 0;
  }
  if (d->g.kind == APX_G_L2)
    return fail(APX_ERR_UNSUPPORTED, "g = NormL2 is not built into the fused step kernel (h = NormL2 is)");

  APX_TRY(make_prox_dev(ctx, d->g, d->dtype, d->n, p->g_lb, p->g_ub, p->g_shift, &p->gd));
  APX_TRY(make_prox_dev(ctx, d->h, d->dtype, std::max<int64_t>(p->m, 1), p->h_lb, p->h_ub, p->h_shift, &p->hd));

  APX_TRY(p->grad.alloc_zero((size_t)p->n_pad * es, ctx->stream));
  APX_TRY(p->xtmp.alloc_zero((size_t)p->n_pad * es, ctx->stream));
  APX_TRY(p->fsc_buf.alloc_zero(sizeof(double) * 8, ctx->stream));
  // n_pad leaves >= 32 spare slots behind A_f->n_pad gradient entries
  p->fsc_packed = p->f_sharded && p->dtype == APX_F64 && p->A_f != nullptr &&
                  p->fsc_off() + 2 <= p->n_pad;
  APX_TRY(p->ctrl_buf.alloc_zero(sizeof(Ctrl), ctx->stream));
  apx_rule r{};
  r.kind = APX_RULE_FIXED;
  r.gamma = 1.0;
  r.t = 1.0;
  APX_TRY(init_ctrl(p.get(), &r, 0.0));
  ctx_retain(ctx);
  comm_retain(p->comm);
  p->held[0] = d->A_f;
  p->held[1] = d->A_lin;
  for (const apx_matrix* M : p->held)
    if (M != nullptr) M->users += 1;
  *out = p.release();
  return APX_OK;
}

int32_t apx_problem_destroy(apx_problem* p) {
  if (p == nullptr || process_exiting()) return APX_OK;      // (exit: matrix.hip ExitMark)
  try {
    (void)hipSetDevice(p->ctx->device);
    (void)hipStreamSynchronize(p->ctx->stream);
    apx_ctx* ctx = p->ctx;
    apx_comm* comm = p->comm;
    for (const apx_matrix* M : p->held)
      if (M != nullptr) M->users -= 1;
    delete p;
    comm_release(comm);
    ctx_release(ctx);
  } catch (...) {
  }
  return APX_OK;
}

int32_t apx_eval(apx_problem* p, const void* x, double* f_x) {
  APX_REQUIRE(p != nullptr && x != nullptr && f_x != nullptr, "apx_eval: NULL argument");
  APX_REQUIRE(p->active == nullptr, "apx_eval: a solver is active on this problem");
  APX_HIP(hipSetDevice(p->ctx->device));
  APX_TRY(h2d(p->ctx, p->xtmp.p, x, (size_t)p->n * p->es()));
  // (a failed team hand-off: the evaluation is repeated on the two passes, and the problem stays on them)
  APX_TRY(with_team_fallback(p, nullptr, [&] { return eval_f(p, p->xtmp.p, true); }));
  Ctrl c;
  APX_TRY(read_ctrl(p, &c));
  *f_x = c.f_x;
  return APX_OK;
}

int32_t apx_problem_team_waits(apx_problem* p, int64_t* out) {
  APX_REQUIRE(p != nullptr && out != nullptr, "apx_problem_team_waits: NULL argument");
  APX_HIP(hipSetDevice(p->ctx->device));
  int64_t pd[4] = {0, 0, 0, 0};
  APX_TRY(slab_team_waits(p, out));
  if (p->active != nullptr) APX_TRY(slab_pd_team_waits(p->active, pd));
  out[0] += pd[0];
  out[1] = std::max(out[1], pd[1]);
  out[2] = std::max(out[2], pd[2]);
  out[3] += pd[3];
  return APX_OK;
}

int32_t apx_problem_team_wait_slots(apx_problem* p, apx_team_wait_slot* out, int32_t cap, int32_t* count) {
  APX_REQUIRE(p != nullptr && count != nullptr && cap >= 0 && (out != nullptr || cap == 0), "apx_problem_team_wait_slots: bad argument");
  APX_HIP(hipSetDevice(p->ctx->device));
  *count = 0;
  for (int kind = 0; kind < 2; ++kind) {
    const TeamWait* w = nullptr;
    if (kind == 0) APX_TRY(slab_team_wait_slots(p, &w));
    else if (p->active != nullptr) APX_TRY(slab_pd_team_wait_slots(p->active, &w));
    if (w == nullptr) continue;
    for (int b = 0; b < 256; ++b) {
      if (w[b].waits == 0) continue;
      if (*count < cap) {
        apx_team_wait_slot& o = out[*count];
        o.waits = w[b].waits;
        o.long_waits = w[b].long_waits;
        o.max_polls = w[b].max_polls;
        o.max_wait_ns = (int64_t)w[b].max_ticks * 10;
        o.at_launch = w[b].at_launch;
        o.at_step = w[b].at_step;
        o.at_clock_ns = (int64_t)w[b].at_clock * 10;
        o.block = b;
        o.team = w[b].team;
        o.member = w[b].member;
        o.xcc = w[b].xcc;
        o.kind = kind;
        o._pad = 0;
      }
      *count += 1;       // (the number there is to report, also beyond cap)
    }
  }
  return APX_OK;
}

int32_t apx_pullback(apx_problem* p, void* grad) {
  APX_REQUIRE(p != nullptr && grad != nullptr, "apx_pullback: NULL argument");
  APX_REQUIRE(p->active == nullptr, "apx_pullback: a solver is active on this problem");
  APX_HIP(hipSetDevice(p->ctx->device));
  APX_TRY(pullback_f(p));
  APX_TRY(d2h(p->ctx, grad, p->grad.p, (size_t)p->n * p->es()));
  return APX_OK;
}

int32_t apx_prox(apx_ctx* ctx, const apx_prox_desc* desc, int32_t conj, int32_t dtype, const void* v, int64_t n,
                 double gamma, void* y, double* g_y) {
  APX_REQUIRE(ctx != nullptr && desc != nullptr && v != nullptr && y != nullptr, "apx_prox: NULL argument");
  APX_REQUIRE(n > 0, "apx_prox: n must be positive");
  APX_REQUIRE(dtype == APX_F64 || dtype == APX_F32, "apx_prox: bad dtype");
  APX_HIP(hipSetDevice(ctx->device));
  const size_t es = dtype_size(dtype);
  DBuf lb, ub, shift, dv, dy;
  ProxDev pd{};
  APX_TRY(make_prox_dev(ctx, *desc, dtype, n, lb, ub, shift, &pd));
  APX_TRY(dv.alloc_zero((size_t)round_up(n, kPad) * es, ctx->stream));
  APX_TRY(dy.alloc_zero((size_t)round_up(n, kPad) * es, ctx->stream));
  APX_TRY(h2d(ctx, dv.p, v, (size_t)n * es));
  APX_TRY(vo_prox(ctx, dtype, dv.p, dy.p, n, pd, gamma, conj != 0, ctx->scal));
  APX_TRY(d2h(ctx, y, dy.p, (size_t)n * es));
  if (g_y != nullptr) {
    double s = 0.0;
    APX_TRY(d2h(ctx, &s, ctx->scal, sizeof(double)));
    *g_y = conj ? std::numeric_limits<double>::quiet_NaN() : vo_gval_final_host(pd, s);
  }
  return APX_OK;
}

int32_t apx_prox_value(apx_ctx* ctx, const apx_prox_desc* desc, int32_t dtype, const void* x, int64_t n,
                       double* value) {
  APX_REQUIRE(ctx != nullptr && desc != nullptr && x != nullptr && value != nullptr, "apx_prox_value: NULL argument");
  APX_REQUIRE(n > 0, "apx_prox_value: n must be positive");
  APX_REQUIRE(dtype == APX_F64 || dtype == APX_F32, "apx_prox_value: bad dtype");
  APX_HIP(hipSetDevice(ctx->device));
  const size_t es = dtype_size(dtype);
  DBuf lb, ub, shift, dx;
  ProxDev pd{};
  APX_TRY(make_prox_dev(ctx, *desc, dtype, n, lb, ub, shift, &pd));
  APX_TRY(dx.alloc_zero((size_t)round_up(n, kPad) * es, ctx->stream));
  APX_TRY(h2d(ctx, dx.p, x, (size_t)n * es));
  APX_TRY(vo_gval(ctx, dtype, dx.p, n, pd, ctx->scal));
  double s = 0.0;
  APX_TRY(d2h(ctx, &s, ctx->scal, sizeof(double)));
  *value = vo_gval_final_host(pd, s);
  return APX_OK;
}

// ---------------------------------------------------------------------------------------
// stepping interface = the generic primal-dual loop
// ---------------------------------------------------------------------------------------
int32_t apx_solver_create(apx_problem* p, const apx_rule* rule, const void* x0, const void* y0, double tol,
                          const apx_solve_opts* opts, apx_solver** out) {
  APX_REQUIRE(p != nullptr && rule != nullptr && x0 != nullptr && out != nullptr, "apx_solver_create: NULL argument");
  APX_REQUIRE(p->active == nullptr, "apx_solver_create: another solver is active on this problem");
  *out = nullptr;
  apx_ctx* ctx = p->ctx;
  APX_HIP(hipSetDevice(ctx->device));
  // use_graph: 0 eager, 1 hipGraph replay, 2 = the library's choice.  Measured on MI355X / ROCm 7.2
  // (tools/small_bench.py): replay is never faster than eager launches here — the two-kernel small path is
  // device-bound at ~12 us / iteration eager vs 18+ us replayed, the generic path ~30 us either way — so
  // the library's choice is eager; replay stays available on request.
  const bool skinny_pg = skinny_pg_eligible(p);                // tall, skinny A_f: row kernel + closing kernel
  const bool small = !skinny_pg && small_eligible(p);
  const bool want_graph = opts != nullptr && opts->use_graph == 1;
  const bool skinny = skinny_pd_eligible(p);                   // tall, skinny A: row kernel + closing kernel
  const bool small_pd = !skinny && !want_graph && small_pd_eligible(p);   // (its buffer roles rotate on the host: no replay)
  if (want_graph && (p->f_kind == APX_F_CALLBACK || p->sharded()))
    return fail(APX_ERR_UNSUPPORTED,
                "hipGraph replay needs a device-only iteration: not with a host-callback f, an all-reduce hook or a communicator");
  const bool lin = p->A_lin != nullptr;
  APX_REQUIRE(!lin || y0 != nullptr, "adaptive_primal_dual needs y0");
  std::unique_ptr<apx_solver> s(new apx_solver());
  s->p = p;
  s->use_graph = want_graph;
  s->rule = *rule;
  s->tol = tol;
  s->poll_every = (opts != nullptr && opts->poll_every > 0) ? opts->poll_every : 16;
  const size_t es = p->es();
  const size_t nb = (size_t)p->n_pad * es;
  APX_TRY(s->x.alloc_zero(nb, ctx->stream));
  APX_TRY(s->xprev.alloc_zero(nb, ctx->stream));
  APX_TRY(s->gprev.alloc_zero(nb, ctx->stream));
  APX_TRY(s->v.alloc_zero(nb, ctx->stream));
  if (lin) {
    const size_t mb = (size_t)p->m_pad * es;
    APX_TRY(s->aty.alloc_zero(nb, ctx->stream));
    APX_TRY(s->y.alloc_zero(mb, ctx->stream));
    APX_TRY(s->ax.alloc_zero(mb, ctx->stream));
    APX_TRY(s->axprev.alloc_zero(mb, ctx->stream));
  }
  if (opts != nullptr && opts->records != nullptr && opts->record_cap > 0) {
    s->rec_cap = opts->record_cap;
    s->host_recs = opts->records;
    APX_TRY(s->recs.alloc_zero((size_t)s->rec_cap * sizeof(apx_record), ctx->stream));
  }
  APX_TRY(init_ctrl(p, rule, tol));                                            // :324
  APX_TRY(h2d(ctx, s->x.p, x0, (size_t)p->n * es));
  if (lin) APX_TRY(h2d(ctx, s->y.p, y0, (size_t)p->m * es));

  if (skinny_pg) {
    APX_TRY(skinny_pg_setup(s.get()));
  } else if (small) {
    APX_TRY(small_setup(s.get()));   // small.hip: pre-loop and iteration as fused kernels
  } else if (skinny) {
    APX_TRY(skinny_pd_setup(s.get(), x0, y0));
  } else if (small_pd) {
    APX_TRY(small_pd_setup(s.get(), x0, y0));
  } else {
    if (lin) {                                                                   // :327  A_x = A * x
      int splits = 0;
      APX_TRY(launch_gemv_n(ctx, p->A_lin, s->x.p, &splits, nullptr));
      APX_TRY(launch_sum_partials_rows(ctx, p->dtype, splits, p->m, p->m_pad, s->ax.p, nullptr));
    }
    APX_TRY(eval_f(p, s->x.p, false));                                           // :328
    APX_TRY(pullback_f(p));
    if (lin) {                                                                   // :329
      APX_TRY(launch_gemv_t(ctx, p->A_lin, s->y.p, s->aty.p, nullptr));
      if (p->lin_sharded) APX_TRY(do_allreduce(p, s->aty.p, p->A_lin->n_pad, p->dtype));
    }
    APX_TRY(vo_step(ctx, p->dtype, s->x.p, s->xprev.p, p->grad.p, s->gprev.p, s->v.p, lin ? s->aty.p : nullptr, p->n,
                    p->gd, p->ctrl()));                                          // :330-332
    if (lin)                                                                     // :331  A_x_prev = A_x
      APX_HIP(hipMemcpyAsync(s->axprev.p, s->ax.p, (size_t)p->m_pad * es, hipMemcpyDeviceToDevice, ctx->stream));
    // (a row-sharded operator: whether a rank CAN take the one pass depends on its own shard, and the two paths enqueue
    //  different collectives — the ranks agree here, once, and take it only if every shard can: csrc/slab.hip does the same for f)
    bool lin_one_pass = lin && slab_pd_eligible(p);
    if (lin && p->lin_sharded) {
      double cannot = lin_one_pass ? 0.0 : 1.0;
      APX_TRY(ranks_sum(p, cannot, &cannot));
      lin_one_pass = cannot == 0.0;
    }
    if (lin_one_pass) {
      // (a rank whose set-up FAILS — an allocation — must still take part in the vote its peers are in: it votes "cannot",
      //  every rank falls back to the two passes, nobody is left waiting in the collective )
      APX_TRY(slab_pd_setup_voted(s.get()));
      if (p->lin_sharded) {       // ... and so does a shard whose kernel turned out not to fit (slab_pd_setup left no state)
        double cannot = s->slab_pd != nullptr ? 0.0 : 1.0;
        APX_TRY(ranks_sum(p, cannot, &cannot));
        if (cannot != 0.0 && s->slab_pd != nullptr) {
          slab_pd_destroy(s->slab_pd);
          s->slab_pd = nullptr;
        }
      }
    }
  }
  p->active = s.get();
  *out = s.release();
  return APX_OK;
}

static int32_t enqueue_iteration(apx_solver* s) {
  if (s->small != nullptr) return small_enqueue(s);
  if (s->small_pd != nullptr) return small_pd_enqueue(s);
  if (s->skinny != nullptr) return skinny_pd_enqueue(s);
  if (s->skinny_pg != nullptr) return skinny_pg_enqueue(s);
  apx_problem* p = s->p;
  apx_ctx* ctx = p->ctx;
  const bool lin = p->A_lin != nullptr;
  // the linear operator's half in ONE pass over A (csrc/slab_pd.hip) while its transposed copy is there
  const bool lin_one_pass = lin && s->slab_pd != nullptr && p->A_lin->dT != nullptr;
  if (lin && !lin_one_pass) {                                                  // :335
    int splits = 0;
    APX_TRY(launch_gemv_n(ctx, p->A_lin, s->x.p, &splits, p->done_ptr()));
    APX_TRY(launch_sum_partials_rows(ctx, p->dtype, splits, p->m, p->m_pad, s->ax.p, p->done_ptr()));
  }
  // (no linear operator: where f and its gradient come from one pass over A, the closing kernel of that pass may take the
  //  residual / step-size reductions and the rule along — csrc/slab.hip SlabTail ; it says so in tail_done)
  SlabTailReq tail{s->x.p, s->xprev.p, s->gprev.p, s->v.p, p->gd, (apx_record*)s->recs.p, s->rec_cap, nullptr, nullptr, 0};
  p->tail_req = lin ? nullptr : &tail;
  p->tail_done = false;
  const int32_t rc_f = eval_f(p, s->x.p, false);                               // :336 (one pass over A where csrc/slab.hip applies)
  p->tail_req = nullptr;
  APX_TRY(rc_f);
  APX_TRY(pullback_f(p));
  if (s->red_ready)      // (the previous iteration's step kernel left k_reduce_n's partials: k_step_red)
    APX_TRY(vo_rule_only(ctx, (const double*)s->red_own.p, p->n, p->gd, p->ctrl(), lin, (apx_record*)s->recs.p, s->rec_cap));
  else if (!p->tail_done)
    APX_TRY(vo_reduce_rule(ctx, p->dtype, s->x.p, s->xprev.p, p->grad.p, s->gprev.p, s->v.p,
                           lin ? s->aty.p : nullptr, p->n, p->gd, p->ctrl(), lin, (apx_record*)s->recs.p,
                           s->rec_cap));                                       // :338-342
  if (lin_one_pass) {
    bool done_here = false;
    APX_TRY(slab_pd_iteration(s, &done_here));                                 // :335, :344-358
    APX_REQUIRE(done_here, "internal: the one-pass primal-dual kernel lost its matrix copy inside an iteration");
  } else if (lin) {
    if (p->lin_sharded) {   // y, A_x, dual_res are row shards: their norms / values are all-reduced
      APX_TRY(vo_dual_sharded(ctx, p->dtype, s->y.p, s->ax.p, s->axprev.p, p->m, p->hd, p->ctrl(),
                              (apx_record*)s->recs.p, s->rec_cap,
                              [p](double* buf, int64_t cnt) { return do_allreduce(p, buf, cnt, APX_F64); }));
    } else {
      APX_TRY(vo_dual(ctx, p->dtype, s->y.p, s->ax.p, s->axprev.p, p->m, p->hd, p->ctrl(), (apx_record*)s->recs.p,
                      s->rec_cap));                                            // :344-354
    }
    APX_TRY(launch_gemv_t(ctx, p->A_lin, s->y.p, s->aty.p, p->done_ptr()));    // :358
    if (p->lin_sharded) APX_TRY(do_allreduce(p, s->aty.p, p->A_lin->n_pad, p->dtype));   // A'y = sum_k A_k' y_k
  }
  if (p->tail_done)      // (the totals of the closing kernel's partials, the rule, the end of the iteration and the step: one kernel)
    return vo_step_rule(ctx, p->dtype, s->x.p, s->xprev.p, p->grad.p, s->gprev.p, s->v.p, p->n, p->gd, p->ctrl(), tail.snap, tail.wred,
                        tail.ldw, (apx_record*)s->recs.p, s->rec_cap);       // :338-342, :348-361
  // f = Zero with a linear operator (the least-absolute-deviation loops): the step also leaves the next iteration's reduction
  // partials (the gradient is zero there as here) — not under hipGraph replay, whose captured iteration must be self-contained
  if (lin && p->f_kind == APX_F_ZERO && !s->use_graph && p->ctx->fused_tail) {
    if (s->red_own.p == nullptr) APX_TRY(s->red_own.alloc_zero(sizeof(double) * kRedQ * kMaxRedBlocks, ctx->stream));
    APX_TRY(vo_step_red(ctx, p->dtype, s->x.p, s->xprev.p, p->grad.p, s->gprev.p, s->v.p, s->aty.p, p->n, p->gd, p->ctrl(),
                        (double*)s->red_own.p));                               // :359-361, and :338, :260-261 of the next iteration
    s->red_ready = true;
    return APX_OK;
  }
  // (APX_OPT_FUSED_TAIL is read per iteration: once it is off, nothing refreshes red_own, and the next iteration must take
  //  its reductions from the vectors again — ADVICE r3)
  s->red_ready = false;
  APX_TRY(vo_step(ctx, p->dtype, s->x.p, s->xprev.p, p->grad.p, s->gprev.p, s->v.p, lin ? s->aty.p : nullptr, p->n,
                  p->gd, p->ctrl()));                                          // :359-361
  return APX_OK;
}

// One iteration captured as a hipGraph and replayed: every pointer and launch geometry of the
// iteration is loop-invariant (prev buffers are rotated by copies inside k_step / k_dual, all
// scalars live in Ctrl), so the graph is instantiated once.  For launch-bound sizes (the
// reference's own 500 x 1000 lasso) this removes the host launch cost of the ~7-12 kernels.
static int32_t capture_iteration(apx_solver* s) {
  apx_ctx* ctx = s->p->ctx;
  if (slab_uses_teams(s->p) || slab_pd_uses_teams(s))
    return fail(APX_ERR_UNSUPPORTED, "hipGraph replay: not with the team kernels of long rows (their launches carry host-side "
                                     "ticket and sequence counters); use eager launches, or apx_ctx_one_pass(ctx, 2)");
  const bool prof = ctx->profile;
  ctx->profile = false;                          // event records would become graph nodes
  hipError_t e = hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal);
  if (e != hipSuccess) {
    ctx->profile = prof;
    return fail(APX_ERR_HIP, std::string("hipStreamBeginCapture: ") + hipGetErrorString(e));
  }
  const int32_t rc = enqueue_iteration(s);
  hipGraph_t graph = nullptr;
  e = hipStreamEndCapture(ctx->stream, &graph);
  ctx->profile = prof;
  if (rc != APX_OK) {
    if (graph) (void)hipGraphDestroy(graph);
    return rc;
  }
  if (e != hipSuccess) return fail(APX_ERR_HIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(e));
  e = hipGraphInstantiate(&s->graph_exec, graph, nullptr, nullptr, 0);
  s->graph = graph;
  if (e != hipSuccess) return fail(APX_ERR_HIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(e));
  return APX_OK;
}

// the error word of the team kernels this solver's iterations ran on ; a failure invalidates the solver for good
static int32_t solver_team_check(apx_solver* s) {
  if (s->poisoned)
    return fail(APX_ERR_HIP, "this solver's iterate was invalidated by a failed team hand-off (apx_last_incident() has the record); "
                             "destroy it — a new solver on the same problem runs on the two GEMV passes");
  int32_t rc = slab_check(s->p);
  if (rc == APX_OK) rc = slab_pd_check(s);
  if (rc != APX_OK) s->poisoned = true;
  return rc;
}

static int32_t poll_done(apx_solver* s, bool* done) {
  apx_ctx* ctx = s->p->ctx;
  int32_t* h = (int32_t*)&ctx->h_scal[40];
  // small path: the closing state (and its done flag) lives in Ctrl B
  const void* cp = s->small != nullptr ? s->small->cb.p : (s->small_pd != nullptr ? small_pd_ctrl(s) : nullptr);
  const int* dp = cp != nullptr ? (const int*)((const char*)cp + offsetof(apx::Ctrl, done)) : s->p->done_ptr();
  APX_HIP(hipMemcpyAsync(h, dp, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
  APX_HIP(hipStreamSynchronize(ctx->stream));
  *done = *h != 0;
  return solver_team_check(s);
}

int32_t apx_solver_step(apx_solver* s, int64_t iters, int32_t sync) {
  APX_REQUIRE(s != nullptr && iters >= 0, "apx_solver_step: bad argument");
  APX_HIP(hipSetDevice(s->p->ctx->device));
  if (s->poisoned) return solver_team_check(s);
  const bool cb = s->p->f_kind == APX_F_CALLBACK;
  if (s->use_graph && s->graph_exec == nullptr && iters > 0 && !s->stopped) APX_TRY(capture_iteration(s));
  for (int64_t k = 0; k < iters && !s->stopped; ++k) {
    if (s->graph_exec != nullptr) APX_HIP(hipGraphLaunch(s->graph_exec, s->p->ctx->stream));
    else {
      const int32_t rc = enqueue_iteration(s);
      if (rc != APX_OK) {
        if (s->p->team_failed) s->poisoned = true;    // (a refused team launch cut the iteration short: the iterate is not one)
        return rc;
      }
    }
    s->iters += 1;
    if (cb || (s->iters % s->poll_every) == 0) {
      bool done = false;
      APX_TRY(poll_done(s, &done));
      if (done) s->stopped = true;
    }
  }
  if (sync) APX_HIP(hipStreamSynchronize(s->p->ctx->stream));
  return APX_OK;
}

int32_t apx_solver_result(apx_solver* s, void* x_out, void* y_out, apx_result* res) {
  APX_REQUIRE(s != nullptr, "apx_solver_result: solver is NULL");
  apx_problem* p = s->p;
  apx_ctx* ctx = p->ctx;
  APX_HIP(hipSetDevice(ctx->device));
  APX_TRY(solver_team_check(s));                   // (a failed team hand-off invalidates the iterate)
  Ctrl c;
  if (s->small != nullptr) APX_TRY(d2h(ctx, &c, s->small->cb.p, sizeof(Ctrl)));
  else if (s->small_pd != nullptr) APX_TRY(d2h(ctx, &c, small_pd_ctrl(s), sizeof(Ctrl)));
  else APX_TRY(read_ctrl(p, &c));
  const bool lin = p->A_lin != nullptr;
  // small path: on convergence x is the tested iterate, otherwise the next one is still staged (Appendix A.8)
  const void* x_src = (s->small != nullptr && c.done == 0) ? s->small->xn.p : s->x.p;
  const void* y_src = s->y.p;
  if (s->small_pd != nullptr) small_pd_result_ptrs(s, c, &x_src, &y_src);
  if (s->skinny != nullptr) {      // x: the tested iterate on convergence, else the next one ; y updated in place
    x_src = s->skinny->x.p;
    y_src = s->skinny->y.p;
  }
  if (x_out != nullptr) APX_TRY(d2h(ctx, x_out, x_src, (size_t)p->n * p->es()));
  if (y_out != nullptr && lin) APX_TRY(d2h(ctx, y_out, y_src, (size_t)p->m * p->es()));
  int64_t nrec = 0;
  if (s->host_recs != nullptr) {
    nrec = std::min<int64_t>(c.it, s->rec_cap);
    APX_TRY(d2h(ctx, s->host_recs, s->recs.p, (size_t)nrec * sizeof(apx_record)));
  }
  if (res != nullptr) {
    memset(res, 0, sizeof(*res));
    const bool conv = c.done != 0;
    const int64_t k = conv ? c.it : s->iters;
    res->numit = k;
    res->converged = conv ? 1 : 0;
    res->flags = c.flags;
    res->gamma = c.gamma;
    res->sigma = c.sigma;
    res->norm_res = c.norm_res;
    res->objective = c.objective;
    // SURVEY §3.4 identities (src/AdaProx.jl:324-363)
    res->f_evals = k + 1;
    res->grad_f_evals = k + 1;
    res->prox_g_evals = conv ? k : k + 1;
    res->prox_h_evals = k;
    res->A_evals = lin ? k + 1 : -1;           // mul_count(0) === nothing (Appendix A.7)
    res->At_evals = lin ? (conv ? k : k + 1) : -1;
    res->n_records = nrec;
    if (c.flags & APX_FLAG_DOMAIN) {
      set_error("a step-size rule took sqrt of a negative number (Julia would throw DomainError)");
      return APX_ERR_DOMAIN;
    }
  }
  return APX_OK;
}

int32_t apx_solver_destroy(apx_solver* s) {
  if (s == nullptr || process_exiting()) return APX_OK;
  try {
    (void)hipSetDevice(s->p->ctx->device);
    (void)hipStreamSynchronize(s->p->ctx->stream);
    if (s->p->active == s) s->p->active = nullptr;
    if (s->graph_exec) (void)hipGraphExecDestroy(s->graph_exec);
    if (s->graph) (void)hipGraphDestroy(s->graph);
    delete s;
  } catch (...) {
  }
  return APX_OK;
}

// ---------------------------------------------------------------------------------------
// one-shot entry points
// ---------------------------------------------------------------------------------------
int32_t apx_adaptive_primal_dual(apx_problem* p, const apx_rule* rule, const void* x0, const void* y0, double tol,
                                 int64_t maxit, const apx_solve_opts* opts, void* x_out, void* y_out,
                                 apx_result* res) {
  APX_REQUIRE(maxit >= 0, "maxit must be non-negative");
  APX_REQUIRE(p != nullptr, "apx_adaptive_primal_dual: problem is NULL");
  // (a team hand-off that failed — the GPU is shared with other work: the 256 workgroups of a launch were not resident
  //  together — discards that run: the solve is repeated from x0 on the two GEMV passes, which need no co-residency, the
  //  problem stays on them and the result carries APX_FLAG_TEAM_FALLBACK ; solver_internal.h with_team_fallback)
  return with_team_fallback(p, res, [&]() -> int32_t {
    apx_solver* s = nullptr;
    APX_TRY(apx_solver_create(p, rule, x0, y0, tol, opts, &s));
    int32_t rc = apx_solver_step(s, maxit, 1);
    if (rc == APX_OK) rc = apx_solver_result(s, x_out, y_out, res);
    apx_solver_destroy(s);
    return rc;
  });
}

int32_t apx_adaptive_proxgrad(apx_problem* p, const apx_rule* rule, const void* x0, double tol, int64_t maxit,
                              const apx_solve_opts* opts, void* x_out, apx_result* res) {
  APX_REQUIRE(p != nullptr, "apx_adaptive_proxgrad: problem is NULL");
  APX_REQUIRE(p->A_lin == nullptr, "adaptive_proxgrad: the problem must have no linear operator (A = 0, h = Zero)");
  return apx_adaptive_primal_dual(p, rule, x0, nullptr, tol, maxit, opts, x_out, nullptr, res);
}

int32_t apx_fixed_proxgrad(apx_problem* p, double gamma, const void* x0, double tol, int64_t maxit,
                           const apx_solve_opts* opts, void* x_out, apx_result* res) {
  apx_rule r{};                      // FixedStepsize(gamma, one(gamma))  :458
  r.kind = APX_RULE_FIXED;
  r.gamma = gamma;
  r.t = 1.0;
  return apx_adaptive_proxgrad(p, &r, x0, tol, maxit, opts, x_out, res);
}

int32_t apx_condat_vu(apx_problem* p, double Lf, double norm_A, const void* x0, const void* y0, double tol,
                      int64_t maxit, const apx_solve_opts* opts, void* x_out, void* y_out, apx_result* res) {
  APX_REQUIRE(p != nullptr && p->A_lin != nullptr, "condat_vu needs a linear operator");
  if (norm_A < 0) {                  // :402-404  norm_A = norm(A)  (Frobenius)
    double ss = 0.0;
    APX_TRY(apx_matrix_sumsq(p->A_lin, &ss));
    norm_A = std::sqrt(ss);
  }
  const double par = 5.0, par2 = 100.0;                                  // :400-401
  const double alpha = (norm_A > par * Lf) ? 1.0 : par2 * norm_A / Lf;   // :405-409
  const double gamma = 1.0 / (Lf / 2 + norm_A / alpha);                  // :410
  const double sigma = 0.99 / (norm_A * alpha);                          // :411
  apx_rule r{};
  r.kind = APX_RULE_FIXED;           // FixedStepsize(gamma, sqrt(sigma / gamma))  :414
  r.gamma = gamma;
  r.t = std::sqrt(sigma / gamma);
  return apx_adaptive_primal_dual(p, &r, x0, y0, tol, maxit, opts, x_out, y_out, res);
}

// adaptive_linesearch_primal_dual ("AdaPDM+") — src/AdaProx.jl:463-550.
// The linesearch test `eta >= ||A'y_next - A'y|| / ||y_next - y||` (:527) is a data-dependent
// host branch per trial, like backtracking: every trial costs one GEMV-T (pass over A) plus a
// 32-byte read-back; the step-size scalars (gamma, eta, Delta, ...) live on the host in fp64.
// Round 4: the FIRST trial of an iteration — nearly always the only one — takes `A_x = A * x` (:502) along: the trial's step
// sizes depend on scalars and on reductions over n-vectors only, so they are known before A x is, and one launch of the one-pass
// primal-dual kernels (csrc/slab_pd.hip slab_pd_trial) then leaves A x AND A' y_next from ONE read of A ; a rejected trial
// continues as before (one GEMV-T per further trial).  50 000 x 20 000 fp64, least absolute deviation: profiles/r04*.
static int32_t adaptive_linesearch_primal_dual_run(apx_problem* p, const apx_linesearch_params* prm, const void* x0,
                                                   const void* y0, double tol, int64_t maxit, const apx_solve_opts* opts,
                                                   void* x_out, void* y_out, apx_result* res);
int32_t apx_adaptive_linesearch_primal_dual(apx_problem* p, const apx_linesearch_params* prm, const void* x0,
                                            const void* y0, double tol, int64_t maxit, const apx_solve_opts* opts,
                                            void* x_out, void* y_out, apx_result* res) {
  APX_REQUIRE(p != nullptr && prm != nullptr && x0 != nullptr && y0 != nullptr && x_out != nullptr,
              "apx_adaptive_linesearch_primal_dual: NULL argument");
  return with_team_fallback(p, res, [&] { return adaptive_linesearch_primal_dual_run(p, prm, x0, y0, tol, maxit, opts, x_out, y_out, res); });
}
static int32_t adaptive_linesearch_primal_dual_run(apx_problem* p, const apx_linesearch_params* prm, const void* x0,
                                                   const void* y0, double tol, int64_t maxit, const apx_solve_opts* opts,
                                                   void* x_out, void* y_out, apx_result* res) {
  APX_REQUIRE(p->A_lin != nullptr, "adaptive_linesearch_primal_dual needs a linear operator");
  // Round 4: a row-sharded A_lin — y, A_x and the trial's y_next are row shards, x and A' y replicated.  Three things cross the
  // ranks per trial: A' y_next (n elements) and the trial's three sums (||y_next - y||^2, ||dual_res||^2, raw h(A_x)) ; the
  // linesearch test is then taken from identical numbers on every rank.  The l2 ball (a norm over all ranks' rows in front of
  // every trial's prox) is not wired for the linesearch loops.
  if (p->lin_sharded && p->hd.kind == APX_G_L2)
    return fail(APX_ERR_UNSUPPORTED, "adaptive_linesearch_primal_dual with a row-sharded linear operator (shard_mask bit 1) takes a "
                                     "separable h only: the l2 ball runs on adaptive_primal_dual / condat_vu");
  APX_REQUIRE(p->active == nullptr, "a solver is active on this problem");
  APX_REQUIRE(maxit >= 0, "maxit must be non-negative");
  double eta = prm->eta;
  const double t = prm->t, delta = prm->delta, Theta = prm->Theta, r = prm->r, R = prm->R;
  APX_REQUIRE(eta > 0, "eta must be positive");                                    // :481
  APX_REQUIRE(Theta > (delta + 1), "must be Theta > (delta + 1)");                 // :482
  double gamma = prm->gamma > 0 ? prm->gamma : 1.0 / (2.0 * Theta * t * eta);      // :484-486
  APX_REQUIRE(gamma <= 1.0 / (2.0 * Theta * t * eta), "gamma is too large");       // :488
  const int64_t max_ls = prm->max_linesearch > 0 ? prm->max_linesearch : 10000;

  apx_ctx* ctx = p->ctx;
  APX_HIP(hipSetDevice(ctx->device));
  // tall, skinny dense A with f = Zero (the reference's LAD / square-root-lasso data): device-resident linesearch
  if (skinny_ls_eligible(p)) return skinny_ls_run(p, prm, gamma, x0, y0, tol, maxit, opts, x_out, y_out, res);
  const size_t es = p->es();
  const size_t nb = (size_t)p->n_pad * es, mb = (size_t)p->m_pad * es;
  DBuf bx, bxprev, bgprev, bv, baty, baty2, by, by2, bax, baxprev;
  for (DBuf* b : {&bx, &bxprev, &bgprev, &bv, &baty, &baty2}) APX_TRY(b->alloc_zero(nb, ctx->stream));
  for (DBuf* b : {&by, &by2, &bax, &baxprev}) APX_TRY(b->alloc_zero(mb, ctx->stream));
  void *y = by.p, *ynext = by2.p, *aty = baty.p, *aty_next = baty2.p, *ax = bax.p, *axprev = baxprev.p;
  apx_rule rr{};
  rr.kind = APX_RULE_FIXED;
  rr.gamma = gamma;
  rr.t = 1.0;
  APX_TRY(init_ctrl(p, &rr, 0.0));
  APX_TRY(h2d(ctx, bx.p, x0, (size_t)p->n * es));
  APX_TRY(h2d(ctx, y, y0, (size_t)p->m * es));
  apx_record* recs = opts ? opts->records : nullptr;
  const int64_t rec_cap = (opts && opts->records) ? opts->record_cap : 0;
  double* d_gamma = (double*)((char*)p->ctrl_buf.p + offsetof(Ctrl, gamma));

  // Ctrl.gamma feeds k_reduce_n / k_step.  No synchronisation behind the upload (round 5: one idle gap of the GPU per iteration
  // less): the value travels from a ring of eight pinned slots, and a slot is written again only eight iterations later — every
  // one of which has synchronised the stream twice (the sums, the trial's read-back).
  int64_t gamma_uploads = 0;
  auto set_gamma = [&](double g) -> int32_t {
    double* slot = &ctx->h_scal[20 + (int)(gamma_uploads++ & 7)];
    *slot = g;
    APX_HIP(hipMemcpyAsync(d_gamma, slot, sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    return APX_OK;
  };
  auto mul_A = [&]() -> int32_t {                 // A_x = A * x
    int splits = 0;
    APX_TRY(launch_gemv_n(ctx, p->A_lin, bx.p, &splits, nullptr));
    return launch_sum_partials_rows(ctx, p->dtype, splits, p->m, p->m_pad, ax, nullptr);
  };

  // the one-pass kernels of the primal-dual loop, when this problem is theirs (a shell of a solver carries their state)
  apx_solver shell;
  shell.p = p;
  if (slab_pd_eligible(p)) APX_TRY(slab_pd_setup_voted(&shell));      // (a shard: a failed set-up is this rank's vote)
  bool one_pass = slab_pd_trial_ok(&shell);
  if (p->lin_sharded) {      // (the two paths enqueue the same collectives here, but the ranks' iterates must come from the same arithmetic)
    double cannot = one_pass ? 0.0 : 1.0;
    APX_TRY(ranks_sum(p, cannot, &cannot));
    one_pass = cannot == 0.0;
  }
  const bool teams = one_pass && slab_pd_uses_teams(&shell);

  double gamma_prev = gamma;                                                       // :491
  const double delta1 = 1.0 + delta;                                               // :490
  APX_TRY(mul_A());                                                                // :494
  APX_TRY(eval_f(p, bx.p, false));                                                 // :495
  APX_TRY(pullback_f(p));
  APX_TRY(launch_gemv_t(ctx, p->A_lin, y, aty, nullptr));                          // :496
  if (p->lin_sharded) APX_TRY(do_allreduce(p, aty, p->A_lin->n_pad, p->dtype));    // A' y = sum_k A_k' y_k
  APX_TRY(vo_step(ctx, p->dtype, bx.p, bxprev.p, p->grad.p, bgprev.p, bv.p, aty, p->n, p->gd, p->ctrl()));   // :497-499
  std::swap(ax, axprev);                                                           // :498  A_x_prev = A_x

  int64_t it_done = maxit, trials_total = 0;
  bool converged = false;
  int32_t flags = 0;
  double sigma = t * t * gamma, norm_res = std::numeric_limits<double>::quiet_NaN(), objective = norm_res;
  for (int64_t it = 1; it <= maxit; ++it) {                                        // :501
    if (!one_pass) APX_TRY(mul_A());                                               // :502 (one pass: with the first trial, below)
    APX_TRY(eval_f(p, bx.p, false));                                               // :503
    APX_TRY(pullback_f(p));
    APX_TRY(vo_reduce_sums(ctx, p->dtype, bx.p, bxprev.p, p->grad.p, bgprev.p, bv.p, aty, p->n, p->gd, p->ctrl(),
                           ctx->scal + 4));                                        // :505, :507-508
    APX_HIP(hipMemcpyAsync(ctx->h_scal + 4, ctx->scal + 4, 5 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    APX_HIP(hipMemcpyAsync(ctx->h_scal + 9, (char*)p->ctrl_buf.p + offsetof(Ctrl, f_x), sizeof(double),
                           hipMemcpyDeviceToHost, ctx->stream));
    APX_HIP(hipStreamSynchronize(ctx->stream));
    APX_TRY(slab_check(p));      // (f's own one-pass kernels, if they run as teams: their error word before f_x and the sums are used)
    const double npr = std::sqrt(ctx->h_scal[4]);
    const double ndg = std::sqrt(ctx->h_scal[5]), dgdx = ctx->h_scal[6], ndx = std::sqrt(ctx->h_scal[7]);
    const double g_x = vo_gval_final_host(p->gd, ctx->h_scal[8]);
    const double f_x = ctx->h_scal[9];
    double C = (ndg * ndg) / dgdx;                                                 // :507
    if (std::isnan(C)) C = 0.0;
    double L = dgdx / (ndx * ndx);                                                 // :508
    if (std::isnan(L)) L = 0.0;
    const double Delta = gamma * L * (gamma * C - 1.0);                            // :509
    const double xi_bar = (t * t) * (gamma * gamma) * (eta * eta) * (delta1 * delta1);   // :510
    const double m4xim1 = 1.0 - 4.0 * xi_bar;                                      // :511
    eta = R * eta;                                                                 // :513
    sigma = t * t * gamma;                                                         // :515
    double ndr2 = 0.0, h_raw = 0.0;
    int64_t trials = 0;
    for (;;) {                                                                     // :516
      auto jsqrt = [&](double v) { if (v < 0) { flags |= APX_FLAG_DOMAIN; return std::numeric_limits<double>::quiet_NaN(); } return std::sqrt(v); };
      auto jmin = [](double a, double b) { return (std::isnan(a) || std::isnan(b)) ? std::numeric_limits<double>::quiet_NaN() : (a < b ? a : b); };
      const double tg = t * eta * gamma;
      const double g1 = gamma * jsqrt(1.0 + gamma / gamma_prev);
      const double g2 = 1.0 / (2.0 * Theta * t * eta);
      const double g3 = gamma * jsqrt(m4xim1 / (2.0 * delta1 * (Delta + jsqrt(Delta * Delta + m4xim1 * (tg * tg)))));
      const double gamma_next = jmin(jmin(g1, g2), g3);                            // :517-521
      const double rho = gamma_next / gamma;                                       // :522
      sigma = t * t * gamma_next;                                                  // :523
      if (one_pass && trials == 0) {
        // A x (:502), w, y_next and A' y_next (:524-526) from one read of A ; y_next and the trial's sums then once more from
        // the stored A x, in the arithmetic of the later trials (O(m))
        APX_TRY(slab_pd_trial(&shell, bx.p, y, axprev, ynext, ax, aty_next, sigma, rho));
        APX_TRY(vo_dual_trial(ctx, p->dtype, y, ax, axprev, ynext, p->m, p->hd, sigma, rho, ctx->scal));
      } else {
        APX_TRY(vo_dual_trial(ctx, p->dtype, y, ax, axprev, ynext, p->m, p->hd, sigma, rho, ctx->scal));   // :524-525
        APX_TRY(launch_gemv_t(ctx, p->A_lin, ynext, aty_next, nullptr));           // :526
      }
      if (p->lin_sharded) {                                                        // A' y_next and the trial's sums over the ranks
        APX_TRY(do_allreduce(p, aty_next, p->A_lin->n_pad, p->dtype));
        APX_TRY(do_allreduce(p, ctx->scal, 3, APX_F64));
      }
      APX_TRY(vo_diff_sumsq(ctx, p->dtype, aty_next, aty, p->n, ctx->scal + 3));
      APX_HIP(hipMemcpyAsync(ctx->h_scal, ctx->scal, 4 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
      APX_HIP(hipStreamSynchronize(ctx->stream));
      if (teams && trials == 0) APX_TRY(slab_pd_check(&shell));                    // (the error word of the team hand-offs)
      trials_total++;
      const double ratio = std::sqrt(ctx->h_scal[3]) / std::sqrt(ctx->h_scal[0]);
      if (eta >= ratio) {                                                          // :527
        gamma_prev = gamma;                                                        // :528
        gamma = gamma_next;
        std::swap(y, ynext);                                                       // :529
        std::swap(aty, aty_next);
        ndr2 = ctx->h_scal[1];
        h_raw = ctx->h_scal[2];
        break;
      }
      eta *= r;                                                                    // :532
      if (++trials >= max_ls)
        return fail(APX_ERR_UNSUPPORTED, "AdaPDM+ linesearch did not terminate (the reference would spin forever)");
    }
    const double ndr = std::sqrt(ndr2);                                            // :535
    norm_res = std::sqrt(npr * npr + ndr * ndr);                                   // :536
    objective = f_x + g_x + vo_gval_final_host(p->hd, h_raw);
    if (recs != nullptr && it <= rec_cap) {                                        // :538-540
      apx_record& q = recs[it - 1];
      q.it = it;
      q.gamma = gamma;
      q.sigma = sigma;
      q.norm_res = norm_res;
      q.objective = objective;
      q.f_x = f_x;
      q.f_evals = it + 1;
      q.grad_evals = 1 + trials_total;      // A' * y_next products so far (+ the one at :496)
    }
    if (norm_res <= tol) {                                                         // :541-543
      it_done = it;
      converged = true;
      break;
    }
    APX_TRY(set_gamma(gamma));
    APX_TRY(vo_step(ctx, p->dtype, bx.p, bxprev.p, p->grad.p, bgprev.p, bv.p, aty, p->n, p->gd, p->ctrl()));   // :545-547
    std::swap(ax, axprev);                                                         // :546  A_x_prev = A_x
  }
  APX_TRY(d2h(ctx, x_out, bx.p, (size_t)p->n * es));
  if (y_out != nullptr) APX_TRY(d2h(ctx, y_out, y, (size_t)p->m * es));
  if (res != nullptr) {
    memset(res, 0, sizeof(*res));
    const int64_t k = it_done;
    res->numit = k;
    res->converged = converged ? 1 : 0;
    res->flags = flags;
    res->gamma = gamma;
    res->sigma = sigma;
    res->norm_res = norm_res;
    res->objective = objective;
    res->f_evals = k + 1;
    res->grad_f_evals = k + 1;
    res->prox_g_evals = converged ? k : k + 1;
    res->prox_h_evals = trials_total;
    res->A_evals = k + 1;
    res->At_evals = 1 + trials_total;
    res->n_records = std::min<int64_t>(k, rec_cap);
    if (flags & APX_FLAG_DOMAIN) {
      set_error("sqrt of a negative number in the AdaPDM+ step-size formula (Julia would throw DomainError)");
      return APX_ERR_DOMAIN;
    }
  }
  return APX_OK;
}

// backtracking_proxgrad — src/AdaProx.jl:34-64
static int32_t backtracking_proxgrad_run(apx_problem* p, const void* x0, double gamma0, double xi, double shrink, double tol,
                                         int64_t maxit, const apx_solve_opts* opts, void* x_out, apx_result* res);
int32_t apx_backtracking_proxgrad(apx_problem* p, const void* x0, double gamma0, double xi, double shrink, double tol,
                                  int64_t maxit, const apx_solve_opts* opts, void* x_out, apx_result* res) {
  APX_REQUIRE(p != nullptr && x0 != nullptr && x_out != nullptr, "apx_backtracking_proxgrad: NULL argument");
  // (a failed team hand-off: the call runs again on the two passes — solver_internal.h with_team_fallback)
  return with_team_fallback(p, res, [&] { return backtracking_proxgrad_run(p, x0, gamma0, xi, shrink, tol, maxit, opts, x_out, res); });
}
static int32_t backtracking_proxgrad_run(apx_problem* p, const void* x0, double gamma0, double xi, double shrink, double tol,
                                         int64_t maxit, const apx_solve_opts* opts, void* x_out, apx_result* res) {
  APX_REQUIRE(p->A_lin == nullptr, "backtracking_proxgrad: the problem must have no linear operator");
  APX_REQUIRE(p->active == nullptr, "a solver is active on this problem");
  APX_REQUIRE(maxit >= 0, "maxit must be non-negative");
  apx_ctx* ctx = p->ctx;
  APX_HIP(hipSetDevice(ctx->device));
  // small dense problems (the reference's own lasso sizes): the linesearch as a device-resident state machine
  if (skinny_bt_eligible(p)) return skinny_bt_run(p, false, x0, gamma0, xi, shrink, tol, maxit, opts, x_out, res);
  if (small_bt_eligible(p)) return small_bt_run(p, false, x0, gamma0, xi, shrink, tol, maxit, opts, x_out, res);
  const size_t es = p->es();
  const size_t nb = (size_t)p->n_pad * es;
  // bgx: grad_x of :52 / :61, private to the loop — every trial evaluates f at z, and where one pass over A forms
  // the gradient along with f (csrc/slab.hip) that evaluation leaves grad f(z) in p->grad; the next shrink trial of the
  // same iteration still needs grad f(x) (:43-45)
  DBuf bx, bz, bgx;
  APX_TRY(bx.alloc_zero(nb, ctx->stream));
  APX_TRY(bz.alloc_zero(nb, ctx->stream));
  APX_TRY(bgx.alloc_zero(nb, ctx->stream));
  void* x = bx.p;
  void* z = bz.p;
  apx_rule r{};
  r.kind = APX_RULE_FIXED;
  r.gamma = 1.0;
  r.t = 1.0;
  APX_TRY(init_ctrl(p, &r, 0.0));
  APX_TRY(h2d(ctx, x, x0, (size_t)p->n * es));
  apx_record* recs = opts ? opts->records : nullptr;
  const int64_t rec_cap = (opts && opts->records) ? opts->record_cap : 0;

  auto read_scalars = [&](double* out4) -> int32_t {   // {dot, ss, graw, f_x}
    APX_HIP(hipMemcpyAsync(ctx->h_scal, ctx->scal, 3 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    APX_HIP(hipMemcpyAsync(ctx->h_scal + 3, (char*)p->ctrl_buf.p + offsetof(Ctrl, f_x), sizeof(double),
                           hipMemcpyDeviceToHost, ctx->stream));
    APX_HIP(hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < 4; ++i) out4[i] = ctx->h_scal[i];
    return APX_OK;
  };

  int64_t f_evals = 0, grad_evals = 0, prox_evals = 0;
  int32_t flags = 0;
  double gamma = gamma0;                                                  // :51
  APX_TRY(eval_f(p, x, true));                                            // :52
  f_evals++;
  double sc[4];
  APX_TRY(read_scalars(sc));
  double f_x = sc[3];
  APX_TRY(pullback_f(p));
  grad_evals++;
  APX_HIP(hipMemcpyAsync(bgx.p, p->grad.p, nb, hipMemcpyDeviceToDevice, ctx->stream));

  void* last_z = x;
  int64_t it_done = maxit;
  bool converged = false;
  double norm_res = std::numeric_limits<double>::quiet_NaN(), objective = norm_res;
  for (int64_t it = 1; it <= maxit; ++it) {                               // :53
    gamma = xi * gamma;                                                   // :54  backtrack_stepsize(xi * gamma, ...)
    double f_z = 0, ub_z = 0, nrm = 0, g_z = 0;
    auto trial = [&]() -> int32_t {
      APX_TRY(vo_bt_trial(ctx, p->dtype, x, bgx.p, z, p->n, p->gd, gamma, ctx->scal));       // :35 / :43
      prox_evals++;
      APX_TRY(eval_f(p, z, true));                                        // :37 / :45
      f_evals++;
      APX_TRY(read_scalars(sc));
      nrm = std::sqrt(sc[1]);
      ub_z = f_x + sc[0] + 1.0 / (2.0 * gamma) * (nrm * nrm);             // :26
      g_z = vo_gval_final_host(p->gd, sc[2]);
      f_z = sc[3];
      return APX_OK;
    };
    APX_TRY(trial());
    while (f_z > ub_z) {                                                  // :38
      gamma *= shrink;                                                    // :39
      if (gamma < 1e-12) flags |= APX_FLAG_TINY_GAMMA;                    // :40-42 (logs, continues)
      APX_TRY(trial());
      if (gamma == 0.0 || !std::isfinite(gamma)) break;                   // the reference would spin forever here
    }
    norm_res = nrm / gamma;                                               // :55
    objective = f_z + g_z;
    if (recs != nullptr && it <= rec_cap) {                               // :56
      apx_record& rr = recs[it - 1];
      rr.it = it;
      rr.gamma = gamma;
      rr.sigma = std::numeric_limits<double>::quiet_NaN();
      rr.norm_res = norm_res;
      rr.objective = objective;
      rr.f_x = f_z;
      rr.f_evals = f_evals;
      rr.grad_evals = grad_evals;
    }
    last_z = z;
    if (norm_res <= tol) {                                                // :57-59
      it_done = it;
      converged = true;
      break;
    }
    std::swap(x, z);                                                      // :60  x, f_x = z, f_z
    f_x = f_z;
    APX_TRY(pullback_f(p));                                               // :61  grad_x = pb()
    grad_evals++;
    APX_HIP(hipMemcpyAsync(bgx.p, p->grad.p, nb, hipMemcpyDeviceToDevice, ctx->stream));
    last_z = x;
  }
  APX_TRY(d2h(ctx, x_out, last_z, (size_t)p->n * es));
  if (res != nullptr) {
    memset(res, 0, sizeof(*res));
    res->numit = it_done;
    res->converged = converged ? 1 : 0;
    res->flags = flags;
    res->gamma = gamma;
    res->sigma = std::numeric_limits<double>::quiet_NaN();
    res->norm_res = norm_res;
    res->objective = objective;
    res->f_evals = f_evals;
    res->grad_f_evals = grad_evals;
    res->prox_g_evals = prox_evals;
    res->prox_h_evals = -1;
    res->A_evals = -1;
    res->At_evals = -1;
    res->n_records = std::min<int64_t>(it_done, rec_cap);
  }
  return APX_OK;
}

}  // extern "C"
