// skinny_ls.hip — AdaPDM+ (`adaptive_linesearch_primal_dual`, src/AdaProx.jl:463-550) as a DEVICE-RESIDENT
// loop for tall, skinny dense A (n <= 124, f = Zero): the method the reference's least_absolute_deviation and
// square_root_lasso experiments showcase (15 of their 31 runs each), on data of 8192 x 13, 4177 x 9, 506 x 14.
// The generic implementation (solver.hip) drives the linesearch from the host — a scalar read-back per trial,
// ~75 us per iteration at these sizes.  Here the linesearch is a small state machine in device memory:
//
//   one TICK = one linesearch trial (:516-533) = k_ls_rows + k_ls_close  (the l2 ball needs its norm first:
//   k_ls_rows<2>, k_ls_rows<3>, k_ls_close<1>)
//     k_ls_rows : first trial of an iteration: primal residual (:505), Delta / xi_bar (:507-511, grad f = 0),
//                 eta = R eta (:513), A_x = <A[i,:], x> (cached for the retries) ; every trial: gamma_next
//                 (:517-521), rho, sigma, w (:524), y_next (:525) into a trial buffer, block partials of
//                 A'y_next, ||y_next - y||^2, ||dual_res||^2, h(A_x)
//     k_ls_close: sums ; the test eta >= ||A'y_next - A'y|| / ||y_next - y|| (:527) ; rejected: eta *= r, next
//                 tick retries ; accepted: gamma, A'y, y (committed lazily by the next row kernel), dual residual,
//                 norm_res, record, termination (:535-543), primal step (:545-547)
//
// The host only enqueues ticks and reads the iteration counter every few dozen ticks; ticks behind the last
// iteration (maxit reached, or converged) do nothing.  All sums have a fixed order; no atomics.
#include <cstdlib>
#include <cstring>
#include <memory>

#include "skinny_dev.h"

namespace apx {

struct LsCtrl {
  // parameters (:463-480)
  double t, delta1, Theta, r, R, tol;
  int64_t maxit, max_ls, rec_cap;
  // state
  double eta, gamma, gamma_prev, sigma;
  double Delta, m4xim1, npr2, g_x;          // of the current outer iteration
  double gamma_next, rho, sigma_next, h_pre;   // of the current trial
  double norm_res, objective;
  int64_t it, trials_total, trials_this;
  int32_t phase;      // 0: the next tick starts a new outer iteration ; 1: it retries the linesearch
  int32_t done;       // 1: converged ; 2: the linesearch did not terminate
  int32_t flags;
  int32_t ycommit;    // the accepted y still sits in the trial buffer
};

struct LsDev {
  const void* A;
  int64_t ld;
  int32_t m, n, npow, _pad;
  int64_t ldpart;
  void *X, *XP, *V;
  double *aty, *aty_next;
  void *Y, *YT;              // y ; trial y_next
  double *axc, *axp, *wbuf;  // A x of this / of the previous outer iteration ; l2 scratch (w)
  double* part;              // [blocks][ldpart]: n partials of A'y_next, ||dr||^2, h(A_x), ||z||^2, ||y_next - y||^2
  LsCtrl *c, *cs;
  ProxDev g, h;
  apx_record* recs;
};

__device__ __forceinline__ double ls_min2(double a, double b) {   // Julia min: NaN-propagating
  if (isnan(a) || isnan(b)) return NAN;
  return a < b ? a : b;
}
__device__ __forceinline__ double ls_sqrt(double v, int32_t* fl) {
  if (v < 0.0) {
    *fl |= APX_FLAG_DOMAIN;
    return NAN;
  }
  return sqrt(v);
}


// PHASE 0: pre-loop (:494-496): A x0 -> axc, block partials of A'y0
// PHASE 1: a whole trial (elementwise conjugate prox)
// PHASE 2: l2 ball, first half: A_x, w -> scratch, partial ||w/sigma + shift||^2
// PHASE 3: l2 ball, second half: y_next from the scratch, partials
template <typename T, int PHASE>
__global__ __launch_bounds__(kBlock) void k_ls_rows(LsDev d) {
  __shared__ double xs[kSkinnyMaxN];
  __shared__ LsCtrl lc;
  __shared__ SkinnyRed wred[kBlock / 64];
  const int tid = threadIdx.x;
  const T* __restrict__ A = (const T*)d.A;
  const int i = blockIdx.x * kBlock + tid;
  const bool row_ok = i < d.m;
  const int ic = row_ok ? i : d.m - 1;
  T* __restrict__ Y = (T*)d.Y;
  T* __restrict__ YT = (T*)d.YT;
  [[maybe_unused]] double h_pre = 0.0;

  if constexpr (PHASE == 0) {
    if (tid < d.n) xs[tid] = (double)((const T*)d.X)[tid];
    __syncthreads();
  } else {
    // the control block is only written by k_ls_close: block-uniform.  Every load of this part goes out before
    // the branch on it (each dependent round trip to memory costs 1-2 us here)
    const bool dead = d.c->done != 0 || d.c->it >= d.c->maxit;
    if constexpr (PHASE == 3) {
      // ||w/sigma + shift|| from the first half's block partials, summed by every block itself
      if (tid == 0) lc = *d.cs;
      double z4[4];
      skinny_partial_norm_issue(d.part, d.ldpart, d.n + 2, (int)gridDim.x, z4);
      if (dead) return;
      h_pre = skinny_partial_norm_finish(d.part, d.ldpart, d.n + 2, (int)gridDim.x, z4);   // (lc is visible behind it)
    } else {
      const bool mine = tid < d.n;
      const T x_t = mine ? ((const T*)d.X)[tid] : (T)0, v_t = mine ? ((const T*)d.V)[tid] : (T)0,
              xp_t = mine ? ((const T*)d.XP)[tid] : (T)0;
      const double aty_t = mine ? d.aty[tid] : 0.0;
      if (tid == 0) lc = *d.c;
      if (dead) return;
      __syncthreads();
      const bool fresh = lc.phase == 0;
      // ---- n side (redundantly per block): only the first trial of an iteration has work here ----------
      double q[3] = {0, 0, 0};
      if (mine) {
        const double xj = (double)x_t;
        xs[tid] = xj;
        if (fresh) {
          const double gam = lc.gamma;
          const double pr = ((double)v_t - xj) / gam + 0.0 + (double)(T)aty_t;   // :505
          const double dx = xj - (double)xp_t;
          q[0] = pr * pr;
          q[1] = dx * dx;
          q[2] = gval_elem<T>(d.g, xj, tid);
        }
      }
      block_sums_t0<3>(q);
      if (tid == 0) {
        int32_t fl = 0;
        const double t = lc.t, gam = lc.gamma;
        if (fresh) {
          lc.npr2 = q[0];
          lc.g_x = gval_final(d.g, q[2]);
          // grad f == 0: ||dg|| = <dg, dx> = 0  (:507-508)
          double C = 0.0 / 0.0;
          if (isnan(C)) C = 0.0;
          const double ndx = sqrt(q[1]);
          double L = 0.0 / (ndx * ndx);
          if (isnan(L)) L = 0.0;
          lc.Delta = gam * L * (gam * C - 1.0);                                              // :509
          const double xi_bar = (t * t) * (gam * gam) * (lc.eta * lc.eta) * (lc.delta1 * lc.delta1);   // :510
          lc.m4xim1 = 1.0 - 4.0 * xi_bar;                                                    // :511
          lc.eta = lc.R * lc.eta;                                                            // :513
        }
        const double tg = t * lc.eta * gam;
        const double g1 = gam * ls_sqrt(1.0 + gam / lc.gamma_prev, &fl);
        const double g2 = 1.0 / (2.0 * lc.Theta * t * lc.eta);
        const double g3 = gam * ls_sqrt(lc.m4xim1 / (2.0 * lc.delta1 *
                                                     (lc.Delta + ls_sqrt(lc.Delta * lc.Delta + lc.m4xim1 * (tg * tg), &fl))), &fl);
        lc.gamma_next = ls_min2(ls_min2(g1, g2), g3);                                       // :517-521
        lc.rho = lc.gamma_next / gam;                                                        // :522
        lc.sigma_next = t * t * lc.gamma_next;                                               // :523
        lc.flags |= fl;
        if (blockIdx.x == 0) *d.cs = lc;        // staged for k_ls_close (and for the second l2 row kernel)
      }
      __syncthreads();
    }
  }

  // ---- this thread's row ------------------------------------------------------------------------
  double a = 0.0, yv = 0.0, yn = 0.0, sdr = 0.0, shv = 0.0, sz = 0.0, sdy = 0.0;
  if constexpr (PHASE == 0) {
    a = skinny_row_dot<T>(A, d.ld, ic, d.n, xs);
    if (row_ok) d.axc[i] = a;
    yn = row_ok ? (double)Y[ic] : 0.0;
  } else {
    if constexpr (PHASE != 3) {
      if (lc.ycommit != 0 && row_ok) Y[i] = YT[i];          // the y accepted by the previous tick
      if (lc.phase == 0) {                                   // first trial: A_x_prev = A_x (:546), new A_x (:502)
        a = skinny_row_dot<T>(A, d.ld, ic, d.n, xs);
        if (row_ok) {
          d.axp[i] = d.axc[i];
          d.axc[i] = a;
        }
      } else {
        a = d.axc[ic];
      }
    } else {
      a = d.axc[ic];
    }
    yv = (double)Y[ic];
    const double sigma = lc.sigma_next, rho = lc.rho;
    const double b = shift_at<T>(d.h, ic);
    double w;
    if constexpr (PHASE == 3) w = d.wbuf[ic];
    else w = yv + sigma * ((1.0 + rho) * a - rho * d.axp[ic]);                            // :524
    if constexpr (PHASE == 2) {
      if (row_ok) {
        d.wbuf[i] = w;
        const double z = w / sigma + b;
        sz = z * z;
      }
    } else {
      const T ynt = (T)prox_conj_elem_s<T>(d.h, w, sigma, ic, h_pre, b);   // :525
      if (row_ok) {
        yn = (double)ynt;
        YT[i] = ynt;
        const double dr = (w - yn) / sigma - a;                                             // :535
        sdr = dr * dr;
        shv = gval_elem_s<T>(d.h, a, ic, b);
        const double dy = yn - yv;
        sdy = dy * dy;
      }
    }
  }

  // ---- block partials: A'y_next, ||dual_res||^2, h(A_x), ||z||^2, ||y_next - y||^2 -----------------------
  const double ex[kSkinnyExtra] = {sdr, shv, sz, sdy};
  const double tot = skinny_block_partials<T>(A, d.ld, ic, d.n, yn, PHASE != 2, ex, wred);
  if (tid < d.n + kSkinnyExtra) {
    const bool is_z = tid == d.n + 2;
    if (PHASE == 2 ? is_z : (PHASE == 3 ? !is_z : true)) d.part[(int64_t)blockIdx.x * d.ldpart + tid] = tot;
  }
}

// MODE 0: pre-loop  v = x0 - gamma A'y0 ; x1 = prox(v)  (:497-499)
// MODE 1: the linesearch test and, when it passes, the rest of the iteration
template <typename T, int MODE>
__global__ __launch_bounds__(kSkinnyCloseBlock) void k_ls_close(LsDev d, int B) {
  __shared__ SkinnyRed red[16];
  __shared__ LsCtrl lc;
  __shared__ double fin[6];
  const int tid = threadIdx.x;
  // every load first (the flags, the partials, A'y, x, the control block), then the branch
  const bool dead = MODE != 0 && (d.c->done != 0 || d.c->it >= d.c->maxit);
  SkinnyColLoads L;
  skinny_colsum_issue<kSkinnyCloseBlock>(d.part, d.ldpart, d.n + kSkinnyExtra, d.npow, B, L);
  T* __restrict__ X = (T*)d.X;
  const double aty_t = MODE == 1 && tid < d.n ? d.aty[tid] : 0.0;
  const T x_t = tid < d.n ? X[tid] : (T)0;
  if (tid == 0) lc = MODE == 1 ? *d.cs : *d.c;
  if (dead) return;
  const double tot = skinny_colsum_finish(d.part, d.ldpart, d.n + kSkinnyExtra, B, L, red);
  {
    if (tid >= d.n && tid < d.n + kSkinnyExtra) fin[tid - d.n] = tot;
    // ||A'y_next - A'y||^2 over the n columns (:527)
    double q[1] = {0.0};
    if (MODE == 1 && tid < d.n) {
      const double da = (double)(T)tot - (double)(T)aty_t;
      q[0] = da * da;
    }
    block_sums_t0<1, kSkinnyCloseBlock>(q);
    if (tid == 0) {
      if constexpr (MODE == 1) {
        const double ratio = sqrt(q[0]) / sqrt(fin[3]);
        lc.trials_total += 1;
        if (lc.eta >= ratio) {                                     // :527  accepted
          lc.gamma_prev = lc.gamma;                                // :528
          lc.gamma = lc.gamma_next;
          lc.sigma = lc.sigma_next;
          lc.ycommit = 1;                                          // :529  y = y_next (copied by the next row kernel)
          lc.trials_this = 0;
          lc.phase = 0;
          lc.it += 1;
          const double npr = sqrt(lc.npr2), ndr = sqrt(fin[0]);    // :535
          lc.norm_res = sqrt(npr * npr + ndr * ndr);               // :536
          lc.objective = 0.0 + lc.g_x + gval_final(d.h, fin[1]);
          if (d.recs != nullptr && lc.it <= lc.rec_cap) {          // :538-540
            apx_record r;
            r.it = lc.it;
            r.gamma = lc.gamma;
            r.sigma = lc.sigma;
            r.norm_res = lc.norm_res;
            r.objective = lc.objective;
            r.f_x = 0.0;
            r.f_evals = lc.it + 1;
            r.grad_evals = 1 + lc.trials_total;                    // A' * y_next products so far (+ the one at :496)
            d.recs[lc.it - 1] = r;
          }
          if (lc.norm_res <= lc.tol) lc.done = 1;                  // :541-543
          fin[4] = 1.0;
        } else {
          lc.eta *= lc.r;                                          // :532
          lc.ycommit = 0;
          lc.phase = 1;
          lc.trials_this += 1;
          if (lc.trials_this >= lc.max_ls) lc.done = 2;            // (the reference would spin forever)
          fin[4] = 0.0;
        }
        *d.c = lc;
      } else {
        fin[4] = 1.0;
      }
    }
    __syncthreads();
    if (fin[4] == 0.0) return;                   // rejected: nothing moves
    if (tid < d.n) {
      const T aty = (T)tot;
      d.aty[tid] = (double)aty;                  // :529  At_y = At_y_next
      if (lc.done == 0) {                        // :545-547
        const double gamma = lc.gamma;
        const double xj = (double)x_t;
        const T vj = (T)(xj - gamma * (0.0 + (double)aty));
        const T xn = (T)prox_elem<T>(d.g, (double)vj, gamma, tid, 0.0);
        ((T*)d.XP)[tid] = x_t;
        X[tid] = xn;
        ((T*)d.V)[tid] = vj;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------
bool skinny_ls_eligible(const apx_problem* p) {
  if (!p->ctx->fused_small) return false;
  if (p->A_lin == nullptr || p->sharded() || p->f_kind != APX_F_ZERO) return false;
  if (no_column_storage(p->A_lin) || p->ctx->skip_zero_cols) return false;
  const int64_t m = p->A_lin->m;
  return p->n <= kSkinnyMaxN && m > 0 && m <= 256 * 1024 && p->n * 4 <= m;
}

template <int PHASE>
static void ls_rows(const LsDev& d, int32_t dtype, int B, hipStream_t q) {
  if (dtype == APX_F32) hipLaunchKernelGGL((k_ls_rows<float, PHASE>), dim3(B), dim3(kBlock), 0, q, d);
  else hipLaunchKernelGGL((k_ls_rows<double, PHASE>), dim3(B), dim3(kBlock), 0, q, d);
}
template <int MODE>
static void ls_close(const LsDev& d, int32_t dtype, int B, hipStream_t q) {
  if (dtype == APX_F32) hipLaunchKernelGGL((k_ls_close<float, MODE>), dim3(1), dim3(kSkinnyCloseBlock), 0, q, d, B);
  else hipLaunchKernelGGL((k_ls_close<double, MODE>), dim3(1), dim3(kSkinnyCloseBlock), 0, q, d, B);
}

int32_t skinny_ls_run(apx_problem* p, const apx_linesearch_params* prm, double gamma, const void* x0, const void* y0,
                      double tol, int64_t maxit, const apx_solve_opts* opts, void* x_out, void* y_out, apx_result* res) {
  apx_ctx* ctx = p->ctx;
  const apx_matrix* A = p->A_lin;
  const size_t es = p->es();
  const size_t nb = (size_t)p->n_pad * es, mb = (size_t)p->m_pad * es, md = (size_t)p->m_pad * sizeof(double);
  const int B = (int)((A->m + kBlock - 1) / kBlock);
  DBuf bx, bxp, bv, baty, by, byt, baxc, baxp, bw, bpart, bc, bcs, brecs;
  LsDev d{};
  d.A = A->d;
  d.ld = A->ld;
  d.m = (int32_t)A->m;
  d.n = (int32_t)p->n;
  d.npow = skinny_npow(d.n);
  d.ldpart = round_up(d.n + kSkinnyExtra, 8);
  for (DBuf* b : {&bx, &bxp, &bv}) APX_TRY(b->alloc_zero(nb, ctx->stream));
  APX_TRY(baty.alloc_zero((size_t)p->n_pad * sizeof(double), ctx->stream));
  for (DBuf* b : {&by, &byt}) APX_TRY(b->alloc_zero(mb, ctx->stream));
  for (DBuf* b : {&baxc, &baxp, &bw}) APX_TRY(b->alloc_zero(md, ctx->stream));
  APX_TRY(bpart.alloc_zero((size_t)B * d.ldpart * sizeof(double), ctx->stream));
  APX_TRY(bc.alloc_zero(sizeof(LsCtrl), ctx->stream));
  APX_TRY(bcs.alloc_zero(sizeof(LsCtrl), ctx->stream));
  apx_record* host_recs = opts ? opts->records : nullptr;
  const int64_t rec_cap = (opts && opts->records) ? opts->record_cap : 0;
  if (rec_cap > 0) APX_TRY(brecs.alloc_zero((size_t)rec_cap * sizeof(apx_record), ctx->stream));
  d.X = bx.p;
  d.XP = bxp.p;
  d.V = bv.p;
  d.aty = (double*)baty.p;
  d.Y = by.p;
  d.YT = byt.p;
  d.axc = (double*)baxc.p;
  d.axp = (double*)baxp.p;
  d.wbuf = (double*)bw.p;
  d.part = (double*)bpart.p;
  d.c = (LsCtrl*)bc.p;
  d.cs = (LsCtrl*)bcs.p;
  d.g = p->gd;
  d.h = p->hd;
  d.recs = (apx_record*)brecs.p;

  LsCtrl c;
  memset(&c, 0, sizeof(c));
  c.t = prm->t;
  c.delta1 = 1.0 + prm->delta;                     // :490
  c.Theta = prm->Theta;
  c.r = prm->r;
  c.R = prm->R;
  c.tol = tol;
  c.maxit = maxit;
  c.max_ls = prm->max_linesearch > 0 ? prm->max_linesearch : 10000;
  c.rec_cap = rec_cap;
  c.eta = prm->eta;
  c.gamma = gamma;
  c.gamma_prev = gamma;                            // :491
  c.sigma = prm->t * prm->t * gamma;
  c.norm_res = c.objective = std::numeric_limits<double>::quiet_NaN();
  APX_TRY(h2d(ctx, bc.p, &c, sizeof(c)));
  APX_TRY(h2d(ctx, bx.p, x0, (size_t)p->n * es));
  APX_TRY(h2d(ctx, by.p, y0, (size_t)p->m * es));
  const int32_t dt = p->dtype;
  hipStream_t q = ctx->stream;
  ls_rows<0>(d, dt, B, q);                          // :494, :496
  ls_close<0>(d, dt, B, q);                         // :497-499
  APX_HIP(hipGetLastError());

  const bool l2 = d.h.kind == APX_G_L2;
  auto tick = [&]() {
    if (l2) {
      ls_rows<2>(d, dt, B, q);
      ls_rows<3>(d, dt, B, q);
    } else {
      ls_rows<1>(d, dt, B, q);
    }
    ls_close<1>(d, dt, B, q);
  };
  // enqueue ticks in chunks and look at the counters between chunks; a chunk never needs to be longer than
  // the iterations still missing (every iteration takes at least one tick)
  while (maxit > 0) {
    const int64_t missing = maxit - c.it;
    const int chunk = (int)std::min<int64_t>(64, std::max<int64_t>(missing, 1));
    for (int k = 0; k < chunk; ++k) tick();
    APX_HIP(hipGetLastError());
    APX_TRY(d2h(ctx, &c, bc.p, sizeof(c)));
    if (c.done != 0 || c.it >= maxit) break;
  }
  if (maxit == 0) APX_TRY(d2h(ctx, &c, bc.p, sizeof(c)));
  if (c.done == 2)
    return fail(APX_ERR_UNSUPPORTED, "AdaPDM+ linesearch did not terminate (the reference would spin forever)");

  APX_TRY(d2h(ctx, x_out, bx.p, (size_t)p->n * es));
  if (y_out != nullptr) APX_TRY(d2h(ctx, y_out, c.ycommit ? byt.p : by.p, (size_t)p->m * es));
  const bool converged = c.done == 1;
  const int64_t k = c.it;
  if (host_recs != nullptr && rec_cap > 0)
    APX_TRY(d2h(ctx, host_recs, brecs.p, (size_t)std::min<int64_t>(k, rec_cap) * sizeof(apx_record)));
  if (res != nullptr) {
    memset(res, 0, sizeof(*res));
    res->numit = k;
    res->converged = converged ? 1 : 0;
    res->flags = c.flags;
    res->gamma = c.gamma;
    res->sigma = c.sigma;
    res->norm_res = c.norm_res;
    res->objective = c.objective;
    res->f_evals = k + 1;
    res->grad_f_evals = k + 1;
    res->prox_g_evals = converged ? k : k + 1;
    res->prox_h_evals = c.trials_total;
    res->A_evals = k + 1;
    res->At_evals = 1 + c.trials_total;
    res->n_records = std::min<int64_t>(k, rec_cap);
    if (c.flags & APX_FLAG_DOMAIN) {
      set_error("sqrt of a negative number in the AdaPDM+ step-size formula (Julia would throw DomainError)");
      return APX_ERR_DOMAIN;
    }
  }
  return APX_OK;
}

}  // namespace apx
