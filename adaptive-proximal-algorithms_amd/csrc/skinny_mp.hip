// skinny_mp.hip — Malitsky-Pock with linesearch (`malitsky_pock` + `backtrack_stepsize_MP`,
// src/AdaProx.jl:555-629) as a DEVICE-RESIDENT loop for tall, skinny dense A (n <= 124, f = Zero): the
// comparison method that takes 15 of the 31 runs of the reference's least_absolute_deviation and
// square_root_lasso experiments.  Same construction as skinny_ls.hip — a thread per row, a one-block closing
// kernel, the control flow as a state machine in device memory — with two kinds of tick:
//
//   dual tick   (:600-606)  rows : w = y + sigma A_x ; y = prox_{sigma h*}(w) in place ; partials of A'y
//                           close: A'y_prev = A'y ; A'y ; sigma_prev = sigma ; sigma *= sqrt(1 + theta) ;
//                                  the first primal trial (:556-560)  x+ = prox_{gamma g}(x - gamma A'ybar)
//   primal tick (:561-577)  rows : A x+ ; partials of ||A x+ - A x||^2, of the dual residual (:617), of h(A x+)
//                           close: lhs > 0.95 ||x+ - x||^2 (:564): sigma /= 2 and the next trial x+ ;
//                                  otherwise residuals, record, termination (:616-626), x = x+
//
// (for h = NormL2 the dual tick needs ||w/sigma + shift|| first: two more launches per tick, idle in primal
// ticks).  The host enqueues ticks in chunks and reads the counters between chunks.
#include <cstdlib>
#include <cstring>
#include <memory>

#include "skinny_dev.h"

namespace apx {

struct MpCtrl {
  double t, tol, theta_outer;
  int64_t maxit, rec_cap;
  double sigma, sigma_prev, gamma, ndx2, h_pre;
  double norm_res, objective;
  int64_t it, trials_total;
  int32_t phase;      // 0: dual tick next ; 1: primal tick next
  int32_t done, flags;
  int32_t axcommit;   // the accepted A x+ still sits in the trial buffer
};

struct MpDev {
  const void* A;
  int64_t ld;
  int32_t m, n, npow, _pad;
  int64_t ldpart;
  void *X, *XT, *V;            // x ; trial x+ ; v of the trial
  double *aty, *atyp;          // A'y, A'y_prev
  void* Y;
  double *w, *axc, *axt;       // w of the dual step ; A x ; trial A x+
  double* part;                // [blocks][ldpart]: n partials of A'y | ||A x+ - A x||^2, ||dual_res||^2, h(A x+), ||z||^2
  MpCtrl* c;
  ProxDev g, h;
  apx_record* recs;
};

// STAGE 0: pre-loop (:597-598): A x0 -> axc, partials of A'y0
// STAGE 1: dual tick: the whole row work (l2: only w and ||z||^2) ; primal tick: the whole row work
// STAGE 3: dual tick, l2 only: y from w, partials of A'y
template <typename T, int STAGE>
__global__ __launch_bounds__(kBlock) void k_mp_rows(MpDev d) {
  __shared__ double xs[kSkinnyMaxN];
  __shared__ SkinnyRed wred[kBlock / 64];
  const int tid = threadIdx.x;
  const T* __restrict__ A = (const T*)d.A;
  const int i = blockIdx.x * kBlock + tid;
  const bool row_ok = i < d.m;
  const int ic = row_ok ? i : d.m - 1;
  T* __restrict__ Y = (T*)d.Y;
  const bool l2 = d.h.kind == APX_G_L2;
  int phase = 0;
  if constexpr (STAGE != 0) {
    // the control block is only written by k_mp_close<1>: uniform over this launch
    if (d.c->done != 0 || d.c->it >= d.c->maxit) return;
    phase = d.c->phase;
    if (STAGE == 3 && (phase != 0 || !l2)) return;
  }
  const bool primal = STAGE == 1 && phase == 1;
  double yn = 0.0, s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  bool aty_partials = false;

  if (STAGE == 0 || primal) {           // a row of A times x0 / the trial x+
    const T* __restrict__ xsrc = (const T*)(STAGE == 0 ? d.X : d.XT);
    if (tid < d.n) xs[tid] = (double)xsrc[tid];
    __syncthreads();
    const double a = skinny_row_dot<T>(A, d.ld, ic, d.n, xs);
    if constexpr (STAGE == 0) {
      if (row_ok) d.axc[i] = a;
      yn = row_ok ? (double)Y[ic] : 0.0;
      aty_partials = true;
    } else if (row_ok) {
      const double sp = d.c->sigma_prev;
      d.axt[i] = a;
      const double da = a - d.axc[i];
      s0 = da * da;                                                     // ||A_x - A_x_prev||^2   (:563)
      const double dr = (d.w[i] - (double)Y[i]) / sp - a;               // :617
      s1 = dr * dr;
      s2 = gval_elem<T>(d.h, a, i);
    }
  } else if (STAGE == 1) {              // dual tick
    if (row_ok && d.c->axcommit != 0) d.axc[i] = d.axt[i];              // A_x of the iterate accepted by the last tick
    const double sigma = d.c->sigma;
    const double a = d.axc[ic];
    const double b = shift_at<T>(d.h, ic);
    const double w = (double)Y[ic] + sigma * a;                          // :601
    if (row_ok) d.w[i] = w;
    if (l2) {
      if (row_ok) {
        const double z = w / sigma + b;
        s3 = z * z;
      }
    } else {
      const T ynt = (T)prox_conj_elem_s<T>(d.h, w, sigma, ic, 0.0, b);   // :602
      if (row_ok) {
        Y[i] = ynt;
        yn = (double)ynt;
      }
      aty_partials = true;
    }
  } else {                              // STAGE 3: l2 ball, second half of the dual tick
    // ||w/sigma + shift|| from the first half's block partials, summed by every block itself
    double z4[4];
    skinny_partial_norm_issue(d.part, d.ldpart, d.n + 3, (int)gridDim.x, z4);
    const double sigma = d.c->sigma;
    const double b = shift_at<T>(d.h, ic);
    const double wv = d.w[ic];
    const double h_pre = skinny_partial_norm_finish(d.part, d.ldpart, d.n + 3, (int)gridDim.x, z4);
    const T ynt = (T)prox_conj_elem_s<T>(d.h, wv, sigma, ic, h_pre, b);
    if (row_ok) {
      Y[i] = ynt;
      yn = (double)ynt;
    }
    aty_partials = true;
  }

  // ---- block partials (aty_partials is uniform over the launch) -----------------------------------------
  const double ex[kSkinnyExtra] = {s0, s1, s2, s3};
  const double tot = skinny_block_partials<T>(A, d.ld, ic, d.n, yn, aty_partials, ex, wred);
  // (STAGE 3 leaves column n + 3 alone: the other blocks of its launch are still reading it)
  if (tid < d.n + kSkinnyExtra && (tid >= d.n || aty_partials) && !(STAGE == 3 && tid == d.n + 3))
    d.part[(int64_t)blockIdx.x * d.ldpart + tid] = tot;
}

// the primal trial for the current sigma (:556-560): theta, gamma, A'ybar, v, x+ ; returns ||x+ - x||^2 on thread 0
template <typename T>
__device__ __forceinline__ void mp_trial(const MpDev& d, MpCtrl& lc, double* ndx2_out) {
  const int tid = threadIdx.x;
  const double theta = lc.sigma / lc.sigma_prev;                         // :556 / :569
  const double gamma = lc.t * lc.t * lc.sigma;                           // :557 / :570
  double q[1] = {0.0};
  if (tid < d.n) {
    const double xp = (double)((const T*)d.X)[tid];
    const double ybar = (1.0 + theta) * (double)(T)d.aty[tid] - theta * (double)(T)d.atyp[tid];   // :558
    const T vi = (T)(xp - gamma * (ybar + 0.0));                         // :559
    const T xi = (T)prox_elem<T>(d.g, (double)vi, gamma, tid, 0.0);      // :560
    const double dd = (double)xi - xp;
    q[0] = dd * dd;
    ((T*)d.V)[tid] = vi;
    ((T*)d.XT)[tid] = xi;
  }
  block_sums_t0<1, kSkinnyCloseBlock>(q);
  if (tid == 0) {
    lc.gamma = gamma;
    *ndx2_out = q[0];
  }
}

// STAGE 0: pre-loop: A'y0 ; STAGE 1: close a dual or a primal tick
template <typename T, int STAGE>
__global__ __launch_bounds__(kSkinnyCloseBlock) void k_mp_close(MpDev d, int B) {
  __shared__ SkinnyRed red[16];
  __shared__ MpCtrl lc;
  __shared__ double fin[kSkinnyExtra + 2];
  const int tid = threadIdx.x;
  if constexpr (STAGE != 0) {
    if (d.c->done != 0 || d.c->it >= d.c->maxit) return;
  }
  const double tot = skinny_colsum<kSkinnyCloseBlock>(d.part, d.ldpart, d.n + kSkinnyExtra, d.npow, B, red);
  if constexpr (STAGE == 0) {
    if (tid < d.n) d.aty[tid] = (double)(T)tot;                          // :598
    return;
  } else {
    if (tid == 0) lc = *d.c;
    if (tid >= d.n && tid < d.n + kSkinnyExtra) fin[tid - d.n] = tot;
    __syncthreads();
    if (lc.phase == 0) {
      // ---- close the dual tick --------------------------------------------------------------------
      if (tid < d.n) {
        d.atyp[tid] = d.aty[tid];                                        // :600
        d.aty[tid] = (double)(T)tot;                                     // :603
      }
      __syncthreads();
      if (tid == 0) {
        lc.sigma_prev = lc.sigma;                                        // :605
        lc.sigma = lc.sigma * sqrt(1.0 + lc.theta_outer);                // :606
        lc.axcommit = 0;
      }
      __syncthreads();
      mp_trial<T>(d, lc, &fin[kSkinnyExtra]);
      if (tid == 0) {
        lc.ndx2 = fin[kSkinnyExtra];
        lc.phase = 1;
        *d.c = lc;
      }
      return;
    }
    // ---- close a primal tick: the linesearch test (:564) ---------------------------------------------
    __shared__ int accept;
    if (tid == 0) {
      const double ndx = sqrt(lc.ndx2), ndax = sqrt(fin[0]);
      const double lhs = lc.gamma * lc.sigma * (ndax * ndax) + 2.0 * lc.gamma * (0.0 - 0.0 - 0.0);   // :563 (f = Zero)
      lc.trials_total += 1;
      accept = (!(lhs > 0.95 * (ndx * ndx)) || lc.sigma == 0.0) ? 1 : 0;
      if (!accept) {
        lc.sigma /= 2;                                                   // :565
        if (lc.sigma < 1e-12) lc.flags |= APX_FLAG_TINY_GAMMA;           // :566-568 (logs, continues)
      }
    }
    __syncthreads();
    if (!accept) {
      mp_trial<T>(d, lc, &fin[kSkinnyExtra]);
      if (tid == 0) {
        lc.ndx2 = fin[kSkinnyExtra];
        *d.c = lc;
      }
      return;
    }
    // accepted: residuals (:616-618), record, termination, x = x+
    double q[2] = {0.0, 0.0};
    if (tid < d.n) {
      const double xi = (double)((const T*)d.XT)[tid];
      double pr = ((double)((const T*)d.V)[tid] - xi) / lc.gamma + 0.0;  // :616
      pr = pr + (double)(T)d.aty[tid];
      q[0] = pr * pr;
      q[1] = gval_elem<T>(d.g, xi, tid);
      ((T*)d.X)[tid] = (T)xi;
    }
    block_sums_t0<2, kSkinnyCloseBlock>(q);
    if (tid == 0) {
      const double npr = sqrt(q[0]), ndr = sqrt(fin[1]);
      lc.norm_res = sqrt(npr * npr + ndr * ndr);                         // :618
      lc.objective = 0.0 + gval_final(d.g, q[1]) + gval_final(d.h, fin[2]);
      lc.it += 1;
      lc.axcommit = 1;
      lc.phase = 0;
      if (d.recs != nullptr && lc.it <= lc.rec_cap) {                    // :620-622
        apx_record r;
        r.it = lc.it;
        r.gamma = lc.gamma;
        r.sigma = lc.sigma;
        r.norm_res = lc.norm_res;
        r.objective = lc.objective;
        r.f_x = 0.0;
        r.f_evals = lc.it + lc.trials_total;                             // one per iteration (:608) + one per trial
        r.grad_evals = 2 * lc.it;                                        // :608, :612
        d.recs[lc.it - 1] = r;
      }
      if (lc.norm_res <= lc.tol) lc.done = 1;                            // :624-626
      *d.c = lc;
    }
  }
}

// ---------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------
bool skinny_mp_eligible(const apx_problem* p) { return skinny_ls_eligible(p); }

template <int STAGE>
static void mp_rows(const MpDev& d, int32_t dtype, int B, hipStream_t q) {
  if (dtype == APX_F32) hipLaunchKernelGGL((k_mp_rows<float, STAGE>), dim3(B), dim3(kBlock), 0, q, d);
  else hipLaunchKernelGGL((k_mp_rows<double, STAGE>), dim3(B), dim3(kBlock), 0, q, d);
}
template <int STAGE>
static void mp_close(const MpDev& d, int32_t dtype, int B, hipStream_t q) {
  if (dtype == APX_F32) hipLaunchKernelGGL((k_mp_close<float, STAGE>), dim3(1), dim3(kSkinnyCloseBlock), 0, q, d, B);
  else hipLaunchKernelGGL((k_mp_close<double, STAGE>), dim3(1), dim3(kSkinnyCloseBlock), 0, q, d, B);
}

int32_t skinny_mp_run(apx_problem* p, const void* x0, const void* y0, double sigma, double t, double tol, int64_t maxit,
                      const apx_solve_opts* opts, void* x_out, void* y_out, apx_result* res) {
  apx_ctx* ctx = p->ctx;
  const apx_matrix* A = p->A_lin;
  const size_t es = p->es();
  const size_t nb = (size_t)p->n_pad * es, mb = (size_t)p->m_pad * es, md = (size_t)p->m_pad * sizeof(double);
  const int B = (int)((A->m + kBlock - 1) / kBlock);
  DBuf bx, bxt, bv, baty, batyp, by, bw, baxc, baxt, bpart, bc, brecs;
  MpDev d{};
  d.A = A->d;
  d.ld = A->ld;
  d.m = (int32_t)A->m;
  d.n = (int32_t)p->n;
  d.npow = skinny_npow(d.n);
  d.ldpart = round_up(d.n + kSkinnyExtra, 8);
  for (DBuf* b : {&bx, &bxt, &bv}) APX_TRY(b->alloc_zero(nb, ctx->stream));
  for (DBuf* b : {&baty, &batyp}) APX_TRY(b->alloc_zero((size_t)p->n_pad * sizeof(double), ctx->stream));
  APX_TRY(by.alloc_zero(mb, ctx->stream));
  for (DBuf* b : {&bw, &baxc, &baxt}) APX_TRY(b->alloc_zero(md, ctx->stream));
  APX_TRY(bpart.alloc_zero((size_t)B * d.ldpart * sizeof(double), ctx->stream));
  APX_TRY(bc.alloc_zero(sizeof(MpCtrl), ctx->stream));
  apx_record* host_recs = opts ? opts->records : nullptr;
  const int64_t rec_cap = (opts && opts->records) ? opts->record_cap : 0;
  if (rec_cap > 0) APX_TRY(brecs.alloc_zero((size_t)rec_cap * sizeof(apx_record), ctx->stream));
  d.X = bx.p;
  d.XT = bxt.p;
  d.V = bv.p;
  d.aty = (double*)baty.p;
  d.atyp = (double*)batyp.p;
  d.Y = by.p;
  d.w = (double*)bw.p;
  d.axc = (double*)baxc.p;
  d.axt = (double*)baxt.p;
  d.part = (double*)bpart.p;
  d.c = (MpCtrl*)bc.p;
  d.g = p->gd;
  d.h = p->hd;
  d.recs = (apx_record*)brecs.p;

  MpCtrl c;
  memset(&c, 0, sizeof(c));
  c.t = t;
  c.tol = tol;
  c.theta_outer = 1.0;                              // :595 `theta = one(sigma)` (never updated in the reference)
  c.maxit = maxit;
  c.rec_cap = rec_cap;
  c.sigma = sigma;
  c.sigma_prev = sigma;
  c.gamma = c.norm_res = c.objective = std::numeric_limits<double>::quiet_NaN();
  APX_TRY(h2d(ctx, bc.p, &c, sizeof(c)));
  APX_TRY(h2d(ctx, bx.p, x0, (size_t)p->n * es));
  APX_TRY(h2d(ctx, by.p, y0, (size_t)p->m * es));
  const int32_t dt = p->dtype;
  hipStream_t q = ctx->stream;
  mp_rows<0>(d, dt, B, q);                           // :597-598
  mp_close<0>(d, dt, B, q);
  APX_HIP(hipGetLastError());

  const bool l2 = d.h.kind == APX_G_L2;
  auto tick = [&]() {
    mp_rows<1>(d, dt, B, q);
    if (l2) mp_rows<3>(d, dt, B, q);
    mp_close<1>(d, dt, B, q);
  };
  while (maxit > 0) {                                // every iteration takes at least two ticks
    const int64_t missing = maxit - c.it;
    const int chunk = (int)std::min<int64_t>(96, std::max<int64_t>(2 * missing, 2));
    for (int k = 0; k < chunk; ++k) tick();
    APX_HIP(hipGetLastError());
    APX_TRY(d2h(ctx, &c, bc.p, sizeof(c)));
    if (c.done != 0 || c.it >= maxit) break;
  }
  if (maxit == 0) APX_TRY(d2h(ctx, &c, bc.p, sizeof(c)));

  APX_TRY(d2h(ctx, x_out, bx.p, (size_t)p->n * es));
  if (y_out != nullptr) APX_TRY(d2h(ctx, y_out, by.p, (size_t)p->m * es));
  const bool conv = c.done == 1;
  const int64_t k = c.it;
  if (host_recs != nullptr && rec_cap > 0)
    APX_TRY(d2h(ctx, host_recs, brecs.p, (size_t)std::min<int64_t>(k, rec_cap) * sizeof(apx_record)));
  if (res != nullptr) {
    memset(res, 0, sizeof(*res));
    res->numit = k;
    res->converged = conv ? 1 : 0;
    res->flags = c.flags;
    res->gamma = c.gamma;
    res->sigma = c.sigma;
    res->norm_res = c.norm_res;
    res->objective = c.objective;
    res->f_evals = k + c.trials_total;
    res->grad_f_evals = 2 * k;
    res->prox_g_evals = c.trials_total;
    res->prox_h_evals = k;
    res->A_evals = 1 + c.trials_total;
    res->At_evals = 1 + k;
    res->n_records = std::min<int64_t>(k, rec_cap);
  }
  return APX_OK;
}

}  // namespace apx
