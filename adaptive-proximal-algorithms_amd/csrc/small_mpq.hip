// small_mpq.hip — Malitsky-Pock with linesearch (`malitsky_pock` + `backtrack_stepsize_MP`, src/AdaProx.jl:555-629)
// as a DEVICE-RESIDENT loop for f = Quadratic(Q, q) with a FEW linear constraints: the dual SVM of the reference
// (experiments/dual_svm/runme.jl:47-90: Q is N x N, A = y' is 1 x N, g = IndBox, h = IndZero), which runs this method
// for twelve values of t per dataset.  The host-driven form reads the linesearch scalars back per trial
// (142 us per iteration at N = 1243, 344 us at mushrooms' N = 8124); here one linesearch trial is one TICK of two
// launches (29 / 230 us per iteration):
//
//   k_mpq_a (many blocks, each redoes the O(n) part): theta, gamma, A'ybar, v, the trial x+ = prox_{gamma g}(v)
//           (:556-560) with ||x+ - x||^2 and <grad_x, x+ - x> ; then the two products as in-block dot products:
//           Q x+ (n dots over the rows of Q) and A x+ (m dots over the rows of A)
//   k_mpq_b (one block): f(x+) = 0.5 <x+, Q x+> + <x+, q> ; the test lhs > 0.95 ||x+ - x||^2 (:563-564) ;
//           rejected: sigma /= 2 (:565) ; accepted: grad f(x+) = Q x+ + q (:612, no second product), residuals,
//           record, termination (:616-626), x = x+ ; and the dual step of the next iteration (:600-609):
//           w = y + sigma A x ; y = prox_{sigma h*}(w) ; A'y (m <= 64 terms per entry) ; sigma *= sqrt(1 + theta) ;
//           f(x), grad f(x) from the Q x already at hand
//
// One writer of the control block (k_mpq_b); k_mpq_a only reads it.  The host enqueues ticks in chunks and reads
// the counters between chunks.
#include <cstdlib>
#include <cstring>
#include <limits>
#include <memory>

#include "small_dev.h"
#include "solver_internal.h"
#include "vecops_dev.h"

namespace apx {

constexpr int kMpqMaxM = 64;          // rows of A: A'y is formed entry by entry in the one-block kernel
constexpr int kMpqBlockB = 1024;
constexpr int kMpqMaxN = 8192;        // (mushrooms' dual SVM is 8124 x 8124)
constexpr int kMpqPerB = kMpqMaxN / kMpqBlockB;      // n-entries per thread of the one-block kernel

struct MpqCtrl {
  double t, tol, theta_outer;
  int64_t maxit, rec_cap;
  double sigma, sigma_prev, gamma, f_prev;
  double norm_res, objective;
  int64_t it, trials_total;
  int32_t done, flags, halt, _pad;
};

struct MpqDev {
  const void* QT;      // rows of Q contiguous (column-major Q')
  const void* AT;      // rows of A contiguous (column-major A')
  const void* A;       // A itself, column-major m x n
  int64_t ldQT, ldAT, ldA;
  int32_t m, n, w, _pad;
  const void* qv;      // q
  void *XP, *XT, *V;   // x (last accepted) ; trial x+ ; v of the trial
  void* Y;
  double *GP;          // grad f(x) = Q x + q, entries rounded to T
  double *QX, *QXP;    // Q x+ as the dots leave it ; Q x (rounded to T)
  double *AX, *AXP;    // A x+ as the dots leave it ; A x (rounded to T)
  double *W;           // w of the dual step
  double *ATY, *ATYP;  // A'y, A'y_prev (rounded to T)
  double* sums;        // ||x+ - x||^2, <grad_x, x+ - x> of the trial in XT
  MpqCtrl* c;
  ProxDev g, h;
  apx_record* recs;
};

// MODE 0: pre-loop: only the products at XT = x0 (:597 A x0 ; Q x0 for the first eval_with_gradient, :608)
// MODE 1: a trial
template <typename T, int MODE>
__global__ __launch_bounds__(kBlock) void k_mpq_a(MpqDev d) {
  extern __shared__ double xs[];                    // the point whose products are formed: n doubles (dynamic: up to 64 KB)
  const int tid = threadIdx.x;
  if constexpr (MODE == 0) {
    const T* __restrict__ X0 = (const T*)d.XT;
    for (int i = tid; i < d.n; i += kBlock) xs[i] = (double)X0[i];
    __syncthreads();
  } else {
    if (d.c->halt != 0) return;                       // written by k_mpq_b of the previous tick: block-uniform
    const double sigma = d.c->sigma, sigma_prev = d.c->sigma_prev, t = d.c->t;
    const double theta = sigma / sigma_prev;                               // :556 / :569
    const double gamma = t * t * sigma;                                    // :557 / :570
    const T* __restrict__ XP = (const T*)d.XP;
    T* __restrict__ XT = (T*)d.XT;
    T* __restrict__ V = (T*)d.V;
    double q[2] = {0.0, 0.0};
    const int per = (d.n + kBlock - 1) / kBlock;
    for (int k0 = 0; k0 < per; k0 += 4) {
      double x4[4], a4[4], ap4[4], g4[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int ic = min(tid + (k0 + u) * kBlock, d.n - 1);
        x4[u] = (double)XP[ic];
        a4[u] = d.ATY[ic];
        ap4[u] = d.ATYP[ic];
        g4[u] = d.GP[ic];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int k = k0 + u;
        const int i = tid + k * kBlock;
        if (i < d.n) {
          const double ybar = (1.0 + theta) * a4[u] - theta * ap4[u];      // :558
          const T vi = (T)(x4[u] - gamma * (ybar + g4[u]));                // :559
          const T xi = (T)prox_elem<T>(d.g, (double)vi, gamma, i, 0.0);    // :560
          const double dd = (double)xi - x4[u];
          q[0] = fma(dd, dd, q[0]);
          q[1] = fma(g4[u], dd, q[1]);
          xs[i] = (double)xi;
          if ((unsigned)k % gridDim.x == blockIdx.x) {                     // chunk k is written by one block
            V[i] = vi;
            XT[i] = xi;
          }
        }
      }
    }
    block_sums_t0<2>(q);                              // (barrier inside: xs is complete behind it)
    if (blockIdx.x == 0 && tid == 0) {
      d.sums[0] = q[0];
      d.sums[1] = q[1];
    }
  }
  small_dots_w<T>(d.w, (const T*)d.QT, d.ldQT, d.n, d.n, xs, d.QX, true);           // temp = f.Q * x  (:25 of the experiment)
  small_dots_w<T>(d.w, (const T*)d.AT, d.ldAT, d.m, d.n, xs, d.AX, true);           // A * x  (:561)
}

// the dual step of the next iteration (:600-609) by the one block: everything it needs is final in global memory
// (XP, QXP, AXP, Y) except what this block itself wrote — hence the barriers around it in the caller
template <typename T>
__device__ __forceinline__ void mpq_dual_step(const MpqDev& d, MpqCtrl& lc, double* ys) {
  const int tid = threadIdx.x;
  const double sigma = lc.sigma;
  T* __restrict__ Y = (T*)d.Y;
  if (tid < d.m) {
    const double b = shift_at<T>(d.h, tid);
    const T w = (T)((double)Y[tid] + sigma * d.AXP[tid]);                                // :601
    const T yn = (T)prox_conj_elem_s<T>(d.h, (double)w, sigma, tid, 0.0, b);             // :602
    d.W[tid] = (double)w;
    Y[tid] = yn;
    ys[tid] = (double)yn;
  }
  __syncthreads();
  const T* __restrict__ A = (const T*)d.A;
  const T* __restrict__ XP = (const T*)d.XP;
  const T* __restrict__ QV = (const T*)d.qv;
  double q[2] = {0.0, 0.0};
#pragma unroll
  for (int k = 0; k < kMpqPerB; ++k) {
    const int j = tid + k * kMpqBlockB;
    if (j < d.n) {
      double s = 0.0;
      for (int i = 0; i < d.m; ++i) s = fma((double)A[(int64_t)j * d.ldA + i], ys[i], s);
      d.ATYP[j] = d.ATY[j];                                                              // :600
      d.ATY[j] = (double)(T)s;                                                           // :603
      const double xj = (double)XP[j], tj = d.QXP[j], qj = (double)QV[j];
      q[0] = fma(xj, tj, q[0]);                                                          // :608  f_x_prev, grad_x_prev
      q[1] = fma(xj, qj, q[1]);
      d.GP[j] = (double)(T)(tj + qj);
    }
  }
  block_sums_t0<2, kMpqBlockB>(q);
  if (tid == 0) {
    lc.f_prev = 0.5 * q[0] + q[1];
    lc.sigma_prev = sigma;                                                               // :605
    lc.sigma = sigma * sqrt(1.0 + lc.theta_outer);                                       // :606
  }
}

// MODE 0: pre-loop (:597-598) and the dual step of iteration 1 ; MODE 1: close a trial
template <typename T, int MODE>
__global__ __launch_bounds__(kMpqBlockB) void k_mpq_b(MpqDev d) {
  __shared__ MpqCtrl lc;
  __shared__ double ys[kMpqMaxM];
  __shared__ int accept;
  const int tid = threadIdx.x;
  const T* __restrict__ XT = (const T*)d.XT;
  const T* __restrict__ QV = (const T*)d.qv;
  T* __restrict__ XP = (T*)d.XP;
  if (MODE == 1 && d.c->halt != 0) return;
  if (tid == 0) lc = *d.c;
  __syncthreads();
  if constexpr (MODE == 0) {
    // A_x = A x0 ; Q x0 ; At_y = A' y0 lands in ATY through the dual step's shift — so form it first
    if (tid < d.m) {
      d.AXP[tid] = (double)(T)d.AX[tid];                                                 // :597
      ys[tid] = (double)((const T*)d.Y)[tid];
    }
    __syncthreads();
    const T* __restrict__ A = (const T*)d.A;
#pragma unroll
    for (int k = 0; k < kMpqPerB; ++k) {
      const int j = tid + k * kMpqBlockB;
      if (j < d.n) {
        double s = 0.0;
        for (int i = 0; i < d.m; ++i) s = fma((double)A[(int64_t)j * d.ldA + i], ys[i], s);
        d.ATY[j] = (double)(T)s;                                                         // :598
        d.QXP[j] = (double)(T)d.QX[j];
      }
    }
    __syncthreads();
    if (lc.maxit > 0) mpq_dual_step<T>(d, lc, ys);
    if (tid == 0) {
      lc.halt = lc.maxit > 0 ? 0 : 1;
      *d.c = lc;
    }
    return;
  } else {
    // ---- f(x+), ||A x+ - A x||^2 -----------------------------------------------------------------------
    double xv[kMpqPerB], tv[kMpqPerB], qv[kMpqPerB];
    double q[3] = {0.0, 0.0, 0.0};
#pragma unroll
    for (int k = 0; k < kMpqPerB; ++k) {
      const int j = tid + k * kMpqBlockB;
      const int jc = min(j, d.n - 1);
      xv[k] = (double)XT[jc];
      tv[k] = (double)(T)d.QX[jc];
      qv[k] = (double)QV[jc];
      if (j < d.n) {
        q[0] = fma(xv[k], tv[k], q[0]);
        q[1] = fma(xv[k], qv[k], q[1]);
      }
    }
    double axn = 0.0;
    if (tid < d.m) {
      axn = (double)(T)d.AX[tid];
      const double da = axn - d.AXP[tid];
      q[2] = da * da;
    }
    block_sums_t0<3, kMpqBlockB>(q);
    if (tid == 0) {
      const double f_x = 0.5 * q[0] + q[1];
      const double gamma = lc.t * lc.t * lc.sigma;
      const double ndx = sqrt(d.sums[0]), ndax = sqrt(q[2]);
      const double lhs = gamma * lc.sigma * (ndax * ndax) + 2.0 * gamma * (f_x - lc.f_prev - d.sums[1]);     // :563
      lc.gamma = gamma;
      lc.trials_total += 1;
      accept = (!(lhs > 0.95 * (ndx * ndx)) || lc.sigma == 0.0) ? 1 : 0;
      if (!accept) {
        lc.sigma /= 2;                                                                   // :565
        if (lc.sigma < 1e-12) lc.flags |= APX_FLAG_TINY_GAMMA;                           // :566-568 (logs, continues)
        *d.c = lc;
      } else {
        lc.objective = f_x;                           // (g(x) and h(A_x) are added below)
      }
    }
    __syncthreads();
    if (!accept) return;
    // ---- accepted: residuals (:616-618), record, termination, x = x+ --------------------------------------
    const double gamma = lc.gamma;
    const T* __restrict__ V = (const T*)d.V;
    double r[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int k = 0; k < kMpqPerB; ++k) {
      const int j = tid + k * kMpqBlockB;
      if (j < d.n) {
        const double gx = (double)(T)(tv[k] + qv[k]);                                    // :612  grad_x = pb()
        double pr = ((double)V[j] - xv[k]) / gamma + gx;                                 // :616
        pr = pr + d.ATY[j];
        r[0] = fma(pr, pr, r[0]);
        r[2] += gval_elem<T>(d.g, xv[k], j);
        XP[j] = (T)xv[k];                                                                // x = x+
        d.QXP[j] = tv[k];
      }
    }
    if (tid < d.m) {
      const double b = shift_at<T>(d.h, tid);
      const double dr = (d.W[tid] - (double)((const T*)d.Y)[tid]) / lc.sigma_prev - axn; // :617
      r[1] = dr * dr;
      r[3] = gval_elem_s<T>(d.h, axn, tid, b);
      d.AXP[tid] = axn;
    }
    block_sums_t0<4, kMpqBlockB>(r);
    if (tid == 0) {
      const double npr = sqrt(r[0]), ndr = sqrt(r[1]);
      lc.norm_res = sqrt(npr * npr + ndr * ndr);                                         // :618
      const double f_x = lc.objective;
      lc.objective = f_x + gval_final(d.g, r[2]) + gval_final(d.h, r[3]);
      lc.it += 1;
      if (d.recs != nullptr && lc.it <= lc.rec_cap) {                                    // :620-622
        apx_record rr;
        rr.it = lc.it;
        rr.gamma = lc.gamma;
        rr.sigma = lc.sigma;
        rr.norm_res = lc.norm_res;
        rr.objective = lc.objective;
        rr.f_x = f_x;
        rr.f_evals = lc.it + lc.trials_total;                                            // one per iteration (:608) + one per trial
        rr.grad_evals = 2 * lc.it;                                                       // :608, :612
        d.recs[lc.it - 1] = rr;
      }
      if (lc.norm_res <= lc.tol) lc.done = 1;                                            // :624-626
      lc.halt = (lc.done != 0 || lc.it >= lc.maxit) ? 1 : 0;
    }
    __syncthreads();                                  // (XP, QXP, AXP of this block are visible to it behind the barrier)
    if (lc.halt == 0) mpq_dual_step<T>(d, lc, ys);
    if (tid == 0) *d.c = lc;
  }
}

// ---------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------
bool small_mpq_eligible(const apx_problem* p) {
  if (!p->ctx->fused_small) return false;
  if (p->A_lin == nullptr || p->sharded() || p->f_kind != APX_F_QUADRATIC) return false;
  if (p->A_f == nullptr || no_column_storage(p->A_f) || no_column_storage(p->A_lin) || p->ctx->skip_zero_cols) return false;
  if (p->hd.kind == APX_G_L2) return false;           // (its conjugate prox needs a norm over all rows first)
  return p->n <= kMpqMaxN && p->n > 0 && p->A_lin->m > 0 && p->A_lin->m <= kMpqMaxM;
}

template <typename T>
static void mpq_tick(const MpqDev& d, int mode, unsigned grid, hipStream_t q) {
  const size_t lds = m_vector_lds_bytes(d.n);
  if (mode == 0) {
    allow_big_lds<k_mpq_a<T, 0>>();
    hipLaunchKernelGGL((k_mpq_a<T, 0>), dim3(grid), dim3(kBlock), lds, q, d);
    hipLaunchKernelGGL((k_mpq_b<T, 0>), dim3(1), dim3(kMpqBlockB), 0, q, d);
  } else {
    allow_big_lds<k_mpq_a<T, 1>>();
    hipLaunchKernelGGL((k_mpq_a<T, 1>), dim3(grid), dim3(kBlock), lds, q, d);
    hipLaunchKernelGGL((k_mpq_b<T, 1>), dim3(1), dim3(kMpqBlockB), 0, q, d);
  }
}

static int32_t transposed_copy(apx_ctx* ctx, const apx_matrix* M, DBuf& buf, const void** out, int64_t* ld) {
  if (M->dT != nullptr) {
    *out = M->dT;
    *ld = M->ldT;
    return APX_OK;
  }
  *ld = round_up(M->n, kPad);                         // private copy: the caller's matrix object is untouched
  APX_TRY(buf.alloc_zero((size_t)*ld * (size_t)M->ld * dtype_size(M->dtype), ctx->stream));
  APX_TRY(transpose_into(ctx, M, buf.p, *ld));
  *out = buf.p;
  return APX_OK;
}

int32_t small_mpq_run(apx_problem* p, const void* x0, const void* y0, double sigma, double t, double tol, int64_t maxit,
                      const apx_solve_opts* opts, void* x_out, void* y_out, apx_result* res) {
  apx_ctx* ctx = p->ctx;
  const apx_matrix *Q = p->A_f, *A = p->A_lin;
  const size_t es = p->es();
  const size_t nb = (size_t)p->n_pad * es, nd = (size_t)p->n_pad * sizeof(double);
  const size_t mb = (size_t)p->m_pad * es, md = (size_t)p->m_pad * sizeof(double);
  DBuf bqt, bat, bxp, bxt, bv, by, bgp, bqx, bqxp, bax, baxp, bw, baty, batyp, bsums, bc, brecs;
  MpqDev d{};
  APX_TRY(transposed_copy(ctx, Q, bqt, &d.QT, &d.ldQT));
  APX_TRY(transposed_copy(ctx, A, bat, &d.AT, &d.ldAT));
  d.A = A->d;
  d.ldA = A->ld;
  d.m = (int32_t)A->m;
  d.n = (int32_t)p->n;
  d.w = pick_waves(d.n);
  d.qv = p->b.p;
  for (DBuf* b : {&bxp, &bxt, &bv}) APX_TRY(b->alloc_zero(nb, ctx->stream));
  for (DBuf* b : {&bgp, &bqx, &bqxp, &baty, &batyp}) APX_TRY(b->alloc_zero(nd, ctx->stream));
  APX_TRY(by.alloc_zero(mb, ctx->stream));
  for (DBuf* b : {&bax, &baxp, &bw}) APX_TRY(b->alloc_zero(md, ctx->stream));
  APX_TRY(bsums.alloc_zero(2 * sizeof(double), ctx->stream));
  APX_TRY(bc.alloc_zero(sizeof(MpqCtrl), ctx->stream));
  apx_record* host_recs = opts ? opts->records : nullptr;
  const int64_t rec_cap = (opts && opts->records) ? opts->record_cap : 0;
  if (rec_cap > 0) APX_TRY(brecs.alloc_zero((size_t)rec_cap * sizeof(apx_record), ctx->stream));
  d.XP = bxp.p;
  d.XT = bxt.p;
  d.V = bv.p;
  d.Y = by.p;
  d.GP = (double*)bgp.p;
  d.QX = (double*)bqx.p;
  d.QXP = (double*)bqxp.p;
  d.AX = (double*)bax.p;
  d.AXP = (double*)baxp.p;
  d.W = (double*)bw.p;
  d.ATY = (double*)baty.p;
  d.ATYP = (double*)batyp.p;
  d.sums = (double*)bsums.p;
  d.c = (MpqCtrl*)bc.p;
  d.g = p->gd;
  d.h = p->hd;
  d.recs = (apx_record*)brecs.p;

  MpqCtrl c;
  memset(&c, 0, sizeof(c));
  c.t = t;
  c.tol = tol;
  c.theta_outer = 1.0;                              // :595 `theta = one(sigma)` (never updated in the reference)
  c.maxit = maxit;
  c.rec_cap = rec_cap;
  c.sigma = sigma;
  c.sigma_prev = sigma;
  c.gamma = c.norm_res = c.objective = c.f_prev = std::numeric_limits<double>::quiet_NaN();
  APX_TRY(h2d(ctx, bc.p, &c, sizeof(c)));
  APX_TRY(h2d(ctx, bxp.p, x0, (size_t)p->n * es));
  APX_TRY(h2d(ctx, bxt.p, x0, (size_t)p->n * es));
  APX_TRY(h2d(ctx, by.p, y0, (size_t)p->m * es));
  hipStream_t q = ctx->stream;
  const unsigned grid = pick_grid(d.n + d.m, d.w);
  const bool f32 = p->dtype == APX_F32;
  if (f32) mpq_tick<float>(d, 0, grid, q);           // :597-598 and the dual step of iteration 1
  else mpq_tick<double>(d, 0, grid, q);
  APX_HIP(hipGetLastError());
  while (maxit > 0) {                                // every iteration takes at least one tick
    const int64_t missing = maxit - c.it;
    const int chunk = (int)std::min<int64_t>(64, std::max<int64_t>(missing, 1));
    for (int k = 0; k < chunk; ++k) {
      if (f32) mpq_tick<float>(d, 1, grid, q);
      else mpq_tick<double>(d, 1, grid, q);
    }
    APX_HIP(hipGetLastError());
    APX_TRY(d2h(ctx, &c, bc.p, sizeof(c)));
    if (c.done != 0 || c.it >= maxit) break;
  }
  if (maxit == 0) APX_TRY(d2h(ctx, &c, bc.p, sizeof(c)));

  APX_TRY(d2h(ctx, x_out, bxp.p, (size_t)p->n * es));
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
