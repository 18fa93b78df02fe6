// small_bt.hip — `backtracking_proxgrad` (src/AdaProx.jl:34-64) as a DEVICE-RESIDENT loop
// for small dense problems (m, n <= 4096: the reference's own lasso sizes, where it runs this method three
// times per experiment, experiments/lasso/runme.jl:96-108).  The generic implementation (solver.hip) reads
// f_z back per trial: 43 - 82 us per iteration at 500 x 1000.  Here one linesearch trial is one TICK of the
// two-kernel scheme of small.hip (every block redoes the O(m + n) work, both products are in-block dots):
//
//   k_bt_a : z = prox_{gamma g}(x - gamma grad_x) (:35 / :43) ; <grad_x, z - x>, ||z - x||^2, g(z) ; A z
//   k_bt_b : res = A z - b (or the logistic link), f_z (:37 / :45) ; the test f_z > ub_z (:38, :26) ;
//            rejected: gamma *= shrink, the next tick retries ; accepted: norm_res, record, termination
//            (:55-59), x = z ; and — speculatively, for every trial — A' res, the gradient at z (:61)
//
// `backtracking_nesterov` (:66-84) runs on the same kernels: its trial ticks start from the extrapolated point,
// need no gradient at z, and every accepted trial is followed by an extrapolation tick (k_bt_a: x = z +
// (theta_prev - 1)/theta (z - z_prev), A x ; k_bt_b: f_x and grad f(x), :78-81).
//
// x / z and grad_x / grad_z live in two buffers each and the control block holds the index of the current one:
// accepting a trial flips the index, nothing is copied.  Two control blocks (A read by k_bt_a and written by
// k_bt_b, B the other way round), so no kernel writes what its own blocks read.  The host enqueues ticks in
// chunks and reads the counters between chunks.
#include <cstdlib>
#include <cstring>
#include <memory>

#include "small_dev.h"

namespace apx {

struct BtCtrl {
  double xi, shrink, tol, n_mean;
  int64_t maxit, rec_cap;
  double gamma, gamma_try, f_x;
  double s_dot, s_ss, s_graw;       // <grad_x, z - x>, ||z - x||^2, raw g(z) of the current trial
  double bias[2];                   // logistic: mean(probs - y) belonging to grad buffer 0 / 1
  double norm_res, objective;
  int64_t it, trials_total;
  int32_t cur;                      // which of the two x / grad buffers (Nesterov: z buffers) is current
  int32_t done, flags;
  int32_t nesterov;                 // backtracking_nesterov (:66-84): trial ticks and extrapolation ticks alternate
  int32_t phase, _pad;              // Nesterov: 0 trial tick, 1 extrapolation tick
  double theta, beta;               // :78-80
};

struct BtDev {
  const void* A;
  const void* AT;
  int64_t ld, ldT;
  int32_t m, ncols, n, w_n, w_t, _pad;
  const void* b;
  void* P[2];          // x / z  (Nesterov: z / z_prev)
  void* XB;            // Nesterov: the extrapolated point x the trials start from
  double* G[2];        // A' res at P[k] (Nesterov: G[0] = A' res at XB), fp64 as the dots leave it
  double* az;          // A z
  BtCtrl *ca, *cb;
  ProxDev g;
  apx_record* recs;
};

// MODE 0: pre-loop (:52): A x0 only ; MODE 2: a trial
template <typename T, int FK, int MODE>
__global__ __launch_bounds__(kBlock) void k_bt_a(BtDev d) {
  __shared__ double zs[kSmallMax];
  __shared__ BtCtrl lc;
  if (threadIdx.x == 0) lc = *d.ca;
  __syncthreads();
  const int cur = lc.cur;
  const bool nes = lc.nesterov != 0;
  const T* __restrict__ X = (const T*)(nes ? d.XB : d.P[cur]);
  if constexpr (MODE == 0) {
    for (int i = threadIdx.x; i < d.n; i += kBlock) zs[i] = (double)X[i];
    __syncthreads();
    small_dots_w<T>(d.w_n, (const T*)d.AT, d.ldT, d.m, d.ncols, zs, d.az, true);
    return;
  } else {
    if (lc.done != 0 || lc.it >= lc.maxit) {        // frozen: hand the state on so that k_bt_b sees it too
      if (blockIdx.x == 0 && threadIdx.x == 0) *d.cb = lc;
      return;
    }
    const int per = (d.n + kBlock - 1) / kBlock;
    if (nes && lc.phase == 1) {                     // extrapolation tick: x = z + beta (z - z_prev) (:80) ; A x
      const T* __restrict__ Zc = (const T*)d.P[cur];
      const T* __restrict__ Zq = (const T*)d.P[1 - cur];
      T* __restrict__ XW = (T*)d.XB;
      const double beta = lc.beta;
      for (int k = 0; k < per; ++k) {
        const int i = threadIdx.x + k * kBlock;
        if (i < d.n) {
          const double zc = (double)Zc[i];
          const T xn = (T)(zc + beta * (zc - (double)Zq[i]));
          zs[i] = (double)xn;
          if ((unsigned)k % gridDim.x == blockIdx.x) XW[i] = xn;
        }
      }
      __syncthreads();
      if (blockIdx.x == 0 && threadIdx.x == 0) *d.cb = lc;
      small_dots_w<T>(d.w_n, (const T*)d.AT, d.ldT, d.m, d.ncols, zs, d.az, true);
      return;
    }
    const int gi_buf = nes ? 0 : cur;
    const double gamma = lc.gamma_try;
    const double* __restrict__ G = d.G[gi_buf];
    T* __restrict__ Z = (T*)d.P[1 - cur];
    double q[3] = {0.0, 0.0, 0.0};
    for (int k0 = 0; k0 < per; k0 += 4) {
      double x4[4], g4[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int ic = min((int)threadIdx.x + (k0 + u) * kBlock, d.n - 1);
        x4[u] = (double)X[ic];
        g4[u] = G[ic];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int k = k0 + u;
        const int i = threadIdx.x + k * kBlock;
        if (i < d.n) {
          double gi;
          if (FK == APX_F_LOGISTIC && i == d.n - 1) {
            gi = (double)(T)lc.bias[gi_buf];                               // grad[end] = mean(probs - y)
          } else {
            const T gs = (T)g4[u];                                         // A' * res
            gi = FK == APX_F_LOGISTIC ? (double)(T)((double)gs / lc.n_mean) : (double)gs;
          }
          const T vi = (T)(x4[u] - gamma * gi);
          const T zi = (T)prox_elem_val<T>(d.g, (double)vi, gamma, i, 0.0, &q[2]);      // :35 / :43
          const double dd = (double)zi - x4[u];
          q[0] = fma(gi, dd, q[0]);                                        // <grad_x, z - x>
          q[1] = fma(dd, dd, q[1]);                                        // ||z - x||^2
          zs[i] = (double)zi;
          if ((unsigned)k % gridDim.x == blockIdx.x) Z[i] = zi;           // chunk k is written by one block
        }
      }
    }
    block_sums_t0<3>(q);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      lc.s_dot = q[0];
      lc.s_ss = q[1];
      lc.s_graw = q[2];
      *d.cb = lc;
    }
    small_dots_w<T>(d.w_n, (const T*)d.AT, d.ldT, d.m, d.ncols, zs, d.az, true);     // (barrier inside block_sums_t0)
  }
}

// MODE 0: pre-loop: f(x0), grad f(x0) into the current buffers ; MODE 2: close a trial
template <typename T, int FK, int MODE>
__global__ __launch_bounds__(kBlock) void k_bt_b(BtDev d) {
  __shared__ double rs[kSmallMax];
  __shared__ BtCtrl lc;
  if (threadIdx.x == 0) lc = *(MODE == 0 ? d.ca : d.cb);
  __syncthreads();
  if (MODE == 2 && (lc.done != 0 || lc.it >= lc.maxit)) return;
  const int cur = lc.cur;
  const bool nes = lc.nesterov != 0;
  const bool extrap = nes && (MODE == 0 || lc.phase == 1);          // the point is XB, its gradient goes to G[0]
  const int tgt = MODE == 0 ? cur : 1 - cur;              // the buffers of the point whose gradient is formed here
  const int gbuf = nes ? 0 : tgt;
  const T* __restrict__ b = (const T*)d.b;
  const T* __restrict__ Zp = (const T*)(extrap ? d.XB : d.P[tgt]);
  double bias = 0.0;
  if constexpr (FK == APX_F_LOGISTIC) bias = (double)Zp[d.n - 1];
  double q[2] = {0.0, 0.0};
  const int per = (d.m + kBlock - 1) / kBlock;
  for (int k0 = 0; k0 < per; k0 += 4) {
    double s4[4], b4[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int ic = min((int)threadIdx.x + (k0 + u) * kBlock, d.m - 1);
      s4[u] = d.az[ic];
      b4[u] = (double)b[ic];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = threadIdx.x + (k0 + u) * kBlock;
      if (i < d.m) {
        if constexpr (FK == APX_F_LEAST_SQUARES) {        // res = A*w - b ; sum res^2   (lasso/runme.jl:22,24)
          const T r = (T)(s4[u] - b4[u]);
          rs[i] = (double)r;
          q[0] = fma((double)r, (double)r, q[0]);
        } else {                                          // sparse_logreg/runme.jl:24-36
          const double logit = s4[u] + bias;
          const double e = 1.0 + exp(-logit);
          q[0] += (b4[u] - 1.0) * logit - log(e);
          const T dd = (T)(1.0 / e - b4[u]);
          rs[i] = (double)dd;
          q[1] += (double)dd;
        }
      }
    }
  }
  block_sums_t0<2>(q);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    double f_z;
    if constexpr (FK == APX_F_LEAST_SQUARES) {
      const double nr = sqrt(q[0]);                       // 0.5 * norm(res)^2 : nrm2 THEN square
      f_z = 0.5 * (nr * nr);
    } else {
      f_z = -(q[0] / lc.n_mean);
      if (!nes || extrap) lc.bias[gbuf] = q[1] / lc.n_mean;
    }
    if (MODE == 0) {
      lc.f_x = f_z;                                       // :52 / :69
    } else if (extrap) {
      lc.f_x = f_z;                                       // :81  f_x, grad_x = eval_with_gradient(f, x)
      lc.phase = 0;
    } else {
      const double gamma = lc.gamma_try;
      const double nrm = sqrt(lc.s_ss);
      const double ub_z = lc.f_x + lc.s_dot + 1.0 / (2.0 * gamma) * (nrm * nrm);        // :26
      lc.trials_total += 1;
      // (gamma == 0 or not finite: the reference would spin forever; the generic path stops there too)
      if (!(f_z > ub_z) || gamma == 0.0 || !isfinite(gamma)) {                          // :38  accepted
        lc.gamma = gamma;
        lc.norm_res = nrm / gamma;                                                      // :55
        lc.objective = f_z + gval_final(d.g, lc.s_graw);
        lc.it += 1;
        if (d.recs != nullptr && lc.it <= lc.rec_cap) {                                 // :56
          apx_record r;
          r.it = lc.it;
          r.gamma = gamma;
          r.sigma = NAN;
          r.norm_res = lc.norm_res;
          r.objective = lc.objective;
          r.f_x = f_z;
          r.f_evals = 1 + lc.trials_total + (nes ? lc.it - 1 : 0);                      // (+ the evaluations at the extrapolated points)
          r.grad_evals = lc.it;
          d.recs[lc.it - 1] = r;
        }
        lc.cur = tgt;                                                                   // :60  x = z (Nesterov: z_prev = z ; z)
        if (lc.norm_res <= lc.tol) lc.done = 1;                                         // :57-59 / :75-77
        if (nes) {
          const double theta_prev = lc.theta;                                           // :78
          lc.theta = (1.0 + sqrt(1.0 + 4.0 * (theta_prev * theta_prev))) / 2.0;         // :79
          lc.beta = (theta_prev - 1.0) / lc.theta;
          lc.gamma_try = gamma;                                                         // (no xi in backtracking_nesterov)
          lc.phase = 1;
        } else {
          lc.f_x = f_z;
          lc.gamma_try = lc.xi * gamma;                                                 // :54 of the next iteration
        }
      } else {
        lc.gamma_try = gamma * lc.shrink;                                               // :39
        if (lc.gamma_try < 1e-12) lc.flags |= APX_FLAG_TINY_GAMMA;                      // :40-42 (logs, continues)
      }
    }
    *d.ca = lc;
  }
  // grad at the trial point, speculatively (:61) — rs is complete behind the barrier of block_sums_t0.
  // Nesterov never needs the gradient at z: only the extrapolation ticks (and the pre-loop) form one.
  if (!nes || extrap) small_dots_w<T>(d.w_t, (const T*)d.A, d.ld, d.ncols, d.m, rs, d.G[gbuf], true);
}

// ---------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------
bool small_bt_eligible(const apx_problem* p) {
  if (!p->ctx->fused_small) return false;
  if (p->A_lin != nullptr || p->sharded()) return false;
  if (p->f_kind != APX_F_LEAST_SQUARES && p->f_kind != APX_F_LOGISTIC) return false;
  if (p->A_f == nullptr || no_column_storage(p->A_f) || p->ctx->skip_zero_cols) return false;
  return p->n <= kSmallMax && p->A_f->m <= kSmallMax && p->A_f->m > 0 && p->A_f->n > 0;
}

template <typename T, int MODE>
static void bt_launch(const BtDev& d, int fk, hipStream_t q) {
  const dim3 ga(pick_grid(d.m, d.w_n)), gb(pick_grid(d.ncols, d.w_t));
  if (fk == APX_F_LEAST_SQUARES) {
    hipLaunchKernelGGL((k_bt_a<T, APX_F_LEAST_SQUARES, MODE>), ga, dim3(kBlock), 0, q, d);
    hipLaunchKernelGGL((k_bt_b<T, APX_F_LEAST_SQUARES, MODE>), gb, dim3(kBlock), 0, q, d);
  } else {
    hipLaunchKernelGGL((k_bt_a<T, APX_F_LOGISTIC, MODE>), ga, dim3(kBlock), 0, q, d);
    hipLaunchKernelGGL((k_bt_b<T, APX_F_LOGISTIC, MODE>), gb, dim3(kBlock), 0, q, d);
  }
}

int32_t small_bt_run(apx_problem* p, bool nesterov, const void* x0, double gamma0, double xi, double shrink, double tol,
                     int64_t maxit, const apx_solve_opts* opts, void* x_out, apx_result* res) {
  apx_ctx* ctx = p->ctx;
  const apx_matrix* A = p->A_f;
  const size_t es = p->es();
  const size_t nb = (size_t)p->n_pad * es, nd = (size_t)p->n_pad * sizeof(double);
  DBuf p0, p1, bxb, g0, g1, baz, bat, bca, bcb, brecs;
  BtDev d{};
  d.A = A->d;
  d.ld = A->ld;
  if (A->dT != nullptr) {
    d.AT = A->dT;
    d.ldT = A->ldT;
  } else {               // private transposed copy: the caller's matrix object is untouched
    d.ldT = round_up(A->n, kPad);
    APX_TRY(bat.alloc_zero((size_t)d.ldT * (size_t)A->ld * es, ctx->stream));
    APX_TRY(transpose_into(ctx, A, bat.p, d.ldT));
    d.AT = bat.p;
  }
  d.m = (int32_t)A->m;
  d.ncols = (int32_t)A->n;
  d.n = (int32_t)p->n;
  d.w_n = pick_waves(d.ncols);
  d.w_t = pick_waves(d.m);
  d.b = p->b.p;
  APX_TRY(p0.alloc_zero(nb, ctx->stream));
  APX_TRY(p1.alloc_zero(nb, ctx->stream));
  APX_TRY(bxb.alloc_zero(nb, ctx->stream));
  APX_TRY(g0.alloc_zero(nd, ctx->stream));
  APX_TRY(g1.alloc_zero(nd, ctx->stream));
  APX_TRY(baz.alloc_zero((size_t)A->ld * sizeof(double), ctx->stream));
  APX_TRY(bca.alloc_zero(sizeof(BtCtrl), ctx->stream));
  APX_TRY(bcb.alloc_zero(sizeof(BtCtrl), ctx->stream));
  apx_record* host_recs = opts ? opts->records : nullptr;
  const int64_t rec_cap = (opts && opts->records) ? opts->record_cap : 0;
  if (rec_cap > 0) APX_TRY(brecs.alloc_zero((size_t)rec_cap * sizeof(apx_record), ctx->stream));
  d.P[0] = p0.p;
  d.P[1] = p1.p;
  d.XB = bxb.p;
  d.G[0] = (double*)g0.p;
  d.G[1] = (double*)g1.p;
  d.az = (double*)baz.p;
  d.ca = (BtCtrl*)bca.p;
  d.cb = (BtCtrl*)bcb.p;
  d.g = p->gd;
  d.recs = (apx_record*)brecs.p;

  BtCtrl c;
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
  APX_TRY(h2d(ctx, bca.p, &c, sizeof(c)));
  APX_TRY(h2d(ctx, p0.p, x0, (size_t)p->n * es));       // x (Nesterov: z) = x0
  if (nesterov) APX_TRY(h2d(ctx, bxb.p, x0, (size_t)p->n * es));
  hipStream_t q = ctx->stream;
  const bool f32 = p->dtype == APX_F32;
  if (f32) bt_launch<float, 0>(d, p->f_kind, q);     // :52  f_x, grad_x = eval_with_gradient(f, x)
  else bt_launch<double, 0>(d, p->f_kind, q);
  APX_HIP(hipGetLastError());
  while (maxit > 0) {                                // every iteration takes at least one tick (Nesterov: two)
    const int64_t missing = (maxit - c.it) * (nesterov ? 2 : 1);
    const int chunk = (int)std::min<int64_t>(64, std::max<int64_t>(missing, 1));
    for (int k = 0; k < chunk; ++k) {
      if (f32) bt_launch<float, 2>(d, p->f_kind, q);
      else bt_launch<double, 2>(d, p->f_kind, q);
    }
    APX_HIP(hipGetLastError());
    APX_TRY(d2h(ctx, &c, bca.p, sizeof(c)));
    if (c.done != 0 || c.it >= maxit) break;
  }
  if (maxit == 0) APX_TRY(d2h(ctx, &c, bca.p, sizeof(c)));

  APX_TRY(d2h(ctx, x_out, d.P[c.cur], (size_t)p->n * es));      // z of the last accepted trial (x0 when maxit == 0)
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
