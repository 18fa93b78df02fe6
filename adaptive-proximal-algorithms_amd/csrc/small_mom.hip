// small_mom.hip — the two comparison methods of the reference's lasso / logistic experiments that have no
// linesearch, `fixed_nesterov` (src/AdaProx.jl:91-142) and `agraal` (:150-192), as DEVICE-RESIDENT loops on the
// two-kernel scheme of small.hip (small dense problems, m, n <= 4096): every block redoes the O(m + n) work, both
// products are in-block dot products, the step-size recurrences run on one device thread.  The generic
// implementations (baselines.hip) read scalars back every iteration: 41 / 66 us per iteration at 500 x 1000.
//
//   aGRAAL, iteration k = one tick:
//     k_mom_a : C, gamma, theta (:175-178) ; x_bar (:179) ; x+ = prox(x_bar - gamma grad_x) (:181) ; norm_res (:182) ; A x+
//     k_mom_b : f(x+) -> record, termination (:183-188) ; grad f(x+) = A' res (:189)
//   fixed Nesterov: tick t finishes iteration t-1 and starts iteration t (the prox step needs the finished gradient):
//     k_mom_a : x+ = prox(z - gamma grad_z), norm_res (:132-133) ; theta, beta (:121-128) ; z+ = x+ + beta (x+ - x)
//               (:129) ; A z+ (and A x+ when records are kept: the objective f(x+) + g(x+), :134-136)
//     k_mom_b : record, termination (:134-139) ; grad f(z+) (:130)
//
// The n-vectors rotate through two / three buffers by iteration parity (roles chosen by the host per tick, as in
// small_pd.hip); two control blocks, A read by k_mom_a and written by k_mom_b, B the other way round.
#include <cstdlib>
#include <cstring>
#include <memory>

#include "small_dev.h"

namespace apx {

struct MomDev {
  const void* A;
  const void* AT;
  int64_t ld, ldT;
  int32_t m, ncols, n, w_n, w_t, method;
  const void* b;
  void* X[3];
  void* AUX[2];        // aGRAAL: x_bar ; Nesterov: the extrapolated point z
  double* G[3];
  double *az, *ax2;    // A (point whose gradient is formed) ; Nesterov with records: A x+
  MomCtrl *ca, *cb;
  ProxDev g;
  apx_record* recs;
};

struct MomRoles {      // buffer indices of this tick
  int32_t x_prev, x_cur, x_next, g_prev, g_cur, g_next, aux_cur, aux_next;
  int32_t first, want_obj;
};

__device__ __forceinline__ double mom_min2(double a, double b) {   // Julia min: NaN-propagating
  if (isnan(a) || isnan(b)) return NAN;
  return a < b ? a : b;
}

// gradient entry i of buffer G (as the dots left it) with the f-specific scaling
template <typename T, int FK>
__device__ __forceinline__ double mom_grad(const double* __restrict__ G, int i, int n, double bias, double n_mean) {
  if (FK == APX_F_LOGISTIC && i == n - 1) return (double)(T)bias;
  const T gs = (T)G[i];
  return FK == APX_F_LOGISTIC ? (double)(T)((double)gs / n_mean) : (double)gs;
}

// MODE 0: only A * X[x_cur] -> az (pre-loop gradients) ; MODE 2: a tick
template <typename T, int FK, int MODE>
__global__ __launch_bounds__(kBlock) void k_mom_a(MomDev d, MomRoles ro) {
  __shared__ double ps[kSmallMax];          // the point whose product is formed
  __shared__ MomCtrl lc;
  const int tid = threadIdx.x;
  if constexpr (MODE == 0) {
    const T* __restrict__ X = (const T*)d.X[ro.x_cur];
    for (int i = tid; i < d.n; i += kBlock) ps[i] = (double)X[i];
    __syncthreads();
    small_dots_w<T>(d.w_n, (const T*)d.AT, d.ldT, d.m, d.ncols, ps, d.az, true);
    return;
  } else {
    if (tid == 0) lc = *d.ca;
    __syncthreads();
    if (lc.done != 0 || lc.it >= lc.maxit) {            // nothing left to do: tell k_mom_b
      if (blockIdx.x == 0 && tid == 0) {
        lc.frozen = 1;
        *d.cb = lc;
      }
      return;
    }
    const int per = (d.n + kBlock - 1) / kBlock;
    if (d.method == MOM_AGRAAL) {
      const T* __restrict__ X = (const T*)d.X[ro.x_cur];
      const T* __restrict__ XP = (const T*)d.X[ro.x_prev];
      const T* __restrict__ XB = (const T*)d.AUX[ro.aux_cur];
      T* __restrict__ XN = (T*)d.X[ro.x_next];
      T* __restrict__ XBN = (T*)d.AUX[ro.aux_next];
      const double* __restrict__ G = d.G[ro.g_cur];
      const double* __restrict__ GP = d.G[ro.g_prev];
      // ---- ||x - x_prev||^2, ||grad_x - grad_x_prev||^2  (:175) ---------------------------------------
      double q[2] = {0.0, 0.0};
      for (int i = tid; i < d.n; i += kBlock) {
        const double dx = (double)X[i] - (double)XP[i];
        const double dg = mom_grad<T, FK>(G, i, d.n, lc.bias[ro.g_cur], lc.n_mean) -
                          mom_grad<T, FK>(GP, i, d.n, lc.bias[ro.g_prev], lc.n_mean);
        q[0] = fma(dx, dx, q[0]);
        q[1] = fma(dg, dg, q[1]);
      }
      block_sums_t0<2>(q);
      if (tid == 0) {
        const double ndx = sqrt(q[0]), ndg = sqrt(q[1]);
        if (lc.need_gamma0 != 0) {                                         // :168-170
          lc.gamma = ndx / ndg;
          lc.need_gamma0 = 0;
        }
        const double C = (ndx * ndx) / (ndg * ndg);                        // :175
        const double gamma_prev = lc.gamma;                                // :176
        lc.gamma = mom_min2(mom_min2(lc.rho * gamma_prev, lc.phi * lc.theta * C / (4.0 * gamma_prev)), lc.gamma_max);
        lc.theta = lc.phi * lc.gamma / gamma_prev;                         // :178
      }
      __syncthreads();
      const double gamma = lc.gamma, phi = lc.phi;
      double s[2] = {0.0, 0.0};
      for (int k = 0; k < per; ++k) {
        const int i = tid + k * kBlock;
        if (i < d.n) {
          const double xi = (double)X[i];
          const T xb = (T)(((phi - 1.0) * xi + (double)XB[i]) / phi);      // :179
          const double gi = mom_grad<T, FK>(G, i, d.n, lc.bias[ro.g_cur], lc.n_mean);
          const T vi = (T)((double)xb - gamma * gi);
          const T xn = (T)prox_elem_val<T>(d.g, (double)vi, gamma, i, 0.0, &s[1]);     // :181
          const double dd = (double)xn - xi;
          s[0] = fma(dd, dd, s[0]);                                        // :182
          ps[i] = (double)xn;
          if ((unsigned)k % gridDim.x == blockIdx.x) {                     // chunk k is written by one block
            XN[i] = xn;
            XBN[i] = xb;
          }
        }
      }
      block_sums_t0<2>(s);
      if (blockIdx.x == 0 && tid == 0) {
        lc.norm_res = sqrt(s[0]) / gamma;
        lc.g_raw = s[1];
        lc.it += 1;
        lc.stop_after = lc.norm_res <= lc.tol ? 1 : 0;                     // :186 (the record of this iteration still needs f)
        *d.cb = lc;
      }
      small_dots_w<T>(d.w_n, (const T*)d.AT, d.ldT, d.m, d.ncols, ps, d.az, true);      // (barrier inside block_sums_t0)
    } else {
      // ---- fixed Nesterov ------------------------------------------------------------------------------
      const T* __restrict__ XC = (const T*)d.X[ro.x_cur];       // x^(t-2)
      T* __restrict__ XN = (T*)d.X[ro.x_next];                   // x^(t-1), formed here
      const T* __restrict__ Z = (const T*)d.AUX[ro.aux_cur];     // z_{t-1}
      T* __restrict__ ZN = (T*)d.AUX[ro.aux_next];               // z_t
      const double* __restrict__ G = d.G[ro.g_cur];              // grad f(z_{t-1})
      const double gamma = lc.gamma;
      // theta, beta of iteration t (:121-128) — identical on every block
      __shared__ double beta_s;
      if (tid == 0) {
        const double theta_prev = lc.theta;
        double theta, beta;
        if (lc.mu == 0.0) {
          theta = (1.0 + sqrt(1.0 + 4.0 * (theta_prev * theta_prev))) / 2.0;
          beta = (theta_prev - 1.0) / theta;
        } else {
          const double a = 1.0 - lc.q * (theta_prev * theta_prev);
          theta = (a + sqrt(a * a + 4.0 * (theta_prev * theta_prev))) / 2.0;
          beta = (theta_prev - 1.0) * (1.0 + gamma * lc.mug - theta * gamma * lc.mu) / theta / (1.0 - gamma * lc.muf);
        }
        lc.theta = theta;
        beta_s = beta;
      }
      __syncthreads();
      const double beta = beta_s;
      double s[2] = {0.0, 0.0};
      __shared__ double xnew_s[kSmallMax];
      for (int k = 0; k < per; ++k) {
        const int i = tid + k * kBlock;
        if (i < d.n) {
          const double xc = (double)XC[i];
          double xn;
          if (ro.first) {
            xn = xc;                                                       // nothing to finish: x^(0) = x^(-1) = x0
          } else {
            const double zi = (double)Z[i];
            const double gi = mom_grad<T, FK>(G, i, d.n, lc.bias[ro.g_cur], lc.n_mean);
            const T vi = (T)(zi - gamma * gi);
            const T xt = (T)prox_elem_val<T>(d.g, (double)vi, gamma, i, 0.0, &s[1]);   // :132
            xn = (double)xt;
            const double dd = xn - zi;
            s[0] = fma(dd, dd, s[0]);                                      // :133
          }
          const T zn = (T)(xn + beta * (xn - xc));                         // :129 of the next iteration
          ps[i] = (double)zn;
          xnew_s[i] = xn;
          if ((unsigned)k % gridDim.x == blockIdx.x) {
            XN[i] = (T)xn;
            ZN[i] = zn;
          }
        }
      }
      block_sums_t0<2>(s);
      if (blockIdx.x == 0 && tid == 0) {
        if (!ro.first) {
          lc.norm_res = sqrt(s[0]) / gamma;
          lc.g_raw = s[1];
          lc.it += 1;
          lc.stop_after = lc.norm_res <= lc.tol ? 1 : 0;                   // :137
        } else {
          lc.stop_after = 0;
        }
        *d.cb = lc;
      }
      small_dots_w<T>(d.w_n, (const T*)d.AT, d.ldT, d.m, d.ncols, ps, d.az, true);
      if (ro.want_obj && !ro.first)                                        // f(x+) for the record (:134-136)
        small_dots_w<T>(d.w_n, (const T*)d.AT, d.ldT, d.m, d.ncols, xnew_s, d.ax2, true);
    }
  }
}

// f and the residual of `az` into rs ; returns the two block sums on thread 0
template <typename T, int FK>
__device__ __forceinline__ void mom_residual(const MomDev& d, const double* __restrict__ az, double bias, double* rs,
                                             double (&q)[2]) {
  const T* __restrict__ b = (const T*)d.b;
  q[0] = q[1] = 0.0;
  const int per = (d.m + kBlock - 1) / kBlock;
  for (int k0 = 0; k0 < per; k0 += 4) {
    double s4[4], b4[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int ic = min((int)threadIdx.x + (k0 + u) * kBlock, d.m - 1);
      s4[u] = az[ic];
      b4[u] = (double)b[ic];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = threadIdx.x + (k0 + u) * kBlock;
      if (i < d.m) {
        if constexpr (FK == APX_F_LEAST_SQUARES) {
          const T r = (T)(s4[u] - b4[u]);
          if (rs != nullptr) rs[i] = (double)r;
          q[0] = fma((double)r, (double)r, q[0]);
        } else {
          const double logit = s4[u] + bias;
          const double e = 1.0 + exp(-logit);
          q[0] += (b4[u] - 1.0) * logit - log(e);
          const T dd = (T)(1.0 / e - b4[u]);
          if (rs != nullptr) rs[i] = (double)dd;
          q[1] += (double)dd;
        }
      }
    }
  }
  block_sums_t0<2>(q);
}

template <int FK>
__device__ __forceinline__ double mom_fval(const double (&q)[2], double n_mean) {
  if (FK == APX_F_LEAST_SQUARES) {
    const double nr = sqrt(q[0]);
    return 0.5 * (nr * nr);
  }
  return -(q[0] / n_mean);
}

// MODE 0: pre-loop: gradient of the point in X[x_cur] -> G[g_next] ; MODE 2: close the tick
template <typename T, int FK, int MODE>
__global__ __launch_bounds__(kBlock) void k_mom_b(MomDev d, MomRoles ro) {
  __shared__ double rs[kSmallMax];
  __shared__ MomCtrl lc;
  const int tid = threadIdx.x;
  if (tid == 0) lc = *(MODE == 0 ? d.ca : d.cb);
  __syncthreads();
  if (MODE == 2 && lc.frozen != 0) return;
  // the point whose gradient is formed: aGRAAL x+ ; Nesterov z+ ; pre-loop X[x_cur]
  const T* __restrict__ P = (const T*)(MODE == 0 ? d.X[ro.x_cur] : (d.method == MOM_AGRAAL ? d.X[ro.x_next] : d.AUX[ro.aux_next]));
  double bias = 0.0;
  if constexpr (FK == APX_F_LOGISTIC) bias = (double)P[d.n - 1];
  double q[2], qx[2] = {0.0, 0.0};
  const bool nesterov_obj = MODE == 2 && d.method == MOM_NESTEROV && ro.want_obj && !ro.first;
  if (nesterov_obj) {                                   // f(x+) from A x+ (no residual kept)
    double bx = 0.0;
    if constexpr (FK == APX_F_LOGISTIC) bx = (double)((const T*)d.X[ro.x_next])[d.n - 1];
    mom_residual<T, FK>(d, d.ax2, bx, nullptr, qx);
    __syncthreads();
  }
  mom_residual<T, FK>(d, d.az, bias, rs, q);
  if (blockIdx.x == 0 && tid == 0) {
    if constexpr (FK == APX_F_LOGISTIC) lc.bias[ro.g_next] = q[1] / lc.n_mean;
    if constexpr (MODE == 2) {
      const bool finished = d.method == MOM_AGRAAL || !ro.first;       // an iteration was completed by this tick
      if (finished) {
        lc.f_new = d.method == MOM_AGRAAL ? mom_fval<FK>(q, lc.n_mean) : (nesterov_obj ? mom_fval<FK>(qx, lc.n_mean) : NAN);
        lc.objective = lc.f_new + gval_final(d.g, lc.g_raw);
        if (d.recs != nullptr && lc.it <= lc.rec_cap) {
          apx_record r;
          r.it = lc.it;
          r.gamma = lc.gamma;
          r.sigma = NAN;
          r.norm_res = lc.norm_res;
          r.objective = lc.objective;
          r.f_x = lc.f_new;
          r.f_evals = d.method == MOM_AGRAAL ? 1 + lc.it : lc.it;       // evaluations counted so far (aGRAAL: 2 before the loop)
          r.grad_evals = r.f_evals;
          d.recs[lc.it - 1] = r;
        }
        if (lc.stop_after != 0) lc.done = 1;
      }
    }
    *d.ca = lc;
  }
  // (rs is complete behind the barrier of block_sums_t0)
  small_dots_w<T>(d.w_t, (const T*)d.A, d.ld, d.ncols, d.m, rs, d.G[ro.g_next], true);
}

// ---------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------
bool small_mom_eligible(const apx_problem* p) {
  if (!p->ctx->fused_small) return false;
  if (p->A_lin != nullptr || p->sharded()) return false;
  if (p->f_kind != APX_F_LEAST_SQUARES && p->f_kind != APX_F_LOGISTIC) return false;
  if (p->A_f == nullptr || no_column_storage(p->A_f) || p->ctx->skip_zero_cols) return false;
  return p->n <= kSmallMax && p->A_f->m <= kSmallMax && p->A_f->m > 0 && p->A_f->n > 0;
}

template <typename T, int MODE>
static void mom_launch(const MomDev& d, int fk, const MomRoles& ro, hipStream_t q) {
  const dim3 ga(pick_grid(d.m, d.w_n)), gb(pick_grid(d.ncols, d.w_t));
  if (fk == APX_F_LEAST_SQUARES) {
    hipLaunchKernelGGL((k_mom_a<T, APX_F_LEAST_SQUARES, MODE>), ga, dim3(kBlock), 0, q, d, ro);
    hipLaunchKernelGGL((k_mom_b<T, APX_F_LEAST_SQUARES, MODE>), gb, dim3(kBlock), 0, q, d, ro);
  } else {
    hipLaunchKernelGGL((k_mom_a<T, APX_F_LOGISTIC, MODE>), ga, dim3(kBlock), 0, q, d, ro);
    hipLaunchKernelGGL((k_mom_b<T, APX_F_LOGISTIC, MODE>), gb, dim3(kBlock), 0, q, d, ro);
  }
}

// method-specific parameters arrive in `c0` (gamma, theta, ... filled by the caller)
int32_t small_mom_run(apx_problem* p, int method, const MomCtrl& c0, const void* x1, const void* x0, int64_t maxit,
                      const apx_solve_opts* opts, void* x_out, apx_result* res) {
  apx_ctx* ctx = p->ctx;
  const apx_matrix* A = p->A_f;
  const size_t es = p->es();
  const size_t nb = (size_t)p->n_pad * es, nd = (size_t)p->n_pad * sizeof(double), md = (size_t)A->ld * sizeof(double);
  DBuf bx[3], baux[2], bg[3], baz, bax2, bat, bca, bcb, brecs;
  MomDev d{};
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
  d.method = method;
  d.b = p->b.p;
  for (int k = 0; k < 3; ++k) {
    APX_TRY(bx[k].alloc_zero(nb, ctx->stream));
    APX_TRY(bg[k].alloc_zero(nd, ctx->stream));
    d.X[k] = bx[k].p;
    d.G[k] = (double*)bg[k].p;
  }
  for (int k = 0; k < 2; ++k) {
    APX_TRY(baux[k].alloc_zero(nb, ctx->stream));
    d.AUX[k] = baux[k].p;
  }
  APX_TRY(baz.alloc_zero(md, ctx->stream));
  APX_TRY(bax2.alloc_zero(md, ctx->stream));
  APX_TRY(bca.alloc_zero(sizeof(MomCtrl), ctx->stream));
  APX_TRY(bcb.alloc_zero(sizeof(MomCtrl), ctx->stream));
  apx_record* host_recs = opts ? opts->records : nullptr;
  const int64_t rec_cap = (opts && opts->records) ? opts->record_cap : 0;
  if (rec_cap > 0) APX_TRY(brecs.alloc_zero((size_t)rec_cap * sizeof(apx_record), ctx->stream));
  d.az = (double*)baz.p;
  d.ax2 = (double*)bax2.p;
  d.ca = (MomCtrl*)bca.p;
  d.cb = (MomCtrl*)bcb.p;
  d.g = p->gd;
  d.recs = (apx_record*)brecs.p;

  MomCtrl c = c0;
  c.n_mean = p->n_mean;
  c.maxit = maxit;
  c.rec_cap = rec_cap;
  c.norm_res = c.objective = c.f_new = std::numeric_limits<double>::quiet_NaN();
  APX_TRY(h2d(ctx, bca.p, &c, sizeof(c)));
  hipStream_t q = ctx->stream;
  const bool f32 = p->dtype == APX_F32;
  const int want_obj = rec_cap > 0 ? 1 : 0;
  auto launch = [&](int mode, const MomRoles& ro) {
    if (mode == 0) {
      if (f32) mom_launch<float, 0>(d, p->f_kind, ro, q);
      else mom_launch<double, 0>(d, p->f_kind, ro, q);
    } else {
      if (f32) mom_launch<float, 2>(d, p->f_kind, ro, q);
      else mom_launch<double, 2>(d, p->f_kind, ro, q);
    }
  };

  int64_t ticks = 0;
  auto roles = [&](int64_t t) {       // tick t (1-based)
    MomRoles ro{};
    ro.want_obj = want_obj;
    if (method == MOM_AGRAAL) {       // iteration k = t: x^(k-2), x^(k-1) -> x^(k) ; g likewise ; x_bar ping-pong
      ro.x_prev = (int)((t + 0) % 3);
      ro.x_cur = (int)((t + 1) % 3);
      ro.x_next = (int)((t + 2) % 3);
      ro.g_prev = ro.x_prev;
      ro.g_cur = ro.x_cur;
      ro.g_next = ro.x_next;
      ro.aux_cur = (int)(t % 2);
      ro.aux_next = (int)((t + 1) % 2);
    } else {                           // tick t: x^(t-2) in X[t % 2], x^(t-1) formed in X[(t + 1) % 2] ; z ping-pong ; one G
      ro.x_cur = (int)(t % 2);
      ro.x_next = (int)((t + 1) % 2);
      ro.aux_cur = (int)((t + 1) % 2);
      ro.aux_next = (int)(t % 2);
      ro.g_cur = ro.g_next = 0;
      ro.first = t == 1 ? 1 : 0;
    }
    return ro;
  };
  if (method == MOM_AGRAAL) {
    // x^(-1) = x0 in X[1], x^(0) = x1 in X[2] (the roles of tick 1), x_bar^(0) = x1 in AUX[1] ; their gradients (:166-167)
    APX_TRY(h2d(ctx, bx[1].p, x0, (size_t)p->n * es));
    APX_TRY(h2d(ctx, bx[2].p, x1, (size_t)p->n * es));
    APX_TRY(h2d(ctx, baux[1].p, x1, (size_t)p->n * es));
    MomRoles pre{};
    pre.x_cur = 1;
    pre.g_next = 1;
    launch(0, pre);
    pre.x_cur = 2;
    pre.g_next = 2;
    launch(0, pre);
  } else {
    APX_TRY(h2d(ctx, bx[0].p, x1, (size_t)p->n * es));     // x^(-1) = x^(0) = x0 (:119)
    APX_TRY(h2d(ctx, bx[1].p, x1, (size_t)p->n * es));
  }
  APX_HIP(hipGetLastError());
  // aGRAAL: one tick per iteration ; Nesterov: iteration k is finished by tick k + 1
  const int64_t extra = method == MOM_NESTEROV ? 1 : 0;
  while (maxit > 0) {
    const int64_t missing = maxit + extra - ticks;
    if (missing <= 0) break;
    const int chunk = (int)std::min<int64_t>(64, missing);
    for (int k = 0; k < chunk; ++k) launch(2, roles(++ticks));
    APX_HIP(hipGetLastError());
    APX_TRY(d2h(ctx, &c, bca.p, sizeof(c)));
    if (c.done != 0 || c.it >= maxit) break;
  }
  if (maxit == 0) APX_TRY(d2h(ctx, &c, bca.p, sizeof(c)));

  // where x^(k) lives, k = c.it finished iterations
  const int64_t k = c.it;
  const void* xsrc = method == MOM_AGRAAL ? d.X[(k + 2) % 3] : d.X[k % 2];
  APX_TRY(d2h(ctx, x_out, xsrc, (size_t)p->n * es));
  const bool conv = c.done == 1;
  if (host_recs != nullptr && rec_cap > 0)
    APX_TRY(d2h(ctx, host_recs, brecs.p, (size_t)std::min<int64_t>(k, rec_cap) * sizeof(apx_record)));
  if (res != nullptr) {
    memset(res, 0, sizeof(*res));
    res->numit = k;
    res->converged = conv ? 1 : 0;
    res->flags = c.flags;
    res->gamma = c.gamma;
    res->sigma = std::numeric_limits<double>::quiet_NaN();
    res->norm_res = c.norm_res;
    res->objective = c.objective;
    // aGRAAL: two gradients before the loop, one after every iteration that goes on (:166-167, :189)
    const int64_t evals = method == MOM_AGRAAL ? 2 + (conv ? k - 1 : k) : k;
    res->f_evals = evals;
    res->grad_f_evals = evals;
    res->prox_g_evals = k;
    res->prox_h_evals = -1;
    res->A_evals = -1;
    res->At_evals = -1;
    res->n_records = std::min<int64_t>(k, rec_cap);
  }
  return APX_OK;
}

}  // namespace apx
