// vecops_dev.h — device-side building blocks shared by the fused vector kernels (vecops.hip) and the
// two-kernel small-problem iteration (small.hip): block reductions, the prox operators in element
// form, the f epilogue, the step-size rules and the closing of an iteration.
//
// Element arithmetic follows the order of the Julia source and is compiled with
// -ffp-contract=off (Julia does not contract `x - gamma*g` into an fma); explicit
// fma() is only used inside dot-product accumulations.
#pragma once
#include <cmath>

#include "apx_internal.h"
#include "gemv_kernels.h"

namespace apx {

// ---------------------------------------------------------------------------------
// block-level helpers
// ---------------------------------------------------------------------------------
template <int Q>
__device__ __forceinline__ void block_partials_store(double (&v)[Q], double* __restrict__ red) {
  __shared__ double sh[kBlock / 64][Q];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    const double s = wave_sum(v[q]);
    if (lane == 0) sh[wv][q] = s;
  }
  __syncthreads();
  if (threadIdx.x < Q) {
    double s = sh[0][threadIdx.x];
#pragma unroll
    for (int w = 1; w < kBlock / 64; ++w) s += sh[w][threadIdx.x];
    red[(int64_t)threadIdx.x * kMaxRedBlocks + blockIdx.x] = s;
  }
}

// single-block: sum `nblk` partials of quantity q (fixed order), result valid on every thread
__device__ __forceinline__ double reduce_partials(const double* __restrict__ red, int q, int nblk) {
  __shared__ double sh[kBlock / 64];
  __shared__ double total;
  double s = 0.0;
  for (int i = threadIdx.x; i < nblk; i += kBlock) s += red[(int64_t)q * kMaxRedBlocks + i];
  s = wave_sum(s);
  __syncthreads();   // protects sh/total across successive calls
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = sh[0];
#pragma unroll
    for (int w = 1; w < kBlock / 64; ++w) t += sh[w];
    total = t;
  }
  __syncthreads();
  return total;
}

// ---------------------------------------------------------------------------------
// prox operators, element form (ProximalOperators.jl / ProximalCore.jl restated,
// SURVEY §8a P1-P4; same forms as oracle/adaprox_oracle.py)
// ---------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ double shift_at(const ProxDev& g, int64_t i) {
  return g.has_shift ? (double)reinterpret_cast<const T*>(g.shift)[i] : 0.0;
}

// prox_{gamma g}(z) for the un-translated term; `aux` = ||z||_2 for NormL2
template <typename T>
__device__ __forceinline__ double prox_core(const ProxDev& g, double z, double gamma, int64_t i, double aux) {
  switch (g.kind) {
    case APX_G_L1: {   // y = z + (z <= -gl ? gl : (z >= gl ? -gl : -z))
      const double gl = gamma * g.lambda;
      return z + (z <= -gl ? gl : (z >= gl ? -gl : -z));
    }
    case APX_G_BOX:
      return z < g.lb ? g.lb : (z > g.ub ? g.ub : z);
    case APX_G_BOXV: {
      const double lb = (double)reinterpret_cast<const T*>(g.lb_vec)[i];
      const double ub = (double)reinterpret_cast<const T*>(g.ub_vec)[i];
      return z < lb ? lb : (z > ub ? ub : z);
    }
    case APX_G_INDZERO:
      return 0.0;
    case APX_G_L2: {   // scale = max(0, 1 - lambda*gamma/||z||)
      const double sc = fmax(0.0, 1.0 - g.lambda * gamma / aux);
      return sc * z;
    }
    default:
      return z;   // Zero
  }
}

// Translate(g, b): z = v + b ; y = prox_g(z) ; y -= b.  The `_s` forms take the shift value b_i already
// loaded (0 without a Translate) so that callers can batch that load with their other operands.
template <typename T>
__device__ __forceinline__ double prox_elem_s(const ProxDev& g, double v, double gamma, int64_t i, double aux, double b) {
  if (g.has_shift) return prox_core<T>(g, v + b, gamma, i, aux) - b;
  return prox_core<T>(g, v, gamma, i, aux);
}
template <typename T>
__device__ __forceinline__ double prox_elem(const ProxDev& g, double v, double gamma, int64_t i, double aux) {
  return prox_elem_s<T>(g, v, gamma, i, aux, shift_at<T>(g, i));
}

// prox of convex_conjugate(h) at w with step sigma (src/AdaProx.jl:325,345):
//   Zero -> IndZero (0) ; IndZero -> Zero (w) ; otherwise Moreau:
//   u = w/sigma ; p = prox_{h,1/sigma}(u) ; y = w - sigma*p      (ProximalCore prox_conjugate!)
template <typename T>
__device__ __forceinline__ double prox_conj_elem_s(const ProxDev& h, double w, double sigma, int64_t i, double aux,
                                                   double b) {
  if (h.kind == APX_G_ZERO && !h.has_shift) return 0.0;
  if (h.kind == APX_G_INDZERO && !h.has_shift) return w;
  const double u = w / sigma;
  const double p = prox_elem_s<T>(h, u, 1.0 / sigma, i, aux, b);
  return w - sigma * p;
}
template <typename T>
__device__ __forceinline__ double prox_conj_elem(const ProxDev& h, double w, double sigma, int64_t i, double aux) {
  return prox_conj_elem_s<T>(h, w, sigma, i, aux, shift_at<T>(h, i));
}

// value g(x): accumulate `gval_elem`, finish with `gval_final`.  gval_core is the
// un-translated term; Translate(g, b)(x) = g(x + b).
template <typename T>
__device__ __forceinline__ double gval_core(const ProxDev& g, double z, int64_t i) {
  switch (g.kind) {
    case APX_G_L1: return fabs(z);
    case APX_G_L2: return z * z;
    case APX_G_BOX: return (z < g.lb || z > g.ub) ? 1.0 : 0.0;
    case APX_G_BOXV: {
      const double lb = (double)reinterpret_cast<const T*>(g.lb_vec)[i];
      const double ub = (double)reinterpret_cast<const T*>(g.ub_vec)[i];
      return (z < lb || z > ub) ? 1.0 : 0.0;
    }
    case APX_G_INDZERO: return z != 0.0 ? 1.0 : 0.0;
    default: return 0.0;
  }
}
template <typename T>
__device__ __forceinline__ double gval_elem(const ProxDev& g, double x, int64_t i) {
  return gval_core<T>(g, x + shift_at<T>(g, i), i);
}
template <typename T>
__device__ __forceinline__ double gval_elem_s(const ProxDev& g, double x, int64_t i, double b) {
  return gval_core<T>(g, x + b, i);
}
// prox that also accumulates the value the prox call returns: for Translate that is the
// value of the INNER prox at the inner point (`v = prox!(y, f.f, z, gamma)`), not g(y + b)
template <typename T>
__device__ __forceinline__ double prox_elem_val(const ProxDev& g, double v, double gamma, int64_t i, double aux,
                                                double* val) {
  const double b = shift_at<T>(g, i);
  const double z = g.has_shift ? v + b : v;
  const T p = (T)prox_core<T>(g, z, gamma, i, aux);
  *val += gval_core<T>(g, (double)p, i);
  return g.has_shift ? (double)p - b : (double)p;
}
__device__ __forceinline__ double gval_final(const ProxDev& g, double s) {
  switch (g.kind) {
    case APX_G_L1: return g.lambda * s;
    case APX_G_L2: return g.lambda * sqrt(s);
    case APX_G_BOX:
    case APX_G_BOXV:
    case APX_G_INDZERO: return s > 0.0 ? INFINITY : 0.0;
    default: return 0.0;
  }
}

// single block: reduce the f partials into fsc[0..1]; FINAL also publishes f_x / bias_grad
__device__ __forceinline__ void f_finalize(Ctrl* c, int fk, const double* fsc) {
  if (fk == APX_F_LEAST_SQUARES) {
    const double nr = sqrt(fsc[0]);          // 0.5 * norm(res)^2 : nrm2 THEN square (Appendix A.2)
    c->f_x = 0.5 * (nr * nr);
  } else if (fk == APX_F_QUADRATIC) {
    c->f_x = 0.5 * fsc[0] + fsc[1];          // 0.5*dot(x, temp) + dot(x, q)
  } else {
    c->f_x = -(fsc[0] / c->n_mean);          // -mean(...)
    c->bias_grad = fsc[1] / c->n_mean;       // mean(probs - y)
  }
}

// ---------------------------------------------------------------------------------
// step-size rules (src/AdaProx.jl:208-308), scalar, one thread
// ---------------------------------------------------------------------------------
__device__ __forceinline__ double jl_min2(double a, double b) {   // Julia min: NaN-propagating
  if (isnan(a) || isnan(b)) return NAN;
  return a < b ? a : b;
}
__device__ __forceinline__ double jl_sqrt(double v, int32_t* flags) {   // Julia: DomainError for v < 0
  if (v < 0.0) { *flags |= APX_FLAG_DOMAIN; return NAN; }
  return sqrt(v);
}
__device__ __forceinline__ double nan_to_zero(double v) { return isnan(v) ? 0.0 : v; }   // :24

__device__ inline void rule_update(Ctrl* c, double s_dg2, double s_dgdx, double s_dx2) {
  int32_t fl = 0;
  const double gamma_prev = c->gamma;
  double gamma = c->gamma, sigma = c->sigma;
  switch (c->rule_kind) {
    case APX_RULE_FIXED:   // :213-215
      gamma = c->r_gamma;
      sigma = c->r_gamma * (c->r_t * c->r_t);
      break;
    case APX_RULE_MM: {    // :226-230
      const double g_prev = c->st0, rho = c->st1;
      const double L = sqrt(s_dg2) / sqrt(s_dx2);
      gamma = jl_min2(jl_sqrt(1.0 + rho, &fl) * g_prev, 1.0 / (2.0 * L));
      sigma = gamma * (c->r_t * c->r_t);
      c->st0 = gamma;
      c->st1 = gamma / g_prev;
      break;
    }
    case APX_RULE_OURS: {  // :258-273
      const double g1 = c->st0, g0 = c->st1;
      const double ndg = sqrt(s_dg2), ndx = sqrt(s_dx2);
      const double xi = (c->r_t * c->r_t) * (g1 * g1) * (c->r_normA * c->r_normA);
      const double C = nan_to_zero((ndg * ndg) / s_dgdx);
      const double L = nan_to_zero(s_dgdx / (ndx * ndx));
      const double D = g1 * L * (g1 * C - 1.0);
      const double d1 = 1.0 + c->r_delta;
      const double inner = 1.0 - 4.0 * xi * (d1 * d1);
      const double t1 = g1 * jl_sqrt(1.0 + g1 / g0, &fl);
      const double t2 = 1.0 / (2.0 * c->r_Theta * c->r_t * c->r_normA);
      const double t3 = g1 * jl_sqrt(inner, &fl) /
                        jl_sqrt(2.0 * d1 * (D + jl_sqrt(D * D + xi * inner, &fl)), &fl);
      gamma = jl_min2(jl_min2(t1, t2), t3);
      sigma = gamma * (c->r_t * c->r_t);
      c->st0 = gamma;
      c->st1 = g1;
      break;
    }
    case APX_RULE_OURS_PLUS: {   // :299-308
      const double g1 = c->st0, g0 = c->st1;
      const double ndg = sqrt(s_dg2), ndx = sqrt(s_dx2);
      const double C = nan_to_zero((ndg * ndg) / s_dgdx);
      const double L = nan_to_zero(s_dgdx / (ndx * ndx));
      const double D = nan_to_zero(1.0 - 2.0 * c->r_r + g1 * L * (g1 * C + 2.0 * (c->r_r - 1.0)));
      const double a = jl_sqrt(1.0 / (c->r_r * (c->r_nu + c->r_xi)) + g1 / g0, &fl);
      const double b = jl_sqrt((c->r_nu * (1.0 + c->r_xi) - 1.0) / (c->r_nu * (c->r_nu + c->r_xi)), &fl) /
                       jl_sqrt(fmax(D, 0.0), &fl);
      gamma = g1 * jl_min2(a, b);
      sigma = gamma;
      c->st0 = gamma;
      c->st1 = g1;
      break;
    }
  }
  if (isnan(gamma)) fl |= APX_FLAG_NAN_GAMMA;
  c->gamma_prev = gamma_prev;
  c->gamma = gamma;
  c->sigma = sigma;
  c->rho = gamma / gamma_prev;     // :342
  c->flags |= fl;
}

__device__ __forceinline__ void finish_iteration(Ctrl* c, apx_record* recs, int64_t rec_cap) {
  // norm_res = sqrt(norm(primal_res)^2 + norm(dual_res)^2)           (:348)
  c->norm_res = sqrt(c->npr2 + c->ndr2);
  c->objective = c->f_x + c->g_x + c->h_Ax;                          // (:351)
  if (recs != nullptr && c->it <= rec_cap) {
    apx_record r;
    r.it = c->it;
    r.gamma = c->gamma;
    r.sigma = c->sigma;
    r.norm_res = c->norm_res;
    r.objective = c->objective;
    r.f_x = c->f_x;
    r.f_evals = c->it + 1;
    r.grad_evals = c->it + 1;
    recs[c->it - 1] = r;
  }
  if (c->norm_res <= c->tol) c->done = 1;                            // (:354)
}

}  // namespace apx
