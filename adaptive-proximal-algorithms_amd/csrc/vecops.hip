// vecops.hip — the O(m) / O(n) part of the iteration as a handful of fused kernels
// (K2, K4-K11 of SURVEY §2.1).  Every reduction is two-stage (block partials in a fixed
// order -> one single-block kernel), no atomics, so iterates are bit-reproducible.
// All loop scalars (gamma, sigma, rho, rule state, residual, done flag) live in the
// device `Ctrl` block: the host never has to read anything back inside the loop.
//
// Element arithmetic follows the order of the Julia source and is compiled with
// -ffp-contract=off (Julia does not contract `x - gamma*g` into an fma); explicit
// fma() is only used inside dot-product accumulations.
#include <algorithm>
#include <cmath>

#include "apx_internal.h"
#include "gemv_kernels.h"
#include "vecops.h"
#include "vecops_dev.h"

namespace apx {



// ---------------------------------------------------------------------------------
// K2 / K4: second stage of gemv_n fused with the f-specific epilogue
//   least squares: res = A*w - b ; sum res^2                     (experiments/lasso/runme.jl:22,24)
//   logistic     : logits = X*w[1:end-1] .+ w[end] ; u = 1 + exp(-logits)
//                  loss terms (y-1)*logits - log(u) ; res = 1/u - y   (sparse_logreg/runme.jl:24-36)
// ---------------------------------------------------------------------------------
template <typename T, int FK>
__global__ __launch_bounds__(kBlock) void k_finish_rows(const double* __restrict__ part, int splits, int64_t ld,
                                                        int64_t m, const T* __restrict__ b,
                                                        const T* __restrict__ x, int64_t n,
                                                        T* __restrict__ res, double* __restrict__ red,
                                                        const int* __restrict__ done) {
  if (done != nullptr && *done != 0) return;
  double q[2] = {0.0, 0.0};
  double bias = 0.0;
  if constexpr (FK == APX_F_LOGISTIC) bias = (double)x[n - 1];
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < m; i += (int64_t)gridDim.x * kBlock) {
    const double s = sum_splits(part, splits, ld, i);
    if constexpr (FK == APX_F_LEAST_SQUARES) {
      const T r = (T)(s - (double)b[i]);
      res[i] = r;
      q[0] = fma((double)r, (double)r, q[0]);
    } else if constexpr (FK == APX_F_QUADRATIC) {
      // temp = Q*x ; f = 0.5*dot(x, temp) + dot(x, q)   (experiments/dual_svm/runme.jl:24-27)
      const T t = (T)s;
      res[i] = t;
      const double xi = (double)x[i];
      q[0] = fma(xi, (double)t, q[0]);
      q[1] = fma(xi, (double)b[i], q[1]);
    } else {
      const double logit = s + bias;
      const double u = 1.0 + exp(-logit);
      const double yi = (double)b[i];
      q[0] += (yi - 1.0) * logit - log(u);
      const T d = (T)(1.0 / u - yi);
      res[i] = d;
      q[1] += (double)d;
    }
  }
  block_partials_store<2>(q, red);
}


__global__ __launch_bounds__(kBlock) void k_f_reduce(const double* __restrict__ red, int nblk, int fk,
                                                     double* __restrict__ fsc, Ctrl* c, int finalize) {
  if (c->done != 0) return;
  const double s0 = reduce_partials(red, 0, nblk);
  const double s1 = reduce_partials(red, 1, nblk);
  if (threadIdx.x == 0) {
    fsc[0] = s0;
    fsc[1] = s1;
    if (finalize) f_finalize(c, fk, fsc);
  }
}

__global__ void k_f_final(int fk, const double* __restrict__ fsc, Ctrl* c) {
  if (c->done != 0) return;
  if (threadIdx.x == 0 && blockIdx.x == 0) f_finalize(c, fk, fsc);
}

// logistic pullback tail: grad[1:end-1] = X'*(probs - y) ./ N ; grad[end] = mean(probs - y)
template <typename T>
__global__ __launch_bounds__(kBlock) void k_logistic_grad(T* __restrict__ grad, int64_t n, const Ctrl* c) {
  if (c->done != 0) return;
  const double N = c->n_mean;
  for (int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x; j < n; j += (int64_t)gridDim.x * kBlock)
    grad[j] = (j == n - 1) ? (T)c->bias_grad : (T)((double)grad[j] / N);
}

// ---------------------------------------------------------------------------------
// K7 + K8 (+ K11): primal residual and the step-size reductions in one pass over n
//   primal_res = (v - x)/gamma + grad_x + At_y                     (src/AdaProx.jl:338)
//   ||dg||^2, <dg,dx>, ||dx||^2, dg = grad_x - grad_x_prev, dx = x - x_prev   (:260-261, :227)
// ---------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(kBlock) void k_reduce_n(const T* __restrict__ x, const T* __restrict__ xprev,
                                                     const T* __restrict__ grad, const T* __restrict__ gprev,
                                                     const T* __restrict__ v, const T* __restrict__ aty,
                                                     int64_t n, ProxDev g, const Ctrl* c,
                                                     double* __restrict__ red) {
  if (c->done != 0) return;
  const double gamma = c->gamma;
  double q[5] = {0, 0, 0, 0, 0};
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
    const double xi = (double)x[i], gi = (double)grad[i];
    double pr = ((double)v[i] - xi) / gamma + gi;
    if (aty != nullptr) pr = pr + (double)aty[i];
    const double dg = gi - (double)gprev[i];
    const double dx = xi - (double)xprev[i];
    q[0] = fma(pr, pr, q[0]);
    q[1] = fma(dg, dg, q[1]);
    q[2] = fma(dg, dx, q[2]);
    q[3] = fma(dx, dx, q[3]);
    q[4] += gval_elem<T>(g, xi, i);
  }
  block_partials_store<5>(q, red);
}


// single block: finish K7/K8, update the rule (:340-342); without a linear operator also
// close the iteration (dual_res == 0 for adaptive_proxgrad: h = Zero, A = 0)
__global__ __launch_bounds__(kBlock) void k_rule(const double* __restrict__ red, int nblk, ProxDev g, Ctrl* c,
                                                 int has_lin, apx_record* recs, int64_t rec_cap) {
  if (c->done != 0) return;
  const double s_pr = reduce_partials(red, 0, nblk);
  const double s_dg2 = reduce_partials(red, 1, nblk);
  const double s_dgdx = reduce_partials(red, 2, nblk);
  const double s_dx2 = reduce_partials(red, 3, nblk);
  const double s_g = reduce_partials(red, 4, nblk);
  if (threadIdx.x != 0) return;
  c->it += 1;
  const double npr = sqrt(s_pr);
  c->npr2 = npr * npr;
  c->g_x = gval_final(g, s_g);
  rule_update(c, s_dg2, s_dgdx, s_dx2);
  if (!has_lin) {
    c->ndr2 = 0.0;
    c->h_Ax = 0.0;
    finish_iteration(c, recs, rec_cap);
  }
}

// ---------------------------------------------------------------------------------
// K9: dual extrapolation, conjugate prox, dual residual (src/AdaProx.jl:344-347)
// ---------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(kBlock) void k_dual_pre(const T* __restrict__ y, const T* __restrict__ ax,
                                                     const T* __restrict__ axprev, int64_t m, ProxDev h,
                                                     const Ctrl* c, double* __restrict__ red) {
  if (c->done != 0) return;
  const double sigma = c->sigma, rho = c->rho;
  double q[1] = {0.0};
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < m; i += (int64_t)gridDim.x * kBlock) {
    const double a = (double)ax[i];
    const double w = (double)y[i] + sigma * ((1.0 + rho) * a - rho * (double)axprev[i]);
    const double z = w / sigma + shift_at<T>(h, i);
    q[0] = fma(z, z, q[0]);
  }
  block_partials_store<1>(q, red);
}

__global__ __launch_bounds__(kBlock) void k_dual_pre_final(const double* __restrict__ red, int nblk, Ctrl* c) {
  if (c->done != 0) return;
  const double s = reduce_partials(red, 0, nblk);
  if (threadIdx.x == 0) c->h_pre = sqrt(s);
}

template <typename T>
__global__ __launch_bounds__(kBlock) void k_dual(T* __restrict__ y, const T* __restrict__ ax,
                                                 T* __restrict__ axprev, int64_t m, ProxDev h, const Ctrl* c,
                                                 double* __restrict__ red) {
  if (c->done != 0) return;
  const double sigma = c->sigma, rho = c->rho, aux = c->h_pre;
  double q[2] = {0.0, 0.0};
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < m; i += (int64_t)gridDim.x * kBlock) {
    const double a = (double)ax[i];
    const double w = (double)y[i] + sigma * ((1.0 + rho) * a - rho * (double)axprev[i]);   // :344
    const T yn = (T)prox_conj_elem<T>(h, w, sigma, i, aux);                                // :345
    const double dr = (w - (double)yn) / sigma - a;                                        // :347
    q[0] = fma(dr, dr, q[0]);
    q[1] += gval_elem<T>(h, a, i);
    y[i] = yn;
    axprev[i] = (T)a;                                                                      // :360 (A_x_prev)
  }
  block_partials_store<2>(q, red);
}

// The same behind a one-pass launch with h = NormL2 (csrc/slab_pd.hip, csrc/team_rows.h team_pd_dual): the row kernel left
// w[i] in y[i]'s place and the new A_x[i] in A_x_prev[i], the closing kernel ||z|| in Ctrl::h_pre — y+ and the dual residual in
// exactly k_dual's arithmetic.
template <typename T>
__global__ __launch_bounds__(kBlock) void k_dual_post(T* __restrict__ y, const T* __restrict__ axnew, int64_t m, ProxDev h, const Ctrl* c,
                                                      double* __restrict__ red) {
  if (c->done != 0) return;
  const double sigma = c->sigma, aux = c->h_pre;
  double q[2] = {0.0, 0.0};
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < m; i += (int64_t)gridDim.x * kBlock) {
    const double a = (double)axnew[i];
    const double w = (double)y[i];
    const T yn = (T)prox_conj_elem<T>(h, w, sigma, i, aux);                                // :345
    const double dr = (w - (double)yn) / sigma - a;                                        // :347
    q[0] = fma(dr, dr, q[0]);
    q[1] += gval_elem<T>(h, a, i);
    y[i] = yn;
  }
  block_partials_store<2>(q, red);
}

__global__ __launch_bounds__(kBlock) void k_final(const double* __restrict__ red, int nblk, ProxDev h, Ctrl* c,
                                                  apx_record* recs, int64_t rec_cap) {
  if (c->done != 0) return;
  const double s_dr = reduce_partials(red, 0, nblk);
  const double s_h = reduce_partials(red, 1, nblk);
  if (threadIdx.x != 0) return;
  const double ndr = sqrt(s_dr);
  c->ndr2 = ndr * ndr;
  c->h_Ax = gval_final(h, s_h);
  finish_iteration(c, recs, rec_cap);
}

// ---------------------------------------------------------------------------------
// K5 + K6: v = x - gamma*(grad_x + At_y) ; rotate prev buffers ; x = prox_{gamma g}(v)
// (src/AdaProx.jl:330-332, :359-361)
// ---------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(kBlock) void k_step(T* __restrict__ x, T* __restrict__ xprev,
                                                 const T* __restrict__ grad, T* __restrict__ gprev,
                                                 T* __restrict__ v, const T* __restrict__ aty, int64_t n,
                                                 ProxDev g, const Ctrl* c) {
  if (c->done != 0) return;
  const double gamma = c->gamma;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
    const double xi = (double)x[i], gi = (double)grad[i];
    double gs = gi;
    if (aty != nullptr) gs = gi + (double)aty[i];
    const T vi = (T)(xi - gamma * gs);
    v[i] = vi;
    xprev[i] = (T)xi;
    gprev[i] = (T)gi;
    x[i] = (T)prox_elem<T>(g, (double)vi, gamma, i, 0.0);
  }
}

// K5 + K6 of the primal-dual loop with f = Zero, and K7 + K8's sums of the NEXT iteration with them: everything k_reduce_n
// reads at the top of iteration k + 1 — x, x_prev, v, A'y, and a gradient that is identically zero — is in this kernel's hands at
// the end of iteration k, so it leaves k_reduce_n's block partials itself (the same grid, the same thread-to-entry mapping, the
// same order: bit-identical sums) and the next iteration starts at k_rule.  One launch fewer per iteration of the loop C5 runs.
template <typename T>
__global__ __launch_bounds__(kBlock) void k_step_red(T* __restrict__ x, T* __restrict__ xprev, const T* __restrict__ grad,
                                                     T* __restrict__ gprev, T* __restrict__ v, const T* __restrict__ aty, int64_t n,
                                                     ProxDev g, const Ctrl* c, double* __restrict__ red) {
  if (c->done != 0) return;
  const double gamma = c->gamma;
  double q[5] = {0, 0, 0, 0, 0};
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
    const double xi = (double)x[i], gi = (double)grad[i];
    double gs = gi;
    if (aty != nullptr) gs = gi + (double)aty[i];
    const T vi = (T)(xi - gamma * gs);
    const T xn = (T)prox_elem<T>(g, (double)vi, gamma, i, 0.0);
    v[i] = vi;
    xprev[i] = (T)xi;
    gprev[i] = (T)gi;
    x[i] = xn;
    // k_reduce_n at the new point (its gamma is still this one: the rule runs behind it ; grad f = 0 there as here)
    const double xn_d = (double)xn;
    double pr = ((double)vi - xn_d) / gamma + gi;
    if (aty != nullptr) pr = pr + (double)aty[i];
    const double dg = gi - (double)(T)gi;
    const double dx = xn_d - (double)(T)xi;
    q[0] = fma(pr, pr, q[0]);
    q[1] = fma(dg, dg, q[1]);
    q[2] = fma(dg, dx, q[2]);
    q[3] = fma(dx, dx, q[3]);
    q[4] += gval_elem<T>(g, xn_d, i);
  }
  block_partials_store<5>(q, red);
}

// K5 + K6 behind the one-pass kernels (csrc/slab.hip SlabTail): the totals of the closing kernel's wave partials — in k_rule's
// order, by EVERY block: the same bits everywhere, no grid-wide synchronisation —, the rule and the end of the iteration
// (:340-342, :348-354) on a private copy of the control block's snapshot, then the step (:359-361) with the new gamma.  Block 0
// writes the control block back ; nobody reads it here except for `done`, which only ever goes from 0 to 1 with the same
// outcome (no step) for a block that sees it early.
template <typename T>
__global__ __launch_bounds__(kBlock) void k_step_rule(T* __restrict__ x, T* __restrict__ xprev, const T* __restrict__ grad,
                                                      T* __restrict__ gprev, T* __restrict__ v, int64_t n, ProxDev g, Ctrl* c,
                                                      const Ctrl* __restrict__ snap, const double* __restrict__ wred, int ldw,
                                                      apx_record* recs, int64_t rec_cap) {
  __shared__ Ctrl lc;
  __shared__ double tsh[kBlock / 64][5];
  if (c->done != 0) return;
  const int tid = threadIdx.x;
  const int nblk = (int)((n + kBlock - 1) / kBlock);
  // blocks of four waves ((W0 + W1) + W2) + W3 — block_partials_store —, thread t the blocks t, t + 256, ..., wave sums, the
  // four waves in order — reduce_partials
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    const double* __restrict__ w = wred + (int64_t)k * ldw;
    double s = 0.0;
    for (int b = tid; b < nblk; b += kBlock) s += ((w[4 * b] + w[4 * b + 1]) + w[4 * b + 2]) + w[4 * b + 3];
    s = wave_sum(s);
    if ((tid & 63) == 0) tsh[tid >> 6][k] = s;
  }
  __syncthreads();
  if (tid == 0) {
    double tot[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) tot[k] = ((tsh[0][k] + tsh[1][k]) + tsh[2][k]) + tsh[3][k];
    lc = *snap;
    lc.it += 1;
    const double npr = sqrt(tot[0]);
    lc.npr2 = npr * npr;
    lc.g_x = gval_final(g, tot[4]);
    rule_update(&lc, tot[1], tot[2], tot[3]);
    lc.ndr2 = 0.0;
    lc.h_Ax = 0.0;
    finish_iteration(&lc, blockIdx.x == 0 ? recs : nullptr, rec_cap);
    if (blockIdx.x == 0) *c = lc;
  }
  __syncthreads();
  if (lc.done != 0) return;                        // (the test passed: x stays the tested iterate)
  const double gamma = lc.gamma;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + tid; i < n; i += (int64_t)gridDim.x * kBlock) {
    const double xi = (double)x[i], gi = (double)grad[i];
    const T vi = (T)(xi - gamma * gi);
    v[i] = vi;
    xprev[i] = (T)xi;
    gprev[i] = (T)gi;
    x[i] = (T)prox_elem<T>(g, (double)vi, gamma, i, 0.0);
  }
}

// ---------------------------------------------------------------------------------
// K10: backtracking trial  z = prox_{gamma g}(x - gamma*grad) ; <grad, z-x> ; ||z-x||^2 ; g(z)
// (src/AdaProx.jl:35-36, :26)
// ---------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(kBlock) void k_bt_trial(const T* __restrict__ x, const T* __restrict__ grad,
                                                     T* __restrict__ z, int64_t n, ProxDev g, double gamma,
                                                     double* __restrict__ red) {
  double q[3] = {0, 0, 0};
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
    const double xi = (double)x[i], gi = (double)grad[i];
    const T vi = (T)(xi - gamma * gi);
    const T zi = (T)prox_elem_val<T>(g, (double)vi, gamma, i, 0.0, &q[2]);
    const double d = (double)zi - xi;
    q[0] = fma(gi, d, q[0]);
    q[1] = fma(d, d, q[1]);
    z[i] = zi;
  }
  block_partials_store<3>(q, red);
}

// single block: raw sums of the first nq quantities -> out[0..nq)
__global__ __launch_bounds__(kBlock) void k_sum_q(const double* __restrict__ red, int nblk, int nq,
                                                  double* __restrict__ out) {
  for (int q = 0; q < nq; ++q) {
    const double s = reduce_partials(red, q, nblk);
    if (threadIdx.x == 0) out[q] = s;
  }
}

// generic prox for the op-level entry point (apx_prox): optional conjugate, optional l2 pre-pass
template <typename T>
__global__ __launch_bounds__(kBlock) void k_sumsq_shift(const T* __restrict__ v, int64_t n, ProxDev g,
                                                        double scale, double* __restrict__ red) {
  double q[1] = {0.0};
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
    const double z = (double)v[i] * scale + shift_at<T>(g, i);
    q[0] = fma(z, z, q[0]);
  }
  block_partials_store<1>(q, red);
}

template <typename T>
__global__ __launch_bounds__(kBlock) void k_prox(const T* __restrict__ v, T* __restrict__ y, int64_t n, ProxDev g,
                                                 double gamma, int conj, const double* __restrict__ aux_sumsq,
                                                 double* __restrict__ red) {
  const double aux = aux_sumsq != nullptr ? sqrt(aux_sumsq[0]) : 0.0;
  double q[1] = {0.0};
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
    const double vi = (double)v[i];
    const T yi = conj ? (T)prox_conj_elem<T>(g, vi, gamma, i, aux)
                      : (T)prox_elem_val<T>(g, vi, gamma, i, aux, &q[0]);
    y[i] = yi;
  }
  block_partials_store<1>(q, red);
}

// ---------------------------------------------------------------------------------
// AdaPDM+ linesearch trial (src/AdaProx.jl:522-527, :535): the dual update with HOST-side
// sigma / rho, written to y_next (y, A_x_prev untouched until the trial is accepted)
//   q0 = ||y_next - y||^2 ; q1 = ||(w - y_next)/sigma - A_x||^2 ; q2 = raw h(A_x)
// ---------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(kBlock) void k_dual_trial(const T* __restrict__ y, const T* __restrict__ ax,
                                                       const T* __restrict__ axprev, T* __restrict__ ynext,
                                                       int64_t m, ProxDev h, double sigma, double rho,
                                                       const double* __restrict__ aux_sumsq,
                                                       double* __restrict__ red) {
  const double aux = aux_sumsq != nullptr ? sqrt(aux_sumsq[0]) : 0.0;
  double q[3] = {0.0, 0.0, 0.0};
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < m; i += (int64_t)gridDim.x * kBlock) {
    const double a = (double)ax[i], yi = (double)y[i];
    const double w = yi + sigma * ((1.0 + rho) * a - rho * (double)axprev[i]);     // :524
    const T yn = (T)prox_conj_elem<T>(h, w, sigma, i, aux);                        // :525
    const double dy = (double)yn - yi;
    const double dr = (w - (double)yn) / sigma - a;                                // :535
    q[0] = fma(dy, dy, q[0]);
    q[1] = fma(dr, dr, q[1]);
    q[2] += gval_elem<T>(h, a, i);
    ynext[i] = yn;
  }
  block_partials_store<3>(q, red);
}

// sum (w/sigma + shift)^2 for the l2-ball conjugate prox of a trial (host sigma / rho)
template <typename T>
__global__ __launch_bounds__(kBlock) void k_dual_pre_v(const T* __restrict__ y, const T* __restrict__ ax,
                                                       const T* __restrict__ axprev, int64_t m, ProxDev h,
                                                       double sigma, double rho, double* __restrict__ red) {
  double q[1] = {0.0};
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < m; i += (int64_t)gridDim.x * kBlock) {
    const double a = (double)ax[i];
    const double w = (double)y[i] + sigma * ((1.0 + rho) * a - rho * (double)axprev[i]);
    const double z = w / sigma + shift_at<T>(h, i);
    q[0] = fma(z, z, q[0]);
  }
  block_partials_store<1>(q, red);
}

// ||a - b||^2   (`norm(At_y_next - At_y)`, :527)
template <typename T>
__global__ __launch_bounds__(kBlock) void k_diff_sumsq(const T* __restrict__ a, const T* __restrict__ b, int64_t n,
                                                       double* __restrict__ red) {
  double q[1] = {0.0};
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
    const double d = (double)a[i] - (double)b[i];
    q[0] = fma(d, d, q[0]);
  }
  block_partials_store<1>(q, red);
}

// ---------------------------------------------------------------------------------
// comparison baselines
// ---------------------------------------------------------------------------------
// out = a + c*(a - b)      Nesterov extrapolation, src/AdaProx.jl:80, :129
template <typename T>
__global__ __launch_bounds__(kBlock) void k_extrap(T* __restrict__ out, const T* __restrict__ a,
                                                   const T* __restrict__ b, double c, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
    const double ai = (double)a[i];
    out[i] = (T)(ai + c * (ai - (double)b[i]));
  }
}

// x_bar = ((phi - 1)*x + x_bar) / phi      aGRAAL, src/AdaProx.jl:179
template <typename T>
__global__ __launch_bounds__(kBlock) void k_xbar(T* __restrict__ xbar, const T* __restrict__ x, double phi,
                                                 int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock)
    xbar[i] = (T)(((phi - 1.0) * (double)x[i] + (double)xbar[i]) / phi);
}

// Malitsky-Pock trial, src/AdaProx.jl:558-560 / :571-573:
//   At_ybar = (1+theta)*At_y - theta*At_y_prev ; v = x_prev - gamma*(At_ybar + grad_prev) ; x = prox_g(v)
//   q0 = ||x - x_prev||^2 ; q1 = <grad_prev, x - x_prev> ; q2 = raw g(x)
template <typename T>
__global__ __launch_bounds__(kBlock) void k_mp_trial(const T* __restrict__ xprev, const T* __restrict__ aty,
                                                     const T* __restrict__ atyprev, const T* __restrict__ gradprev,
                                                     T* __restrict__ v, T* __restrict__ x, int64_t n, ProxDev g,
                                                     double theta, double gamma, double* __restrict__ red) {
  double q[3] = {0, 0, 0};
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
    const double xp = (double)xprev[i], gp = (double)gradprev[i];
    const double ybar = (1.0 + theta) * (double)aty[i] - theta * (double)atyprev[i];
    const T vi = (T)(xp - gamma * (ybar + gp));
    const T xi = (T)prox_elem_val<T>(g, (double)vi, gamma, i, 0.0, &q[2]);
    const double d = (double)xi - xp;
    q[0] = fma(d, d, q[0]);
    q[1] = fma(gp, d, q[1]);
    v[i] = vi;
    x[i] = xi;
  }
  block_partials_store<3>(q, red);
}

// Malitsky-Pock dual residual, src/AdaProx.jl:601, :617:  w = y_old + sigma_prev*A_x_old ;
//   dual_res = (w - y_new)/sigma_prev - A_x_new ;  q0 = ||dual_res||^2 ; q1 = raw h(A_x_new)
template <typename T>
__global__ __launch_bounds__(kBlock) void k_mp_dualres(const T* __restrict__ yold, const T* __restrict__ axold,
                                                       const T* __restrict__ ynew, const T* __restrict__ axnew,
                                                       int64_t m, double sigma_prev, ProxDev h,
                                                       double* __restrict__ red) {
  double q[2] = {0, 0};
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < m; i += (int64_t)gridDim.x * kBlock) {
    const double w = (double)yold[i] + sigma_prev * (double)axold[i];
    const double an = (double)axnew[i];
    const double dr = (w - (double)ynew[i]) / sigma_prev - an;
    q[0] = fma(dr, dr, q[0]);
    q[1] += gval_elem<T>(h, an, i);
  }
  block_partials_store<2>(q, red);
}

// primal_res = (v - x)/gamma + grad + At_y with a HOST gamma:  q0 = ||primal_res||^2 ; q1 = raw g(x)
template <typename T>
__global__ __launch_bounds__(kBlock) void k_primal_res(const T* __restrict__ v, const T* __restrict__ x,
                                                       const T* __restrict__ grad, const T* __restrict__ aty,
                                                       int64_t n, double gamma, ProxDev g,
                                                       double* __restrict__ red) {
  double q[2] = {0, 0};
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
    const double xi = (double)x[i];
    double pr = ((double)v[i] - xi) / gamma + (double)grad[i];
    if (aty != nullptr) pr = pr + (double)aty[i];
    q[0] = fma(pr, pr, q[0]);
    q[1] += gval_elem<T>(g, xi, i);
  }
  block_partials_store<2>(q, red);
}

// g(x): raw accumulator of the value of a proximable term (finished by gval_final / vo_gval_final_host)
template <typename T>
__global__ __launch_bounds__(kBlock) void k_gval(const T* __restrict__ x, int64_t n, ProxDev g,
                                                 double* __restrict__ red) {
  double q[1] = {0.0};
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock)
    q[0] += gval_elem<T>(g, (double)x[i], i);
  block_partials_store<1>(q, red);
}

// ---------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------
static inline unsigned vgrid(int64_t n) {
  return (unsigned)std::max<int64_t>(1, std::min<int64_t>(kMaxRedBlocks, (n + kBlock - 1) / kBlock));
}

#define DISPATCH_T(dtype, CALL_F64, CALL_F32) \
  do {                                        \
    if ((dtype) == APX_F32) { CALL_F32; }     \
    else { CALL_F64; }                        \
  } while (0)

int32_t vo_finish_rows(apx_ctx* ctx, int32_t dtype, int fk, int splits, int64_t ld, int64_t m, const void* b,
                       const void* x, int64_t n, void* res, Ctrl* c, double* fsc, bool finalize) {
  const unsigned g = vgrid(m);
  const int* done = &c->done;
  if (fk == APX_F_LEAST_SQUARES) {
    DISPATCH_T(dtype,
               hipLaunchKernelGGL((k_finish_rows<double, APX_F_LEAST_SQUARES>), dim3(g), dim3(kBlock), 0, ctx->stream,
                                  ctx->part, splits, ld, m, (const double*)b, (const double*)x, n, (double*)res,
                                  ctx->red, done),
               hipLaunchKernelGGL((k_finish_rows<float, APX_F_LEAST_SQUARES>), dim3(g), dim3(kBlock), 0, ctx->stream,
                                  ctx->part, splits, ld, m, (const float*)b, (const float*)x, n, (float*)res,
                                  ctx->red, done));
  } else if (fk == APX_F_QUADRATIC) {
    DISPATCH_T(dtype,
               hipLaunchKernelGGL((k_finish_rows<double, APX_F_QUADRATIC>), dim3(g), dim3(kBlock), 0, ctx->stream,
                                  ctx->part, splits, ld, m, (const double*)b, (const double*)x, n, (double*)res,
                                  ctx->red, done),
               hipLaunchKernelGGL((k_finish_rows<float, APX_F_QUADRATIC>), dim3(g), dim3(kBlock), 0, ctx->stream,
                                  ctx->part, splits, ld, m, (const float*)b, (const float*)x, n, (float*)res,
                                  ctx->red, done));
  } else {
    DISPATCH_T(dtype,
               hipLaunchKernelGGL((k_finish_rows<double, APX_F_LOGISTIC>), dim3(g), dim3(kBlock), 0, ctx->stream,
                                  ctx->part, splits, ld, m, (const double*)b, (const double*)x, n, (double*)res,
                                  ctx->red, done),
               hipLaunchKernelGGL((k_finish_rows<float, APX_F_LOGISTIC>), dim3(g), dim3(kBlock), 0, ctx->stream,
                                  ctx->part, splits, ld, m, (const float*)b, (const float*)x, n, (float*)res,
                                  ctx->red, done));
  }
  hipLaunchKernelGGL(k_f_reduce, dim3(1), dim3(kBlock), 0, ctx->stream, ctx->red, (int)g, fk, fsc, c,
                     finalize ? 1 : 0);
  APX_HIP(hipGetLastError());
  return APX_OK;
}

int32_t vo_f_final(apx_ctx* ctx, int fk, const double* fsc, Ctrl* c) {
  hipLaunchKernelGGL(k_f_final, dim3(1), dim3(64), 0, ctx->stream, fk, fsc, c);
  APX_HIP(hipGetLastError());
  return APX_OK;
}

int32_t vo_logistic_grad(apx_ctx* ctx, int32_t dtype, void* grad, int64_t n, const Ctrl* c) {
  const unsigned g = vgrid(n);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(k_logistic_grad<double>, dim3(g), dim3(kBlock), 0, ctx->stream, (double*)grad, n, c),
             hipLaunchKernelGGL(k_logistic_grad<float>, dim3(g), dim3(kBlock), 0, ctx->stream, (float*)grad, n, c));
  APX_HIP(hipGetLastError());
  return APX_OK;
}

int32_t vo_reduce_rule(apx_ctx* ctx, int32_t dtype, const void* x, const void* xprev, const void* grad,
                       const void* gprev, const void* v, const void* aty, int64_t n, const ProxDev& g, Ctrl* c,
                       bool has_lin, apx_record* recs, int64_t rec_cap) {
  const unsigned gr = vgrid(n);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(k_reduce_n<double>, dim3(gr), dim3(kBlock), 0, ctx->stream, (const double*)x,
                                (const double*)xprev, (const double*)grad, (const double*)gprev, (const double*)v,
                                (const double*)aty, n, g, c, ctx->red),
             hipLaunchKernelGGL(k_reduce_n<float>, dim3(gr), dim3(kBlock), 0, ctx->stream, (const float*)x,
                                (const float*)xprev, (const float*)grad, (const float*)gprev, (const float*)v,
                                (const float*)aty, n, g, c, ctx->red));
  hipLaunchKernelGGL(k_rule, dim3(1), dim3(kBlock), 0, ctx->stream, ctx->red, (int)gr, g, c, has_lin ? 1 : 0, recs,
                     rec_cap);
  APX_HIP(hipGetLastError());
  return APX_OK;
}

int32_t vo_dual(apx_ctx* ctx, int32_t dtype, void* y, const void* ax, void* axprev, int64_t m, const ProxDev& h,
                Ctrl* c, apx_record* recs, int64_t rec_cap) {
  const unsigned gr = vgrid(m);
  if (h.kind == APX_G_L2) {
    DISPATCH_T(dtype,
               hipLaunchKernelGGL(k_dual_pre<double>, dim3(gr), dim3(kBlock), 0, ctx->stream, (const double*)y,
                                  (const double*)ax, (const double*)axprev, m, h, c, ctx->red),
               hipLaunchKernelGGL(k_dual_pre<float>, dim3(gr), dim3(kBlock), 0, ctx->stream, (const float*)y,
                                  (const float*)ax, (const float*)axprev, m, h, c, ctx->red));
    hipLaunchKernelGGL(k_dual_pre_final, dim3(1), dim3(kBlock), 0, ctx->stream, ctx->red, (int)gr, c);
  }
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(k_dual<double>, dim3(gr), dim3(kBlock), 0, ctx->stream, (double*)y, (const double*)ax,
                                (double*)axprev, m, h, c, ctx->red),
             hipLaunchKernelGGL(k_dual<float>, dim3(gr), dim3(kBlock), 0, ctx->stream, (float*)y, (const float*)ax,
                                (float*)axprev, m, h, c, ctx->red));
  hipLaunchKernelGGL(k_final, dim3(1), dim3(kBlock), 0, ctx->stream, ctx->red, (int)gr, h, c, recs, rec_cap);
  APX_HIP(hipGetLastError());
  return APX_OK;
}

int32_t vo_dual_post(apx_ctx* ctx, int32_t dtype, void* y, const void* axnew, int64_t m, const ProxDev& h, Ctrl* c, apx_record* recs,
                     int64_t rec_cap) {
  const unsigned gr = vgrid(m);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(k_dual_post<double>, dim3(gr), dim3(kBlock), 0, ctx->stream, (double*)y, (const double*)axnew, m, h, c, ctx->red),
             hipLaunchKernelGGL(k_dual_post<float>, dim3(gr), dim3(kBlock), 0, ctx->stream, (float*)y, (const float*)axnew, m, h, c, ctx->red));
  hipLaunchKernelGGL(k_final, dim3(1), dim3(kBlock), 0, ctx->stream, ctx->red, (int)gr, h, c, recs, rec_cap);
  APX_HIP(hipGetLastError());
  return APX_OK;
}

// the same with the rows of A (hence y, A_x, dual_res) sharded over ranks: the block partials are
// summed to device scalars, all-reduced by `reduce`, and the iteration is closed from the global sums
__global__ void k_dual_pre_final_s(const double* __restrict__ sums, Ctrl* c) {
  if (c->done != 0) return;
  if (threadIdx.x == 0 && blockIdx.x == 0) c->h_pre = sqrt(sums[0]);
}
__global__ void k_final_s(const double* __restrict__ sums, ProxDev h, Ctrl* c, apx_record* recs, int64_t rec_cap) {
  if (c->done != 0) return;
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const double ndr = sqrt(sums[0]);
  c->ndr2 = ndr * ndr;
  c->h_Ax = gval_final(h, sums[1]);
  finish_iteration(c, recs, rec_cap);
}

// A' y+ = sigma (1 - sc) A' z for the l2 ball on a row shard of the operator: `aty` holds the all-reduced A' z, sums[0] the
// all-reduced sum z^2 (csrc/slab_pd.hip) ; leaves ||z|| in Ctrl::h_pre for k_dual_post
template <typename T>
__global__ __launch_bounds__(kBlock) void k_l2_scale(T* __restrict__ aty, int64_t n, const double* __restrict__ sums, ProxDev h, Ctrl* c) {
  if (c->done != 0) return;
  const double sigma = c->sigma, aux = sqrt(sums[0]);
  const double sc = fmax(0.0, 1.0 - h.lambda * (1.0 / sigma) / aux);                       // csrc/vecops_dev.h prox_core, gamma = 1 / sigma
  const double fac = sigma * (1.0 - sc);
  for (int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x; j < n; j += (int64_t)gridDim.x * kBlock) aty[j] = (T)((double)aty[j] * fac);
  if (blockIdx.x == 0 && threadIdx.x == 0) c->h_pre = aux;
}
int32_t vo_l2_scale(apx_ctx* ctx, int32_t dtype, void* aty, int64_t n, const double* sums, const ProxDev& h, Ctrl* c) {
  const unsigned gr = vgrid(n);
  // (h_pre is written by block 0 and read by the NEXT kernel only)
  DISPATCH_T(dtype, hipLaunchKernelGGL(k_l2_scale<double>, dim3(gr), dim3(kBlock), 0, ctx->stream, (double*)aty, n, sums, h, c),
             hipLaunchKernelGGL(k_l2_scale<float>, dim3(gr), dim3(kBlock), 0, ctx->stream, (float*)aty, n, sums, h, c));
  APX_HIP(hipGetLastError());
  return APX_OK;
}
// k_dual_post on a row shard: this rank's two sums (sum dual_res^2, raw h(A_x)) to `out2` on the device, for the all-reduce
int32_t vo_dual_post_sums(apx_ctx* ctx, int32_t dtype, void* y, const void* axnew, int64_t m, const ProxDev& h, Ctrl* c, double* out2) {
  const unsigned gr = vgrid(m);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(k_dual_post<double>, dim3(gr), dim3(kBlock), 0, ctx->stream, (double*)y, (const double*)axnew, m, h, c, ctx->red),
             hipLaunchKernelGGL(k_dual_post<float>, dim3(gr), dim3(kBlock), 0, ctx->stream, (float*)y, (const float*)axnew, m, h, c, ctx->red));
  hipLaunchKernelGGL(k_sum_q, dim3(1), dim3(kBlock), 0, ctx->stream, ctx->red, (int)gr, 2, out2);
  APX_HIP(hipGetLastError());
  return APX_OK;
}

// the end of the iteration from two all-reduced sums (sum dual_res^2, raw h(A_x)) on the device: behind a one-pass launch on a
// row shard of the operator (csrc/slab_pd.hip)
int32_t vo_final_sums(apx_ctx* ctx, const double* sums, const ProxDev& h, Ctrl* c, apx_record* recs, int64_t rec_cap) {
  hipLaunchKernelGGL(k_final_s, dim3(1), dim3(64), 0, ctx->stream, sums, h, c, recs, rec_cap);
  APX_HIP(hipGetLastError());
  return APX_OK;
}

int32_t vo_dual_sharded(apx_ctx* ctx, int32_t dtype, void* y, const void* ax, void* axprev, int64_t m,
                        const ProxDev& h, Ctrl* c, apx_record* recs, int64_t rec_cap,
                        const std::function<int32_t(double*, int64_t)>& reduce) {
  const unsigned gr = vgrid(m);
  if (h.kind == APX_G_L2) {
    DISPATCH_T(dtype,
               hipLaunchKernelGGL(k_dual_pre<double>, dim3(gr), dim3(kBlock), 0, ctx->stream, (const double*)y,
                                  (const double*)ax, (const double*)axprev, m, h, c, ctx->red),
               hipLaunchKernelGGL(k_dual_pre<float>, dim3(gr), dim3(kBlock), 0, ctx->stream, (const float*)y,
                                  (const float*)ax, (const float*)axprev, m, h, c, ctx->red));
    hipLaunchKernelGGL(k_sum_q, dim3(1), dim3(kBlock), 0, ctx->stream, ctx->red, (int)gr, 1, ctx->scal + 12);
    APX_TRY(reduce(ctx->scal + 12, 1));
    hipLaunchKernelGGL(k_dual_pre_final_s, dim3(1), dim3(64), 0, ctx->stream, ctx->scal + 12, c);
  }
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(k_dual<double>, dim3(gr), dim3(kBlock), 0, ctx->stream, (double*)y, (const double*)ax,
                                (double*)axprev, m, h, c, ctx->red),
             hipLaunchKernelGGL(k_dual<float>, dim3(gr), dim3(kBlock), 0, ctx->stream, (float*)y, (const float*)ax,
                                (float*)axprev, m, h, c, ctx->red));
  hipLaunchKernelGGL(k_sum_q, dim3(1), dim3(kBlock), 0, ctx->stream, ctx->red, (int)gr, 2, ctx->scal + 13);
  APX_TRY(reduce(ctx->scal + 13, 2));
  hipLaunchKernelGGL(k_final_s, dim3(1), dim3(64), 0, ctx->stream, ctx->scal + 13, h, c, recs, rec_cap);
  APX_HIP(hipGetLastError());
  return APX_OK;
}

int32_t vo_step(apx_ctx* ctx, int32_t dtype, void* x, void* xprev, const void* grad, void* gprev, void* v,
                const void* aty, int64_t n, const ProxDev& g, const Ctrl* c) {
  const unsigned gr = vgrid(n);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(k_step<double>, dim3(gr), dim3(kBlock), 0, ctx->stream, (double*)x, (double*)xprev,
                                (const double*)grad, (double*)gprev, (double*)v, (const double*)aty, n, g, c),
             hipLaunchKernelGGL(k_step<float>, dim3(gr), dim3(kBlock), 0, ctx->stream, (float*)x, (float*)xprev,
                                (const float*)grad, (float*)gprev, (float*)v, (const float*)aty, n, g, c));
  APX_HIP(hipGetLastError());
  return APX_OK;
}

int32_t vo_step_red(apx_ctx* ctx, int32_t dtype, void* x, void* xprev, const void* grad, void* gprev, void* v, const void* aty,
                    int64_t n, const ProxDev& g, const Ctrl* c, double* red) {
  const unsigned gr = vgrid(n);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(k_step_red<double>, dim3(gr), dim3(kBlock), 0, ctx->stream, (double*)x, (double*)xprev,
                                (const double*)grad, (double*)gprev, (double*)v, (const double*)aty, n, g, c, red),
             hipLaunchKernelGGL(k_step_red<float>, dim3(gr), dim3(kBlock), 0, ctx->stream, (float*)x, (float*)xprev,
                                (const float*)grad, (float*)gprev, (float*)v, (const float*)aty, n, g, c, red));
  APX_HIP(hipGetLastError());
  return APX_OK;
}

// k_rule on block partials that k_step_red left (the grid of both is vgrid(n))
int32_t vo_rule_only(apx_ctx* ctx, const double* red, int64_t n, const ProxDev& g, Ctrl* c, bool has_lin, apx_record* recs, int64_t rec_cap) {
  hipLaunchKernelGGL(k_rule, dim3(1), dim3(kBlock), 0, ctx->stream, red, (int)vgrid(n), g, c, has_lin ? 1 : 0, recs, rec_cap);
  APX_HIP(hipGetLastError());
  return APX_OK;
}

int32_t vo_step_rule(apx_ctx* ctx, int32_t dtype, void* x, void* xprev, const void* grad, void* gprev, void* v, int64_t n,
                     const ProxDev& g, Ctrl* c, const Ctrl* snap, const double* wred, int32_t ldw, apx_record* recs, int64_t rec_cap) {
  const unsigned gr = vgrid(n);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(k_step_rule<double>, dim3(gr), dim3(kBlock), 0, ctx->stream, (double*)x, (double*)xprev,
                                (const double*)grad, (double*)gprev, (double*)v, n, g, c, snap, wred, (int)ldw, recs, rec_cap),
             hipLaunchKernelGGL(k_step_rule<float>, dim3(gr), dim3(kBlock), 0, ctx->stream, (float*)x, (float*)xprev,
                                (const float*)grad, (float*)gprev, (float*)v, n, g, c, snap, wred, (int)ldw, recs, rec_cap));
  APX_HIP(hipGetLastError());
  return APX_OK;
}

int32_t vo_bt_trial(apx_ctx* ctx, int32_t dtype, const void* x, const void* grad, void* z, int64_t n,
                    const ProxDev& g, double gamma, double* out3) {
  const unsigned gr = vgrid(n);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(k_bt_trial<double>, dim3(gr), dim3(kBlock), 0, ctx->stream, (const double*)x,
                                (const double*)grad, (double*)z, n, g, gamma, ctx->red),
             hipLaunchKernelGGL(k_bt_trial<float>, dim3(gr), dim3(kBlock), 0, ctx->stream, (const float*)x,
                                (const float*)grad, (float*)z, n, g, gamma, ctx->red));
  hipLaunchKernelGGL(k_sum_q, dim3(1), dim3(kBlock), 0, ctx->stream, ctx->red, (int)gr, 3, out3);
  APX_HIP(hipGetLastError());
  return APX_OK;
}

int32_t vo_prox(apx_ctx* ctx, int32_t dtype, const void* v, void* y, int64_t n, const ProxDev& g, double gamma,
                bool conj, double* out_val_sum) {
  const unsigned gr = vgrid(n);
  double* aux = nullptr;
  if (g.kind == APX_G_L2) {
    // ||z||, z = v (+shift) for the plain prox, z = v/gamma (+shift) for the conjugate
    const double scale = conj ? 1.0 / gamma : 1.0;
    DISPATCH_T(dtype,
               hipLaunchKernelGGL(k_sumsq_shift<double>, dim3(gr), dim3(kBlock), 0, ctx->stream, (const double*)v, n,
                                  g, scale, ctx->red),
               hipLaunchKernelGGL(k_sumsq_shift<float>, dim3(gr), dim3(kBlock), 0, ctx->stream, (const float*)v, n,
                                  g, scale, ctx->red));
    hipLaunchKernelGGL(k_sum_q, dim3(1), dim3(kBlock), 0, ctx->stream, ctx->red, (int)gr, 1, ctx->scal + 8);
    aux = ctx->scal + 8;
  }
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(k_prox<double>, dim3(gr), dim3(kBlock), 0, ctx->stream, (const double*)v, (double*)y,
                                n, g, gamma, conj ? 1 : 0, aux, ctx->red),
             hipLaunchKernelGGL(k_prox<float>, dim3(gr), dim3(kBlock), 0, ctx->stream, (const float*)v, (float*)y, n,
                                g, gamma, conj ? 1 : 0, aux, ctx->red));
  hipLaunchKernelGGL(k_sum_q, dim3(1), dim3(kBlock), 0, ctx->stream, ctx->red, (int)gr, 1, out_val_sum);
  APX_HIP(hipGetLastError());
  return APX_OK;
}

// K7/K8 sums for a HOST-driven loop: out5 (device) = raw { pr^2, dg^2, dg.dx, dx^2, g(x) }; gamma from Ctrl
int32_t vo_reduce_sums(apx_ctx* ctx, int32_t dtype, const void* x, const void* xprev, const void* grad,
                       const void* gprev, const void* v, const void* aty, int64_t n, const ProxDev& g, const Ctrl* c,
                       double* out5) {
  const unsigned gr = vgrid(n);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(k_reduce_n<double>, dim3(gr), dim3(kBlock), 0, ctx->stream, (const double*)x,
                                (const double*)xprev, (const double*)grad, (const double*)gprev, (const double*)v,
                                (const double*)aty, n, g, c, ctx->red),
             hipLaunchKernelGGL(k_reduce_n<float>, dim3(gr), dim3(kBlock), 0, ctx->stream, (const float*)x,
                                (const float*)xprev, (const float*)grad, (const float*)gprev, (const float*)v,
                                (const float*)aty, n, g, c, ctx->red));
  hipLaunchKernelGGL(k_sum_q, dim3(1), dim3(kBlock), 0, ctx->stream, ctx->red, (int)gr, 5, out5);
  APX_HIP(hipGetLastError());
  return APX_OK;
}

// AdaPDM+ trial: out3 (device) = { ||y_next - y||^2, ||dual_res||^2, raw h(A_x) }
int32_t vo_dual_trial(apx_ctx* ctx, int32_t dtype, const void* y, const void* ax, const void* axprev, void* ynext,
                      int64_t m, const ProxDev& h, double sigma, double rho, double* out3) {
  const unsigned gr = vgrid(m);
  double* aux = nullptr;
  if (h.kind == APX_G_L2) {
    DISPATCH_T(dtype,
               hipLaunchKernelGGL(k_dual_pre_v<double>, dim3(gr), dim3(kBlock), 0, ctx->stream, (const double*)y,
                                  (const double*)ax, (const double*)axprev, m, h, sigma, rho, ctx->red),
               hipLaunchKernelGGL(k_dual_pre_v<float>, dim3(gr), dim3(kBlock), 0, ctx->stream, (const float*)y,
                                  (const float*)ax, (const float*)axprev, m, h, sigma, rho, ctx->red));
    hipLaunchKernelGGL(k_sum_q, dim3(1), dim3(kBlock), 0, ctx->stream, ctx->red, (int)gr, 1, ctx->scal + 12);
    aux = ctx->scal + 12;
  }
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(k_dual_trial<double>, dim3(gr), dim3(kBlock), 0, ctx->stream, (const double*)y,
                                (const double*)ax, (const double*)axprev, (double*)ynext, m, h, sigma, rho, aux,
                                ctx->red),
             hipLaunchKernelGGL(k_dual_trial<float>, dim3(gr), dim3(kBlock), 0, ctx->stream, (const float*)y,
                                (const float*)ax, (const float*)axprev, (float*)ynext, m, h, sigma, rho, aux,
                                ctx->red));
  hipLaunchKernelGGL(k_sum_q, dim3(1), dim3(kBlock), 0, ctx->stream, ctx->red, (int)gr, 3, out3);
  APX_HIP(hipGetLastError());
  return APX_OK;
}

int32_t vo_diff_sumsq(apx_ctx* ctx, int32_t dtype, const void* a, const void* b, int64_t n, double* out1) {
  const unsigned gr = vgrid(n);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(k_diff_sumsq<double>, dim3(gr), dim3(kBlock), 0, ctx->stream, (const double*)a,
                                (const double*)b, n, ctx->red),
             hipLaunchKernelGGL(k_diff_sumsq<float>, dim3(gr), dim3(kBlock), 0, ctx->stream, (const float*)a,
                                (const float*)b, n, ctx->red));
  hipLaunchKernelGGL(k_sum_q, dim3(1), dim3(kBlock), 0, ctx->stream, ctx->red, (int)gr, 1, out1);
  APX_HIP(hipGetLastError());
  return APX_OK;
}

// out = a + b   (`temp + f.q`, the Quadratic pullback of experiments/dual_svm/runme.jl:26)
template <typename T>
__global__ __launch_bounds__(kBlock) void k_add(T* __restrict__ out, const T* __restrict__ a, const T* __restrict__ b,
                                                int64_t n, const int* __restrict__ done) {
  if (done != nullptr && *done != 0) return;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock)
    out[i] = (T)((double)a[i] + (double)b[i]);
}

int32_t vo_add(apx_ctx* ctx, int32_t dtype, void* out, const void* a, const void* b, int64_t n, const int* done) {
  const unsigned gr = vgrid(n);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(k_add<double>, dim3(gr), dim3(kBlock), 0, ctx->stream, (double*)out, (const double*)a,
                                (const double*)b, n, done),
             hipLaunchKernelGGL(k_add<float>, dim3(gr), dim3(kBlock), 0, ctx->stream, (float*)out, (const float*)a,
                                (const float*)b, n, done));
  APX_HIP(hipGetLastError());
  return APX_OK;
}

int32_t vo_extrap(apx_ctx* ctx, int32_t dtype, void* out, const void* a, const void* b, double c, int64_t n) {
  const unsigned gr = vgrid(n);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(k_extrap<double>, dim3(gr), dim3(kBlock), 0, ctx->stream, (double*)out,
                                (const double*)a, (const double*)b, c, n),
             hipLaunchKernelGGL(k_extrap<float>, dim3(gr), dim3(kBlock), 0, ctx->stream, (float*)out, (const float*)a,
                                (const float*)b, c, n));
  APX_HIP(hipGetLastError());
  return APX_OK;
}

int32_t vo_xbar(apx_ctx* ctx, int32_t dtype, void* xbar, const void* x, double phi, int64_t n) {
  const unsigned gr = vgrid(n);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(k_xbar<double>, dim3(gr), dim3(kBlock), 0, ctx->stream, (double*)xbar,
                                (const double*)x, phi, n),
             hipLaunchKernelGGL(k_xbar<float>, dim3(gr), dim3(kBlock), 0, ctx->stream, (float*)xbar, (const float*)x,
                                phi, n));
  APX_HIP(hipGetLastError());
  return APX_OK;
}

int32_t vo_mp_trial(apx_ctx* ctx, int32_t dtype, const void* xprev, const void* aty, const void* atyprev,
                    const void* gradprev, void* v, void* x, int64_t n, const ProxDev& g, double theta, double gamma,
                    double* out3) {
  const unsigned gr = vgrid(n);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(k_mp_trial<double>, dim3(gr), dim3(kBlock), 0, ctx->stream, (const double*)xprev,
                                (const double*)aty, (const double*)atyprev, (const double*)gradprev, (double*)v,
                                (double*)x, n, g, theta, gamma, ctx->red),
             hipLaunchKernelGGL(k_mp_trial<float>, dim3(gr), dim3(kBlock), 0, ctx->stream, (const float*)xprev,
                                (const float*)aty, (const float*)atyprev, (const float*)gradprev, (float*)v,
                                (float*)x, n, g, theta, gamma, ctx->red));
  hipLaunchKernelGGL(k_sum_q, dim3(1), dim3(kBlock), 0, ctx->stream, ctx->red, (int)gr, 3, out3);
  APX_HIP(hipGetLastError());
  return APX_OK;
}

int32_t vo_mp_dualres(apx_ctx* ctx, int32_t dtype, const void* yold, const void* axold, const void* ynew,
                      const void* axnew, int64_t m, double sigma_prev, const ProxDev& h, double* out2) {
  const unsigned gr = vgrid(m);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(k_mp_dualres<double>, dim3(gr), dim3(kBlock), 0, ctx->stream, (const double*)yold,
                                (const double*)axold, (const double*)ynew, (const double*)axnew, m, sigma_prev, h,
                                ctx->red),
             hipLaunchKernelGGL(k_mp_dualres<float>, dim3(gr), dim3(kBlock), 0, ctx->stream, (const float*)yold,
                                (const float*)axold, (const float*)ynew, (const float*)axnew, m, sigma_prev, h,
                                ctx->red));
  hipLaunchKernelGGL(k_sum_q, dim3(1), dim3(kBlock), 0, ctx->stream, ctx->red, (int)gr, 2, out2);
  APX_HIP(hipGetLastError());
  return APX_OK;
}

int32_t vo_primal_res(apx_ctx* ctx, int32_t dtype, const void* v, const void* x, const void* grad, const void* aty,
                      int64_t n, double gamma, const ProxDev& g, double* out2) {
  const unsigned gr = vgrid(n);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(k_primal_res<double>, dim3(gr), dim3(kBlock), 0, ctx->stream, (const double*)v,
                                (const double*)x, (const double*)grad, (const double*)aty, n, gamma, g, ctx->red),
             hipLaunchKernelGGL(k_primal_res<float>, dim3(gr), dim3(kBlock), 0, ctx->stream, (const float*)v,
                                (const float*)x, (const float*)grad, (const float*)aty, n, gamma, g, ctx->red));
  hipLaunchKernelGGL(k_sum_q, dim3(1), dim3(kBlock), 0, ctx->stream, ctx->red, (int)gr, 2, out2);
  APX_HIP(hipGetLastError());
  return APX_OK;
}

int32_t vo_gval(apx_ctx* ctx, int32_t dtype, const void* x, int64_t n, const ProxDev& g, double* out_val_sum) {
  const unsigned gr = vgrid(n);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(k_gval<double>, dim3(gr), dim3(kBlock), 0, ctx->stream, (const double*)x, n, g,
                                ctx->red),
             hipLaunchKernelGGL(k_gval<float>, dim3(gr), dim3(kBlock), 0, ctx->stream, (const float*)x, n, g,
                                ctx->red));
  hipLaunchKernelGGL(k_sum_q, dim3(1), dim3(kBlock), 0, ctx->stream, ctx->red, (int)gr, 1, out_val_sum);
  APX_HIP(hipGetLastError());
  return APX_OK;
}

double vo_gval_final_host(const ProxDev& g, double s) {
  switch (g.kind) {
    case APX_G_L1: return g.lambda * s;
    case APX_G_L2: return g.lambda * std::sqrt(s);
    case APX_G_BOX:
    case APX_G_BOXV:
    case APX_G_INDZERO: return s > 0.0 ? INFINITY : 0.0;
    default: return 0.0;
  }
}

}  // namespace apx
