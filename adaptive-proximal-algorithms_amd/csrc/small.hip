// small.hip — the proximal-gradient iteration of a SMALL dense problem in TWO kernels.
//
// The reference's own experiments are small (lasso 100 x 300, 500 x 1000, 4000 x 1000 —
// experiments/lasso/runme.jl:191-195): on the GPU an iteration of src/AdaProx.jl:334-362 is then
// bound by kernel boundaries and memory round trips, not bytes (the generic path needs 8 launches,
// ~4 us each, for 4 MB of matrix).  Here every workgroup REDUNDANTLY does the O(m) / O(n) work of the
// iteration — the residual and step-size reductions, the rule, the prox — so the only grid-wide
// dependencies left are the two products:
//
//   k_small_b : r = A x - b (or the logistic link) ; f(x) ; A' r ; extra blocks rotate the
//               x / grad / v buffers
//   k_small_a : primal residual, ||dg||^2, <dg,dx>, ||dx||^2, g(x) (:338, :260-261) ; rule update
//               (:340) ; termination test (:354) ; record ; v = x - gamma*grad ; x+ = prox(v)
//               (:359-361) ; A x+
//
// Both products are complete dot products inside one block (lanes along the reduction index: the
// transposed copy of A for A x, A itself for A' r; the vector operand in LDS), so there are no split
// partials and no second stage.  Every block computes the same sums in the same order, so all blocks
// take identical decisions without any cross-block communication; block 0 alone writes the control
// block.  No atomics: results are bit-reproducible run to run.
//
// State that every block reads in full (x, xprev, gprev, v, Ctrl) is never written by the kernel
// that reads it: k_small_a writes staging buffers (xn, vn, gn, Ctrl B), k_small_b rotates them in
// (each element owned by exactly one thread) and writes Ctrl A.
//
// Measured (tools/small_bench.py, MI355X): 500 x 1000 fp64 lasso 12.5 us / iteration against 31 us
// on the generic kernels; 100 x 300: 11.4 vs 30.6 us; logistic 2000 x 120 fp32: 14.8 vs 28 us.
#include <algorithm>
#include <cstdlib>
#include <memory>

#include "small_dev.h"
#include "solver_internal.h"
#include "vecops_dev.h"

namespace apx {

// Both kernels are chains of dependent memory round trips (~1 us each: every operand was written by
// another XCD in the previous kernel), so each is organised as ONE batch of loads (clamped indices, no
// branch between them), arithmetic, then the dot products; stores never wait on further loads.

// k_small_b: blocks [0, dot_blocks) finish A x and form A' r ; the blocks behind them rotate the staged
// iterate in (x_prev = x ; x = x+ ; grad_prev = grad ; v = v+, :359-361), one element per thread
template <typename T, int FK, int BS>
__global__ __launch_bounds__(BS) void k_small_b(SmallDev d, int dot_blocks) {
  extern __shared__ double rs[];              // the residual, m doubles (dynamic: up to 128 KB)
  const bool live = d.cb->done == 0;          // Ctrl B is not written by this kernel: block-uniform
  if ((int)blockIdx.x >= dot_blocks) {
    const int i = ((int)blockIdx.x - dot_blocks) * BS + threadIdx.x;
    if (live && i < d.n) {
      T* __restrict__ X = (T*)d.X;
      const T xo = X[i], xn = ((const T*)d.XN)[i], gn = ((const T*)d.GN)[i], vn = ((const T*)d.VN)[i];
      ((T*)d.XP)[i] = xo;
      X[i] = xn;
      ((T*)d.GP)[i] = gn;
      ((T*)d.V)[i] = vn;
    }
    return;
  }
  const T* __restrict__ b = (const T*)d.b;
  __shared__ Ctrl lt;
  const bool writer = blockIdx.x == 0 && threadIdx.x == 0;
  double bias = 0.0;
  if constexpr (FK == APX_F_LOGISTIC) bias = (double)((const T*)d.XN)[d.n - 1];
  double q[2] = {0.0, 0.0};
  const int per = (d.m + BS - 1) / BS;
  constexpr int MB = 8;             // chunks per batch (16 loads in flight per lane)
  for (int k0 = 0; k0 < per; k0 += MB) {
    double s4[MB], b4[MB];
#pragma unroll
    for (int u = 0; u < MB; ++u) {
      const int ic = min((int)threadIdx.x + (k0 + u) * BS, d.m - 1);
      s4[u] = d.axp[ic];
      b4[u] = (double)b[ic];
    }
    // the control block goes straight to LDS, BEHIND the first batch of loads: a private copy would live in
    // scratch, and its wave would wait a whole round trip for it before issuing anything else
    if (k0 == 0 && writer) lt = *d.cb;
#pragma unroll
    for (int u = 0; u < MB; ++u) {
      const int i = threadIdx.x + (k0 + u) * BS;
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
  block_sums_t0<2, BS>(q);             // (barrier inside: rs is complete behind it)
  if (writer && live) {
    f_finalize(&lt, FK, q);
    *d.ca = lt;
  }
  // (only the first dot_blocks blocks of this grid form dot products: the rotation blocks behind them have left)
  small_dots_w<T, BS>(d.w_t, (const T*)d.A, d.ld, d.ncols, d.m, rs, d.grp, live, dot_blocks);       // A' r: columns of A
}

// MODE 0: only A x for the staged x (start of the pre-loop, :327-328)
// MODE 1: pre-loop step  v = x0 - gamma*grad f(x0) ; x1 = prox(v)  (:330-332) — no residual, no rule
// MODE 2: the loop body
template <typename T, int FK, int MODE>
__global__ __launch_bounds__(kBlock) void k_small_a(SmallDev d) {
  __shared__ double xs[kSmallMax];
  if constexpr (MODE == 0) {
    const T* __restrict__ XN = (const T*)d.XN;
    for (int i = threadIdx.x; i < d.n; i += kBlock) xs[i] = (double)XN[i];
    __syncthreads();
    small_dots_w<T>(d.w_n, (const T*)d.AT, d.ldT, d.m, d.ncols, xs, d.axp, true);    // A x: rows of A = columns of A'
    return;
  } else {
    __shared__ Ctrl lc;
    // thread 0 alone reads the done flag (block 0 of THIS launch may set it while a later block starts:
    // that block then leaves, which is what it would decide for itself anyway) and the control block
    int done_in = 0;
    const T* __restrict__ X = (const T*)d.X;
    [[maybe_unused]] const T* __restrict__ XP = (const T*)d.XP;
    [[maybe_unused]] const T* __restrict__ GP = (const T*)d.GP;
    [[maybe_unused]] const T* __restrict__ V = (const T*)d.V;
    const double gamma = d.ca->gamma;
    double N = 1.0, bias_grad = 0.0;
    if constexpr (FK == APX_F_LOGISTIC) {
      N = d.ca->n_mean;
      bias_grad = d.ca->bias_grad;
    }
    double gl[kSmallPer], gb[kSmallPer];     // gradient and the shift of a translated g, per owned element
    double q[5] = {0, 0, 0, 0, 0};
    const int per = (d.n + kBlock - 1) / kBlock;
#pragma unroll
    for (int k0 = 0; k0 < kSmallPer; k0 += 4) {
#pragma unroll
      for (int u = 0; u < 4; ++u) gl[k0 + u] = gb[k0 + u] = 0.0;
      if (k0 < per) {
        double g4[4], x4[4], v4[4], gp4[4], xp4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int ic = min((int)threadIdx.x + (k0 + u) * kBlock, d.n - 1);
          g4[u] = d.grp[ic];
          x4[u] = (double)X[ic];
          gb[k0 + u] = shift_at<T>(d.g, ic);
          if constexpr (MODE == 2) {
            v4[u] = (double)V[ic];
            gp4[u] = (double)GP[ic];
            xp4[u] = (double)XP[ic];
          }
        }
        if (k0 == 0 && threadIdx.x == 0) {     // behind the first batch of loads, straight to LDS (see k_small_b)
          done_in = d.cb->done;
          if constexpr (MODE == 2) lc = *d.ca;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int i = threadIdx.x + (k0 + u) * kBlock;
          if (i < d.n) {
            double gi;
            if (FK == APX_F_LOGISTIC && i == d.n - 1) {
              gi = (double)(T)bias_grad;                             // grad[end] = mean(probs - y)
            } else {
              const T gs = (T)g4[u];                                 // A' * res
              gi = FK == APX_F_LOGISTIC ? (double)(T)((double)gs / N) : (double)gs;
            }
            const double xi = x4[u];
            if constexpr (MODE == 2) {
              const double pr = (v4[u] - xi) / gamma + gi;           // :338 (A'y == 0)
              const double dg = gi - gp4[u];
              const double dx = xi - xp4[u];
              q[0] = fma(pr, pr, q[0]);
              q[1] = fma(dg, dg, q[1]);
              q[2] = fma(dg, dx, q[2]);
              q[3] = fma(dx, dx, q[3]);
              q[4] += gval_elem_s<T>(d.g, xi, i, gb[k0 + u]);
            }
            gl[k0 + u] = gi;
            xs[i] = xi;
          }
        }
      }
    }
    if constexpr (MODE == 2) block_sums_t0<5>(q);
    if (threadIdx.x == 0) {
      if (done_in != 0) {
        lc.done = 1;
      } else if constexpr (MODE == 1) {
        lc.done = 0;
        lc.gamma = gamma;
      } else {
        lc.it += 1;
        const double npr = sqrt(q[0]);
        lc.npr2 = npr * npr;
        lc.g_x = gval_final(d.g, q[4]);
        rule_update(&lc, q[1], q[2], q[3]);                          // :340-342
        lc.ndr2 = 0.0;                                               // h = Zero, A = 0: dual_res == 0
        lc.h_Ax = 0.0;
        finish_iteration(&lc, blockIdx.x == 0 ? d.recs : nullptr, d.rec_cap);   // :348-354
        if (blockIdx.x == 0) *d.cb = lc;
      }
    }
    __syncthreads();
    if (lc.done != 0) return;      // converged: x is the tested iterate, nothing moves (:355)
    const double gnew = lc.gamma;
    T* __restrict__ XNw = (T*)d.XN;
    T* __restrict__ VNw = (T*)d.VN;
    T* __restrict__ GNw = (T*)d.GN;
#pragma unroll
    for (int k = 0; k < kSmallPer; ++k) {
      const int i = threadIdx.x + k * kBlock;
      if (i < d.n) {
        const T vi = (T)(xs[i] - gnew * gl[k]);                      // :359
        const T xn = (T)prox_elem_s<T>(d.g, (double)vi, gnew, i, 0.0, gb[k]);   // :361
        xs[i] = (double)xn;
        if ((unsigned)k % gridDim.x == blockIdx.x) {     // chunk k of the staging buffers is written by one block
          XNw[i] = xn;
          VNw[i] = vi;
          GNw[i] = (T)gl[k];
        }
      }
    }
    __syncthreads();
    small_dots_w<T>(d.w_n, (const T*)d.AT, d.ldT, d.m, d.ncols, xs, d.axp, true);
  }
}

// ---------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------
bool small_eligible(const apx_problem* p) {
  if (!p->ctx->fused_small) return false;
  if (p->A_lin != nullptr || p->sharded()) return false;
  if (p->f_kind != APX_F_LEAST_SQUARES && p->f_kind != APX_F_LOGISTIC) return false;
  if (p->A_f == nullptr || no_column_storage(p->A_f)) return false;
  if (p->ctx->skip_zero_cols) return false;
  return small_shape_ok(p->A_f->m, p->n) && p->A_f->n > 0;
}

template <typename T, int BS>
static void launch_b_inst(const SmallDev& d, int fk, hipStream_t st) {
  const int dot_blocks = (int)pick_grid(d.ncols, d.w_t, BS);
  const dim3 grid((unsigned)(dot_blocks + (d.n + BS - 1) / BS));   // + the rotation blocks
  const size_t lds = m_vector_lds_bytes(d.m);
  if (fk == APX_F_LEAST_SQUARES) {
    allow_big_lds<k_small_b<T, APX_F_LEAST_SQUARES, BS>>();
    hipLaunchKernelGGL((k_small_b<T, APX_F_LEAST_SQUARES, BS>), grid, dim3(BS), lds, st, d, dot_blocks);
  } else {
    allow_big_lds<k_small_b<T, APX_F_LOGISTIC, BS>>();
    hipLaunchKernelGGL((k_small_b<T, APX_F_LOGISTIC, BS>), grid, dim3(BS), lds, st, d, dot_blocks);
  }
}

template <typename T>
static void launch_b(const SmallDev& d, int fk, hipStream_t st) {
  if (d.m > kSmallMax) launch_b_inst<T, kTallBlock>(d, fk, st);
  else launch_b_inst<T, kBlock>(d, fk, st);
}

template <typename T, int MODE>
static void launch_a(const SmallDev& d, int fk, hipStream_t st) {
  const dim3 grid(pick_grid(d.m, d.w_n));
  if (fk == APX_F_LEAST_SQUARES)
    hipLaunchKernelGGL((k_small_a<T, APX_F_LEAST_SQUARES, MODE>), grid, dim3(kBlock), 0, st, d);
  else
    hipLaunchKernelGGL((k_small_a<T, APX_F_LOGISTIC, MODE>), grid, dim3(kBlock), 0, st, d);
}

// called by apx_solver_create INSTEAD of the generic pre-loop (:324-332), with x = x0 uploaded and Ctrl
// initialised: the gradient at x0 goes through the same kernels as every later one, so an iterate that
// does not move reproduces its gradient bit for bit (dg == 0 exactly, as in the reference)
int32_t small_setup(apx_solver* s) {
  apx_problem* p = s->p;
  apx_ctx* ctx = p->ctx;
  const apx_matrix* A = p->A_f;
  std::unique_ptr<SmallState> st(new SmallState());
  const size_t es = p->es();
  const size_t nb = (size_t)p->n_pad * es;
  APX_TRY(st->xn.alloc_zero(nb, ctx->stream));
  APX_TRY(st->vn.alloc_zero(nb, ctx->stream));
  APX_TRY(st->gn.alloc_zero(nb, ctx->stream));
  APX_TRY(st->cb.alloc_zero(sizeof(Ctrl), ctx->stream));
  SmallDev d{};
  d.A = A->d;
  d.ld = A->ld;
  if (A->dT != nullptr) {
    d.AT = A->dT;
    d.ldT = A->ldT;
  } else {               // private transposed copy (a few MB at most): the caller's matrix object is untouched
    d.ldT = round_up(A->n, kPad);
    APX_TRY(st->at.alloc_zero((size_t)d.ldT * (size_t)A->ld * es, ctx->stream));
    APX_TRY(transpose_into(ctx, A, st->at.p, d.ldT));
    d.AT = st->at.p;
  }
  d.m = (int32_t)A->m;
  d.ncols = (int32_t)A->n;
  d.n = (int32_t)p->n;
  d.w_n = pick_waves(d.ncols);
  d.w_t = pick_waves(d.m, m_side_block(d.m));
  APX_TRY(st->axp.alloc_zero((size_t)A->ld * sizeof(double), ctx->stream));
  APX_TRY(st->grp.alloc_zero((size_t)p->n_pad * sizeof(double), ctx->stream));
  d.b = p->b.p;
  d.X = s->x.p;
  d.XP = s->xprev.p;
  d.GP = s->gprev.p;
  d.V = s->v.p;
  d.XN = st->xn.p;
  d.VN = st->vn.p;
  d.GN = st->gn.p;
  d.axp = (double*)st->axp.p;
  d.grp = (double*)st->grp.p;
  d.ca = p->ctrl();
  d.cb = (Ctrl*)st->cb.p;
  d.g = p->gd;
  d.recs = (apx_record*)s->recs.p;
  d.rec_cap = s->rec_cap;
  // pre-loop: A x0 -> f(x0), A' r -> grad f(x0) -> v, x1 staged (rotated in by the first loop k_small_b)
  APX_HIP(hipMemcpyAsync(st->xn.p, s->x.p, nb, hipMemcpyDeviceToDevice, ctx->stream));
  APX_HIP(hipMemcpyAsync(st->cb.p, p->ctrl_buf.p, sizeof(Ctrl), hipMemcpyDeviceToDevice, ctx->stream));
  if (p->dtype == APX_F32) {
    launch_a<float, 0>(d, p->f_kind, ctx->stream);
    launch_b<float>(d, p->f_kind, ctx->stream);
    launch_a<float, 1>(d, p->f_kind, ctx->stream);
  } else {
    launch_a<double, 0>(d, p->f_kind, ctx->stream);
    launch_b<double>(d, p->f_kind, ctx->stream);
    launch_a<double, 1>(d, p->f_kind, ctx->stream);
  }
  APX_HIP(hipGetLastError());
  st->dev = d;
  s->small = st.release();
  return APX_OK;
}

int32_t small_enqueue(apx_solver* s) {
  apx_problem* p = s->p;
  const SmallState* st = s->small;
  const SmallDev& d = st->dev;
  hipStream_t q = p->ctx->stream;
  if (p->dtype == APX_F32) {
    launch_b<float>(d, p->f_kind, q);
    launch_a<float, 2>(d, p->f_kind, q);
  } else {
    launch_b<double>(d, p->f_kind, q);
    launch_a<double, 2>(d, p->f_kind, q);
  }
  APX_HIP(hipGetLastError());
  return APX_OK;
}

}  // namespace apx
