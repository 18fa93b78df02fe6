// small_pd.hip — the primal-dual iteration (src/AdaProx.jl:334-362 with a linear operator) of a SMALL
// dense problem in TWO kernels, for the sizes of the reference's own LAD / square-root-lasso experiments
// (experiments/least_absolute_deviation/runme.jl:39-62: f = Zero, g = NormL1, h = Translate(NormL1 |
// NormL2, -b), A dense with a few hundred rows) and of its dual SVM (experiments/dual_svm/runme.jl:47-76:
// f = Quadratic(Q, q), g = IndBox, h = IndZero, A = y' — the third product Q x+ rides in kernel A, its
// output rotating through three buffers like A x).  Same idea as small.hip: every workgroup redundantly does
// the O(m + n) work, the two products are in-block dot products, nothing else crosses blocks.
//
//   k_small_pd_b (iteration k): primal_res = (v - x)/gamma + A'y (:338) ; ||dx||^2 ... (:260-261, grad f = 0)
//        ; rule (:340-342) ; w = y + sigma((1+rho) A_x - rho A_x_prev) ; y+ = prox_{sigma h*}(w) (:344-345,
//        the l2 ball needs ||w/sigma + shift|| first) ; dual_res (:347) ; norm_res, test, record (:348-354)
//        ; A'y+ (:358)
//   k_small_pd_a (iteration k): v+ = x - gamma A'y+ ; x+ = prox_{gamma g}(v+) (:359-361) ; A x+ (:335 of k+1)
//
// No kernel writes a buffer that it (or the other kernel of the same iteration) still reads: x and A_x
// live in three buffers (prev / cur / next), v, y, A'y and the control block in two, and the HOST rotates
// the roles per iteration (index mod 3 / mod 2) — no copies, no races, no atomics.  A converged control
// block is forwarded unchanged by every later k_small_pd_b, so iterations enqueued behind the termination
// do nothing and the host finds the tested iterate from Ctrl.it.
#include <cstdlib>
#include <memory>

#include "small_dev.h"

namespace apx {


// MODE 0: only A' y_cur -> aty_next (pre-loop, :329) ; MODE 2: the loop body
// MREG: m <= 4096, the m-side operands stay in registers between the passes (their loads are in flight during
// the rule); otherwise they are re-read per pass (L2 hits) and only y+ lives on chip
template <typename T, int MODE, int FK, bool MREG, int BS>
__global__ __launch_bounds__(BS) void k_small_pd_b(PdDev d) {
  extern __shared__ double ys[];              // y (MODE 0) / y+ : m doubles (dynamic: up to 128 KB)
  if constexpr (MODE == 0) {
    const T* __restrict__ Y = (const T*)d.y_cur;
    for (int i = threadIdx.x; i < d.m; i += BS) ys[i] = (double)Y[i];
    __syncthreads();
    small_dots_w<T, BS>(d.w_t, (const T*)d.A, d.ld, d.n, d.m, ys, d.aty_next, true);
    return;
  } else {
    __shared__ Ctrl lc;
    const bool writer = blockIdx.x == 0 && threadIdx.x == 0;
    const int done_in = d.c_cur->done;          // c_cur is not written by this kernel: block-uniform
    const double gamma = d.c_cur->gamma;
    const T* __restrict__ X = (const T*)d.x_cur;
    const T* __restrict__ XP = (const T*)d.x_prev;
    const T* __restrict__ V = (const T*)d.v_cur;
    const T* __restrict__ Y = (const T*)d.y_cur;
    // ---- n side: primal residual and the rule's sums (f = Zero: grad f == 0, dg == 0 ; Quadratic:
    //      grad = Q x + q from kernel A's third product, f = 0.5 <x, Q x> + <x, q>) -----------------
    double q[5] = {0, 0, 0, 0, 0};
    double fq[2] = {0.0, 0.0};
    [[maybe_unused]] const T* __restrict__ QV = (const T*)d.qvec;
    const int per_n = (d.n + BS - 1) / BS;
    for (int k0 = 0; k0 < per_n; k0 += 4) {
      double x4[4], xp4[4], v4[4], a4[4], gb4[4];
      [[maybe_unused]] double t4[4], tp4[4], qv4[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int ic = min((int)threadIdx.x + (k0 + u) * BS, d.n - 1);
        x4[u] = (double)X[ic];
        xp4[u] = (double)XP[ic];
        v4[u] = (double)V[ic];
        a4[u] = (double)(T)d.aty_cur[ic];
        gb4[u] = shift_at<T>(d.g, ic);
        if constexpr (FK == APX_F_QUADRATIC) {
          t4[u] = (double)(T)d.qx_cur[ic];
          tp4[u] = (double)(T)d.qx_prev[ic];
          qv4[u] = (double)QV[ic];
        }
      }
      // the control block goes straight to LDS, BEHIND the first batch of loads: a private copy would live in
      // scratch, and its wave would wait a whole round trip for it before issuing anything else
      if (k0 == 0 && threadIdx.x == 0) lc = *d.c_cur;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = threadIdx.x + (k0 + u) * BS;
        if (i < d.n) {
          double gi = 0.0, gp = 0.0;
          if constexpr (FK == APX_F_QUADRATIC) {         // quadratic_pullback() = temp + q  (dual_svm/runme.jl:24-27)
            gi = (double)(T)(t4[u] + qv4[u]);
            gp = (double)(T)(tp4[u] + qv4[u]);
            fq[0] = fma(x4[u], t4[u], fq[0]);
            fq[1] = fma(x4[u], qv4[u], fq[1]);
          }
          double pr = (v4[u] - x4[u]) / gamma + gi;      // (v - x)/gamma + grad_x   (:338)
          pr = pr + a4[u];                               //                + At_y
          const double dg = gi - gp;
          const double dx = x4[u] - xp4[u];
          q[0] = fma(pr, pr, q[0]);
          q[1] = fma(dg, dg, q[1]);
          q[2] = fma(dg, dx, q[2]);
          q[3] = fma(dx, dx, q[3]);
          q[4] += gval_elem_s<T>(d.g, x4[u], i, gb4[u]);
        }
      }
    }
    // ---- m side operands, loaded now so that they are in flight during the rule ----------------
    static_assert(!MREG || BS == kBlock, "the register-resident m side is written for 256-thread blocks");
    constexpr int MP = MREG ? kSmallPer : 1;
    [[maybe_unused]] double yl[MP], al[MP], apl[MP], hb[MP];
    const int per_m = (d.m + BS - 1) / BS;
    if constexpr (MREG) {
#pragma unroll
      for (int k = 0; k < kSmallPer; ++k) {
        yl[k] = al[k] = apl[k] = hb[k] = 0.0;
        if (k < per_m) {
          const int ic = min((int)threadIdx.x + k * BS, d.m - 1);
          yl[k] = (double)Y[ic];
          al[k] = (double)(T)d.ax_cur[ic];
          apl[k] = (double)(T)d.ax_prev[ic];
          hb[k] = shift_at<T>(d.h, ic);                    // Translate(h, -b): the data vector
        }
      }
    }
    block_sums_t0<5, BS>(q);
    if constexpr (FK == APX_F_QUADRATIC) {
      __syncthreads();
      block_sums_t0<2, BS>(fq);
    }
    if (threadIdx.x == 0) {
      if (done_in != 0) {
        lc.done = 2;                                     // already converged: forward the control block
      } else {
        if constexpr (FK == APX_F_QUADRATIC) lc.f_x = 0.5 * fq[0] + fq[1];
        lc.it += 1;
        const double npr = sqrt(q[0]);
        lc.npr2 = npr * npr;
        lc.g_x = gval_final(d.g, q[4]);
        rule_update(&lc, q[1], q[2], q[3]);              // :340-342
      }
    }
    __syncthreads();
    if (lc.done == 2) {
      if (writer) {
        *d.c_next = lc;
        d.c_next->done = done_in;                        // (lc.done is the marker 2 here)
      }
      return;
    }
    const double sigma = lc.sigma, rho = lc.rho;
    T* __restrict__ YN = (T*)d.y_next;
    double r2[2] = {0.0, 0.0};
    // one element of the dual update: w (:344), y+ = prox_{sigma h*}(w) (:345), dual residual (:347)
    auto dual_elem = [&](int i, int k, double yv, double a, double ap, double b, double aux) {
      const double w = yv + sigma * ((1.0 + rho) * a - rho * ap);
      const T yn = (T)prox_conj_elem_s<T>(d.h, w, sigma, i, aux, b);
      const double dr = (w - (double)yn) / sigma - a;
      r2[0] = fma(dr, dr, r2[0]);
      r2[1] += gval_elem_s<T>(d.h, a, i, b);
      ys[i] = (double)yn;
      if ((unsigned)k % gridDim.x == blockIdx.x) YN[i] = yn;       // chunk k is written by one block
    };
    if constexpr (MREG) {
      if (d.h.kind == APX_G_L2) {                // the l2 ball needs ||w/sigma + shift|| first
        double z2[1] = {0.0};
#pragma unroll
        for (int k = 0; k < kSmallPer; ++k) {
          const int i = threadIdx.x + k * BS;
          if (i < d.m) {
            const double w = yl[k] + sigma * ((1.0 + rho) * al[k] - rho * apl[k]);
            const double z = w / sigma + hb[k];
            z2[0] = fma(z, z, z2[0]);
          }
        }
        block_sums_t0<1, BS>(z2);
        if (threadIdx.x == 0) lc.h_pre = sqrt(z2[0]);
        __syncthreads();
      }
      const double aux = lc.h_pre;
#pragma unroll
      for (int k = 0; k < kSmallPer; ++k) {
        const int i = threadIdx.x + k * BS;
        if (i < d.m) dual_elem(i, k, yl[k], al[k], apl[k], hb[k], aux);
      }
    } else {
      constexpr int MB = 8;       // chunks per batch: 32 loads in flight per lane, 4 batches at m = 8192
      if (d.h.kind == APX_G_L2) {
        double z2[1] = {0.0};
        for (int k0 = 0; k0 < per_m; k0 += MB) {
          double y4[MB], a4[MB], p4[MB], b4[MB];
#pragma unroll
          for (int u = 0; u < MB; ++u) {
            const int ic = min((int)threadIdx.x + (k0 + u) * BS, d.m - 1);
            y4[u] = (double)Y[ic];
            a4[u] = (double)(T)d.ax_cur[ic];
            p4[u] = (double)(T)d.ax_prev[ic];
            b4[u] = shift_at<T>(d.h, ic);
          }
#pragma unroll
          for (int u = 0; u < MB; ++u)
            if ((int)threadIdx.x + (k0 + u) * BS < d.m) {
              const double w = y4[u] + sigma * ((1.0 + rho) * a4[u] - rho * p4[u]);
              const double z = w / sigma + b4[u];
              z2[0] = fma(z, z, z2[0]);
            }
        }
        block_sums_t0<1, BS>(z2);
        if (threadIdx.x == 0) lc.h_pre = sqrt(z2[0]);
        __syncthreads();
      }
      const double aux = lc.h_pre;
      for (int k0 = 0; k0 < per_m; k0 += MB) {
        double y4[MB], a4[MB], p4[MB], b4[MB];
#pragma unroll
        for (int u = 0; u < MB; ++u) {
          const int ic = min((int)threadIdx.x + (k0 + u) * BS, d.m - 1);
          y4[u] = (double)Y[ic];
          a4[u] = (double)(T)d.ax_cur[ic];
          p4[u] = (double)(T)d.ax_prev[ic];
          b4[u] = shift_at<T>(d.h, ic);
        }
#pragma unroll
        for (int u = 0; u < MB; ++u) {
          const int i = threadIdx.x + (k0 + u) * BS;
          if (i < d.m) dual_elem(i, k0 + u, y4[u], a4[u], p4[u], b4[u], aux);
        }
      }
    }
    block_sums_t0<2, BS>(r2);
    if (threadIdx.x == 0) {
      const double ndr = sqrt(r2[0]);
      lc.ndr2 = ndr * ndr;
      lc.h_Ax = gval_final(d.h, r2[1]);
      finish_iteration(&lc, blockIdx.x == 0 ? d.recs : nullptr, d.rec_cap);      // :348-354
      if (blockIdx.x == 0) *d.c_next = lc;
    }
    __syncthreads();
    if (lc.done != 0) return;      // converged: (x, y+) is what the reference returns (:355)
    small_dots_w<T, BS>(d.w_t, (const T*)d.A, d.ld, d.n, d.m, ys, d.aty_next, true);    // :358  At_y = A' * y
  }
}

// MODE 0: only A x_cur -> ax_next (pre-loop, :327) ; MODE 2: x+ and A x+
template <typename T, int MODE, int FK>
__global__ __launch_bounds__(kBlock) void k_small_pd_a(PdDev d) {
  __shared__ double xs[kSmallMax];
  const T* __restrict__ X = (const T*)d.x_cur;
  if constexpr (MODE == 0) {
    for (int i = threadIdx.x; i < d.n; i += kBlock) xs[i] = (double)X[i];
    __syncthreads();
    small_dots_w<T>(d.w_n, (const T*)d.AT, d.ldT, d.m, d.n, xs, d.ax_next, true);
    if constexpr (FK == APX_F_QUADRATIC)
      small_dots_w<T>(d.w_n, (const T*)d.QT, d.ldQ, d.n, d.n, xs, d.qx_next, true);   // temp = Q * x
    return;
  } else {
    const int done = d.c_next->done;            // written by the previous kernel: block-uniform
    const double gamma = d.c_next->gamma;
    T* __restrict__ XN = (T*)d.x_next;
    T* __restrict__ VN = (T*)d.v_next;
    const int per_n = (d.n + kBlock - 1) / kBlock;
    for (int k0 = 0; k0 < per_n; k0 += 4) {
      double x4[4], a4[4], gb4[4];
      [[maybe_unused]] double t4[4], qv4[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int ic = min((int)threadIdx.x + (k0 + u) * kBlock, d.n - 1);
        x4[u] = (double)X[ic];
        a4[u] = (double)(T)d.aty_next[ic];
        gb4[u] = shift_at<T>(d.g, ic);
        if constexpr (FK == APX_F_QUADRATIC) {
          t4[u] = (double)(T)d.qx_cur[ic];
          qv4[u] = (double)((const T*)d.qvec)[ic];
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int k = k0 + u;
        const int i = threadIdx.x + k * kBlock;
        if (i < d.n) {
          double gi = 0.0;
          if constexpr (FK == APX_F_QUADRATIC) gi = (double)(T)(t4[u] + qv4[u]);
          const double gs = gi + a4[u];                                      // grad_x + At_y
          const T vi = (T)(x4[u] - gamma * gs);                              // :359
          const T xn = (T)prox_elem_s<T>(d.g, (double)vi, gamma, i, 0.0, gb4[u]);   // :361
          xs[i] = (double)xn;
          if (done == 0 && (unsigned)k % gridDim.x == blockIdx.x) {
            XN[i] = xn;
            VN[i] = vi;
          }
        }
      }
    }
    __syncthreads();
    small_dots_w<T>(d.w_n, (const T*)d.AT, d.ldT, d.m, d.n, xs, d.ax_next, done == 0);   // :335 of the next iteration
    if constexpr (FK == APX_F_QUADRATIC)
      small_dots_w<T>(d.w_n, (const T*)d.QT, d.ldQ, d.n, d.n, xs, d.qx_next, done == 0);  // :336  temp = Q * x+
  }
}

// ---------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------
bool small_pd_eligible(const apx_problem* p) {
  if (!p->ctx->fused_small) return false;
  if (p->A_lin == nullptr || p->sharded()) return false;
  if (p->f_kind != APX_F_ZERO && p->f_kind != APX_F_QUADRATIC) return false;
  if (p->f_kind == APX_F_QUADRATIC && (p->A_f == nullptr || no_column_storage(p->A_f))) return false;
  if (no_column_storage(p->A_lin) || p->ctx->skip_zero_cols) return false;
  return small_shape_ok(p->A_lin->m, p->n);
}

// roles of the rotating buffers in loop iteration k (1-based)
static PdDev roles(const apx_solver* s, int64_t k) {
  const SmallPdState* st = s->small_pd;
  PdDev d = st->base;
  const int k3 = (int)(k % 3), k2 = (int)(k % 2);
  d.x_prev = st->xb[k3].p;
  d.x_cur = st->xb[(k3 + 1) % 3].p;
  d.x_next = st->xb[(k3 + 2) % 3].p;
  d.ax_prev = (const double*)st->axb[k3].p;
  d.ax_cur = (const double*)st->axb[(k3 + 1) % 3].p;
  d.ax_next = (double*)st->axb[(k3 + 2) % 3].p;
  d.qx_prev = (const double*)st->qxb[k3].p;
  d.qx_cur = (const double*)st->qxb[(k3 + 1) % 3].p;
  d.qx_next = (double*)st->qxb[(k3 + 2) % 3].p;
  d.v_cur = st->vb[k2].p;
  d.v_next = st->vb[1 - k2].p;
  d.y_cur = st->yb[k2].p;
  d.y_next = st->yb[1 - k2].p;
  d.aty_cur = (const double*)st->atyb[k2].p;
  d.aty_next = (double*)st->atyb[1 - k2].p;
  d.c_cur = (const Ctrl*)st->cb[k2].p;
  d.c_next = (Ctrl*)st->cb[1 - k2].p;
  return d;
}

template <typename T, int MODE, int FK, bool MREG>
static void launch_pd_b_inst(const PdDev& d, hipStream_t q) {
  constexpr int BS = MREG ? kBlock : kTallBlock;
  allow_big_lds<k_small_pd_b<T, MODE, FK, MREG, BS>>();
  hipLaunchKernelGGL((k_small_pd_b<T, MODE, FK, MREG, BS>), dim3(pick_grid(d.n, d.w_t, BS)), dim3(BS), m_vector_lds_bytes(d.m),
                     q, d);
}
template <typename T, int MODE>
static void launch_pd_b_t(const PdDev& d, hipStream_t q) {
  const bool quad = d.fk == APX_F_QUADRATIC, mreg = d.m <= kSmallMax;
  if (quad) {
    if (mreg) launch_pd_b_inst<T, MODE, APX_F_QUADRATIC, true>(d, q);
    else launch_pd_b_inst<T, MODE, APX_F_QUADRATIC, false>(d, q);
  } else {
    if (mreg) launch_pd_b_inst<T, MODE, APX_F_ZERO, true>(d, q);
    else launch_pd_b_inst<T, MODE, APX_F_ZERO, false>(d, q);
  }
}
template <int MODE>
static void launch_pd_b(const PdDev& d, int32_t dtype, hipStream_t q) {
  if (dtype == APX_F32) launch_pd_b_t<float, MODE>(d, q);
  else launch_pd_b_t<double, MODE>(d, q);
}
template <int MODE>
static void launch_pd_a(const PdDev& d, int32_t dtype, hipStream_t q) {
  const bool quad = d.fk == APX_F_QUADRATIC;
  const dim3 grid(pick_grid(quad ? std::max(d.m, d.n) : d.m, d.w_n));     // (the Q x product has n outputs)
  if (dtype == APX_F32) {
    if (quad) hipLaunchKernelGGL((k_small_pd_a<float, MODE, APX_F_QUADRATIC>), grid, dim3(kBlock), 0, q, d);
    else hipLaunchKernelGGL((k_small_pd_a<float, MODE, APX_F_ZERO>), grid, dim3(kBlock), 0, q, d);
  } else {
    if (quad) hipLaunchKernelGGL((k_small_pd_a<double, MODE, APX_F_QUADRATIC>), grid, dim3(kBlock), 0, q, d);
    else hipLaunchKernelGGL((k_small_pd_a<double, MODE, APX_F_ZERO>), grid, dim3(kBlock), 0, q, d);
  }
}

// called by apx_solver_create instead of the generic pre-loop (:324-332); Ctrl is initialised in p->ctrl_buf
int32_t small_pd_setup(apx_solver* s, const void* x0, const void* y0) {
  apx_problem* p = s->p;
  apx_ctx* ctx = p->ctx;
  const apx_matrix* A = p->A_lin;
  std::unique_ptr<SmallPdState> st(new SmallPdState());
  const size_t es = p->es();
  const size_t nb = (size_t)p->n_pad * es, mb = (size_t)p->m_pad * es;
  for (auto& b : st->xb) APX_TRY(b.alloc_zero(nb, ctx->stream));
  for (auto& b : st->vb) APX_TRY(b.alloc_zero(nb, ctx->stream));
  for (auto& b : st->yb) APX_TRY(b.alloc_zero(mb, ctx->stream));
  for (auto& b : st->axb) APX_TRY(b.alloc_zero((size_t)p->m_pad * sizeof(double), ctx->stream));
  for (auto& b : st->qxb) APX_TRY(b.alloc_zero((size_t)p->n_pad * sizeof(double), ctx->stream));
  for (auto& b : st->atyb) APX_TRY(b.alloc_zero((size_t)p->n_pad * sizeof(double), ctx->stream));
  for (auto& b : st->cb) APX_TRY(b.alloc_zero(sizeof(Ctrl), ctx->stream));
  PdDev d{};
  d.A = A->d;
  d.ld = A->ld;
  if (A->dT != nullptr) {
    d.AT = A->dT;
    d.ldT = A->ldT;
  } else {               // private transposed copy: the caller's matrix object is untouched
    d.ldT = round_up(A->n, kPad);
    APX_TRY(st->at.alloc_zero((size_t)d.ldT * (size_t)A->ld * es, ctx->stream));
    APX_TRY(transpose_into(ctx, A, st->at.p, d.ldT));
    d.AT = st->at.p;
  }
  d.fk = p->f_kind;
  if (p->f_kind == APX_F_QUADRATIC) {   // rows of Q as contiguous columns: (Q x)_i = <Q[i, :], x> exactly as gemv_n forms it
    const apx_matrix* Q = p->A_f;
    if (Q->dT != nullptr) {
      d.QT = Q->dT;
      d.ldQ = Q->ldT;
    } else {
      d.ldQ = round_up(Q->n, kPad);
      APX_TRY(st->qt.alloc_zero((size_t)d.ldQ * (size_t)Q->ld * es, ctx->stream));
      APX_TRY(transpose_into(ctx, Q, st->qt.p, d.ldQ));
      d.QT = st->qt.p;
    }
    d.qvec = p->b.p;
  }
  d.m = (int32_t)A->m;
  d.n = (int32_t)p->n;
  d.w_n = pick_waves(d.n);
  d.w_t = pick_waves(d.m, m_side_block(d.m));
  d.g = p->gd;
  d.h = p->hd;
  d.recs = (apx_record*)s->recs.p;
  d.rec_cap = s->rec_cap;
  st->base = d;
  s->small_pd = st.release();
  const SmallPdState* S = s->small_pd;
  // roles of iteration 1: x_prev = xb[1] = x0, x_cur = xb[2] = x1, ax_prev = axb[1] = A x0, ax_cur = axb[2] = A x1,
  // v_cur = vb[1], y_cur = yb[1] = y0, aty_cur = atyb[1] = A' y0, c_cur = cb[1] = the initial control block
  APX_TRY(h2d(ctx, S->xb[1].p, x0, (size_t)p->n * es));
  APX_TRY(h2d(ctx, S->yb[1].p, y0, (size_t)p->m * es));
  APX_HIP(hipMemcpyAsync(S->cb[1].p, p->ctrl_buf.p, sizeof(Ctrl), hipMemcpyDeviceToDevice, ctx->stream));
  PdDev pre = S->base;
  pre.x_cur = S->xb[1].p;                                  // :327-328  A_x = A * x0 (and Q * x0)
  pre.ax_next = (double*)S->axb[1].p;
  pre.qx_next = (double*)S->qxb[1].p;
  launch_pd_a<0>(pre, p->dtype, ctx->stream);
  pre.y_cur = S->yb[1].p;                                  // :329  At_y = A' * y0
  pre.aty_next = (double*)S->atyb[1].p;
  launch_pd_b<0>(pre, p->dtype, ctx->stream);
  pre.c_next = (Ctrl*)S->cb[1].p;                          // :330-332  v, x1 (and A x1 for iteration 1)
  pre.x_next = S->xb[2].p;
  pre.v_next = S->vb[1].p;
  pre.ax_next = (double*)S->axb[2].p;
  pre.qx_cur = (const double*)S->qxb[1].p;
  pre.qx_next = (double*)S->qxb[2].p;
  launch_pd_a<2>(pre, p->dtype, ctx->stream);
  APX_HIP(hipGetLastError());
  return APX_OK;
}

int32_t small_pd_enqueue(apx_solver* s) {
  const PdDev d = roles(s, s->iters + 1);
  launch_pd_b<2>(d, s->p->dtype, s->p->ctx->stream);
  launch_pd_a<2>(d, s->p->dtype, s->p->ctx->stream);
  APX_HIP(hipGetLastError());
  return APX_OK;
}

// the control block written last (by the k_small_pd_b of iteration s->iters; the initial one before that)
const void* small_pd_ctrl(const apx_solver* s) { return s->small_pd->cb[(s->iters + 1) % 2].p; }

// where the iterates the reference would return live (Appendix A.8): on convergence at iteration k the
// tested x and the already updated y; otherwise the next iterates
void small_pd_result_ptrs(const apx_solver* s, const Ctrl& c, const void** x, const void** y) {
  const SmallPdState* st = s->small_pd;
  if (c.done != 0) {
    *x = st->xb[(c.it + 1) % 3].p;       // x_cur(k)
    *y = st->yb[(c.it + 1) % 2].p;       // y_next(k)
  } else {
    *x = st->xb[(s->iters + 2) % 3].p;   // x_next(K)
    *y = st->yb[(s->iters + 1) % 2].p;   // y_next(K)
  }
}

}  // namespace apx
