// skinny_pd.hip — the primal-dual iteration (src/AdaProx.jl:334-362, f = Zero) for TALL, SKINNY dense A:
// n <= 124 columns, any reasonable number of rows — exactly the reference's own LAD / square-root-lasso
// data (experiments/least_absolute_deviation/runme.jl:128-160: cpusmall_scale 8192 x 13, abalone 4177 x 9,
// housing_scale 506 x 14).  With so few columns x fits in a handful of LDS words, so nothing about the m
// side has to be done redundantly (small_pd.hip's cost on these shapes) and A x needs no kernel of its own:
//
//   k_skinny_rows  (one thread per row, 256 rows per block)
//        every block: primal_res, ||dx||^2, g(x) over the n columns (:338, :260-261), rule (:340-342)
//        its rows   : A_x[i] = <A[i,:], x> (:335, coalesced down the columns of A) ; w (:344) ; y+ (:345) ;
//                     dual_res (:347) ; y, A_x updated in place (each entry owned by one thread) ;
//                     block partials of A'y+ (n numbers), ||dual_res||^2, h(A_x)
//   k_skinny_close (one block)
//        sums the block partials in a fixed order ; norm_res, termination, record (:348-354) ;
//        v = x - gamma A'y+ ; x+ = prox_{gamma g}(v) (:358-361)
//
// For h = NormL2 the projection needs ||w/sigma + shift|| over all rows before any y+ exists, so the row
// kernel is split in two (w and its partial norms ; then every block sums those partials itself and applies the
// prox).  Two (three) launches per iteration, the
// control block is written by one thread, every sum has a fixed order: bit-reproducible, no atomics.
#include <cstdlib>
#include <memory>

#include "skinny_dev.h"

namespace apx {

// PHASE 0: pre-loop (:327-329)  A_x = A x0 into ax, block partials of A' y0 ; nothing else
// PHASE 1: the whole row work of an iteration (elementwise conjugate prox)
// PHASE 2: l2 ball, first half  : A_x, w -> scratch, partial ||w/sigma + shift||^2
// PHASE 3: l2 ball, second half : y+ from the scratch, dual residual, partials
template <typename T, int PHASE>
__global__ __launch_bounds__(kBlock) void k_skinny_rows(SkinnyDev d) {
  __shared__ double xs[kSkinnyMaxN];
  __shared__ Ctrl lc;
  __shared__ SkinnyRed wred[kBlock / 64];
  const int tid = threadIdx.x;
  const T* __restrict__ A = (const T*)d.A;
  const int i = blockIdx.x * kBlock + tid;
  const bool row_ok = i < d.m;
  const int ic = row_ok ? i : d.m - 1;
  T* __restrict__ Y = (T*)d.Y;
  [[maybe_unused]] double h_pre = 0.0;

  if constexpr (PHASE == 0) {
    if (tid < d.n) xs[tid] = (double)((const T*)d.X)[tid];
    __syncthreads();
  } else {
    // (every load goes out before the branch on the done flag, and the control block goes straight to LDS behind
    //  them: each dependent round trip to memory costs 1-2 us here, and a private copy would live in scratch)
    const int done = d.c->done;                // written by an earlier launch: block-uniform
    if constexpr (PHASE == 1 || PHASE == 2) {
      // ---- n side, redundantly per block (n <= 124 values): residual, rule ------------------------
      const double gamma = d.c->gamma;
      const bool mine = tid < d.n;
      const T x_t = mine ? ((const T*)d.X)[tid] : (T)0, v_t = mine ? ((const T*)d.V)[tid] : (T)0,
              xp_t = mine ? ((const T*)d.XP)[tid] : (T)0;
      const double aty_t = mine ? d.aty[tid] : 0.0;
      if (tid == 0) lc = *d.c;
      if (done != 0) return;
      double q[5] = {0, 0, 0, 0, 0};
      if (mine) {
        const double xj = (double)x_t;
        const double pr = ((double)v_t - xj) / gamma + 0.0 + (double)(T)aty_t;   // :338
        const double dx = xj - (double)xp_t;
        q[0] = pr * pr;
        q[3] = dx * dx;
        q[4] = gval_elem<T>(d.g, xj, tid);
        xs[tid] = xj;
      }
      block_sums_t0<5>(q);
      if (tid == 0) {
        lc.it += 1;
        const double npr = sqrt(q[0]);
        lc.npr2 = npr * npr;
        lc.g_x = gval_final(d.g, q[4]);
        rule_update(&lc, q[1], q[2], q[3]);                 // :340-342  (grad f == 0: dg == 0)
        if (blockIdx.x == 0) *d.cs = lc;                    // staged for k_skinny_close
      }
      __syncthreads();
    } else {
      // l2 ball, second half: ||w/sigma + shift|| from the first half's block partials, summed by every block in
      // the same fixed order (no launch of its own for it)
      if (tid == 0) lc = *d.cs;
      double z4[4];
      skinny_partial_norm_issue(d.part, d.ldpart, d.n + 2, (int)gridDim.x, z4);
      if (done != 0) return;
      h_pre = skinny_partial_norm_finish(d.part, d.ldpart, d.n + 2, (int)gridDim.x, z4);   // (lc is visible behind it)
    }
  }

  // ---- this thread's row ------------------------------------------------------------------------
  double a = 0.0;
  if constexpr (PHASE != 3) a = skinny_row_dot<T>(A, d.ld, ic, d.n, xs);            // A_x[i] = <A[i,:], x>
  double yn = 0.0, sdr = 0.0, shv = 0.0, sz = 0.0;
  if constexpr (PHASE == 0) {
    if (row_ok) d.ax[i] = a;
    yn = row_ok ? (double)Y[ic] : 0.0;
  } else {
    const double sigma = lc.sigma, rho = lc.rho;
    const double b = shift_at<T>(d.h, ic);
    if constexpr (PHASE == 2) {
      const double w = (double)Y[ic] + sigma * ((1.0 + rho) * a - rho * d.ax[ic]);     // :344
      if (row_ok) {
        d.wbuf[i] = w;
        d.axn[i] = a;
        const double z = w / sigma + b;
        sz = z * z;
      }
    } else {
      double w;
      if constexpr (PHASE == 1) {
        w = (double)Y[ic] + sigma * ((1.0 + rho) * a - rho * d.ax[ic]);                // :344
      } else {
        w = d.wbuf[ic];
        a = d.axn[ic];
      }
      const double aux = h_pre;
      const T ynt = (T)prox_conj_elem_s<T>(d.h, w, sigma, ic, aux, b);                  // :345
      if (row_ok) {
        yn = (double)ynt;
        const double dr = (w - yn) / sigma - a;                                         // :347
        sdr = dr * dr;
        shv = gval_elem_s<T>(d.h, a, ic, b);
        Y[i] = ynt;
        d.ax[i] = a;                                                                    // A_x_prev for the next iteration
      }
    }
  }

  // ---- block partials: A' y (n numbers, A re-read from L1 / L2), ||dual_res||^2, h(A_x), ||z||^2 ------
  const double ex[kSkinnyExtra] = {sdr, shv, sz, 0.0};
  const double tot = skinny_block_partials<T>(A, d.ld, ic, d.n, yn, PHASE != 2, ex, wred);
  if (tid < d.n + kSkinnyExtra) {
    const bool is_z = tid == d.n + 2;
    if (PHASE == 2 ? is_z : (PHASE == 3 ? !is_z : true)) d.part[(int64_t)blockIdx.x * d.ldpart + tid] = tot;
  }
}

// MODE 0: pre-loop  v = x0 - gamma A'y0 ; x1 = prox(v)  (:330-332)
// MODE 1: close the iteration and take the primal step
template <typename T, int MODE>
__global__ __launch_bounds__(kSkinnyCloseBlock) void k_skinny_close(SkinnyDev d, int B) {
  __shared__ SkinnyRed red[16];
  __shared__ Ctrl lc;
  __shared__ double fin[4];
  const int tid = threadIdx.x;
  // every load first (the done flag, the partials, x, the control block), then the branch
  const int done = MODE != 0 ? d.c->done : 0;
  SkinnyColLoads L;
  skinny_colsum_issue<kSkinnyCloseBlock>(d.part, d.ldpart, d.n + kSkinnyExtra, d.npow, B, L);
  T* __restrict__ X = (T*)d.X;
  const T x_t = tid < d.n ? X[tid] : (T)0;
  if (tid == 0) lc = MODE == 1 ? *d.cs : *d.c;
  if (done != 0) return;
  const double tot = skinny_colsum_finish(d.part, d.ldpart, d.n + kSkinnyExtra, B, L, red);
  {
    if (tid == d.n) fin[0] = tot;
    if (tid == d.n + 1) fin[1] = tot;
    __syncthreads();
    if constexpr (MODE == 1) {
      if (tid == 0) {
        const double ndr = sqrt(fin[0]);
        lc.ndr2 = ndr * ndr;
        lc.h_Ax = gval_final(d.h, fin[1]);
        finish_iteration(&lc, d.recs, d.rec_cap);            // :348-354
        *d.c = lc;
      }
      __syncthreads();
      if (lc.done != 0) return;                              // (x, y+) is what the reference returns (:355)
    }
    if (tid < d.n) {
      const double gamma = lc.gamma;
      const double xj = (double)x_t;
      const T aty = (T)tot;                                  // :358  At_y = A' * y
      const T vj = (T)(xj - gamma * (0.0 + (double)aty));    // :359
      const T xn = (T)prox_elem<T>(d.g, (double)vj, gamma, tid, 0.0);     // :361
      ((T*)d.XP)[tid] = x_t;
      X[tid] = xn;
      ((T*)d.V)[tid] = vj;
      d.aty[tid] = (double)aty;
    }
  }
}

// ---------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------
bool skinny_pd_eligible(const apx_problem* p) {
  if (!p->ctx->fused_small) return false;
  if (p->A_lin == nullptr || p->sharded() || p->f_kind != APX_F_ZERO) return false;
  if (no_column_storage(p->A_lin) || p->ctx->skip_zero_cols) return false;
  const int64_t m = p->A_lin->m;
  // a block per 256 rows; the closing block sums that many partials per column.  Below ~1000 rows (3000 for the
  // three-launch l2 form) small_pd.hip's redundant m side is cheaper (tools/small_bench.py: 506 x 14 12.2 vs 14.3 us)
  const int64_t m_min = p->hd.kind == APX_G_L2 ? 3072 : 1024;
  return p->n <= kSkinnyMaxN && m >= m_min && m <= 256 * 1024 && p->n * 4 <= m;
}

template <int PHASE>
static void launch_rows(const SkinnyDev& d, int32_t dtype, int B, hipStream_t q) {
  if (dtype == APX_F32) hipLaunchKernelGGL((k_skinny_rows<float, PHASE>), dim3(B), dim3(kBlock), 0, q, d);
  else hipLaunchKernelGGL((k_skinny_rows<double, PHASE>), dim3(B), dim3(kBlock), 0, q, d);
}
template <int MODE>
static void launch_close(const SkinnyDev& d, int32_t dtype, int B, hipStream_t q) {
  if (dtype == APX_F32) hipLaunchKernelGGL((k_skinny_close<float, MODE>), dim3(1), dim3(kSkinnyCloseBlock), 0, q, d, B);
  else hipLaunchKernelGGL((k_skinny_close<double, MODE>), dim3(1), dim3(kSkinnyCloseBlock), 0, q, d, B);
}

// called by apx_solver_create instead of the generic pre-loop (:324-332); Ctrl is initialised in p->ctrl_buf
int32_t skinny_pd_setup(apx_solver* s, const void* x0, const void* y0) {
  apx_problem* p = s->p;
  apx_ctx* ctx = p->ctx;
  const apx_matrix* A = p->A_lin;
  std::unique_ptr<SkinnyState> st(new SkinnyState());
  const size_t es = p->es();
  const size_t nb = (size_t)p->n_pad * es, mb = (size_t)p->m_pad * es, md = (size_t)p->m_pad * sizeof(double);
  st->B = (int)((A->m + kBlock - 1) / kBlock);
  SkinnyDev d{};
  d.A = A->d;
  d.ld = A->ld;
  d.m = (int32_t)A->m;
  d.n = (int32_t)p->n;
  d.npow = skinny_npow(d.n);
  d.ldpart = round_up(d.n + kSkinnyExtra, 8);
  APX_TRY(st->x.alloc_zero(nb, ctx->stream));
  APX_TRY(st->xp.alloc_zero(nb, ctx->stream));
  APX_TRY(st->v.alloc_zero(nb, ctx->stream));
  APX_TRY(st->aty.alloc_zero((size_t)p->n_pad * sizeof(double), ctx->stream));
  APX_TRY(st->y.alloc_zero(mb, ctx->stream));
  APX_TRY(st->ax.alloc_zero(md, ctx->stream));
  APX_TRY(st->axn.alloc_zero(md, ctx->stream));
  APX_TRY(st->wbuf.alloc_zero(md, ctx->stream));
  APX_TRY(st->part.alloc_zero((size_t)st->B * d.ldpart * sizeof(double), ctx->stream));
  APX_TRY(st->cs.alloc_zero(sizeof(Ctrl), ctx->stream));
  d.X = st->x.p;
  d.XP = st->xp.p;
  d.V = st->v.p;
  d.aty = (double*)st->aty.p;
  d.Y = st->y.p;
  d.ax = (double*)st->ax.p;
  d.axn = (double*)st->axn.p;
  d.wbuf = (double*)st->wbuf.p;
  d.part = (double*)st->part.p;
  d.c = p->ctrl();
  d.cs = (Ctrl*)st->cs.p;
  d.g = p->gd;
  d.h = p->hd;
  d.recs = (apx_record*)s->recs.p;
  d.rec_cap = s->rec_cap;
  st->dev = d;
  APX_TRY(h2d(ctx, st->x.p, x0, (size_t)p->n * es));
  APX_TRY(h2d(ctx, st->y.p, y0, (size_t)p->m * es));
  launch_rows<0>(d, p->dtype, st->B, ctx->stream);          // :327, :329
  launch_close<0>(d, p->dtype, st->B, ctx->stream);         // :330-332
  APX_HIP(hipGetLastError());
  s->skinny = st.release();
  return APX_OK;
}

int32_t skinny_pd_enqueue(apx_solver* s) {
  const SkinnyState* st = s->skinny;
  const SkinnyDev& d = st->dev;
  hipStream_t q = s->p->ctx->stream;
  const int32_t dt = s->p->dtype;
  if (d.h.kind == APX_G_L2) {
    launch_rows<2>(d, dt, st->B, q);
    launch_rows<3>(d, dt, st->B, q);
  } else {
    launch_rows<1>(d, dt, st->B, q);
  }
  launch_close<1>(d, dt, st->B, q);
  APX_HIP(hipGetLastError());
  return APX_OK;
}

}  // namespace apx
