// skinny_pg.hip — the proximal-gradient iteration (`adaptive_proxgrad` / `fixed_proxgrad`, src/AdaProx.jl:418-421,
// 457-459 over the loop :334-362) for TALL, SKINNY dense A: at most 124 unknowns, thousands of rows — the shape of the
// reference's logistic-regression data once densified (mushrooms 8124 x 112, phishing 11055 x 68,
// experiments/sparse_logreg/runme.jl:177-204) and of any tall least-squares problem.  With so few columns the
// gradient is n block-partial sums and the whole n side fits one workgroup:
//
//   k_spg_rows  a block owns groups of 64 rows (lane = row) and splits the COLUMNS over its waves, CPW per wave:
//               every wave loads its CPW entries of the row once, into registers (coalesced down each column, all
//               loads in flight together) ; the waves' partial dots meet in LDS -> A_x[i] (+ bias) ; res = A_x - b
//               or the logistic link (lasso/runme.jl:22 ; sparse_logreg/runme.jl:24-36) ; A' res is accumulated
//               per lane from the SAME registers and wave-summed once per block ; block partials of ||res||^2
//               or the loss terms, of sum(res)
//   k_spg_close (one block of 1024): partials summed in a fixed order -> f(x), grad f(x) ; primal residual and
//               the step-size sums (:338, :260-261) ; rule (:340) ; termination, record (:348-354) ;
//               v = x - gamma grad ; x+ = prox_{gamma g}(v) (:359-361)
//
// Two launches per iteration, A read once, no redundant work, every sum in a fixed order (bit-reproducible); the
// control block is the generic one (`Ctrl`), so polling, results and hipGraph replay work as in the generic path.
// (A thread per whole row — skinny_pd.hip's scheme, right for n ~ 10 — leaves a chain of n dependent loads and n
//  wave sums per thread: 51 us per iteration at 8124 x 112.)
#include <cstdlib>
#include <memory>

#include "skinny_rows_dev.h"

namespace apx {

struct SkinnyPgState {
  DBuf part;
  SkinnyRowsPlan plan;
  SkinnyPgDev dev{};
};
void skinny_pg_destroy(SkinnyPgState* st) { delete st; }

// MODE 0: pre-loop (:328-332) grad f(x0) ; v = x0 - gamma grad ; x1 = prox(v)   MODE 1: close iteration k and step
template <typename T, int FK, int MODE>
__global__ __launch_bounds__(kSpgCloseBlock) void k_spg_close(SkinnyPgDev d, int B) {
  __shared__ SkinnyRed red[16];
  __shared__ Ctrl lc;
  __shared__ double fin[2];
  __shared__ double qs[2][8];
  const int tid = threadIdx.x;
  T* __restrict__ X = (T*)d.X;
  T* __restrict__ XP = (T*)d.XP;
  T* __restrict__ GP = (T*)d.GP;
  T* __restrict__ V = (T*)d.V;
  // ---- every load of this kernel -------------------------------------------------------------------
  const int done = d.c->done;
  const int nc = d.ncols + 2;
  SkinnyColLoads L;
  skinny_colsum_issue<kSpgCloseBlock>(d.part, d.ldpart, nc, d.npow, B, L);
  const bool mine = tid < d.n;
  const T x_t = mine ? X[tid] : (T)0, v_t = mine ? V[tid] : (T)0, gp_t = mine ? GP[tid] : (T)0, xp_t = mine ? XP[tid] : (T)0;
  if (tid == 0) lc = *d.c;
  if (done != 0) return;
  // ---- column sums of the block partials, in a fixed order --------------------------------------------
  const double tot = skinny_colsum_finish(d.part, d.ldpart, nc, B, L, red);
  if (tid == d.ncols) fin[0] = tot;
  if (tid == d.ncols + 1) fin[1] = tot;
  __syncthreads();
  // the gradient entry of this thread
  double gj = 0.0;
  if (mine) {
    if (FK == APX_F_LOGISTIC && tid == d.n - 1) {
      gj = (double)(T)(fin[1] / lc.n_mean);                        // grad[end] = mean(probs - y)
    } else {
      const T gs = (T)tot;                                          // A' * res
      gj = FK == APX_F_LOGISTIC ? (double)(T)((double)gs / lc.n_mean) : (double)gs;
    }
  }
  const double xj = (double)x_t;
  if constexpr (MODE == 1) {
    if (tid < 128) {                                                // n <= 124: the first two waves hold it all
      double q[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      if (mine) {
        const double pr = ((double)v_t - xj) / lc.gamma + gj;      // :338 (A'y == 0)
        const double dg = gj - (double)gp_t;
        const double dx = xj - (double)xp_t;
        q[0] = pr * pr;
        q[1] = dg * dg;
        q[2] = dg * dx;
        q[3] = dx * dx;
        q[4] = gval_elem<T>(d.g, xj, tid);
      }
      wave_sum_multi<8>(q);                                         // lane l: sum l / 8
      if ((tid & 7) == 0) qs[tid >> 6][(tid & 63) >> 3] = q[0];
    }
    __syncthreads();
    if (tid == 0) {
      if constexpr (FK == APX_F_LEAST_SQUARES) {
        const double nr = sqrt(fin[0]);                             // 0.5 * norm(res)^2 : nrm2 THEN square
        lc.f_x = 0.5 * (nr * nr);
      } else {
        lc.f_x = -(fin[0] / lc.n_mean);
        lc.bias_grad = fin[1] / lc.n_mean;
      }
      lc.it += 1;
      const double npr = sqrt(qs[0][0] + qs[1][0]);
      lc.npr2 = npr * npr;
      lc.g_x = gval_final(d.g, qs[0][4] + qs[1][4]);
      rule_update(&lc, qs[0][1] + qs[1][1], qs[0][2] + qs[1][2], qs[0][3] + qs[1][3]);   // :340-342
      lc.ndr2 = 0.0;                                                // h = Zero, A = 0: dual_res == 0
      lc.h_Ax = 0.0;
      finish_iteration(&lc, d.recs, d.rec_cap);                     // :348-354
      *d.c = lc;
    }
    __syncthreads();
    if (lc.done != 0) return;      // converged: x is the tested iterate, nothing moves (:355)
  }
  if (mine) {
    const double gamma = lc.gamma;
    const T vj = (T)(xj - gamma * gj);                              // :330 / :359
    const T xn = (T)prox_elem<T>(d.g, (double)vj, gamma, tid, 0.0); // :332 / :361
    XP[tid] = x_t;
    GP[tid] = (T)gj;
    V[tid] = vj;
    X[tid] = xn;
  }
}

// ---------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------
bool skinny_pg_eligible(const apx_problem* p) { return skinny_rows_eligible(p); }

template <typename T, int FK, int MODE>
static void spg_tick_t(const SkinnyPgState* st, hipStream_t q) {
  const SkinnyPgDev& d = st->dev;
  skinny_rows_launch<T, FK>(st->plan, d.X, &d.c->done, q);
  hipLaunchKernelGGL((k_spg_close<T, FK, MODE>), dim3(1), dim3(kSpgCloseBlock), 0, q, d, st->plan.B);
}
template <int MODE>
static void spg_tick(const SkinnyPgState* st, int32_t dtype, int fk, hipStream_t q) {
  const bool ls = fk == APX_F_LEAST_SQUARES;
  if (dtype == APX_F32) {
    if (ls) spg_tick_t<float, APX_F_LEAST_SQUARES, MODE>(st, q);
    else spg_tick_t<float, APX_F_LOGISTIC, MODE>(st, q);
  } else {
    if (ls) spg_tick_t<double, APX_F_LEAST_SQUARES, MODE>(st, q);
    else spg_tick_t<double, APX_F_LOGISTIC, MODE>(st, q);
  }
}

// called by apx_solver_create instead of the generic pre-loop (:324-332); Ctrl is initialised in p->ctrl_buf, x0 in s->x
int32_t skinny_pg_setup(apx_solver* s) {
  apx_problem* p = s->p;
  apx_ctx* ctx = p->ctx;
  const apx_matrix* A = p->A_f;
  std::unique_ptr<SkinnyPgState> st(new SkinnyPgState());
  skinny_rows_plan(A, p->n, p->b.p, &st->plan);
  APX_TRY(st->part.alloc_zero((size_t)st->plan.B * st->plan.dev.ldpart * sizeof(double), ctx->stream));
  st->plan.dev.part = (double*)st->part.p;
  SkinnyPgDev d{};
  d.ncols = st->plan.dev.ncols;
  d.n = st->plan.dev.n;
  d.npow = st->plan.npow;
  d.ldpart = st->plan.dev.ldpart;
  d.X = s->x.p;                 // the solver's own n-vectors: results and stepping work as in the generic path
  d.XP = s->xprev.p;
  d.GP = s->gprev.p;
  d.V = s->v.p;
  d.part = st->plan.dev.part;
  d.c = p->ctrl();
  d.g = p->gd;
  d.recs = (apx_record*)s->recs.p;
  d.rec_cap = s->rec_cap;
  st->dev = d;
  spg_tick<0>(st.get(), p->dtype, p->f_kind, ctx->stream);
  APX_HIP(hipGetLastError());
  s->skinny_pg = st.release();
  return APX_OK;
}

int32_t skinny_pg_enqueue(apx_solver* s) {
  const SkinnyPgState* st = s->skinny_pg;
  spg_tick<1>(st, s->p->dtype, s->p->f_kind, s->p->ctx->stream);
  APX_HIP(hipGetLastError());
  return APX_OK;
}

}  // namespace apx
