// skinny_mom.hip — `fixed_nesterov` (src/AdaProx.jl:91-142) and `agraal` (:150-192) as DEVICE-RESIDENT loops for
// tall, skinny dense A (at most 124 unknowns, 2048 rows or more): the shape of all three datasets of the
// reference's logistic experiment, which runs both methods on each (experiments/sparse_logreg/runme.jl:104-124).
// The host-driven forms read scalars back every iteration (53 / 80 us per iteration at 8124 x 112); here an
// iteration is one or two TICKS of the two launches of skinny_pg.hip:
//
//   k_spg_rows   (skinny_rows_dev.h) at the point XT: block partials of A' res, of the loss, of sum(res)
//   k_smom_close (one block): f and grad f at XT from the partials, then
//     aGRAAL   : finish iteration k (objective f(x+) + g(x+), record, termination :183-188 ; x_prev, grad_x_prev = x,
//                grad_x ; x = x+ ; grad_x = this pass's gradient :189) and start iteration k + 1 (C, gamma, theta
//                :175-178 ; x_bar :179 ; x+ = prox(x_bar - gamma grad_x) :181 ; norm_res :182) -> XT = x+
//     Nesterov : XT = z: x+ = prox(z - gamma grad_z), norm_res (:130-133) ; with records a second tick at XT = x+ for
//                the objective f(x+) + g(x+) (:134-136, not counted) ; record, termination (:134-139) ; theta, beta
//                (:121-128) and z = x+ + beta (x+ - x) (:129) -> XT
//
// With n <= 124 the whole n side lives in one workgroup: plain n-vectors copied by the closing block, one writer of
// the control block.  The host enqueues ticks in chunks and reads the counters between chunks.
#include <cstdlib>
#include <cstring>
#include <limits>
#include <memory>

#include "skinny_rows_dev.h"

namespace apx {

struct SmomDev {
  int32_t ncols, n, npow, method;
  int64_t ldpart;
  const double* part;
  void* XT;            // the point the row kernel evaluates
  void *X, *XP;        // x, x_prev
  void* AUX;           // aGRAAL: x_bar ; Nesterov: z
  double *G, *GP;      // aGRAAL: grad f(x), grad f(x_prev), entries rounded to T
  MomCtrl* c;
  ProxDev g;
  apx_record* recs;
  int32_t want_obj, _pad;
};

constexpr int kSmomBlock = 1024;

__device__ __forceinline__ double smom_min2(double a, double b) {   // Julia min: NaN-propagating
  if (isnan(a) || isnan(b)) return NAN;
  return a < b ? a : b;
}

// sums of up to eight per-thread values over the first two waves (n <= 124): call from tid < 128, then a barrier,
// then qs[0][k] + qs[1][k]
__device__ __forceinline__ void smom_sums(double (&q)[8], double (*qs)[8]) {
  const int tid = threadIdx.x;
  wave_sum_multi<8>(q);                               // lane l: sum l / 8
  if ((tid & 7) == 0) qs[tid >> 6][(tid & 63) >> 3] = q[0];
}

// MODE 0: aGRAAL: grad f(x_prev) of the pre-loop (:167), XT = x1 next ; Nesterov: only theta, beta, z of iteration 1
//         (launched without a row pass)
// MODE 2: close a tick
template <typename T, int FK, int MODE>
__global__ __launch_bounds__(kSmomBlock) void k_smom_close(SmomDev d, int B) {
  __shared__ SkinnyRed red[16];
  __shared__ MomCtrl lc;
  __shared__ double fin[2];
  __shared__ double qs[2][8];
  __shared__ int go;                   // the loop goes on: form the next point
  __shared__ double beta_s;
  const int tid = threadIdx.x;
  const bool agraal = d.method == MOM_AGRAAL;
  T* __restrict__ XT = (T*)d.XT;
  T* __restrict__ X = (T*)d.X;
  T* __restrict__ XP = (T*)d.XP;
  T* __restrict__ AUX = (T*)d.AUX;
  // ---- every load of this kernel, then the branch ---------------------------------------------------
  const int frozen = MODE == 2 ? d.c->frozen : 0;
  const int nc = d.ncols + 2;
  SkinnyColLoads L;
  skinny_colsum_issue<kSmomBlock>(d.part, d.ldpart, nc, d.npow, B, L);      // (B == 0: no row pass before this launch)
  const bool mine = tid < d.n;
  const T xt_t = mine ? XT[tid] : (T)0, x_t = mine ? X[tid] : (T)0, xp_t = mine ? XP[tid] : (T)0, aux_t = mine ? AUX[tid] : (T)0;
  const double g_t = mine ? d.G[tid] : 0.0, gp_t = mine ? d.GP[tid] : 0.0;
  if (tid == 0) lc = *d.c;
  if (frozen != 0) return;
  const double tot = skinny_colsum_finish(d.part, d.ldpart, nc, B, L, red);
  if (tid == d.ncols) fin[0] = tot;
  if (tid == d.ncols + 1) fin[1] = tot;
  __syncthreads();
  // ---- f and grad f at XT ----------------------------------------------------------------------------
  double gj = 0.0;
  if (mine) {
    if (FK == APX_F_LOGISTIC && tid == d.n - 1) {
      gj = (double)(T)(fin[1] / lc.n_mean);                        // grad[end] = mean(probs - y)
    } else {
      const T gs = (T)tot;                                          // A' * res
      gj = FK == APX_F_LOGISTIC ? (double)(T)((double)gs / lc.n_mean) : (double)gs;
    }
  }
  double f_pt = 0.0;
  if (tid == 0) {
    if constexpr (FK == APX_F_LEAST_SQUARES) {
      const double nr = sqrt(fin[0]);                               // 0.5 * norm(res)^2 : nrm2 THEN square
      f_pt = 0.5 * (nr * nr);
    } else {
      f_pt = -(fin[0] / lc.n_mean);
    }
  }
  auto record = [&](double f_new) {                                 // thread 0: iteration lc.it is complete
    lc.f_new = f_new;
    lc.objective = f_new + gval_final(d.g, lc.g_raw);
    if (d.recs != nullptr && lc.it <= lc.rec_cap) {
      apx_record r;
      r.it = lc.it;
      r.gamma = lc.gamma;
      r.sigma = NAN;
      r.norm_res = lc.norm_res;
      r.objective = lc.objective;
      r.f_x = f_new;
      r.f_evals = agraal ? 1 + lc.it : lc.it;                       // evaluations counted so far (aGRAAL: 2 before the loop)
      r.grad_evals = r.f_evals;
      d.recs[lc.it - 1] = r;
    }
    if (lc.stop_after != 0) lc.done = 1;
  };

  const int phase = lc.phase;                         // (read by everyone before thread 0 may change it)
  __syncthreads();

  if (agraal) {
    if constexpr (MODE == 0) {
      if (mine) {
        d.GP[tid] = gj;                                             // :167  grad f(x_prev)
        XT[tid] = x_t;                                              // next: grad f(x) at x = x1 (:166)
      }
      return;
    } else {
      const bool pending = phase != 0;                              // XT is the x+ of an unfinished iteration
      if (tid == 0) {
        if (pending) {
          lc.it += 1;
          record(f_pt);                                             // :183-188
        }
        go = (lc.done == 0 && lc.it < lc.maxit) ? 1 : 0;
      }
      __syncthreads();
      // x_prev, grad_x_prev = x, grad_x ; x = x+ ; grad_x = grad f(x+)  (:180, :189) — the pre-loop pass only stores grad_x
      const double xc = pending ? (double)xt_t : (double)x_t, xpv = pending ? (double)x_t : (double)xp_t;
      const double gc = gj, gpv = pending ? g_t : gp_t;
      if (mine) {
        if (pending) {
          XP[tid] = x_t;
          X[tid] = xt_t;
          d.GP[tid] = g_t;
        }
        d.G[tid] = gj;
      }
      if (go != 0) {
        if (tid < 128) {
          double q[8] = {0, 0, 0, 0, 0, 0, 0, 0};
          if (mine) {
            const double dx = xc - xpv, dg = gc - gpv;
            q[0] = dx * dx;
            q[1] = dg * dg;
          }
          smom_sums(q, qs);
        }
        __syncthreads();
        if (tid == 0) {
          const double ndx = sqrt(qs[0][0] + qs[1][0]), ndg = sqrt(qs[0][1] + qs[1][1]);
          if (lc.need_gamma0 != 0) {                                         // :168-170
            lc.gamma = ndx / ndg;
            lc.need_gamma0 = 0;
          }
          const double C = (ndx * ndx) / (ndg * ndg);                        // :175
          const double gamma_prev = lc.gamma;                                // :176
          lc.gamma = smom_min2(smom_min2(lc.rho * gamma_prev, lc.phi * lc.theta * C / (4.0 * gamma_prev)), lc.gamma_max);
          lc.theta = lc.phi * lc.gamma / gamma_prev;                         // :178
        }
        __syncthreads();
        const double gamma = lc.gamma, phi = lc.phi;
        if (tid < 128) {
          double s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
          if (mine) {
            const T xb = (T)(((phi - 1.0) * xc + (double)aux_t) / phi);      // :179
            const T vi = (T)((double)xb - gamma * gc);
            const T xn = (T)prox_elem_val<T>(d.g, (double)vi, gamma, tid, 0.0, &s[1]);     // :181
            const double dd = (double)xn - xc;
            s[0] = dd * dd;                                                  // :182
            XT[tid] = xn;
            AUX[tid] = xb;
          }
          smom_sums(s, qs);
        }
        __syncthreads();
        if (tid == 0) {
          lc.norm_res = sqrt(qs[0][0] + qs[1][0]) / gamma;
          lc.g_raw = qs[0][1] + qs[1][1];
          lc.stop_after = lc.norm_res <= lc.tol ? 1 : 0;                     // :186 (the record of this iteration still needs f)
          lc.phase = 1;
        }
      }
      if (tid == 0) {
        lc.frozen = go != 0 ? 0 : 1;
        *d.c = lc;
      }
    }
  } else {
    // ---- fixed Nesterov --------------------------------------------------------------------------------
    const double gamma = lc.gamma;
    double xs = (double)x_t, xps = (double)xp_t;      // x, x_prev the next z is formed from
    if constexpr (MODE == 0) {
      if (tid == 0) go = lc.maxit > 0 ? 1 : 0;
      __syncthreads();
    } else {
      if (phase == 0) {                               // XT = z was evaluated: x+ = prox(z - gamma grad_z), norm_res (:130-133)
        double xn = 0.0;
        if (tid < 128) {
          double s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
          if (mine) {
            const double zi = (double)xt_t;
            const T vi = (T)(zi - gamma * gj);
            const T xq = (T)prox_elem_val<T>(d.g, (double)vi, gamma, tid, 0.0, &s[1]);   // :132
            xn = (double)xq;
            const double dd = xn - zi;
            s[0] = dd * dd;                                                  // :133
            XP[tid] = x_t;                                                   // :131
            X[tid] = xq;
            if (d.want_obj != 0) XT[tid] = xq;                               // next tick: f(x+) for the record
          }
          smom_sums(s, qs);
        }
        __syncthreads();
        if (tid == 0) {
          lc.norm_res = sqrt(qs[0][0] + qs[1][0]) / gamma;
          lc.g_raw = qs[0][1] + qs[1][1];
          lc.stop_after = lc.norm_res <= lc.tol ? 1 : 0;                     // :137
          if (d.want_obj != 0) {
            lc.phase = 1;
            go = 0;
          } else {
            lc.it += 1;
            record(NAN);                                                     // (no record sink: only the termination test)
            go = (lc.done == 0 && lc.it < lc.maxit) ? 1 : 0;
          }
        }
        xps = (double)x_t;
        xs = xn;
        __syncthreads();
      } else {                                        // XT = x+ was evaluated: the objective (:134-136), record, termination
        if (tid == 0) {
          lc.it += 1;
          record(f_pt);
          lc.phase = 0;
          go = (lc.done == 0 && lc.it < lc.maxit) ? 1 : 0;
        }
        __syncthreads();
      }
    }
    const bool waiting = MODE == 2 && lc.phase == 1;  // the objective tick comes first
    if (go != 0) {
      if (tid == 0) {                                 // theta, beta of the next iteration (:121-128)
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
      if (mine) {
        const T zn = (T)(xs + beta_s * (xs - xps));                          // :129
        XT[tid] = zn;
        AUX[tid] = zn;
      }
    }
    if (tid == 0) {
      lc.frozen = (go != 0 || waiting) ? 0 : 1;
      *d.c = lc;
    }
  }
}

// ---------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------
bool skinny_mom_eligible(const apx_problem* p) { return skinny_rows_eligible(p); }

template <typename T, int FK, int MODE>
static void smom_tick_t(const SkinnyRowsPlan& pl, const SmomDev& d, bool rows, hipStream_t q) {
  if (rows) skinny_rows_launch<T, FK>(pl, d.XT, &d.c->frozen, q);
  hipLaunchKernelGGL((k_smom_close<T, FK, MODE>), dim3(1), dim3(kSmomBlock), 0, q, d, rows ? pl.B : 0);
}
template <int MODE>
static void smom_tick(const SkinnyRowsPlan& pl, const SmomDev& d, int32_t dtype, int fk, bool rows, hipStream_t q) {
  const bool ls = fk == APX_F_LEAST_SQUARES;
  if (dtype == APX_F32) {
    if (ls) smom_tick_t<float, APX_F_LEAST_SQUARES, MODE>(pl, d, rows, q);
    else smom_tick_t<float, APX_F_LOGISTIC, MODE>(pl, d, rows, q);
  } else {
    if (ls) smom_tick_t<double, APX_F_LEAST_SQUARES, MODE>(pl, d, rows, q);
    else smom_tick_t<double, APX_F_LOGISTIC, MODE>(pl, d, rows, q);
  }
}

// method-specific parameters arrive in `c0` (gamma, theta, ... filled by the caller), as for small_mom_run
int32_t skinny_mom_run(apx_problem* p, int method, const MomCtrl& c0, const void* x1, const void* x0, int64_t maxit,
                       const apx_solve_opts* opts, void* x_out, apx_result* res) {
  apx_ctx* ctx = p->ctx;
  const size_t es = p->es();
  const size_t nb = (size_t)p->n_pad * es, nd = (size_t)p->n_pad * sizeof(double);
  DBuf bxt, bx, bxp, baux, bg, bgp, bpart, bc, brecs;
  SkinnyRowsPlan pl;
  skinny_rows_plan(p->A_f, p->n, p->b.p, &pl);
  APX_TRY(bpart.alloc_zero((size_t)pl.B * pl.dev.ldpart * sizeof(double), ctx->stream));
  pl.dev.part = (double*)bpart.p;
  for (DBuf* b : {&bxt, &bx, &bxp, &baux}) APX_TRY(b->alloc_zero(nb, ctx->stream));
  APX_TRY(bg.alloc_zero(nd, ctx->stream));
  APX_TRY(bgp.alloc_zero(nd, ctx->stream));
  APX_TRY(bc.alloc_zero(sizeof(MomCtrl), ctx->stream));
  apx_record* host_recs = opts ? opts->records : nullptr;
  const int64_t rec_cap = (opts && opts->records) ? opts->record_cap : 0;
  if (rec_cap > 0) APX_TRY(brecs.alloc_zero((size_t)rec_cap * sizeof(apx_record), ctx->stream));
  SmomDev d{};
  d.ncols = pl.dev.ncols;
  d.n = pl.dev.n;
  d.npow = pl.npow;
  d.method = method;
  d.ldpart = pl.dev.ldpart;
  d.part = pl.dev.part;
  d.XT = bxt.p;
  d.X = bx.p;
  d.XP = bxp.p;
  d.AUX = baux.p;
  d.G = (double*)bg.p;
  d.GP = (double*)bgp.p;
  d.c = (MomCtrl*)bc.p;
  d.g = p->gd;
  d.recs = (apx_record*)brecs.p;
  d.want_obj = rec_cap > 0 ? 1 : 0;

  MomCtrl c = c0;
  c.n_mean = p->n_mean;
  c.maxit = maxit;
  c.rec_cap = rec_cap;
  c.norm_res = c.objective = c.f_new = std::numeric_limits<double>::quiet_NaN();
  c.frozen = maxit > 0 ? 0 : 1;
  APX_TRY(h2d(ctx, bc.p, &c, sizeof(c)));
  hipStream_t q = ctx->stream;
  const bool ag = method == MOM_AGRAAL;
  APX_TRY(h2d(ctx, bx.p, x1, (size_t)p->n * es));                  // x = x1 (Nesterov: x = x_prev = x0, passed as x1)
  APX_TRY(h2d(ctx, bxp.p, ag ? x0 : x1, (size_t)p->n * es));
  if (ag) {
    APX_TRY(h2d(ctx, baux.p, x1, (size_t)p->n * es));              // :165  x_bar = x1
    APX_TRY(h2d(ctx, bxt.p, x0, (size_t)p->n * es));               // first pass: grad f(x_prev)
  }
  if (maxit > 0) {
    // aGRAAL: grad f(x_prev) (:167) ; then the pass at x1 (:166) is the first regular tick.  Nesterov: z of iteration 1.
    smom_tick<0>(pl, d, p->dtype, p->f_kind, ag, q);
    if (ag) smom_tick<2>(pl, d, p->dtype, p->f_kind, true, q);
  }
  APX_HIP(hipGetLastError());
  const int per_it = (!ag && d.want_obj != 0) ? 2 : 1;             // Nesterov with records: a second pass for f(x+)
  while (maxit > 0) {
    const int64_t missing = (maxit - c.it) * per_it;
    const int chunk = (int)std::min<int64_t>(64, std::max<int64_t>(missing, 1));
    for (int k = 0; k < chunk; ++k) smom_tick<2>(pl, d, p->dtype, p->f_kind, true, q);
    APX_HIP(hipGetLastError());
    APX_TRY(d2h(ctx, &c, bc.p, sizeof(c)));
    if (c.done != 0 || c.it >= maxit) break;
  }

  const int64_t k = c.it;
  APX_TRY(d2h(ctx, x_out, bx.p, (size_t)p->n * es));
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
    const int64_t evals = ag ? 2 + (conv ? k - 1 : k) : k;
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
