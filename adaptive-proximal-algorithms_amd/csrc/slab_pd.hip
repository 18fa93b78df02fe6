// slab_pd.hip — the linear-operator half of the primal-dual iteration (src/AdaProx.jl:335, :344-347, :358) in ONE pass
// over A: A_x = A x ; w = y + sigma ((1 + rho) A_x - rho A_x_prev) ; y = prox_{sigma h*}(w) ; dual_res = (w - y) / sigma
// - A_x ; At_y = A' y — for a separable h (the least-absolute-deviation clamp, experiments/least_absolute_deviation/
// runme.jl:39-41) with rows of at most 10 752 entries.  The generic path streams A twice per iteration (A x, then
// A' y); but y[i] depends on row i of A only, so a block that owns a slab of rows and keeps R of them at a time in
// registers (csrc/slab.hip's scheme: thread t holds the columns t, t + BS, ..., read from the transposed copy) can
// form A_x[i], update y[i] and add y[i] * A[i,:] to a private column accumulator from the SAME registers.  The B
// partial n-vectors and the dual-residual sums are closed by a second small kernel, which also runs the end of the
// iteration (norm_res, record, termination: :348-354).  No atomics: bit-reproducible.
#include <atomic>
#include <cstdlib>
#include <memory>

#include "skinny_dev.h"
#include "solver_internal.h"
#include "team_dev.h"
#include "team_rows.h"
#include "vecops_dev.h"

namespace apx {

struct SlabPdDev {
  const void* AT;         // A' column-major: row i of A at AT + i * ldT, contiguous
  int64_t ldT;
  int32_t m, n, rows_per_block, ncols;      // (ncols = n: the name csrc/team_rows.h reads)
  int32_t deal_stride, deal_log;            // k_multi_rows: the rows dealt in chunks (csrc/team_dev.h) ; k_slabpd_rows keeps its slabs
  void* Y;                // y (T), updated in place
  void* AXP;              // A_x_prev (T), updated in place (:360)
  // where a row's y+ and A x go: y and A_x_prev themselves in the loop above ; buffers of their own in a linesearch TRIAL
  // (adaptive_linesearch_primal_dual, src/AdaProx.jl:516-533: y and A_x_prev must survive a rejected trial), whose sigma and
  // rho are the host's (csrc/team_rows.h reads YN / AXN / trial through the row functions of this loop only)
  void* YN;
  void* AXN;
  double sigma_arg, rho_arg;
  int32_t trial;
  // a row shard of the operator (round 4): the launch's sums — A' y, sum dual_res^2, raw h(A_x) — are this rank's share ; the
  // closing kernel leaves the two scalars in `sums_out` and the end of the iteration to the kernel behind the all-reduce
  int32_t no_finish;
  double* sums_out;
  double* part;           // [blocks][ldpart]: n column partials of A' y, then sum dual_res^2, raw h(A_x)
  int64_t ldpart;
  Ctrl* c;
  ProxDev h;
};

struct SlabPdState {
  DBuf part;
  SlabPdDev dev{};
  int B = 0, cpt = 0, bs = 0, rows = 0;
  bool nt = true;         // non-temporal loads of A (a matrix beyond the Infinity Cache)
  // rows of more than 10 752 Float64 entries: k_team_rows (csrc/team_rows.h) — one workgroup, or teams of 2 ... 16
  TeamPlan plan;          // kv = 0: not k_team_rows ; members, vectors per thread, the three-slot shape, compute waves, an experimental slot shape
  int multi_waves = 0;    // k_multi_rows: 0 = by fit, 10 / 11 compute waves forced (APX_OPT_MULTI_WAVES)
  TeamHost team;          // ring, tickets, the error word and its diagnostic record, wait statistics (csrc/team_rows.h)
  std::string kname;      // the row kernel as launched (apx_ctx_kernel_name)
};
void slab_pd_destroy(SlabPdState* st) { delete st; }

// H rows of A into registers (csrc/slab.hip's slab_load) together with the per-row state of the rows this thread may
// finish: y, A_x_prev and the shift of h — every thread, unconverted, so that nothing waits for the loads here
template <typename T, int C, int H, bool NT>
__device__ __forceinline__ void slabpd_load(T (&a)[H][C], T& y_in, T& axp_in, T& b_in, const SlabPdDev& d,
                                            const unsigned (&off)[C], int r0, int row_end, int tid) {
  const T* __restrict__ AT = (const T*)d.AT;
#pragma unroll
  for (int r = 0; r < H; ++r) {
    const int ri = __builtin_amdgcn_readfirstlane(min(r0 + r, row_end - 1));
    const T* __restrict__ row = AT + (int64_t)ri * d.ldT;
#pragma unroll
    for (int k = 0; k < C; ++k)                  // scalar row base + 32-bit BYTE offset: one address register per load
      a[r][k] = NT ? __builtin_nontemporal_load((const T*)((const char*)row + off[k]))   // (read once: past the caches,
                   : *(const T*)((const char*)row + off[k]);                             //  unless A fits the Infinity Cache)
  }
  const int ic = min(r0 + min(tid, H - 1), row_end - 1);
  y_in = ((const T*)d.Y)[ic];
  axp_in = ((const T*)d.AXP)[ic];
  b_in = d.h.has_shift ? ((const T*)d.h.shift)[ic] : (T)0;
}

// the H rows r0 .. r0 + H - 1 held in a: A_x, the dual step of each (thread r finishes row r0 + r: :344-347, :360) and
// y[i] * A[i,:] into the column accumulator ; two barriers
template <typename T, int C, int H, int BS, bool TIGHT>
__device__ __forceinline__ void slabpd_group(const T (&a)[H][C], T y_in, T axp_in, T b_in, const T (&xr)[C], double (&acc)[C],
                                             const SlabPdDev& d, double sigma, double rho, double& e0, double& e1, int r0,
                                             int row_end, double (*wsum)[(H < 2 ? 2 : H)], double* y_s, int tid) {
  constexpr int RP = H < 2 ? 2 : H;                // (the multi-value wave sum wants at least two)
  const int lane = tid & 63, wv = tid >> 6;
  T* __restrict__ Y = (T*)d.Y;
  T* __restrict__ AXP = (T*)d.AXP;
  double p[RP];
#pragma unroll
  for (int r = 0; r < RP; ++r) p[r] = 0.0;
#pragma unroll
  for (int k = 0; k < C; ++k) {
    T xv = xr[k];
    if constexpr (sizeof(T) == 4) asm volatile("" : "+v"(xv));        // (x stays in T: see csrc/slab.hip)
    const double xd = (double)xv;
#pragma unroll
    for (int r = 0; r < H; ++r) p[r] = fma((double)a[r][k], xd, p[r]);
    if constexpr (TIGHT && sizeof(T) == 4) __builtin_amdgcn_sched_barrier(0);
  }
  wave_sum_multi<RP>(p);                           // lane l: the wave's sum for row l / (64 / RP)
  if (lane % (64 / RP) == 0) wsum[wv][lane / (64 / RP)] = p[0];
  __syncthreads();
  if (tid < H) {                                   // thread r finishes row r0 + r
    double dot = wsum[0][tid];
#pragma unroll
    for (int w = 1; w < BS / 64; ++w) dot += wsum[w][tid];
    const int i = r0 + tid;
    double yn = 0.0;
    if (i < row_end) {
      const double ax = (double)(T)dot;                                                   // :335, rounded to T like every stored A x
      const double b = (double)b_in;
      const double w = (double)y_in + sigma * ((1.0 + rho) * ax - rho * (double)axp_in);  // :344
      const T ynt = (T)prox_conj_elem_s<T>(d.h, w, sigma, i, 0.0, b);                     // :345
      yn = (double)ynt;
      const double dr = (w - yn) / sigma - ax;                                            // :347
      e0 += dr * dr;
      e1 += gval_elem_s<T>(d.h, ax, i, b);
      Y[i] = ynt;
      AXP[i] = (T)ax;                                                                     // :360
    }
    y_s[tid] = yn;
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < H; ++r) {
    const double yy = y_s[r];
#pragma unroll
    for (int k = 0; k < C; ++k) {
      T av = a[r][k];
      if constexpr (sizeof(T) == 4) asm volatile("" : "+v"(av));      // (convert again: see csrc/slab.hip)
      acc[k] = fma((double)av, yy, acc[k]);
    }
    if constexpr (TIGHT && sizeof(T) == 4) __builtin_amdgcn_sched_barrier(0);
  }
  if constexpr (TIGHT) {                           // (the sums complete HERE, not sunk past the next loads: csrc/slab.hip)
#pragma unroll
    for (int k = 0; k < C; ++k) asm volatile("" : "+v"(acc[k]));
  }
}

// PIPE: the R rows as two groups of R / 2 that alternate, one computing while the other's loads arrive (csrc/slab.hip)
template <typename T, int C, int R, int BS, bool PIPE, bool NT>
__global__ __launch_bounds__(BS) void k_slabpd_rows(SlabPdDev d, const void* xv) {
  constexpr int H = PIPE ? R / 2 : R;
  static_assert(!PIPE || R % 2 == 0, "two groups");
  constexpr int RP = H < 2 ? 2 : H;
  __shared__ double wsum[BS / 64][RP];
  __shared__ double y_s[RP];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (d.c->done != 0) return;                      // written by an earlier launch: block-uniform
  const double sigma = d.c->sigma, rho = d.c->rho; // this iteration's, set by the rule kernel before this launch
  const T* __restrict__ X = (const T*)xv;
  T xr[C];
  double acc[C];
  unsigned off[C];
#pragma unroll
  for (int k = 0; k < C; ++k) {
    const int j = tid + k * BS;
    off[k] = (unsigned)min(j, d.n - 1) * (unsigned)sizeof(T);         // in bytes ; clamped: a valid address, weight 0
    xr[k] = j < d.n ? X[j] : (T)0;
    acc[k] = 0.0;
  }
  double e0 = 0.0, e1 = 0.0;                       // sum dual_res^2, raw h(A_x) of the rows this thread finishes
  const int row_begin = blockIdx.x * d.rows_per_block;
  const int row_end = min(row_begin + d.rows_per_block, d.m);
  if constexpr (!PIPE) {
    for (int r0 = row_begin; r0 < row_end; r0 += R) {
      T a[H][C];
      T y_in, axp_in, b_in;
      slabpd_load<T, C, H, NT>(a, y_in, axp_in, b_in, d, off, r0, row_end, tid);
      slabpd_group<T, C, H, BS, false>(a, y_in, axp_in, b_in, xr, acc, d, sigma, rho, e0, e1, r0, row_end, wsum, y_s, tid);
    }
  } else {
    T a0[H][C], a1[H][C];
    T y0, y1, p0, p1, b0, b1;
    __builtin_amdgcn_sched_barrier(0);
    slabpd_load<T, C, H, NT>(a0, y0, p0, b0, d, off, row_begin, row_end, tid);
    __builtin_amdgcn_sched_barrier(0);             // (in this order: the waits of the loop count the loads issued after a group's)
    slabpd_load<T, C, H, NT>(a1, y1, p1, b1, d, off, row_begin + H, row_end, tid);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
    for (int r0 = row_begin; r0 < row_end; r0 += R) {
      slabpd_group<T, C, H, BS, true>(a0, y0, p0, b0, xr, acc, d, sigma, rho, e0, e1, r0, row_end, wsum, y_s, tid);
      __builtin_amdgcn_sched_barrier(0);
      slabpd_load<T, C, H, NT>(a0, y0, p0, b0, d, off, r0 + R, row_end, tid);
      __builtin_amdgcn_sched_barrier(0);
      slabpd_group<T, C, H, BS, true>(a1, y1, p1, b1, xr, acc, d, sigma, rho, e0, e1, r0 + H, row_end, wsum, y_s, tid);
      __builtin_amdgcn_sched_barrier(0);
      slabpd_load<T, C, H, NT>(a1, y1, p1, b1, d, off, r0 + R + H, row_end, tid);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  double* __restrict__ out = d.part + (int64_t)blockIdx.x * d.ldpart;
#pragma unroll
  for (int k = 0; k < C; ++k) {
    const int j = tid + k * BS;
    if (j < d.n) out[j] = acc[k];
  }
  if (wv == 0) {                                   // the rows were finished by threads 0 .. H-1 (one wave)
    double ex[2] = {tid < H ? e0 : 0.0, tid < H ? e1 : 0.0};
    wave_sum_multi<2>(ex);                         // lane l: ex[l / 32]
    if ((lane & 31) == 0) out[d.n + (lane >> 5)] = ex[0];
  }
}

// (rows too long for one workgroup's registers: k_team_rows<kTeamFkPd, ...>, csrc/team_rows.h — its communication wave takes
//  the dual step of a row ONCE, where no row of A is held, and the compute waves multiply the row by y+[i])

// A'y from the block partials (64 columns per block, sixteen loads in flight, fixed order) ; the dual residual and the
// end of the iteration (:347-354) by block 0
constexpr int kSlabPdCloseCols = 64;
constexpr int kSlabPdCloseBlock = 1024;            // (sixteen threads per column: all of a thread's loads in flight at once — csrc/slab.hip)
// L2 (h = NormL2, round 4): the launch left A' z and sum z^2 (csrc/team_rows.h team_pd_dual) ; EVERY block adds up the B
// partial sums of z^2 — in one fixed order, so all of them scale by the same factor —, A' y+ = sigma (1 - sc) A' z, and the
// end of the iteration waits for k_dual_post, which forms y+ and the dual residual from the norm left in Ctrl::h_pre.
template <typename T, bool L2 = false>
__global__ __launch_bounds__(kSlabPdCloseBlock) void k_slabpd_close(SlabPdDev d, int B, T* __restrict__ aty, apx_record* recs, int64_t rec_cap) {
  constexpr int S = kSlabPdCloseBlock / kSlabPdCloseCols;
  __shared__ double fs[2];
  __shared__ double sub[S][kSlabPdCloseCols];
  if (d.c->done != 0) return;
  const int tid = threadIdx.x;
  const int c = tid % kSlabPdCloseCols, sl = tid / kSlabPdCloseCols;
  const int j = blockIdx.x * kSlabPdCloseCols + c;
  const double part_sum = j < d.n ? skinny_strided_sum(d.part, d.ldpart, j, sl, S, B) : 0.0;
  if ((L2 || blockIdx.x == 0) && tid < 128) {
    const int which = tid >> 6, lane = tid & 63;
    double s = 0.0;
    for (int b = lane; b < B; b += 64) s += d.part[(int64_t)b * d.ldpart + d.n + which];
    s = wave_sum(s);
    if (lane == 0) fs[which] = s;
  }
  sub[sl][c] = part_sum;
  __syncthreads();
  if (sl == 0 && j < d.n) {
    double g = sub[0][c];
#pragma unroll
    for (int k = 1; k < S; ++k) g += sub[k][c];
    if constexpr (L2) {
      if (d.no_finish == 0) {      // (a row shard: the norm is over ALL ranks' rows — scaled behind the all-reduce, vo_l2_scale)
        const double sigma = d.trial ? d.sigma_arg : d.c->sigma, aux = sqrt(fs[0]);
        const double sc = fmax(0.0, 1.0 - d.h.lambda * (1.0 / sigma) / aux);               // csrc/vecops_dev.h prox_core, gamma = 1 / sigma
        g *= sigma * (1.0 - sc);
      }
    }
    aty[j] = (T)g;                                                                         // :358  At_y = A' * y
  }
  if (blockIdx.x == 0 && tid == 0 && d.sums_out != nullptr) {
    d.sums_out[0] = fs[0];
    d.sums_out[1] = fs[1];
  }
  if (blockIdx.x == 0 && tid == 0 && d.trial == 0 && d.no_finish == 0) {      // (a linesearch trial: the host's loop closes the iteration)
    if constexpr (L2) {
      d.c->h_pre = sqrt(fs[0]);                                                            // ||z||: what k_dual_post's conjugate prox scales by
    } else {
      const double ndr = sqrt(fs[0]);
      d.c->ndr2 = ndr * ndr;
      d.c->h_Ax = gval_final(d.h, fs[1]);
      finish_iteration(d.c, recs, rec_cap);                                                // :348-354
    }
  }
}
template <typename T>
static void slabpd_close_launch(const SlabPdDev& d, int nparts, T* aty, apx_record* recs, int64_t rec_cap, hipStream_t q) {
  const dim3 grid((unsigned)((d.n + kSlabPdCloseCols - 1) / kSlabPdCloseCols));
  if (d.h.kind == APX_G_L2) hipLaunchKernelGGL((k_slabpd_close<T, true>), grid, dim3(kSlabPdCloseBlock), 0, q, d, nparts, aty, recs, rec_cap);
  else hipLaunchKernelGGL((k_slabpd_close<T, false>), grid, dim3(kSlabPdCloseBlock), 0, q, d, nparts, aty, recs, rec_cap);
}

// ---------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------
struct SlabPdShape {
  int bs, cpt, rows;
};
static SlabPdShape slab_pd_shape(const apx_problem* p) {
  const int64_t cols = p->A_lin->n;
  const bool f32 = p->dtype == APX_F32;
  // Float64 rows of up to 5632 entries: k_multi_rows (csrc/team_rows.h ; the A/B switch APX_OPT_MULTI_ROWS_MASK, bit 2: csrc/slab.hip
  // slab_shape) — against the shapes below, TB/s of the bytes read at 6 GB: 600: 4.0 -> 5.7, 1000: 5.8 -> 6.3, 2100: 4.2 -> 6.3, 5632: 6.0 -> 6.9
  const int multi = p->ctx->multi_mask;
  if (!f32 && cols <= 2 * 4 * 704 && (multi & 4)) return {768, 27, 1};
  if (f32 && cols <= 4 * 4 * 704 && (multi & 4) && (multi & 8)) return {768, 27, 1};      // (Float32, round 3: bit 3 of the mask — csrc/slab.hip slab_shape)
  if (cols <= 1024) return {1024, 1, 4};           // (one column per thread, sixteen rows per step — 32 spill here: csrc/slab.hip slab_shape)
  if (cols <= 2 * 768) return {768, 2, 4};         // (1025 ... 1536 entries: csrc/slab.hip slab_shape)
  if (cols <= 2 * 1024) return {1024, 2, 4};
  if (cols <= 3 * 1024) return {1024, 3, 4};       // (three / four columns per thread: csrc/slab.hip slab_shape, measured)
  if (cols <= 4 * 1024) return {1024, 4, 4};
  if (cols <= 5 * 1024) return {1024, 5, 4};
  // (between five columns of 1024 threads and fourteen of 768: seven and ten — a row of 5200 entries kept 37 % of the
  //  fourteen-column shape's lanes busy, 3.6 TB/s ; tools/shape_sweep.py)
  if (cols <= 7 * 768) return {768, 7, 4};
  // Float64 rows beyond that: k_team_rows (csrc/team_rows.h), whose slice width adapts to the row — one workgroup with x in LDS
  // up to 11 264 columns (6.2 - 6.9 TB/s throughout ; the ten / fourteen-column shapes below leave a sawtooth — 5400: 4.9,
  // 7700: 5.0 — and stay for Float32), teams of 2 ... 16 workgroups, a slice of every row each, up to 180 224 ; the teams
  // need a part of 8 XCDs x 32 CUs
  // (one workgroup up to 11 264 columns hands nothing over: it stays when the team kernels proper are switched off)
  if (!f32 && (cols <= 2 * kTeamSingleKv * 704 || (cols <= kTeamMaxCols && p->ctx->num_cu == 256 && p->ctx->teams && !p->teams_dead()))) return {768, 28, 1};
  if (cols <= 10 * 768) return {768, 10, f32 ? 4 : 2};
  if (cols <= 14 * 768) return {768, 14, f32 ? 4 : 2};
  // Float32 rows beyond one workgroup (round 5): the teams of csrc/team_rows.h in their Float32 form — two to five vectors per
  // thread, four slots, teams of 2 ... 16 up to 225 280 entries — as f's gradient has had them since round 3 (csrc/slab.hip)
  if (f32 && cols <= kTeamMaxColsF32 && p->ctx->num_cu == 256 && p->ctx->teams && !p->teams_dead() && (multi & 8)) return {768, 28, 1};
  // (C5's row length, 20 000 fp64, is out of reach of ONE workgroup's row kernel: x, the accumulator and one row are 480 KB of
  //  a CU's 512 KB of VGPRs — 1024 threads x 20 columns x 1 row spilled 432 B and ran at 3.4 ms against 2.36 ms for the two passes)
  return {0, 0, 0};
}

static bool slab_pd_shape_ok(const apx_problem* p) {
  if (!p->ctx->one_pass) return false;
  if (p->A_lin == nullptr) return false;
  // a row-sharded OPERATOR is as good as a whole one (round 4): the rows are local ; A' y and the two residual sums travel — for
  // the l2 ball A' z and sum z^2 first, the scaling and y+ behind that all-reduce, then the residual sums (three collectives, as
  // on the two passes).  A sharded f next to a replicated operator is nobody's use and keeps the generic kernels
  if (p->sharded() && !(p->lin_sharded && !p->f_sharded)) return false;
  if (p->A_lin->sparse || p->ctx->skip_zero_cols) return false;
  // (h = NormL2: its conjugate prox needs a norm over all rows first, but is linear in the rows once the norm is known — the
  //  kernels of csrc/team_rows.h leave A' z and sum z^2 and the closing kernel scales, round 4 ; the first-generation row kernels
  //  of this file do not)
  if (p->hd.kind == APX_G_L2 && (slab_pd_shape(p).cpt != 27 && slab_pd_shape(p).cpt != 28)) return false;
  if ((double)p->A_lin->m * (double)p->A_lin->n * (double)p->es() < (double)p->ctx->one_pass_min_bytes) return false;   // (see slab.hip)
  // (very short rows: with two columns per thread the two passes were faster below 512 entries — 130: 7.0 against 8.8 ms at
  //  6 GB ; with one column per thread and sixteen rows per step: 130: 5.5, 200: 3.6, 450: 1.9 ms against 7.0, 4.9, 2.8)
  if (p->A_lin->n < 128) return false;
  return p->A_lin->m >= p->ctx->one_pass_min_rows && p->A_lin->m < ((int64_t)1 << 30) && slab_pd_shape(p).bs != 0;
}

// (reads only the transposed copy, which a matrix of up to 16 GB is given here: see slab_eligible)
bool slab_pd_eligible(const apx_problem* p) {
  if (!slab_pd_shape_ok(p)) return false;
  const apx_matrix* A = p->A_lin;
  if (A->dT == nullptr && p->ctx->auto_transpose && (double)A->ld * (double)A->n_pad * (double)p->es() <= 16.0 * 1024 * 1024 * 1024) {
    if (apx_matrix_build_transpose(const_cast<apx_matrix*>(A)) != APX_OK) (void)hipGetLastError();
  }
  return A->dT != nullptr;
}

// On a row shard the ranks VOTE on the path after the set-up (csrc/solver.hip, AdaPDM+, Malitsky-Pock): a rank whose set-up fails
// — team_alloc, a partial buffer, the pinned diagnostics — must reach that vote like everybody else, or its peers wait in the
// all-reduce forever.  Here the failure becomes "this rank cannot" (no state left behind): every rank then takes the
// two passes.  An unsharded problem gets the error.
int32_t slab_pd_setup_voted(apx_solver* s) {
  const int32_t rc = slab_pd_setup(s);
  if (rc == APX_OK || !s->p->lin_sharded) return rc;
  (void)hipGetLastError();
  if (s->slab_pd != nullptr) {
    slab_pd_destroy(s->slab_pd);
    s->slab_pd = nullptr;
  }
  return APX_OK;
}

int32_t slab_pd_setup(apx_solver* s) {
  apx_problem* p = s->p;
  apx_ctx* ctx = p->ctx;
  if (ctx->fault_setup & 1) return fail(APX_ERR_HIP, "slab_pd_setup: allocation failed (APX_OPT_FAULT_SETUP, a test hook)");
  const apx_matrix* A = p->A_lin;
  std::unique_ptr<SlabPdState> st(new SlabPdState());
  const SlabPdShape sh = slab_pd_shape(p);
  st->cpt = sh.cpt;
  st->bs = sh.bs;
  st->rows = sh.rows;
  st->multi_waves = ctx->multi_waves;
  int64_t B = (int64_t)ctx->num_cu * (st->cpt <= 2 ? 2 : 1);
  int64_t rpb = (A->m + B - 1) / B;                // dealt evenly, nothing rounded up: a block's last step may be partial
  B = (A->m + rpb - 1) / rpb;                      // (csrc/slab.hip: rounding to 16 left CUs without a block)
  if (st->cpt == 28) {                             // k_team_rows: slabs = teams, ALL P x B workgroups launched
    const bool f32 = p->dtype == APX_F32;
    const TeamPlan pl = f32 ? team_plan(A->n, TeamTune{0, ctx->multi_waves, 0, true}, 4)
                            : team_plan(A->n, TeamTune{ctx->team_members, ctx->multi_waves, ctx->team_slots, true});
    if (pl.P == 0) return APX_OK;                    // (no plan for this length: the generic kernels)
    st->plan = pl;
    B = ctx->num_cu / pl.P;
    rpb = (A->m + B - 1) / B;
    const int per_cu = pl.P > 1 ? (f32 ? team_blocks_per_cu<kTeamFkPd, SlabPdDev, float>(pl, true) : team_blocks_per_cu<kTeamFkPd, SlabPdDev>(pl, true)) : 1;
    if (per_cu < 1) return APX_OK;                   // (it does not fit a CU: the generic kernels)
    APX_TRY(team_alloc(ctx, &st->team, (int)B, pl.P, rpb, A->n * (int64_t)dtype_size(p->dtype)));
  }
  st->B = (int)B;
  SlabPdDev d{};
  d.AT = A->dT;
  d.ldT = A->ldT;
  d.m = (int32_t)A->m;
  d.n = (int32_t)p->n;
  d.ncols = d.n;
  d.rows_per_block = (int32_t)rpb;
  d.deal_stride = 1;
  d.deal_log = 0;
  if (st->cpt == 27 && ctx->row_deal != 0) {      // (k_multi_rows ; the team kernels carry their deal in TeamDev)
    d.deal_stride = (int32_t)B;
    if (ctx->row_deal < 0) d.deal_log = deal_chunk_log(A->n * (int64_t)dtype_size(p->dtype), rpb);
    for (int r = ctx->row_deal; r > 1; r >>= 1) d.deal_log += 1;
  }
  d.Y = s->y.p;
  d.AXP = s->axprev.p;
  d.YN = d.Y;
  d.AXN = d.AXP;
  d.trial = 0;
  d.no_finish = p->lin_sharded ? 1 : 0;
  d.sums_out = p->lin_sharded ? ctx->scal + 13 : nullptr;
  d.ldpart = round_up(d.n + 2, 8);
  const int64_t nparts = B * (st->cpt == 27 ? multi_groups(A->n, p->dtype == APX_F32 ? 4 : 2) : 1);      // (k_multi_rows on very short rows: csrc/team_rows.h)
  APX_TRY(st->part.alloc_zero((size_t)nparts * d.ldpart * sizeof(double), ctx->stream));
  d.part = (double*)st->part.p;
  d.c = p->ctrl();
  d.h = p->hd;
  st->nt = (int64_t)A->m * A->n * (int64_t)dtype_size(p->dtype) >= ((int64_t)384 << 20);     // (csrc/slab.hip: measured)
  st->dev = d;
  s->slab_pd = st.release();
  return APX_OK;
}

template <typename T>
static bool slab_pd_launch_t(SlabPdState* st, bool cooperative, const SlabPdDev& d, const void* x, void* aty, apx_record* recs,
                             int64_t rec_cap, hipStream_t q) {
  const dim3 grid((unsigned)st->B);
  constexpr bool f32 = sizeof(T) == 4;
  if (st->cpt == 27) {                             // k_multi_rows: H rows per step, the dual steps of a step in the row wave's lanes
    if (st->nt) multi_launch<T, kTeamFkPd, true, SlabPdDev>(d.n, st->B, st->multi_waves, d, x, q);
    else multi_launch<T, kTeamFkPd, false, SlabPdDev>(d.n, st->B, st->multi_waves, d, x, q);
    slabpd_close_launch<T>(d, st->B * multi_groups(d.n, 16 / (int)sizeof(T)), (T*)aty, recs, rec_cap, q);
    return true;
  }
  if (st->plan.kv != 0) {
    const bool ok = team_launch<kTeamFkPd, SlabPdDev, T>(st->plan, st->B, st->nt, cooperative, d, st->team.dev, x, q);
    if (!ok) return false;                         // (a refused launch: NOTHING runs — not the closing kernel on the last launch's partials either)
    slabpd_close_launch<T>(d, st->B, (T*)aty, recs, rec_cap, q);
    return true;
  }
  // two row groups in flight only where it measured faster here — Float32 at two columns per thread (400 000 x 2000: 0.572 ->
  // 0.552 ms) ; with the per-row state of y and A_x_prev on top of the rows, the other shapes lose (200 000 x 5000 fp64
  // 1.31 -> 1.38, 100 000 x 10 000 fp32 0.68 -> 0.71)
  // (from ~16 steps per block on, as in csrc/slab.hip ; non-temporal loads for a matrix beyond the Infinity Cache: st->nt)
  const bool pipe = f32 && d.rows_per_block / 16 >= 16;
  if (st->nt) {
    if (st->cpt == 1) hipLaunchKernelGGL((k_slabpd_rows<T, 1, 16, 1024, false, true>), grid, dim3(1024), 0, q, d, x);
    else if (st->cpt == 2 && st->bs == 768) hipLaunchKernelGGL((k_slabpd_rows<T, 2, 16, 768, false, true>), grid, dim3(768), 0, q, d, x);
    else if (st->cpt == 2 && pipe) hipLaunchKernelGGL((k_slabpd_rows<T, 2, 16, 1024, f32, true>), grid, dim3(1024), 0, q, d, x);
    else if (st->cpt == 2) hipLaunchKernelGGL((k_slabpd_rows<T, 2, 16, 1024, false, true>), grid, dim3(1024), 0, q, d, x);
    else if (st->cpt == 3) hipLaunchKernelGGL((k_slabpd_rows<T, 3, 4, 1024, false, true>), grid, dim3(1024), 0, q, d, x);
    else if (st->cpt == 4) hipLaunchKernelGGL((k_slabpd_rows<T, 4, 4, 1024, false, true>), grid, dim3(1024), 0, q, d, x);
    else if (st->cpt == 5) hipLaunchKernelGGL((k_slabpd_rows<T, 5, 4, 1024, false, true>), grid, dim3(1024), 0, q, d, x);
    else if (st->cpt == 7) hipLaunchKernelGGL((k_slabpd_rows<T, 7, 4, 768, false, true>), grid, dim3(768), 0, q, d, x);
    else if (st->cpt == 10) hipLaunchKernelGGL((k_slabpd_rows<T, 10, (f32 ? 4 : 2), 768, false, true>), grid, dim3(768), 0, q, d, x);
    else hipLaunchKernelGGL((k_slabpd_rows<T, 14, (f32 ? 4 : 2), 768, false, true>), grid, dim3(768), 0, q, d, x);
  } else {
    if (st->cpt == 1) hipLaunchKernelGGL((k_slabpd_rows<T, 1, 16, 1024, false, false>), grid, dim3(1024), 0, q, d, x);
    else if (st->cpt == 2 && st->bs == 768) hipLaunchKernelGGL((k_slabpd_rows<T, 2, 16, 768, false, false>), grid, dim3(768), 0, q, d, x);
    else if (st->cpt == 2 && pipe) hipLaunchKernelGGL((k_slabpd_rows<T, 2, 16, 1024, f32, false>), grid, dim3(1024), 0, q, d, x);
    else if (st->cpt == 2) hipLaunchKernelGGL((k_slabpd_rows<T, 2, 16, 1024, false, false>), grid, dim3(1024), 0, q, d, x);
    else if (st->cpt == 3) hipLaunchKernelGGL((k_slabpd_rows<T, 3, 4, 1024, false, false>), grid, dim3(1024), 0, q, d, x);
    else if (st->cpt == 4) hipLaunchKernelGGL((k_slabpd_rows<T, 4, 4, 1024, false, false>), grid, dim3(1024), 0, q, d, x);
    else if (st->cpt == 5) hipLaunchKernelGGL((k_slabpd_rows<T, 5, 4, 1024, false, false>), grid, dim3(1024), 0, q, d, x);
    else if (st->cpt == 7) hipLaunchKernelGGL((k_slabpd_rows<T, 7, 4, 768, false, false>), grid, dim3(768), 0, q, d, x);
    else if (st->cpt == 10) hipLaunchKernelGGL((k_slabpd_rows<T, 10, (f32 ? 4 : 2), 768, false, false>), grid, dim3(768), 0, q, d, x);
    else hipLaunchKernelGGL((k_slabpd_rows<T, 14, (f32 ? 4 : 2), 768, false, false>), grid, dim3(768), 0, q, d, x);
  }
  slabpd_close_launch<T>(d, st->B, (T*)aty, recs, rec_cap, q);
  return true;
}

// what gemv_n(A, x) + the dual step + gemv_t(A, y) leave behind: y, A_x_prev updated, At_y in s->aty, the iteration closed.
// false in *done_here: the transposed copy is gone (A edited in place) — the caller runs the generic kernels
int32_t slab_pd_iteration(apx_solver* s, bool* done_here) {
  apx_problem* p = s->p;
  apx_ctx* ctx = p->ctx;
  SlabPdState* st = s->slab_pd;
  *done_here = false;
  if (p->A_lin->dT == nullptr) return APX_OK;
  SlabPdDev d = st->dev;
  d.AT = p->A_lin->dT;
  d.ldT = p->A_lin->ldT;
  if (st->kname.empty()) {
    char b[256];
    if (st->plan.kv != 0)
      snprintf(b, sizeof b, "k_team_rows<primal_dual, %d member%s%s, %d vectors per thread, %d compute waves, %d row slots, %d row%s of slack> + k_slabpd_close",
               st->plan.P, st->plan.P == 1 ? "" : "s", st->team.dev.interleave ? " (shares interleaved)" : "", std::max(st->plan.kv, st->plan.P == 1 ? 4 : 2), st->plan.waves,
               st->plan.alt ? st->plan.alt / 10 : ((st->plan.big || st->plan.P == 1) ? 3 : 4),
               st->plan.alt ? st->plan.alt % 10 : ((st->plan.big || st->plan.P == 1) ? 1 : 2), (st->plan.big || st->plan.P == 1) ? "" : "s");
    else if (st->cpt == 27)
      snprintf(b, sizeof b, "k_multi_rows<%s, primal_dual, %d row group(s)> + k_slabpd_close", p->dtype == APX_F32 ? "float" : "double",
               multi_groups(d.n, p->dtype == APX_F32 ? 4 : 2));
    else
      snprintf(b, sizeof b, "k_slabpd_rows<%s, %d columns per thread, %d threads> + k_slabpd_close", p->dtype == APX_F32 ? "float" : "double", st->cpt, st->bs);
    st->kname = b;
  }
  ctx->kname[APX_KSTAT_ONE_PASS] = st->kname;
  APX_TRY(prof_begin(ctx, APX_KSTAT_ONE_PASS, (double)p->A_lin->m * (double)p->A_lin->n * (double)p->es()));
  const bool launched =
      p->dtype == APX_F32 ? slab_pd_launch_t<float>(st, (ctx->team_cooperative && !p->sharded()), d, s->x.p, s->aty.p, (apx_record*)s->recs.p, s->rec_cap, ctx->stream)
                          : slab_pd_launch_t<double>(st, (ctx->team_cooperative && !p->sharded()), d, s->x.p, s->aty.p, (apx_record*)s->recs.p, s->rec_cap, ctx->stream);
  APX_TRY(prof_end(ctx));
  if (!launched) return team_launch_refused(p, "one-pass primal-dual kernel (teams)");
  const bool l2 = p->hd.kind == APX_G_L2;
  if (l2 && !p->lin_sharded)       // y+ and the dual residual from w (left in y's place) and ||z|| ; the end of the iteration (:345-354)
    APX_TRY(vo_dual_post(ctx, p->dtype, s->y.p, s->axprev.p, p->m, p->hd, p->ctrl(), (apx_record*)s->recs.p, s->rec_cap));
  if (p->lin_sharded) {            // A' y = sum_k A_k' y_k, the residual sums likewise ; the iteration closed from the global sums
    APX_TRY(problem_allreduce(p, s->aty.p, p->A_lin->n_pad, p->dtype));
    APX_TRY(problem_allreduce(p, ctx->scal + 13, 2, APX_F64));
    if (l2) {                      // so far A' z and sum z^2: the factor from the global norm, y+ and this rank's residual sums, summed again
      APX_TRY(vo_l2_scale(ctx, p->dtype, s->aty.p, p->n, ctx->scal + 13, p->hd, p->ctrl()));
      APX_TRY(vo_dual_post_sums(ctx, p->dtype, s->y.p, s->axprev.p, p->m, p->hd, p->ctrl(), ctx->scal + 13));
      APX_TRY(problem_allreduce(p, ctx->scal + 13, 2, APX_F64));
    }
    APX_TRY(vo_final_sums(ctx, ctx->scal + 13, p->hd, p->ctrl(), (apx_record*)s->recs.p, s->rec_cap));
  }
  APX_HIP(hipGetLastError());
  *done_here = true;
  return APX_OK;
}

// One linesearch trial of adaptive_linesearch_primal_dual (src/AdaProx.jl:502, :522-526) in one pass over A: A_x = A x into
// `axnew`, w and y_next = prox_{sigma h*}(w) row by row (into `ynext` ; for the l2 ball: w — the caller's vo_dual_trial forms
// y_next from A_x in the two passes' own arithmetic either way), A' y_next into `aty_next`.  y and A_x_prev are only read.
// Only the kernels of csrc/team_rows.h (k_multi_rows, k_team_rows): slab_pd_trial_ok.
bool slab_pd_trial_ok(const apx_solver* s) {
  return s->slab_pd != nullptr && (s->slab_pd->cpt == 27 || s->slab_pd->cpt == 28) && s->p->A_lin->dT != nullptr;
}
int32_t slab_pd_trial(apx_solver* s, const void* x, const void* y, const void* axprev, void* ynext, void* axnew, void* aty_next,
                      double sigma, double rho) {
  apx_problem* p = s->p;
  apx_ctx* ctx = p->ctx;
  SlabPdState* st = s->slab_pd;
  APX_REQUIRE(slab_pd_trial_ok(s), "internal: slab_pd_trial without its kernels");
  SlabPdDev d = st->dev;
  d.AT = p->A_lin->dT;
  d.ldT = p->A_lin->ldT;
  d.Y = const_cast<void*>(y);
  d.AXP = const_cast<void*>(axprev);
  d.YN = ynext;
  d.AXN = axnew;
  d.sigma_arg = sigma;
  d.rho_arg = rho;
  d.trial = 1;
  ctx->kname[APX_KSTAT_ONE_PASS] = st->cpt == 27 ? "k_multi_rows<primal_dual> (linesearch trial) + k_slabpd_close" : "k_team_rows<primal_dual> (linesearch trial) + k_slabpd_close";
  APX_TRY(prof_begin(ctx, APX_KSTAT_ONE_PASS, (double)p->A_lin->m * (double)p->A_lin->n * (double)p->es()));
  const bool launched = p->dtype == APX_F32 ? slab_pd_launch_t<float>(st, (ctx->team_cooperative && !p->sharded()), d, x, aty_next, nullptr, 0, ctx->stream)
                                            : slab_pd_launch_t<double>(st, (ctx->team_cooperative && !p->sharded()), d, x, aty_next, nullptr, 0, ctx->stream);
  APX_TRY(prof_end(ctx));
  if (!launched) return team_launch_refused(p, "one-pass primal-dual kernel (teams)");
  APX_HIP(hipGetLastError());
  return APX_OK;
}

bool slab_pd_uses_teams(const apx_solver* s) { return s->slab_pd != nullptr && s->slab_pd->plan.P > 1; }

// the error word of the team hand-off (0: fine), read with the convergence flag and with the result ; the message carries the
// diagnostic record of the wave that gave up (csrc/team_rows.h team_describe)
int32_t slab_pd_check(apx_solver* s) {
  SlabPdState* st = s->slab_pd;
  if (st == nullptr || st->plan.P <= 1) return APX_OK;
  int32_t err = 0;
  APX_TRY(team_read_err(s->p->ctx, &st->team, &err));
  double any = err != 0 ? 1.0 : 0.0;
  APX_TRY(ranks_sum(s->p, any, &any));             // (a row shard: every rank reads the word at the same points — csrc/slab.hip slab_check)
  if (any == 0.0) return APX_OK;
  s->p->team_failed = true;
  const std::string what = err != 0 ? team_describe(st->team.pinned->diag, "one-pass primal-dual kernel (teams)")
                                    : std::string("one-pass primal-dual kernel (teams): a team hand-off failed on another rank of this row-sharded "
                                                  "problem (its message has the details); the result is invalid and was discarded");
  note_incident(what);
  return fail(APX_ERR_HIP, what);
}

int32_t slab_pd_team_waits(apx_solver* s, int64_t out[4]) {
  out[0] = out[1] = out[2] = out[3] = 0;
  SlabPdState* st = s->slab_pd;
  if (st == nullptr || st->plan.P <= 1) return APX_OK;
  APX_TRY(team_read_waits(s->p->ctx, &st->team, out));
  out[2] *= 10;                                    // ticks of 10 ns
  out[3] = st->team.dev.launch;
  return APX_OK;
}

int32_t slab_pd_team_wait_slots(apx_solver* s, const TeamWait** slots) {
  *slots = nullptr;
  SlabPdState* st = s->slab_pd;
  if (st == nullptr || st->plan.P <= 1) return APX_OK;
  return team_read_wait_slots(s->p->ctx, &st->team, slots);
}

}  // namespace apx
