// skinny_rows_dev.h — the row kernel of the tall-skinny proximal-gradient family (skinny_pg.hip: adaptive / fixed
// proximal gradient ; skinny_bt.hip: the backtracking methods ; skinny_mom.hip: fixed Nesterov, aGRAAL): one pass over
// a dense A with at most 124 unknowns that leaves, for the point X it is given, the block partials of A' res, of
// ||res||^2 or the logistic loss terms and of sum(res) — everything f(X) and grad f(X) are made of.
#pragma once
#include "skinny_dev.h"

namespace apx {

struct SkinnyRowsDev {
  const void* A;          // m_pad x ncols, column-major
  int64_t ld;
  int32_t m, ncols, n, groups;   // n = ncols (+ 1: the logistic bias) ; groups of 64 rows per block
  int64_t ldpart;
  const void* b;
  double* part;           // [blocks][ldpart]: ncols partials of A' res, ||res||^2 or the loss sum, sum(res)
};

struct SkinnyRowsPlan {
  SkinnyRowsDev dev{};
  int B = 0, waves = 0, cpw = 0, npow = 0;
};

constexpr int kSpgMaxWaves = 8;       // column slices per block
constexpr int kSpgCloseBlock = 1024;
constexpr int kSpgMaxBlocks = 512;    // row blocks: what the closing block sums per column

// (Every dependent trip to memory costs 1-2 us here — the partials and the control block were written by the previous
//  launch, on other XCDs — so both kernels issue ALL their loads first, the done flag among them, and only then branch.)
template <typename T, int FK, int CPW>
__global__ __launch_bounds__(64 * kSpgMaxWaves) void k_spg_rows(SkinnyRowsDev d, const void* X, const int* halt) {
  __shared__ double xs[kSkinnyMaxN + 8];
  __shared__ double pd[2][kSpgMaxWaves][64];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int W = blockDim.x >> 6;
  const T* __restrict__ A = (const T*)d.A;
  const T* __restrict__ bv = (const T*)d.b;
  const int done = *halt;                      // written by an earlier launch: block-uniform
  const T xv = tid < d.n ? ((const T*)X)[tid] : (T)0;
  const int c0 = wv * CPW;
  int64_t co[CPW];
#pragma unroll
  for (int u = 0; u < CPW; ++u) co[u] = (int64_t)min(c0 + u, d.ncols - 1) * d.ld;   // clamped: a valid address, weight 0
  const int row0 = blockIdx.x * d.groups * 64 + lane;
  T av[CPW], nv[CPW] = {};
  T bi, nb = (T)0;
  {
    const int ic = min(row0, d.m - 1);
#pragma unroll
    for (int u = 0; u < CPW; ++u) av[u] = A[co[u] + ic];
    bi = bv[ic];
  }
  if (done != 0) return;
  if (tid < d.n) xs[tid] = (double)xv;
  __syncthreads();
  double xr[CPW], acc[CPW];
#pragma unroll
  for (int u = 0; u < CPW; ++u) {
    xr[u] = c0 + u < d.ncols ? xs[c0 + u] : 0.0;
    acc[u] = 0.0;
  }
  double ex[2] = {0.0, 0.0};
  for (int g = 0; g < d.groups; ++g) {
    const int i = row0 + g * 64;
    if (g + 1 < d.groups) {                    // the next group's loads go out before this group's arithmetic
      const int ic = min(i + 64, d.m - 1);
#pragma unroll
      for (int u = 0; u < CPW; ++u) nv[u] = A[co[u] + ic];
      nb = bv[ic];
    }
    double part = 0.0;
#pragma unroll
    for (int u = 0; u < CPW; ++u) part = fma((double)av[u], xr[u], part);
    pd[g & 1][wv][lane] = part;
    __syncthreads();                           // (two buffers: one barrier per group is enough)
    double a = pd[g & 1][0][lane];
    for (int w = 1; w < W; ++w) a += pd[g & 1][w][lane];
    a = (double)(T)a;                          // A_x rounded to T like every stored A x
    double res = 0.0;
    if (i < d.m) {
      if constexpr (FK == APX_F_LEAST_SQUARES) {
        res = (double)(T)(a - (double)bi);
        ex[0] += res * res;
      } else {
        const double logit = a + xs[d.n - 1];
        const double e = 1.0 + exp(-logit);
        ex[0] += ((double)bi - 1.0) * logit - log(e);
        res = (double)(T)(1.0 / e - (double)bi);
        ex[1] += res;
      }
    }
#pragma unroll
    for (int u = 0; u < CPW; ++u) {
      acc[u] = fma((double)av[u], res, acc[u]);
      av[u] = nv[u];
    }
    bi = nb;
  }
  double* __restrict__ out = d.part + (int64_t)blockIdx.x * d.ldpart;
  wave_sum_multi<CPW>(acc);                    // lane l: the total of column c0 + l / (64 / CPW)
  if (wv == 0) wave_sum_multi<2>(ex);          // (every wave holds the same two sums) lane l: ex[l / 32]
  constexpr int kPer = 64 / CPW;
  const int col = c0 + lane / kPer;
  if (lane % kPer == 0 && col < d.ncols) out[col] = acc[0];
  if (wv == 0 && (lane & 31) == 0) out[d.ncols + (lane >> 5)] = ex[0];
}

// columns per wave: 4 up to 32 columns (more waves per row group), else 16 ; row groups per block so that at most
// kSpgMaxBlocks partial rows are left for the closing block ; `part` must hold B * ldpart doubles
inline void skinny_rows_plan(const apx_matrix* A, int64_t n, const void* b, SkinnyRowsPlan* pl) {
  pl->cpw = A->n <= 4 * kSpgMaxWaves ? 4 : 16;
  pl->waves = (int)((A->n + pl->cpw - 1) / pl->cpw);
  const int64_t ngroups = (A->m + 63) / 64;
  const int groups = (int)((ngroups + kSpgMaxBlocks - 1) / kSpgMaxBlocks);
  pl->B = (int)((ngroups + groups - 1) / groups);
  pl->npow = skinny_npow((int)A->n);       // (>= ncols + 4 >= the ncols + 2 partial columns used)
  SkinnyRowsDev& d = pl->dev;
  d.groups = groups;
  d.A = A->d;
  d.ld = A->ld;
  d.m = (int32_t)A->m;
  d.ncols = (int32_t)A->n;
  d.n = (int32_t)n;
  d.ldpart = round_up(d.ncols + kSkinnyExtra, 8);
  d.b = b;
}

template <typename T, int FK>
inline void skinny_rows_launch(const SkinnyRowsPlan& pl, const void* X, const int* halt, hipStream_t q) {
  if (pl.cpw == 4) hipLaunchKernelGGL((k_spg_rows<T, FK, 4>), dim3(pl.B), dim3(64 * pl.waves), 0, q, pl.dev, X, halt);
  else hipLaunchKernelGGL((k_spg_rows<T, FK, 16>), dim3(pl.B), dim3(64 * pl.waves), 0, q, pl.dev, X, halt);
}

// shapes the family takes: dense LinearLeastSquares / LogisticLoss, no linear operator, no all-reduce hook
inline bool skinny_rows_eligible(const apx_problem* p) {
  if (!p->ctx->fused_small) return false;
  if (p->A_lin != nullptr || p->sharded()) return false;
  if (p->f_kind != APX_F_LEAST_SQUARES && p->f_kind != APX_F_LOGISTIC) return false;
  if (p->A_f == nullptr || no_column_storage(p->A_f) || p->ctx->skip_zero_cols) return false;
  const int64_t m = p->A_f->m;
  // below ~2000 rows small.hip's redundant scheme is as fast (logistic 2000 x 120: 14.8 us)
  // (no upper limit that matters: a block walks as many groups of 64 rows as it takes to leave 512 partial rows, and
  //  this ONE pass over A per iteration is half the traffic of the two GEMV passes whatever the size)
  return p->n <= kSkinnyMaxN && p->A_f->n > 0 && m >= 2048 && m <= ((int64_t)1 << 24) && p->n * 4 <= m;
}

}  // namespace apx
