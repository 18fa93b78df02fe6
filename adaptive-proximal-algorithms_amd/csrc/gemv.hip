// gemv.hip — launch geometry, compiled variants and HIP-event timing of the two GEMV kernels.
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "apx_internal.h"
#include "gemv_kernels.h"

namespace apx {

using NLaunch = void (*)(dim3, hipStream_t, const void*, int64_t, const void*, double*, int64_t, int64_t,
                         int64_t, const int*);
using TLaunch = void (*)(dim3, hipStream_t, const void*, int64_t, const void*, void*, double*, int64_t, int64_t,
                         int64_t, const int*);

template <typename T, int RPT, int UNR, bool NT, int XM, bool PIPE>
static void n_launch(dim3 grid, hipStream_t s, const void* A, int64_t ld, const void* x, double* part,
                     int64_t m_pad, int64_t n, int64_t cps, const int* done) {
  hipLaunchKernelGGL((gemv_n_kernel<T, RPT, UNR, NT, XM, PIPE>), grid, dim3(kBlock), 0, s, (const T*)A, ld,
                     (const T*)x, part, m_pad, n, cps, done);
}

// interleaved-chunk variant: `n` carries n_pad, `cps` is unused
template <typename T, int RPT, int UNR, bool NT>
static void n_launch_il(dim3 grid, hipStream_t s, const void* A, int64_t ld, const void* x, double* part,
                        int64_t m_pad, int64_t n_pad, int64_t, const int* done) {
  hipLaunchKernelGGL((gemv_n_il_kernel<T, RPT, UNR, NT>), grid, dim3(kBlock), 0, s, (const T*)A, ld, (const T*)x,
                     part, m_pad, n_pad, done);
}

template <typename T, int CPB, int UNR, bool NT, bool PIPE>
static void t_launch(dim3 grid, hipStream_t s, const void* A, int64_t ld, const void* r, void* out, double* part,
                     int64_t rps, int64_t m_pad, int64_t n_pad, const int* done) {
  hipLaunchKernelGGL((gemv_t_kernel<T, CPB, UNR, NT, PIPE>), grid, dim3(kBlock), 0, s, (const T*)A, ld,
                     (const T*)r, (T*)out, part, rps, m_pad, n_pad, done);
}

struct NVar { const char* name; int rpt, unr; NLaunch f64, f32; bool il; };
struct TVar { const char* name; int cpb, unr; TLaunch f64, f32; };

#define NV(RPT, UNR, NT, XM, PIPE)                                                        \
  { "n_rpt" #RPT "_unr" #UNR "_nt" #NT "_x" #XM "_p" #PIPE, RPT, UNR,                    \
    n_launch<double, RPT, UNR, NT, XM, PIPE>, n_launch<float, RPT, UNR, NT, XM, PIPE>, false }
#define NVI(RPT, UNR, NT)                                                                 \
  { "n_il_rpt" #RPT "_unr" #UNR "_nt" #NT, RPT, UNR,                                      \
    n_launch_il<double, RPT, UNR, NT>, n_launch_il<float, RPT, UNR, NT>, true }
#define TV(CPB, UNR, NT, PIPE)                                                            \
  { "t_cpb" #CPB "_unr" #UNR "_nt" #NT "_p" #PIPE, CPB, UNR,                              \
    t_launch<double, CPB, UNR, NT, PIPE>, t_launch<float, CPB, UNR, NT, PIPE> }

// Variant 0 is the default; the others exist so that one gpurun call can sweep them
// (apx_ctx_set_gemv_config / tools/gemv_sweep.py) — the choice is by measurement on MI355X.
static const NVar kNVars[] = {
    // non-temporal loads of A win by ~9 % on MI355X (6.75 vs 6.2 TB/s); tile shape / unroll are +-1 %
    NV(1, 8, true, 0, true),   NV(2, 8, true, 0, true),   NV(4, 4, true, 0, true),   NV(1, 16, true, 0, true),
    NV(2, 16, true, 0, true),  NV(1, 8, true, 1, true),   NV(2, 4, true, 0, true),   NV(1, 4, true, 0, true),
    NV(4, 8, true, 0, true),   NV(1, 8, true, 0, false),  NV(2, 8, false, 0, true),  NV(1, 8, false, 0, true),
    NVI(2, 16, true),          NVI(1, 16, true),          NVI(2, 8, true),           NVI(4, 8, true),
    NVI(1, 8, true),           NVI(4, 16, true),          NVI(2, 16, false),
};
static const TVar kTVars[] = {
    TV(4, 4, true, true),   TV(8, 2, true, true),   TV(4, 8, true, true),   TV(2, 8, true, true),
    TV(4, 2, true, true),   TV(8, 4, true, true),   TV(2, 16, true, true),  TV(16, 1, true, true),
    TV(4, 4, true, false),  TV(8, 2, false, true),  TV(4, 4, false, true),
};

int gemv_n_variant_count() { return (int)(sizeof(kNVars) / sizeof(kNVars[0])); }
int gemv_t_variant_count() { return (int)(sizeof(kTVars) / sizeof(kTVars[0])); }
const char* gemv_n_variant_name(int i) { return kNVars[i].name; }
const char* gemv_t_variant_name(int i) { return kTVars[i].name; }

int32_t ensure_part(apx_ctx* ctx, size_t bytes) {
  if (bytes <= ctx->part_bytes) return APX_OK;
  // growing the scratch must not race with kernels still using the old buffer
  APX_HIP(hipStreamSynchronize(ctx->stream));
  if (ctx->part) (void)hipFree(ctx->part);
  ctx->part = nullptr;
  ctx->part_bytes = 0;
  const size_t want = bytes + bytes / 4;
  APX_HIP(hipMalloc((void**)&ctx->part, want));
  ctx->part_bytes = want;
  return APX_OK;
}

// ---- HIP-event timing of single launches (bench.py roofline; off by default) ---------
int32_t prof_begin(apx_ctx* ctx, int kind, double bytes) {
  if (!ctx->profile) return APX_OK;
  std::pair<hipEvent_t, hipEvent_t> ev;
  if (!ctx->ev_free.empty()) {
    ev = ctx->ev_free.back();
    ctx->ev_free.pop_back();
  } else {
    APX_HIP(hipEventCreate(&ev.first));
    APX_HIP(hipEventCreate(&ev.second));
  }
  APX_HIP(hipEventRecord(ev.first, ctx->stream));
  ctx->ev_pending.push_back({ev.first, ev.second, kind, bytes});
  return APX_OK;
}
int32_t prof_end(apx_ctx* ctx) {
  if (!ctx->profile) return APX_OK;
  APX_HIP(hipEventRecord(ctx->ev_pending.back().b, ctx->stream));
  return APX_OK;
}

static int32_t launch_t_core(apx_ctx* ctx, const void* d, int64_t ld, int64_t n_pad, int32_t dtype, double bytes,
                             int stat_kind, const void* r, void* out, int* splits_out, const int* done);

// opt-in sparse-x path: compact the non-zero entries of x, then stream only those columns
static int32_t launch_gemv_n_nz(apx_ctx* ctx, const apx_matrix* A, const void* x, int* splits_out, const int* done) {
  if (ctx->nz_cap < A->n) {
    APX_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->nz_idx) (void)hipFree(ctx->nz_idx);
    ctx->nz_idx = nullptr;
    ctx->nz_cap = 0;
    APX_HIP(hipMalloc((void**)&ctx->nz_idx, (size_t)(A->n + 16) * sizeof(int32_t)));
    ctx->nz_cap = A->n;
  }
  int32_t* count = ctx->nz_idx + ctx->nz_cap;
  constexpr int RPT = 2, UNR = 8;
  const int V = A->dtype == APX_F32 ? 4 : 2;
  const int64_t rows_per_block = (int64_t)kBlock * V * RPT;
  const int64_t row_tiles = (A->ld + rows_per_block - 1) / rows_per_block;
  int64_t S = std::max<int64_t>(1, (ctx->cfg.n_target_blocks + row_tiles - 1) / row_tiles);
  S = std::min<int64_t>(S, std::max<int64_t>(1, A->n / UNR));
  APX_REQUIRE(S <= 65535, "gemv_n grid too large");
  APX_TRY(ensure_part(ctx, (size_t)S * A->ld * sizeof(double)));
  APX_TRY(prof_begin(ctx, APX_KSTAT_GEMV_N, (double)A->m * (double)A->n * dtype_size(A->dtype)));
  dim3 grid((unsigned)row_tiles, (unsigned)S);
  if (A->dtype == APX_F32) {
    hipLaunchKernelGGL(compact_nonzero_kernel<float>, dim3(1), dim3(1024), 0, ctx->stream, (const float*)x, A->n,
                       ctx->nz_idx, count, done);
    hipLaunchKernelGGL((gemv_n_nz_kernel<float, RPT, UNR, true>), grid, dim3(kBlock), 0, ctx->stream,
                       (const float*)A->d, A->ld, (const float*)x, ctx->nz_idx, count, ctx->part, A->ld, done);
  } else {
    hipLaunchKernelGGL(compact_nonzero_kernel<double>, dim3(1), dim3(1024), 0, ctx->stream, (const double*)x, A->n,
                       ctx->nz_idx, count, done);
    hipLaunchKernelGGL((gemv_n_nz_kernel<double, RPT, UNR, true>), grid, dim3(kBlock), 0, ctx->stream,
                       (const double*)A->d, A->ld, (const double*)x, ctx->nz_idx, count, ctx->part, A->ld, done);
  }
  APX_TRY(prof_end(ctx));
  APX_HIP(hipGetLastError());
  *splits_out = (int)S;
  return APX_OK;
}

// One launch of the row-tiled kernel (gemv_n_kernel: block = a tile of rows x one split of the columns, the vector staged through
// LDS) on a dense column-major operand (d, ld = padded rows, ncols columns, ncols_pad their padded count): fp64 partials
// [splits][ld] left in ctx->part.  `A*x` on the column-major storage ; `A'*r` on the ROW storage of a rows-only matrix (operand A').
static int32_t launch_n_core(apx_ctx* ctx, const void* d, int64_t ld, int64_t ncols, int64_t ncols_pad, int32_t dtype, double bytes,
                             int stat_kind, const void* x, int* splits_out, const int* done) {
  const int V = dtype == APX_F32 ? 4 : 2;
  int vi = ctx->cfg.n_variant;
  int64_t target = ctx->cfg.n_target_blocks;
  if (ctx->cfg.n_auto && ncols * (int64_t)dtype_size(dtype) <= 256 * 1024) {
    // short rows (C5: 160 KB, C3: 40 KB per row of A): the narrow 512-row tile with ~8 blocks per
    // CU measured 4 % (C5) / 1 % (C3) faster than the wide-tile default (gpurun_out/sweep_c5.log)
    vi = ctx->cfg.n_variant_narrow;
    target = 8 * (int64_t)ctx->num_cu;
  }
  int64_t rows_per_block = (int64_t)kBlock * V * kNVars[vi].rpt;
  int64_t row_tiles = (ld + rows_per_block - 1) / rows_per_block;
  int64_t S = std::max<int64_t>(1, (target + row_tiles - 1) / row_tiles);
  if (ctx->cfg.n_auto && (double)ld * (double)ncols * (double)dtype_size(dtype) > 256e6) {
    // (cache-resident matrices are latency-bound: there more blocks matter more than partial traffic)
    // every column split writes and re-reads an m-vector of fp64 partials: keep that below ~1 %
    // of the A stream (S*16 B per row against n*sizeof(T)); when that leaves too few blocks for a
    // wide row tile, fall back to the narrow-tile variant (short-and-wide problems such as C3)
    const int64_t s_cap = std::max<int64_t>(1, ncols * (int64_t)dtype_size(dtype) / 1600);
    if (S > s_cap) {
      S = s_cap;
      if (row_tiles * S < 8 * (int64_t)ctx->num_cu && kNVars[vi].rpt > 1) {
        vi = ctx->cfg.n_variant_narrow;
        rows_per_block = (int64_t)kBlock * V * kNVars[vi].rpt;
        row_tiles = (ld + rows_per_block - 1) / rows_per_block;
      }
    }
  }
  const NVar& v = kNVars[vi];
  int64_t cps = round_up((ncols + S - 1) / S, v.unr);
  if (cps < v.unr) cps = v.unr;
  S = (ncols + cps - 1) / cps;
  if (S < 1) S = 1;
  int64_t n_arg = ncols;
  if (v.il) {   // chunks of v.unr columns dealt round-robin to the S splits; ncols_pad % unr == 0
    n_arg = ncols_pad;
    S = std::min<int64_t>(S, ncols_pad / v.unr);
  }
  APX_REQUIRE(S <= 65535 && row_tiles <= 2147483647LL, "gemv_n grid too large");
  APX_TRY(ensure_part(ctx, (size_t)S * ld * sizeof(double)));
  APX_TRY(prof_begin(ctx, stat_kind, bytes));
  dim3 grid((unsigned)row_tiles, (unsigned)S);
  (dtype == APX_F32 ? v.f32 : v.f64)(grid, ctx->stream, d, ld, x, ctx->part, ld, n_arg, cps, done);
  APX_TRY(prof_end(ctx));
  APX_HIP(hipGetLastError());
  *splits_out = (int)S;
  return APX_OK;
}

int32_t launch_gemv_n(apx_ctx* ctx, const apx_matrix* A, const void* x, int* splits_out, const int* done) {
  if (A->sparse) return launch_spmv_n(ctx, A, x, splits_out, done);
  const double bytes = (double)A->m * (double)A->n * dtype_size(A->dtype);
  if (ctx->skip_zero_cols && A->d != nullptr) return launch_gemv_n_nz(ctx, A, x, splits_out, done);
  if (A->dT != nullptr)   // the row storage (dual layout, or a rows-only matrix): A*x as the column-streaming kernel over A'
    return launch_t_core(ctx, A->dT, A->ldT, A->ld, A->dtype, bytes, APX_KSTAT_GEMV_N, x, nullptr, splits_out, done);
  return launch_n_core(ctx, A->d, A->ld, A->n, A->n_pad, A->dtype, bytes, APX_KSTAT_GEMV_N, x, splits_out, done);
}

// One launch of the column-streaming kernel on a dense column-major operand (d, ld, rows m_pad,
// columns n_pad): out != nullptr -> the product as T (gemv_t proper, with a finish pass when the
// rows are split); out == nullptr -> fp64 partials [splits][n_pad] left in ctx->part for a fused
// epilogue (used by gemv_n on the transposed copy).
static int32_t launch_t_core(apx_ctx* ctx, const void* d, int64_t ld, int64_t n_pad, int32_t dtype, double bytes,
                             int stat_kind, const void* r, void* out, int* splits_out, const int* done) {
  const TVar& v = kTVars[ctx->cfg.t_variant];
  const int V = dtype == APX_F32 ? 4 : 2;
  const int64_t col_blocks = n_pad / v.cpb;
  const int64_t step = (int64_t)kBlock * V * v.unr;
  int64_t SR = std::max<int64_t>(1, ctx->cfg.t_target_blocks / std::max<int64_t>(1, col_blocks));
  int64_t rps = round_up((ld + SR - 1) / SR, step);
  SR = (ld + rps - 1) / rps;
  APX_REQUIRE(SR <= 65535 && col_blocks <= 2147483647LL, "gemv_t grid too large");
  if (SR > 1 || out == nullptr) APX_TRY(ensure_part(ctx, (size_t)SR * n_pad * sizeof(double)));
  APX_TRY(prof_begin(ctx, stat_kind, bytes));
  dim3 grid((unsigned)col_blocks, (unsigned)SR);
  (dtype == APX_F32 ? v.f32 : v.f64)(grid, ctx->stream, d, ld, r, out, ctx->part, rps, ld, n_pad, done);
  APX_TRY(prof_end(ctx));
  if (SR > 1 && out != nullptr) {
    const unsigned g = (unsigned)((n_pad + kBlock - 1) / kBlock);
    if (dtype == APX_F32)
      hipLaunchKernelGGL(gemv_t_finish<float>, dim3(g), dim3(kBlock), 0, ctx->stream, ctx->part, (int)SR, n_pad,
                         (float*)out, done);
    else
      hipLaunchKernelGGL(gemv_t_finish<double>, dim3(g), dim3(kBlock), 0, ctx->stream, ctx->part, (int)SR, n_pad,
                         (double*)out, done);
  }
  APX_HIP(hipGetLastError());
  if (splits_out) *splits_out = (int)SR;
  return APX_OK;
}

int32_t launch_gemv_t(apx_ctx* ctx, const apx_matrix* A, const void* r, void* out, const int* done) {
  if (A->sparse) return launch_spmv_t(ctx, A, r, out, done);
  if (A->d == nullptr) {
    // a rows-only matrix: A'*r = (the row storage as an n_pad x m column-major operand) * r — the row-tiled kernel with the roles
    // of the two dimensions swapped, its column-split partials summed in a fixed order (r is padded to ld = the operand's ncols_pad)
    int S = 0;
    APX_TRY(launch_n_core(ctx, A->dT, A->ldT, A->m, A->ld, A->dtype, (double)A->m * (double)A->n * dtype_size(A->dtype),
                          APX_KSTAT_GEMV_T, r, &S, done));
    return launch_sum_partials_rows(ctx, A->dtype, S, A->ldT, A->ldT, out, done);
  }
  return launch_t_core(ctx, A->d, A->ld, A->n_pad, A->dtype, (double)A->m * (double)A->n * dtype_size(A->dtype),
                       APX_KSTAT_GEMV_T, r, out, nullptr, done);
}

template <typename T>
__global__ __launch_bounds__(kBlock) void sum_partials_rows(const double* __restrict__ part, int splits, int64_t ld,
                                                            int64_t m, T* __restrict__ y,
                                                            const int* __restrict__ done) {
  if (done != nullptr && *done != 0) return;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < m; i += (int64_t)gridDim.x * kBlock) {
    y[i] = (T)sum_splits(part, splits, ld, i);
  }
}

// The same sum for a FEW rows and many splits (A = y' of the dual SVM is 1 x N: ~1000 column splits and one
// output — a thread per row would walk them as a chain of dependent loads, 38 us): a block per row, the splits
// dealt to its threads, then a fixed-order block sum.
template <typename T>
__global__ __launch_bounds__(kBlock) void sum_partials_few_rows(const double* __restrict__ part, int splits, int64_t ld,
                                                                T* __restrict__ y, const int* __restrict__ done) {
  __shared__ double sh[kBlock / 64];
  if (done != nullptr && *done != 0) return;
  const int64_t i = blockIdx.x;
  double a[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int k = (int)threadIdx.x + u * kBlock;
    a[u] = k < splits ? part[(int64_t)k * ld + i] : 0.0;
  }
  double s = (a[0] + a[1]) + (a[2] + a[3]);
  for (int k = (int)threadIdx.x + 4 * kBlock; k < splits; k += kBlock) s += part[(int64_t)k * ld + i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = sh[0];
#pragma unroll
    for (int w = 1; w < kBlock / 64; ++w) t += sh[w];
    y[i] = (T)t;
  }
}

int32_t launch_sum_partials_rows(apx_ctx* ctx, int32_t dtype, int splits, int64_t m, int64_t ld, void* y,
                                 const int* done) {
  if (m <= 64 && splits > 16) {
    if (dtype == APX_F32)
      hipLaunchKernelGGL(sum_partials_few_rows<float>, dim3((unsigned)m), dim3(kBlock), 0, ctx->stream, ctx->part, splits, ld,
                         (float*)y, done);
    else
      hipLaunchKernelGGL(sum_partials_few_rows<double>, dim3((unsigned)m), dim3(kBlock), 0, ctx->stream, ctx->part, splits, ld,
                         (double*)y, done);
    APX_HIP(hipGetLastError());
    return APX_OK;
  }
  const unsigned g = (unsigned)std::min<int64_t>(kMaxRedBlocks, (m + kBlock - 1) / kBlock);
  if (dtype == APX_F32)
    hipLaunchKernelGGL(sum_partials_rows<float>, dim3(g), dim3(kBlock), 0, ctx->stream, ctx->part, splits, ld, m,
                       (float*)y, done);
  else
    hipLaunchKernelGGL(sum_partials_rows<double>, dim3(g), dim3(kBlock), 0, ctx->stream, ctx->part, splits, ld, m,
                       (double*)y, done);
  APX_HIP(hipGetLastError());
  return APX_OK;
}

}  // namespace apx
