// gemv_kernels.h — the two bandwidth-bound kernels of the hot path, gfx950 only.
//
//   gemv_n :  y = A x      (K1 of SURVEY §2.1;  `f.A * w`,  experiments/lasso/runme.jl:22, src/AdaProx.jl:335)
//   gemv_t :  g = A' r     (K3;  `f.A' * res`, experiments/lasso/runme.jl:23, src/AdaProx.jl:358)
//
// A is column-major (Julia layout) with leading dimension `ld`; ld and the padded
// column count are multiples of 16 elements and every pad entry is ZERO, so neither
// kernel needs a tail path: all accesses are 16 B / lane, 1 KiB / wave-instruction,
// 128-B aligned.  A is streamed exactly once per launch, straight into VGPRs with a
// deep unroll (GUIDE-hip §5 table, row "GEMV / M <= 16": LDS round trip for the
// streamed operand is pure overhead); only the SHARED vector (x tile) goes through
// LDS.  MFMA is deliberately not used: 0.25 flop/B (fp64) is three orders below the
// ridge.  Reductions are two-stage and atomics-free => bit-reproducible.
//
// Accumulation is always in fp64 (fp32 storage is widened on the fly: VALU has >5x
// headroom at the HBM-bound rate of ~10 B/clk/CU).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace apx {

typedef double dvec2 __attribute__((ext_vector_type(2)));
typedef float fvec4 __attribute__((ext_vector_type(4)));

template <typename T> struct V16;
template <> struct V16<double> { using type = dvec2; static constexpr int N = 2; };
template <> struct V16<float> { using type = fvec4; static constexpr int N = 4; };

template <typename T, bool NT>
__device__ __forceinline__ typename V16<T>::type ld16(const T* p) {
  using VT = typename V16<T>::type;
  if constexpr (NT) return __builtin_nontemporal_load(reinterpret_cast<const VT*>(p));
  else return *reinterpret_cast<const VT*>(p);
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

constexpr int kBlock = 256;   // 4 waves; one per SIMD

// Second stage of a split product: sum over the `splits` fp64 partials of element i
// (part[k*stride + i]).  Eight independent accumulators keep eight loads in flight (a plain
// running sum is a chain of dependent L2 round trips: 24-36 us at 84 splits); the order is fixed
// — accumulator u takes k = u mod 8, then a fixed tree — so results stay bit-reproducible.
__device__ __forceinline__ double sum_splits(const double* __restrict__ part, int splits, int64_t stride, int64_t i) {
  double a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int k = 0;
  for (; k + 8 <= splits; k += 8) {
#pragma unroll
    for (int u = 0; u < 8; ++u) a[u] += part[(int64_t)(k + u) * stride + i];
  }
  double tail = 0.0;   // (a scalar, not a[k & 7]: runtime-indexed register arrays go to scratch)
  for (; k < splits; ++k) tail += part[(int64_t)k * stride + i];
  return (((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]))) + tail;
}

// ----------------------------------------------------------------------------------
// gemv_t: block = CPB consecutive columns x one row split.  Every thread walks down
// the rows (16 B per column per step), re-using its r fragment for all CPB columns,
// so r traffic (L2-resident, m*8 B) is 1/CPB of the A stream.  grid = (n_pad/CPB, SR).
//   SR == 1 : out[j] written directly (as T)
//   SR  > 1 : part[split*n_pad + j] (double), summed in order by gemv_t_finish
// ----------------------------------------------------------------------------------
template <typename T, int CPB, int UNR, bool NT, bool PIPE>
__global__ __launch_bounds__(kBlock) void gemv_t_kernel(
    const T* __restrict__ A, int64_t ld, const T* __restrict__ r, T* __restrict__ out,
    double* __restrict__ part, int64_t rows_per_split, int64_t m_pad, int64_t n_pad,
    const int* __restrict__ done) {
  if (done != nullptr && *done != 0) return;
  constexpr int V = V16<T>::N;
  constexpr int64_t STEP = (int64_t)kBlock * V;
  const int tid = threadIdx.x;
  const int64_t j0 = (int64_t)blockIdx.x * CPB;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_split;
  int64_t r1 = r0 + rows_per_split;
  if (r1 > m_pad) r1 = m_pad;

  double acc[CPB];
#pragma unroll
  for (int c = 0; c < CPB; ++c) acc[c] = 0.0;

  const T* Ab = A + j0 * ld;
  int64_t row = r0 + (int64_t)tid * V;
  const int64_t nfull = (r1 - r0) / (STEP * UNR);
  for (int64_t it = 0; it < nfull; ++it) {
    // issue every load of the step first (UNR*(CPB+1) 16-B loads per lane in flight), then the FMAs
    typename V16<T>::type av[UNR][CPB], rv[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int64_t rr = row + u * STEP;
      rv[u] = ld16<T, false>(r + rr);
#pragma unroll
      for (int c = 0; c < CPB; ++c) av[u][c] = ld16<T, NT>(Ab + c * ld + rr);
    }
    if constexpr (PIPE) __builtin_amdgcn_sched_barrier(0);   // keep hipcc from sinking loads between the FMAs
#pragma unroll
    for (int u = 0; u < UNR; ++u)
#pragma unroll
      for (int c = 0; c < CPB; ++c)
#pragma unroll
        for (int e = 0; e < V; ++e) acc[c] = fma((double)av[u][c][e], (double)rv[u][e], acc[c]);
    row += STEP * UNR;
  }
  for (; row < r1; row += STEP) {   // ragged tail of the split (whole 16-B chunks: m_pad % 16 == 0)
    const auto rv = ld16<T, false>(r + row);
#pragma unroll
    for (int c = 0; c < CPB; ++c) {
      const auto av = ld16<T, NT>(Ab + c * ld + row);
#pragma unroll
      for (int e = 0; e < V; ++e) acc[c] = fma((double)av[e], (double)rv[e], acc[c]);
    }
  }

  __shared__ double red[kBlock / 64][CPB];
  const int lane = tid & 63, wv = tid >> 6;
#pragma unroll
  for (int c = 0; c < CPB; ++c) {
    const double s = wave_sum(acc[c]);
    if (lane == 0) red[wv][c] = s;
  }
  __syncthreads();
  if (tid < CPB) {
    double s = red[0][tid];
#pragma unroll
    for (int w = 1; w < kBlock / 64; ++w) s += red[w][tid];
    if (gridDim.y == 1 && out != nullptr) out[j0 + tid] = (T)s;
    else part[(int64_t)blockIdx.y * n_pad + j0 + tid] = s;   // fp64 partials (row splits, or out == nullptr)
  }
}

template <typename T>
__global__ __launch_bounds__(kBlock) void gemv_t_finish(const double* __restrict__ part, int splits,
                                                        int64_t n_pad, T* __restrict__ out,
                                                        const int* __restrict__ done) {
  if (done != nullptr && *done != 0) return;
  const int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (j >= n_pad) return;
  out[j] = (T)sum_splits(part, splits, n_pad, j);
}

// ----------------------------------------------------------------------------------
// gemv_n: block = (256*V*RPT rows) x (one column split).  Coalesced down the rows; the
// x slice of the split is staged in LDS in XT-column tiles (XMODE 0, one broadcast
// ds_read_b64 per column) or fetched through the scalar cache (XMODE 1: the column
// index is wave-uniform, so hipcc emits s_load).  Each thread keeps RPT*V fp64
// accumulators; UNR columns are in flight together (UNR*RPT 16-B loads per lane).
//   part[split*m_pad + row] (double); the f-specific epilogue (residual, loss,
//   sum of squares) is fused into the second-stage kernel (vecops.hip, K2/K4).
// grid = (ceil(m_pad / (256*V*RPT)), S)
// ----------------------------------------------------------------------------------
constexpr int kXTile = 2048;   // x columns per LDS tile (16 KiB of fp64)

template <typename T, int RPT, int UNR, bool NT, int XMODE, bool PIPE>
__global__ __launch_bounds__(kBlock) void gemv_n_kernel(
    const T* __restrict__ A, int64_t ld, const T* __restrict__ x, double* __restrict__ part,
    int64_t m_pad, int64_t n, int64_t cols_per_split, const int* __restrict__ done) {
  if (done != nullptr && *done != 0) return;
  constexpr int V = V16<T>::N;
  constexpr int64_t STEP = (int64_t)kBlock * V;
  const int tid = threadIdx.x;
  const int64_t row_base = (int64_t)blockIdx.x * (STEP * RPT) + (int64_t)tid * V;
  const int64_t c0 = (int64_t)blockIdx.y * cols_per_split;
  int64_t c1 = c0 + cols_per_split;
  if (c1 > n) c1 = n;

  // rows past m_pad are redirected to row 0 (valid memory) and dropped at the end, so
  // the streaming loop carries no per-load branch (GUIDE-hip §5 trap 4c)
  int64_t rowk[RPT];
  bool live[RPT];
#pragma unroll
  for (int k = 0; k < RPT; ++k) {
    const int64_t rr = row_base + k * STEP;
    live[k] = rr < m_pad;
    rowk[k] = live[k] ? rr : 0;
  }

  double acc[RPT][V];
#pragma unroll
  for (int k = 0; k < RPT; ++k)
#pragma unroll
    for (int e = 0; e < V; ++e) acc[k][e] = 0.0;

  __shared__ double xs[XMODE == 0 ? kXTile : 1];

  for (int64_t t0 = c0; t0 < c1; t0 += kXTile) {
    int64_t t1 = t0 + kXTile;
    if (t1 > c1) t1 = c1;
    if constexpr (XMODE == 0) {
      __syncthreads();
      for (int64_t j = t0 + tid; j < t1; j += kBlock) xs[j - t0] = (double)x[j];
      __syncthreads();
    }
    const T* Ac = A + t0 * ld;
    int64_t j = t0;
    for (; j + UNR <= t1; j += UNR) {
      // all UNR*RPT loads of the step are issued before the first FMA consumes one
      typename V16<T>::type av[UNR][RPT];
      double xj[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
#pragma unroll
        for (int k = 0; k < RPT; ++k) av[u][k] = ld16<T, NT>(Ac + u * ld + rowk[k]);
        if constexpr (XMODE == 0) xj[u] = xs[j + u - t0];
        else xj[u] = (double)x[j + u];
      }
      if constexpr (PIPE) __builtin_amdgcn_sched_barrier(0);   // keep hipcc from sinking loads between the FMAs
#pragma unroll
      for (int u = 0; u < UNR; ++u)
#pragma unroll
        for (int k = 0; k < RPT; ++k)
#pragma unroll
          for (int e = 0; e < V; ++e) acc[k][e] = fma((double)av[u][k][e], xj[u], acc[k][e]);
      Ac += UNR * ld;
    }
    for (; j < t1; ++j) {
      double xj;
      if constexpr (XMODE == 0) xj = xs[j - t0];
      else xj = (double)x[j];
#pragma unroll
      for (int k = 0; k < RPT; ++k) {
        const auto av = ld16<T, NT>(Ac + rowk[k]);
#pragma unroll
        for (int e = 0; e < V; ++e) acc[k][e] = fma((double)av[e], xj, acc[k][e]);
      }
      Ac += ld;
    }
  }

  double* po = part + (int64_t)blockIdx.y * m_pad;
#pragma unroll
  for (int k = 0; k < RPT; ++k) {
    if (live[k]) {
      dvec2* dst = reinterpret_cast<dvec2*>(po + row_base + k * STEP);
#pragma unroll
      for (int e = 0; e < V; e += 2) {
        dvec2 o;
        o[0] = acc[k][e];
        o[1] = acc[k][e + 1];
        dst[e / 2] = o;
      }
    }
  }
}

// ----------------------------------------------------------------------------------
// gemv_n, interleaved column chunks ("il"): split s of S owns the UNR-column chunks
// s, s+S, s+2S, ... instead of one contiguous band of n/S columns.  All resident blocks then
// sweep the matrix together, so the addresses in flight chip-wide span S*UNR columns
// (~0.5 GB at C2) instead of half the matrix — friendlier to the TLBs and DRAM pages, same
// bytes, same partial layout.  x is read through the scalar cache (uniform index); the
// x vector must be readable (zero-padded) up to n_pad, which is a multiple of UNR.
// grid = (ceil(m_pad / (256*V*RPT)), S)
// ----------------------------------------------------------------------------------
template <typename T, int RPT, int UNR, bool NT>
__global__ __launch_bounds__(kBlock) void gemv_n_il_kernel(
    const T* __restrict__ A, int64_t ld, const T* __restrict__ x, double* __restrict__ part,
    int64_t m_pad, int64_t n_pad, const int* __restrict__ done) {
  if (done != nullptr && *done != 0) return;
  constexpr int V = V16<T>::N;
  constexpr int64_t STEP = (int64_t)kBlock * V;
  const int tid = threadIdx.x;
  const int64_t row_base = (int64_t)blockIdx.x * (STEP * RPT) + (int64_t)tid * V;
  int64_t rowk[RPT];
  bool live[RPT];
#pragma unroll
  for (int k = 0; k < RPT; ++k) {
    const int64_t rr = row_base + k * STEP;
    live[k] = rr < m_pad;
    rowk[k] = live[k] ? rr : 0;
  }
  double acc[RPT][V];
#pragma unroll
  for (int k = 0; k < RPT; ++k)
#pragma unroll
    for (int e = 0; e < V; ++e) acc[k][e] = 0.0;

  const int64_t nchunks = n_pad / UNR;
  const int64_t S = gridDim.y;
  for (int64_t chunk = blockIdx.y; chunk < nchunks; chunk += S) {
    const int64_t j0 = chunk * UNR;
    const T* Ac = A + j0 * ld;
    typename V16<T>::type av[UNR][RPT];
    double xj[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
#pragma unroll
      for (int k = 0; k < RPT; ++k) av[u][k] = ld16<T, NT>(Ac + u * ld + rowk[k]);
      xj[u] = (double)x[j0 + u];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < UNR; ++u)
#pragma unroll
      for (int k = 0; k < RPT; ++k)
#pragma unroll
        for (int e = 0; e < V; ++e) acc[k][e] = fma((double)av[u][k][e], xj[u], acc[k][e]);
  }

  double* po = part + (int64_t)blockIdx.y * m_pad;
#pragma unroll
  for (int k = 0; k < RPT; ++k) {
    if (live[k]) {
      dvec2* dst = reinterpret_cast<dvec2*>(po + row_base + k * STEP);
#pragma unroll
      for (int e = 0; e < V; e += 2) {
        dvec2 o;
        o[0] = acc[k][e];
        o[1] = acc[k][e + 1];
        dst[e / 2] = o;
      }
    }
  }
}

// ----------------------------------------------------------------------------------
// Opt-in: exploit the exact zeros of x (a lasso iterate after the soft-threshold).
// Reference BLAS `dgemv` itself skips columns with x_j == 0; here a one-block pass compacts
// the indices of the non-zero entries (ascending) and gemv_n_nz streams only those columns of
// the column-major A.  Same kernel structure, same partial layout, fixed order.
// ----------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(1024) void compact_nonzero_kernel(const T* __restrict__ x, int64_t n,
                                                               int32_t* __restrict__ idx, int32_t* __restrict__ count,
                                                               const int* __restrict__ done) {
  if (done != nullptr && *done != 0) return;
  __shared__ int32_t wsum[16];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int64_t per = (n + 1023) / 1024;
  const int64_t lo = (int64_t)tid * per;
  const int64_t hi = lo + per < n ? lo + per : n;
  int32_t c = 0;
  for (int64_t i = lo; i < hi; ++i) c += (x[i] != (T)0) ? 1 : 0;
  int32_t incl = c;                                   // inclusive scan inside the wave
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int32_t v = __shfl_up(incl, off, 64);
    if (lane >= off) incl += v;
  }
  if (lane == 63) wsum[wv] = incl;
  __syncthreads();
  int32_t base = 0;
  for (int w = 0; w < wv; ++w) base += wsum[w];
  int32_t pos = base + incl - c;
  for (int64_t i = lo; i < hi; ++i)
    if (x[i] != (T)0) idx[pos++] = (int32_t)i;
  if (tid == 1023) *count = base + incl;
}

template <typename T, int RPT, int UNR, bool NT>
__global__ __launch_bounds__(kBlock) void gemv_n_nz_kernel(
    const T* __restrict__ A, int64_t ld, const T* __restrict__ x, const int32_t* __restrict__ idx,
    const int32_t* __restrict__ count, double* __restrict__ part, int64_t m_pad, const int* __restrict__ done) {
  if (done != nullptr && *done != 0) return;
  constexpr int V = V16<T>::N;
  constexpr int64_t STEP = (int64_t)kBlock * V;
  constexpr int TILE = 1024;
  const int tid = threadIdx.x;
  const int64_t row_base = (int64_t)blockIdx.x * (STEP * RPT) + (int64_t)tid * V;
  const int64_t total = *count;
  int64_t cps = (total + gridDim.y - 1) / gridDim.y;
  cps = (cps + UNR - 1) / UNR * UNR;
  const int64_t c0 = (int64_t)blockIdx.y * cps;
  int64_t c1 = c0 + cps;
  if (c1 > total) c1 = total;

  int64_t rowk[RPT];
  bool live[RPT];
#pragma unroll
  for (int k = 0; k < RPT; ++k) {
    const int64_t rr = row_base + k * STEP;
    live[k] = rr < m_pad;
    rowk[k] = live[k] ? rr : 0;
  }
  double acc[RPT][V];
#pragma unroll
  for (int k = 0; k < RPT; ++k)
#pragma unroll
    for (int e = 0; e < V; ++e) acc[k][e] = 0.0;

  __shared__ double xs[TILE];
  __shared__ int32_t js[TILE];
  for (int64_t t0 = c0; t0 < c1; t0 += TILE) {
    int64_t t1 = t0 + TILE;
    if (t1 > c1) t1 = c1;
    __syncthreads();
    for (int64_t j = t0 + tid; j < t1; j += kBlock) {
      const int32_t col = idx[j];
      js[j - t0] = col;
      xs[j - t0] = (double)x[col];
    }
    __syncthreads();
    int64_t j = t0;
    for (; j + UNR <= t1; j += UNR) {
      typename V16<T>::type av[UNR][RPT];
      double xj[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const T* Ac = A + (int64_t)js[j + u - t0] * ld;
#pragma unroll
        for (int k = 0; k < RPT; ++k) av[u][k] = ld16<T, NT>(Ac + rowk[k]);
        xj[u] = xs[j + u - t0];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < UNR; ++u)
#pragma unroll
        for (int k = 0; k < RPT; ++k)
#pragma unroll
          for (int e = 0; e < V; ++e) acc[k][e] = fma((double)av[u][k][e], xj[u], acc[k][e]);
    }
    for (; j < t1; ++j) {
      const T* Ac = A + (int64_t)js[j - t0] * ld;
      const double xj = xs[j - t0];
#pragma unroll
      for (int k = 0; k < RPT; ++k) {
        const auto av = ld16<T, NT>(Ac + rowk[k]);
#pragma unroll
        for (int e = 0; e < V; ++e) acc[k][e] = fma((double)av[e], xj, acc[k][e]);
      }
    }
  }

  double* po = part + (int64_t)blockIdx.y * m_pad;
#pragma unroll
  for (int k = 0; k < RPT; ++k) {
    if (live[k]) {
      dvec2* dst = reinterpret_cast<dvec2*>(po + row_base + k * STEP);
#pragma unroll
      for (int e = 0; e < V; e += 2) {
        dvec2 o;
        o[0] = acc[k][e];
        o[1] = acc[k][e + 1];
        dst[e / 2] = o;
      }
    }
  }
}

}  // namespace apx
