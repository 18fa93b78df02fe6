// small_dev.h — pieces shared by the two-kernel small-problem iterations (small.hip: proximal gradient,
// small_pd.hip: primal-dual): in-block dot products with the vector operand in LDS, block sums, grid sizing.
#pragma once
#include <algorithm>

#include "solver_internal.h"
#include "vecops_dev.h"

namespace apx {

static_assert(kSmallMax % kBlock == 0, "kSmallMax must be a multiple of the block size");
constexpr int kSmallPer = kSmallMax / kBlock;


// out[i] = dot(M[:, i], vec), i in [0, n_out): M is column-major (ld x n_out, columns contiguous), vec (length
// len) sits in LDS.  W waves of the block share one dot product (lanes along the reduction index, coalesced
// 512 B per wave-load), 4/W dot products per block and pass, blocks stride over the rest.  Every dot product
// is finished inside its block — no split partials, no second stage.
constexpr int kDotUnroll = 16;   // loads in flight per lane: a 1024-element dot product is one round trip

template <typename T, int W>
__device__ __forceinline__ void small_dots(const T* __restrict__ M, int64_t ld, int n_out, int len,
                                           const double* __restrict__ vec, double* __restrict__ out, bool live) {
  constexpr int DPB = (kBlock / 64) / W;
  __shared__ double wsum[kBlock / 64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int part = wv % W;
  constexpr int step = W * 64;
  for (int d0 = blockIdx.x * DPB; d0 < n_out; d0 += gridDim.x * DPB) {
    const int dot = d0 + wv / W;
    double acc = 0.0;
    if (dot < n_out) {
      const T* __restrict__ col = M + (int64_t)dot * ld;
      for (int r = part * 64 + lane; r < len; r += kDotUnroll * step) {
        T a[kDotUnroll];
        double xv[kDotUnroll];
#pragma unroll
        for (int u = 0; u < kDotUnroll; ++u) {   // predicated: the tail reads element 0 and multiplies by 0
          const int rr = r + u * step;
          const bool ok = rr < len;
          a[u] = col[ok ? rr : 0];
          xv[u] = ok ? vec[rr] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < kDotUnroll; ++u) acc = fma((double)a[u], xv[u], acc);
      }
    }
    acc = wave_sum(acc);
    if constexpr (W == 1) {
      if (live && lane == 0 && dot < n_out) out[dot] = acc;
    } else {
      __syncthreads();                         // wsum free again (previous pass)
      if (lane == 0) wsum[wv] = acc;
      __syncthreads();
      if (live && lane == 0 && part == 0 && dot < n_out) {
        double t = wsum[wv];
#pragma unroll
        for (int w = 1; w < W; ++w) t += wsum[wv + w];
        out[dot] = t;
      }
    }
  }
}

template <typename T>
__device__ __forceinline__ void small_dots_w(int W, const T* __restrict__ M, int64_t ld, int n_out, int len,
                                             const double* __restrict__ vec, double* __restrict__ out, bool live) {
  if (W == 1) small_dots<T, 1>(M, ld, n_out, len, vec, out, live);
  else if (W == 2) small_dots<T, 2>(M, ld, n_out, len, vec, out, live);
  else small_dots<T, 4>(M, ld, n_out, len, vec, out, live);
}

// block-wide sums of Q quantities, valid on thread 0 only
template <int Q>
__device__ __forceinline__ void block_sums_t0(double (&v)[Q]) {
  __shared__ double sh[kBlock / 64][Q];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    const double s = wave_sum(v[q]);
    if (lane == 0) sh[wv][q] = s;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int q = 0; q < Q; ++q) v[q] = ((sh[0][q] + sh[1][q]) + sh[2][q]) + sh[3][q];
  }
}

// waves per dot product: <= 16 loads per lane
inline int pick_waves(int len) { return len > 2048 ? 4 : (len > 1024 ? 2 : 1); }   // (64 * kDotUnroll per wave)
inline unsigned pick_grid(int n_out, int W) {
  const int dpb = (kBlock / 64) / W;
  return (unsigned)std::max(1, std::min(1024, (n_out + dpb - 1) / dpb));
}

}  // namespace apx
