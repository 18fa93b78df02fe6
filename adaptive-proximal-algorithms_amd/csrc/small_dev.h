// small_dev.h — pieces shared by the two-kernel small-problem iterations (small.hip: proximal gradient,
// small_pd.hip: primal-dual): in-block dot products with the vector operand in LDS, block sums, grid sizing.
#pragma once
#include <algorithm>
#include <atomic>

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

// `nblk`: the blocks [0, nblk) of the launch share the dot products (0: all of them).  A kernel whose grid carries
// other blocks behind them must say so: striding by gridDim.x would skip outputs as soon as a second pass is needed.
template <typename T, int W, int BS = kBlock>
__device__ __forceinline__ void small_dots(const T* __restrict__ M, int64_t ld, int n_out, int len,
                                           const double* __restrict__ vec, double* __restrict__ out, bool live,
                                           int nblk = 0) {
  constexpr int DPB = (BS / 64) / W;
  __shared__ double wsum[BS / 64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int part = wv % W;
  constexpr int step = W * 64;
  const int nb = nblk > 0 ? nblk : (int)gridDim.x;
  for (int d0 = blockIdx.x * DPB; d0 < n_out; d0 += nb * DPB) {
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

template <typename T, int BS = kBlock>
__device__ __forceinline__ void small_dots_w(int W, const T* __restrict__ M, int64_t ld, int n_out, int len,
                                             const double* __restrict__ vec, double* __restrict__ out, bool live,
                                             int nblk = 0) {
  if (W == 1) small_dots<T, 1, BS>(M, ld, n_out, len, vec, out, live, nblk);
  else if (W == 2) small_dots<T, 2, BS>(M, ld, n_out, len, vec, out, live, nblk);
  else if (W == 4 || BS / 64 <= 4) small_dots<T, 4, BS>(M, ld, n_out, len, vec, out, live, nblk);
  else if constexpr (BS / 64 >= 16) {         // 1024-thread blocks (tall matrices): up to 16 waves per dot
    if (W == 8) small_dots<T, 8, BS>(M, ld, n_out, len, vec, out, live, nblk);
    else small_dots<T, 16, BS>(M, ld, n_out, len, vec, out, live, nblk);
  }
}

// block-wide sums of Q quantities, valid on thread 0 only.  BS must be the block size of the calling kernel: the
// staging rows and the (unrolled) final sum are sized by it; waves beyond BS / 64 are not stored, so a mismatch
// cannot write outside the staging rows.
template <int Q, int BS = kBlock>
__device__ __forceinline__ void block_sums_t0(double (&v)[Q]) {
  __shared__ double sh[BS / 64][Q];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    const double s = wave_sum(v[q]);
    if (lane == 0 && wv < BS / 64) sh[wv][q] = s;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      double t = sh[0][q];
#pragma unroll
      for (int w = 1; w < BS / 64; ++w) t += sh[w][q];
      v[q] = t;
    }
  }
}

// a workgroup of gfx950 may allocate all 160 KB of LDS (tools/lds_probe.hip); dynamic allocations above 64 KB
// are opted into per kernel — 144 KB: the kernels also hold a few KB of static LDS, and dynamic + static
// beyond 160 KB is refused
// (the attribute belongs to the function ON A DEVICE: once per device of the process, from whichever host thread drives it)
template <auto KERNEL>
inline void allow_big_lds() {
  static std::atomic<unsigned long long> done_devs{0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  const unsigned long long bit = 1ull << (dev & 63);
  if ((done_devs.load(std::memory_order_relaxed) & bit) == 0) {
    if (hipFuncSetAttribute((const void*)KERNEL, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024) != hipSuccess)
      (void)hipGetLastError();       // (not sticky: a launch that really needs more than the default fails by itself)
    done_devs.fetch_or(bit, std::memory_order_relaxed);
  }
}
inline size_t m_vector_lds_bytes(int m) { return (size_t)round_up(m, 32) * sizeof(double); }
// tall matrices are taken while they stay small in bytes (beyond that the products are bandwidth-bound and
// the generic kernels are the right tool)
inline bool small_shape_ok(int64_t m, int64_t n) {
  return m > 0 && n > 0 && n <= kSmallMax && m <= kSmallMaxM && (m <= kSmallMax || m * n <= (int64_t)1 << 22);
}

// waves per dot product: <= 16 loads per lane
inline int pick_waves(int len, int bs = kBlock) {      // (64 * kDotUnroll elements per wave and round trip)
  int w = 1;
  while (w < bs / 64 && len > 64 * kDotUnroll * w) w *= 2;
  return w;
}
inline unsigned pick_grid(int n_out, int W, int bs = kBlock) {
  const int dpb = (bs / 64) / W;
  return (unsigned)std::max(1, std::min(1024, (n_out + dpb - 1) / dpb));
}
// tall matrices (m > kSmallMax): the kernel that does the O(m) work redundantly runs 1024-thread blocks
constexpr int kTallBlock = 1024;
inline int m_side_block(int m) { return m > kSmallMax ? kTallBlock : kBlock; }

}  // namespace apx
