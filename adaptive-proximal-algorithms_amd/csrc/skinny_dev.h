// skinny_dev.h — pieces shared by the tall-skinny kernels (skinny_pd.hip, skinny_pg.hip, skinny_ls.hip,
// skinny_mp.hip): a thread per row of A, block partials of A'y and of a few scalars, and their fixed-order sum in
// the one-block closing kernels.
#pragma once
#include "small_dev.h"

namespace apx {

constexpr int kSkinnyExtra = 4;                       // scalar partial columns behind the n columns of A'y
using SkinnyRed = double[kSkinnyMaxN + kSkinnyExtra];
constexpr int kSkinnyCloseBlock = 1024;               // threads of the one-block closing kernels

// <A[i,:], x> for row ic, rounded to T like every stored A_x: neighbouring lanes hold neighbouring rows, so each
// column load is coalesced; 8 columns in flight
template <typename T>
__device__ __forceinline__ double skinny_row_dot(const T* __restrict__ A, int64_t ld, int ic, int n,
                                                 const double* __restrict__ xs) {
  const T* __restrict__ row = A + ic;
  double a = 0.0;
  int j = 0;
  for (; j + 8 <= n; j += 8) {
    T av[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) av[u] = row[(int64_t)(j + u) * ld];
#pragma unroll
    for (int u = 0; u < 8; ++u) a = fma((double)av[u], xs[j + u], a);
  }
  for (; j < n; ++j) a = fma((double)row[(int64_t)j * ld], xs[j], a);
  return (double)(T)a;
}

// Wave totals of N per-lane values with N - 1 + log2(64 / N) exchanges instead of 6 N: at each of the first log2 N
// steps a lane keeps half of its values and hands the other half to its partner.  Afterwards v[0] of lane l is the
// total of value l / (64 / N).
template <int H, int N>
__device__ __forceinline__ void wave_sum_multi_step(double (&v)[N], int lane) {
  if constexpr (H >= 1) {                      // (a template per step: every register index is a constant — as a
    constexpr int kOff = 64 * H / N;           //  loop over H this compiled to ~900 v_cndmask, 3 us per call)
    const bool upper = (lane & kOff) != 0;
#pragma unroll
    for (int u = 0; u < H; ++u) {
      const double send = upper ? v[u] : v[u + H];
      const double keep = upper ? v[u + H] : v[u];
      v[u] = keep + __shfl_xor(send, kOff, 64);
    }
    wave_sum_multi_step<H / 2, N>(v, lane);
  }
}
template <int N>
__device__ __forceinline__ void wave_sum_multi(double (&v)[N]) {
  static_assert(N == 1 || N == 2 || N == 4 || N == 8 || N == 16 || N == 32, "power of two");
  wave_sum_multi_step<N / 2, N>(v, threadIdx.x & 63);
#pragma unroll
  for (int off = 32 / N; off >= 1; off >>= 1) v[0] += __shfl_xor(v[0], off, 64);
}

// Block totals of A[i,j] * yn over the block's rows (j < n, when with_aty) and of the kSkinnyExtra scalars ex[]:
// wave shuffles, one LDS row per wave, a barrier, then thread c < n + kSkinnyExtra returns the total of column c
// (waves summed in order).  Threads of rows beyond m pass yn = 0 and zeros.
template <typename T>
__device__ __forceinline__ double skinny_block_partials(const T* __restrict__ A, int64_t ld, int ic, int n, double yn,
                                                        bool with_aty, const double (&ex)[kSkinnyExtra],
                                                        SkinnyRed* wred) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (with_aty) {
    const T* __restrict__ row = A + ic;
    int j = 0;
    for (; j + 8 <= n; j += 8) {               // eight columns per step: eight loads in flight, ten exchanges
      double p[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) p[u] = (double)row[(int64_t)(j + u) * ld] * yn;
      wave_sum_multi<8>(p);                    // lane l: the total of column j + l / 8
      if ((lane & 7) == 0) wred[wv][j + (lane >> 3)] = p[0];
    }
    for (; j < n; ++j) {
      const double p = wave_sum((double)row[(int64_t)j * ld] * yn);
      if (lane == 0) wred[wv][j] = p;
    }
  }
  {
    double e4[kSkinnyExtra];
#pragma unroll
    for (int e = 0; e < kSkinnyExtra; ++e) e4[e] = ex[e];
    wave_sum_multi<kSkinnyExtra>(e4);          // lane l: the total of scalar l / 16
    if ((lane & 15) == 0) wred[wv][n + (lane >> 4)] = e4[0];
  }
  __syncthreads();
  double s = 0.0;
  if (tid < n + kSkinnyExtra) {
    s = wred[0][tid];
#pragma unroll
    for (int w = 1; w < kBlock / 64; ++w) s += wred[w][tid];
  }
  return s;
}

// Blocks s, s + S, s + 2 S, ... of column c summed with sixteen independent accumulators (a plain running sum is a
// chain of dependent L2 round trips, ~1 us each); the order is fixed: accumulator u takes every sixteenth term,
// then a fixed tree.
__device__ __forceinline__ double skinny_strided_sum(const double* __restrict__ part, int64_t ldpart, int c, int s, int S,
                                                     int B) {
  double a[16];
#pragma unroll
  for (int u = 0; u < 16; ++u) a[u] = 0.0;
  int b = s;
  for (; b + 15 * S < B; b += 16 * S) {
#pragma unroll
    for (int u = 0; u < 16; ++u) a[u] += part[(int64_t)(b + u * S) * ldpart + c];
  }
  double tail[4] = {0.0, 0.0, 0.0, 0.0};       // (at most 15 terms left: four chains)
  for (; b + 3 * S < B; b += 4 * S) {
#pragma unroll
    for (int u = 0; u < 4; ++u) tail[u] += part[(int64_t)(b + u * S) * ldpart + c];
  }
  double last = 0.0;
  for (; b < B; b += S) last += part[(int64_t)b * ldpart + c];
  return ((((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]))) +
          (((a[8] + a[9]) + (a[10] + a[11])) + ((a[12] + a[13]) + (a[14] + a[15])))) +
         (((tail[0] + tail[1]) + (tail[2] + tail[3])) + last);
}

// Column c of the block partials summed over the B blocks in a fixed order by all BS threads: thread (c, s) takes
// blocks s, s + S, ... (S = min(BS / npow, 16)), then thread c < nc adds the S sub-sums in order and returns the total.
// In two halves, so that a kernel can issue these loads together with its others BEFORE its first branch: `issue`
// puts the first sixteen terms of every thread in flight, `finish` adds them up (and walks on when B > 16 S).
struct SkinnyColLoads {
  double pa[16];
  int c, sg, S;
  bool on;
};
template <int BS>
__device__ __forceinline__ void skinny_colsum_issue(const double* __restrict__ part, int64_t ldpart, int nc, int npow, int B,
                                                    SkinnyColLoads& L) {
  const int tid = threadIdx.x;
  L.S = min(BS / npow, 16);
  L.c = tid % npow;
  L.sg = tid / npow;
  L.on = L.c < nc && L.sg < L.S;
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    const int b = L.sg + u * L.S;
    L.pa[u] = L.on && b < B ? part[(int64_t)b * ldpart + L.c] : 0.0;
  }
}
__device__ __forceinline__ double skinny_colsum_finish(const double* __restrict__ part, int64_t ldpart, int nc, int B,
                                                       const SkinnyColLoads& L, SkinnyRed* red) {
  const int tid = threadIdx.x;
  if (L.on) {
    const double* pa = L.pa;
    double t = (((pa[0] + pa[1]) + (pa[2] + pa[3])) + ((pa[4] + pa[5]) + (pa[6] + pa[7]))) +
               (((pa[8] + pa[9]) + (pa[10] + pa[11])) + ((pa[12] + pa[13]) + (pa[14] + pa[15])));
    if (B > 16 * L.S) t += skinny_strided_sum(part, ldpart, L.c, L.sg + 16 * L.S, L.S, B);
    red[L.sg][L.c] = t;
  }
  __syncthreads();
  double tot = 0.0;
  if (tid < nc)
    for (int k = 0; k < L.S; ++k) tot += red[k][tid];
  return tot;
}
template <int BS>
__device__ __forceinline__ double skinny_colsum(const double* __restrict__ part, int64_t ldpart, int nc, int npow, int B,
                                                SkinnyRed* red) {
  SkinnyColLoads L;
  skinny_colsum_issue<BS>(part, ldpart, nc, npow, B, L);
  return skinny_colsum_finish(part, ldpart, nc, B, L, red);
}

// sqrt of the sum over the B blocks of column `col` of the block partials (the l2 ball's ||w/sigma + shift||),
// formed by a 256-thread block itself in a fixed order — every block of a launch gets the same bits, and the norm
// needs no launch of its own.  `issue` puts the loads in flight, `finish` (barriers inside) returns it to all threads.
__device__ __forceinline__ void skinny_partial_norm_issue(const double* __restrict__ part, int64_t ldpart, int col, int B,
                                                          double (&z4)[4]) {
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int b = (int)threadIdx.x + u * kBlock;
    z4[u] = b < B ? part[(int64_t)b * ldpart + col] : 0.0;
  }
}
__device__ __forceinline__ double skinny_partial_norm_finish(const double* __restrict__ part, int64_t ldpart, int col, int B,
                                                             const double (&z4)[4]) {
  __shared__ double out;
  double zz[1] = {(z4[0] + z4[1]) + (z4[2] + z4[3])};
  for (int b = (int)threadIdx.x + 4 * kBlock; b < B; b += kBlock) zz[0] += part[(int64_t)b * ldpart + col];
  block_sums_t0<1>(zz);
  if (threadIdx.x == 0) out = sqrt(zz[0]);
  __syncthreads();
  return out;
}

inline int skinny_npow(int n) {
  int npow = 16;
  while (npow < n + kSkinnyExtra) npow *= 2;
  return npow;
}

}  // namespace apx
