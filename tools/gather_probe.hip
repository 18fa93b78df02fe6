// gather_probe.hip — where is the ceiling of the sparse products' access pattern on MI355X ?  
// A sparse `A*x` / `A'*r` in compressed rows streams nnz * (sizeof(T) + 4) bytes of values and indices and GATHERS one operand
// entry per stored entry.  This probe takes the row structure away and measures the three ingredients on 60 M entries
// (profiles/r01f_sparse_bench.jsonl's large case), fp64 values, int32 indices:
//   stream      values + indices only, 16-byte loads, nothing gathered                     -> what HBM gives the 12 B / entry
//   gather(N)   + one 8-byte gather per entry from an operand of N doubles, uniformly random -> what the L2 / MALL gather costs
//                 (N = 100 000: 0.8 MB, L2-resident ; N = 1 000 000: 8 MB, beyond an XCD's 4 MB L2)
//   lds(N)      the same with the operand in LDS tiles of 16 384 doubles (indices taken modulo the tile: the blocked layout's
//                 best case — every gather a bank-conflicted LDS read instead of an L2 transaction) ; with blocks of 256 lanes, ONE
//                 per CU beside 128 KB of LDS, four waves cannot keep the stream going (2.3 TB/s) — the extra rows repeat it with
//                 blocks of 1024 lanes (16 waves per CU): 5.4 - 5.9 TB/s, the rate of the stream alone.  csrc/sparse.hip's
//                 spmv_tiles is built on that row.
//   sorted(N)   gather with indices sorted within each wave's 64-entry stride (neighbouring lanes hit neighbouring lines:
//                 what a row-local column ordering could buy)
// Prints GB/s of 12 B / entry.   hipcc -O3 --offload-arch=gfx950 tools/gather_probe.hip -o gather_probe && ./gather_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(_e), __LINE__); exit(1); } } while (0)

template <int MODE, int U, int kBlock = 256, int kTile = 16384>      // MODE 0 stream, 1 gather from global, 2 gather from LDS tile
__global__ __launch_bounds__(kBlock) void k_probe(const int32_t* __restrict__ ind, const double* __restrict__ val, const double* __restrict__ x,
                                                  int64_t nnz, int n_x, double* __restrict__ out) {
  extern __shared__ double tile[];
  if (MODE == 2) {
    for (int i = threadIdx.x; i < kTile; i += kBlock) tile[i] = x[i % n_x];
    __syncthreads();
  }
  double s = 0.0;
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x; k < nnz; k += stride * U) {
    int32_t j[U];
    double v[U], xv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t kk = k + u * stride;
      const bool ok = kk < nnz;
      j[u] = ok ? ind[kk] : 0;
      v[u] = ok ? val[kk] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (MODE == 0) xv[u] = (double)j[u];
      else if (MODE == 1) xv[u] = x[j[u]];
      else xv[u] = tile[j[u] & (kTile - 1)];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) s = fma(v[u], xv[u], s);
  }
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if ((threadIdx.x & 63) == 0) out[(int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6)] = s;
}

template <int MODE, int U, int kBlock = 256, int kTile = 16384>
static double run(const int32_t* ind, const double* val, const double* x, int64_t nnz, int n_x, double* out, int blocks) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  const size_t lds = MODE == 2 ? kTile * sizeof(double) : 0;
  if (MODE == 2) CK(hipFuncSetAttribute((const void*)k_probe<MODE, U, kBlock, kTile>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k_probe<MODE, U, kBlock, kTile>), dim3(blocks), dim3(kBlock), lds, 0, ind, val, x, nnz, n_x, out);
  CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((k_probe<MODE, U, kBlock, kTile>), dim3(blocks), dim3(kBlock), lds, 0, ind, val, x, nnz, n_x, out);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    best = std::min(best, ms);
  }
  return (double)nnz * 12.0 / (best * 1e-3) / 1e9;
}

int main() {
  const int64_t nnz = 60'000'000;
  std::vector<double> hv(nnz);
  std::mt19937_64 rng(1);
  for (auto& v : hv) v = (double)(rng() >> 11) * (1.0 / 9007199254740992.0);
  double *val, *x, *out;
  int32_t* ind;
  CK(hipMalloc(&val, nnz * 8));
  CK(hipMalloc(&ind, nnz * 4));
  CK(hipMalloc(&x, 16'000'000 * 8));
  CK(hipMalloc(&out, 1 << 22));
  CK(hipMemcpy(val, hv.data(), nnz * 8, hipMemcpyHostToDevice));
  CK(hipMemset(x, 0, 16'000'000 * 8));
  std::vector<int32_t> hi(nnz);
  for (int blocks : {1024, 2048, 4096, 8192}) {
    printf("grid %d blocks of 256:\n", blocks);
    printf("  stream only                U=4 %7.0f  U=8 %7.0f GB/s\n", run<0, 4>(ind, val, x, nnz, 1, out, blocks), run<0, 8>(ind, val, x, nnz, 1, out, blocks));
    for (int n_x : {2000, 100'000, 262'144, 524'288, 1'000'000, 16'000'000}) {
      for (auto& j : hi) j = (int32_t)(rng() % (uint64_t)n_x);
      CK(hipMemcpy(ind, hi.data(), nnz * 4, hipMemcpyHostToDevice));
      printf("  gather, operand %9d  U=4 %7.0f  U=8 %7.0f  U=16 %7.0f GB/s", n_x, run<1, 4>(ind, val, x, nnz, n_x, out, blocks),
             run<1, 8>(ind, val, x, nnz, n_x, out, blocks), run<1, 16>(ind, val, x, nnz, n_x, out, blocks));
      printf("   | LDS tile U=8 %7.0f\n", run<2, 8>(ind, val, x, nnz, n_x, out, blocks));
      if (blocks == 2048 && n_x == 100'000) {      // the LDS tile with full occupancy: blocks of 1024 lanes, one or two per CU
        printf("          LDS tile, blocks of 1024: 128 KB x 256 blocks U=4 %7.0f U=8 %7.0f | x 512 U=8 %7.0f | 64 KB x 512 U=4 %7.0f U=8 %7.0f | x 1024 U=8 %7.0f\n",
               run<2, 4, 1024, 16384>(ind, val, x, nnz, n_x, out, 256), run<2, 8, 1024, 16384>(ind, val, x, nnz, n_x, out, 256),
               run<2, 8, 1024, 16384>(ind, val, x, nnz, n_x, out, 512), run<2, 4, 1024, 8192>(ind, val, x, nnz, n_x, out, 512),
               run<2, 8, 1024, 8192>(ind, val, x, nnz, n_x, out, 512), run<2, 8, 1024, 8192>(ind, val, x, nnz, n_x, out, 1024));
        printf("          global gather, blocks of 1024 x 512 U=8 %7.0f | stream, 1024 x 512 U=8 %7.0f\n",
               run<1, 8, 1024>(ind, val, x, nnz, n_x, out, 512), run<0, 8, 1024>(ind, val, x, nnz, n_x, out, 512));
      }
      // indices sorted within every run of 64 consecutive entries (one wave's lanes of one load)
      for (int64_t k = 0; k + 64 <= nnz; k += 64) std::sort(hi.begin() + k, hi.begin() + k + 64);
      CK(hipMemcpy(ind, hi.data(), nnz * 4, hipMemcpyHostToDevice));
      printf("          sorted per wave-load  U=8 %7.0f GB/s\n", run<1, 8>(ind, val, x, nnz, n_x, out, blocks));
      if (blocks != 2048) break;      // (the operand sweep once)
    }
  }
  return 0;
}
