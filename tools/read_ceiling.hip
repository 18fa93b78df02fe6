// read_ceiling.hip — what is the fastest a kernel can READ 40 GB out of HBM on this part?
// The two GEMV passes of the iteration are pure streaming reads of A (the vector operand is L2-resident),
// so the best rate found here is the ceiling their roofline fraction should be judged against besides the
// 8 TB/s spec number.  Sweeps: access pattern (block-contiguous chunk / grid-strided 4 KiB pieces / one
// "column" of 400 KB per block pass like gemv_t), loads in flight per lane, grid size, cache policy.
// build: hipcc -O3 --offload-arch=gfx950 tools/read_ceiling.hip -o gpurun_out/read_ceiling
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double dvec2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <bool NT>
__device__ __forceinline__ dvec2 ld(const dvec2* p) {
  if constexpr (NT) return __builtin_nontemporal_load(p);
  else return *p;
}

// PATTERN 0: grid-strided: piece j of 4 KiB (256 lanes x 16 B) goes to block j % G
// PATTERN 1: block-contiguous: block b owns [b * chunk, (b + 1) * chunk)
// PATTERN 2: "columns": the buffer is ncol columns of colv vectors; block b takes columns b, b + G, ...
template <int PATTERN, int U, bool NT>
__global__ __launch_bounds__(256) void reader(const dvec2* __restrict__ v, size_t nvec, size_t colv, double* out) {
  double s0 = 0, s1 = 0;
  if constexpr (PATTERN == 0) {
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < nvec; i += U * stride) {
      dvec2 a[U];
#pragma unroll
      for (int u = 0; u < U; ++u) a[u] = ld<NT>(v + i + u * stride);
#pragma unroll
      for (int u = 0; u < U; ++u) { s0 += a[u][0]; s1 += a[u][1]; }
    }
  } else if constexpr (PATTERN == 1) {
    const size_t chunk = nvec / gridDim.x;
    const dvec2* p = v + (size_t)blockIdx.x * chunk;
    size_t i = threadIdx.x;
    for (; i + (U - 1) * 256 < chunk; i += U * 256) {
      dvec2 a[U];
#pragma unroll
      for (int u = 0; u < U; ++u) a[u] = ld<NT>(p + i + u * 256);
#pragma unroll
      for (int u = 0; u < U; ++u) { s0 += a[u][0]; s1 += a[u][1]; }
    }
  } else {
    const size_t ncol = nvec / colv;
    for (size_t c = blockIdx.x; c < ncol; c += gridDim.x) {
      const dvec2* p = v + c * colv;
      size_t i = threadIdx.x;
      for (; i + (U - 1) * 256 < colv; i += U * 256) {
        dvec2 a[U];
#pragma unroll
        for (int u = 0; u < U; ++u) a[u] = ld<NT>(p + i + u * 256);
#pragma unroll
        for (int u = 0; u < U; ++u) { s0 += a[u][0]; s1 += a[u][1]; }
      }
    }
  }
  if (s0 + s1 == 123.456) out[0] = s0;   // keep the loads alive
}

template <int PATTERN, int U, bool NT>
static double run(const dvec2* buf, size_t nvec, size_t colv, double* out, int grid, hipStream_t s) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((reader<PATTERN, U, NT>), dim3(grid), dim3(256), 0, s, buf, nvec, colv, out);
  float best = 1e30f;
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipEventRecord(e0, s));
    hipLaunchKernelGGL((reader<PATTERN, U, NT>), dim3(grid), dim3(256), 0, s, buf, nvec, colv, out);
    CK(hipEventRecord(e1, s));
    CK(hipStreamSynchronize(s));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
  return (double)nvec * 16 / (best * 1e-3) / 1e9;
}

int main() {
  const size_t colv = 50016 / 2;                       // one column of the C2 matrix: 50 016 doubles
  const size_t ncol = 100000;
  const size_t nvec = colv * ncol;                     // 40.0 GB
  dvec2* buf; double* out;
  CK(hipMalloc(&buf, nvec * 16)); CK(hipMalloc(&out, 64));
  CK(hipMemset(buf, 0, nvec * 16));
  hipStream_t s; CK(hipStreamCreate(&s));
  printf("reading %.1f GB; best of 4 launches each\n", nvec * 16 / 1e9);
  for (int grid : {1024, 2048, 4096, 8192, 16384}) {
    printf("grid %5d | strided  U4 %5.0f  U8 %5.0f  U16 %5.0f | strided NT  U4 %5.0f  U8 %5.0f  U16 %5.0f\n", grid,
           run<0, 4, false>(buf, nvec, colv, out, grid, s), run<0, 8, false>(buf, nvec, colv, out, grid, s),
           run<0, 16, false>(buf, nvec, colv, out, grid, s), run<0, 4, true>(buf, nvec, colv, out, grid, s),
           run<0, 8, true>(buf, nvec, colv, out, grid, s), run<0, 16, true>(buf, nvec, colv, out, grid, s));
    printf("           | contig   U4 %5.0f  U8 %5.0f  U16 %5.0f | contig  NT  U4 %5.0f  U8 %5.0f  U16 %5.0f\n",
           run<1, 4, false>(buf, nvec, colv, out, grid, s), run<1, 8, false>(buf, nvec, colv, out, grid, s),
           run<1, 16, false>(buf, nvec, colv, out, grid, s), run<1, 4, true>(buf, nvec, colv, out, grid, s),
           run<1, 8, true>(buf, nvec, colv, out, grid, s), run<1, 16, true>(buf, nvec, colv, out, grid, s));
    printf("           | columns  U4 %5.0f  U8 %5.0f  U16 %5.0f | columns NT  U4 %5.0f  U8 %5.0f  U16 %5.0f   (GB/s)\n",
           run<2, 4, false>(buf, nvec, colv, out, grid, s), run<2, 8, false>(buf, nvec, colv, out, grid, s),
           run<2, 16, false>(buf, nvec, colv, out, grid, s), run<2, 4, true>(buf, nvec, colv, out, grid, s),
           run<2, 8, true>(buf, nvec, colv, out, grid, s), run<2, 16, true>(buf, nvec, colv, out, grid, s));
    fflush(stdout);
  }
  return 0;
}
