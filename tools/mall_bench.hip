// mall_bench.hip — how fast is the SECOND read of a panel that was just streamed from HBM?
// (feasibility probe for Infinity-Cache blocking of the fused A x / A' r iteration)
// build: hipcc -O3 --offload-arch=gfx950 tools/mall_bench.hip -o gpurun_out/mall_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double dvec2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <bool NT>
__global__ __launch_bounds__(256) void read_sum(const double* __restrict__ p, size_t nvec, double* out) {
  const dvec2* v = (const dvec2*)p;
  double s0 = 0, s1 = 0;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * 256;
  for (; i + 3 * stride < nvec; i += 4 * stride) {
    dvec2 a, b, c, d;
    if (NT) { a = __builtin_nontemporal_load(v + i); b = __builtin_nontemporal_load(v + i + stride);
              c = __builtin_nontemporal_load(v + i + 2 * stride); d = __builtin_nontemporal_load(v + i + 3 * stride); }
    else { a = v[i]; b = v[i + stride]; c = v[i + 2 * stride]; d = v[i + 3 * stride]; }
    s0 += a[0] + b[0] + c[0] + d[0];
    s1 += a[1] + b[1] + c[1] + d[1];
  }
  for (; i < nvec; i += stride) { dvec2 a = v[i]; s0 += a[0]; s1 += a[1]; }
  if (s0 + s1 == 123.456) out[0] = s0;   // keep the loads alive
}

int main() {
  const size_t TOTAL = (size_t)4 << 30;   // 4 GiB arena
  double* buf; double* out;
  CK(hipMalloc(&buf, TOTAL)); CK(hipMalloc(&out, 64));
  CK(hipMemset(buf, 0, TOTAL));
  hipStream_t s1, s2; CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
  hipEvent_t e0, e1, e2; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
  const int grid = 256 * 8;
  for (size_t mb : {16, 32, 64, 96, 128, 160, 192, 224, 256, 320}) {
    const size_t bytes = mb << 20, nvec = bytes / 16;
    const int K = (int)(TOTAL / bytes);
    double t_first = 0, t_second = 0, t_third = 0; int cnt = 0;
    for (int rep = 0; rep < 2; ++rep)
      for (int k = 0; k < K; ++k) {
        const double* p = buf + (size_t)k * (bytes / 8);
        CK(hipEventRecord(e0, s1));
        hipLaunchKernelGGL(read_sum<false>, dim3(grid), dim3(256), 0, s1, p, nvec, out);
        CK(hipEventRecord(e1, s1));
        hipLaunchKernelGGL(read_sum<false>, dim3(grid), dim3(256), 0, s1, p, nvec, out);
        CK(hipEventRecord(e2, s1));
        CK(hipStreamSynchronize(s1));
        float a, b; CK(hipEventElapsedTime(&a, e0, e1)); CK(hipEventElapsedTime(&b, e1, e2));
        if (rep == 1) { t_first += a; t_second += b; cnt++; }
      }
    printf("panel %4zu MiB: first read %7.0f GB/s   immediate re-read %7.0f GB/s\n", mb,
           bytes / (t_first / cnt * 1e-3) / 1e9, bytes / (t_second / cnt * 1e-3) / 1e9);
  }
  // overlap: stream 1 re-reads panel k (hot) while stream 2 first-reads panel k+1 (cold)
  for (size_t mb : {64, 96, 128}) {
    const size_t bytes = mb << 20, nvec = bytes / 16;
    const int K = (int)(TOTAL / bytes);
    // prime panel 0
    hipLaunchKernelGGL(read_sum<false>, dim3(grid), dim3(256), 0, s1, buf, nvec, out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, s1));
    for (int k = 0; k + 1 < K; ++k) {
      const double* hot = buf + (size_t)k * (bytes / 8);
      const double* cold = buf + (size_t)(k + 1) * (bytes / 8);
      hipLaunchKernelGGL(read_sum<false>, dim3(grid / 2), dim3(256), 0, s1, hot, nvec, out);
      hipLaunchKernelGGL(read_sum<false>, dim3(grid / 2), dim3(256), 0, s2, cold, nvec, out + 8);
      CK(hipEventRecord(e1, s2)); CK(hipStreamWaitEvent(s1, e1, 0));
      CK(hipEventRecord(e2, s1)); CK(hipStreamWaitEvent(s2, e2, 0));
    }
    CK(hipEventRecord(e2, s1));
    CK(hipDeviceSynchronize());
    float t; CK(hipEventElapsedTime(&t, e0, e2));
    printf("overlap %4zu MiB panels: %d steps, %.3f ms/step, effective (hot+cold bytes)/time = %.0f GB/s, cold-only rate %.0f GB/s\n",
           mb, K - 1, t / (K - 1), 2.0 * bytes / (t / (K - 1) * 1e-3) / 1e9, 1.0 * bytes / (t / (K - 1) * 1e-3) / 1e9);
  }
  return 0;
}
