// lds_probe.hip — how much LDS can one workgroup of this device allocate (static limit, opt-in dynamic limit)?
// build: hipcc -O2 --offload-arch=gfx950 tools/lds_probe.hip -o gpurun_out/lds_probe
#include <hip/hip_runtime.h>
#include <cstdio>
extern __shared__ double dyn[];
__global__ void k(double* out, int n) {
  for (int i = threadIdx.x; i < n; i += blockDim.x) dyn[i] = i;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = dyn[n - 1];
}
int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  int optin = 0, maxblk = 0;
  hipDeviceGetAttribute(&maxblk, hipDeviceAttributeMaxSharedMemoryPerBlock, 0);
  hipDeviceGetAttribute(&optin, hipDeviceAttributeSharedMemPerBlockOptin, 0);
  printf("%s: sharedMemPerBlock %zu  maxSharedMemoryPerMultiProcessor %zu  attr MaxSharedMemoryPerBlock %d  optin %d\n", p.name,
         p.sharedMemPerBlock, p.maxSharedMemoryPerMultiProcessor, maxblk, optin);
  double* out;
  hipMalloc(&out, 8);
  for (int kb : {32, 64, 96, 128, 144, 160}) {
    hipError_t e1 = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, kb * 1024);
    hipLaunchKernelGGL(k, dim3(1), dim3(256), kb * 1024, 0, out, kb * 128);
    hipError_t e2 = hipGetLastError();
    hipError_t e3 = hipDeviceSynchronize();
    double h = -1;
    hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost);
    printf("%3d KB dynamic: setattr %s, launch %s, sync %s, last element %.0f (expect %d)\n", kb, hipGetErrorString(e1),
           hipGetErrorString(e2), hipGetErrorString(e3), h, kb * 128 - 1);
  }
  return 0;
}
