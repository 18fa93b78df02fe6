// stream_pattern_probe.hip — pure reads of 40 GB (no arithmetic, no synchronisation) in the address patterns the one-pass row
// kernels can generate, to tell what the PATTERN costs apart from what the kernel costs (round 3: the deal of the rows).
//   slab      : block (team, member) streams the member's 50 KB slice of every row of the team's SLAB of consecutive rows
//   chunkR    : ... of the team's rows dealt in chunks of R rows, the chunks cyclically (csrc/team_dev.h deal_row)
//   blockedR  : the same chunks, but the member's R slices of a chunk stored CONTIGUOUSLY (a member-blocked copy of A)
// build: hipcc -O3 --offload-arch=gfx950 tools/stream_pattern_probe.hip -o gpurun_out/stream_pattern_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double dvec2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

// rows of rowv vectors ; P members per team ; member slice = rowv / P vectors ; T teams = gridDim.x / P
// mode 0 slab, 1 chunked (R rows per chunk), 2 member-blocked chunks
template <int U>
__global__ __launch_bounds__(768) void reader(const dvec2* __restrict__ v, int m, int rowv, int P, int R, int mode, double* out) {
  const int T = gridDim.x / P, team = blockIdx.x / P, s = blockIdx.x % P;
  const int nvs = rowv / P;
  const int BS = blockDim.x;
  double s0 = 0, s1 = 0;
  const int rpt = (m + T - 1) / T;
  const int chunks = (m + R - 1) / R;
  const int mine = team < chunks ? (chunks - team + T - 1) / T : 0;
  const int nrows = mode == 0 ? max(min((team + 1) * rpt, m) - team * rpt, 0) : mine * R;
  for (int k = 0; k < nrows; ++k) {
    const dvec2* p;
    if (mode == 0) p = v + (size_t)(team * rpt + k) * rowv + (size_t)s * nvs;
    else {
      const int c = (k / R) * T + team, rin = k % R;
      const int row = min(c * R + rin, m - 1);
      if (mode == 1) p = v + (size_t)row * rowv + (size_t)s * nvs;
      else if (mode == 2) p = v + ((size_t)c * P + s) * (size_t)R * nvs + (size_t)rin * nvs;      // [chunk][member][row in chunk][slice]
      else p = v + (size_t)row * rowv;               // 3: the members' slices INTERLEAVED in pieces of one workgroup's load (BS vectors)
    }
    if (mode == 3) {          // piece j of the row goes to member j % P: per load instruction the team covers P * BS contiguous vectors
      for (int j = s; j * BS < rowv; j += U * P) {
        dvec2 a[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { const int i = (j + u * P) * BS + threadIdx.x; a[u] = (i < rowv) ? __builtin_nontemporal_load(p + i) : dvec2{0, 0}; }
#pragma unroll
        for (int u = 0; u < U; ++u) { s0 += a[u][0]; s1 += a[u][1]; }
      }
      continue;
    }
    for (int i = threadIdx.x; i < nvs; i += U * BS) {
      dvec2 a[U];
#pragma unroll
      for (int u = 0; u < U; ++u) a[u] = (i + u * BS < nvs) ? __builtin_nontemporal_load(p + i + u * BS) : dvec2{0, 0};
#pragma unroll
      for (int u = 0; u < U; ++u) { s0 += a[u][0]; s1 += a[u][1]; }
    }
  }
  if (s0 + s1 == 123.456) out[0] = s0;
}

template <int U = 5>
static double run(const dvec2* buf, int m, int rowv, int P, int R, int mode, int threads, double* out, hipStream_t s, int grid = 256) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipEventRecord(e0, s));
    hipLaunchKernelGGL((reader<U>), dim3(grid), dim3(threads), 0, s, buf, m, rowv, P, R, mode, out);
    CK(hipEventRecord(e1, s));
    CK(hipStreamSynchronize(s));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep > 0 && ms < best) best = ms;
  }
  CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
  return (double)m * rowv * 16 / (best * 1e-3) / 1e9;
}

int main(int argc, char** argv) {
  // C2: 50 000 rows of 50 000 vectors (800 KB), teams of 16 ; C5: 50 000 rows of 10 000 vectors (160 KB), teams of 2
  struct Cfg { const char* name; int m, rowv, P; } cfgs[] = {{"C2 rows 800 KB, teams of 16", 50000, 50000, 16}, {"C5 rows 160 KB, teams of 2", 50000, 10000, 2},
                                                            {"C3 rows 40 KB (fp32), one workgroup", 100000, 2500, 1}};
  hipStream_t s; CK(hipStreamCreate(&s));
  double* out; CK(hipMalloc(&out, 64));
  if (argc > 1) {      // second set (round 3): would FEWER, wider members or interleaved slices read C2's rows faster ?  and the harness's ceiling
    for (int rep = 0; rep < 2; ++rep) {
      const Cfg c = cfgs[0];
      dvec2* buf;
      const size_t bytes = (size_t)c.m * c.rowv * 16 + (64 << 20);
      CK(hipMalloc(&buf, bytes));
      CK(hipMemset(buf, 0, bytes));
      printf("%s (allocation %d), chunk8, members per team:", c.name, rep);
      for (int P : {1, 2, 4, 8, 16, 32}) printf(" P%-2d %5.0f", P, run(buf, c.m, c.rowv, P, 8, 1, 640, out, s));
      printf("\n  the members' slices interleaved in pieces of 10 KB (chunk8):");
      for (int P : {2, 4, 8, 16}) printf(" P%-2d %5.0f", P, run(buf, c.m, c.rowv, P, 8, 3, 640, out, s));
      printf("\n  ... chunk1:");
      for (int P : {2, 4, 8, 16}) printf(" P%-2d %5.0f", P, run(buf, c.m, c.rowv, P, 1, 3, 640, out, s));
      // the ceiling: the whole buffer as ONE row dealt in 10 KB pieces to 256 members (a linear stream of the chip), and 256 contiguous 1/256ths
      printf("\n  the buffer as one linear stream of the chip (10 KB pieces dealt to 256 workgroups) %5.0f ; as 256 contiguous parts %5.0f\n",
             run(buf, 2, c.m / 2 * c.rowv, 256, 1, 3, 640, out, s), run(buf, 2, c.m / 2 * c.rowv, 256, 1, 1, 640, out, s));
      fflush(stdout);
      CK(hipFree(buf));
    }
    return 0;
  }
  for (int rep = 0; rep < 2; ++rep)
    for (const Cfg& c : cfgs) {
      dvec2* buf;
      const size_t bytes = (size_t)c.m * c.rowv * 16 + (64 << 20);
      CK(hipMalloc(&buf, bytes));
      CK(hipMemset(buf, 0, bytes));
      printf("%s (allocation %d): 256 workgroups of 640 threads, pure non-temporal reads, GB/s\n", c.name, rep);
      printf("  slab %5.0f", run(buf, c.m, c.rowv, c.P, 1, 0, 640, out, s));
      for (int R : {1, 2, 4, 8, 16, 32}) printf(" | chunk%-2d %5.0f blocked %5.0f", R, run(buf, c.m, c.rowv, c.P, R, 1, 640, out, s), run(buf, c.m, c.rowv, c.P, R, 2, 640, out, s));
      printf("\n");
      // the same bytes in flight per CU dealt to MORE, thinner workgroups (members): 512 x 320 threads, 1024 x 192 (slices half / a quarter as wide)
      printf("  chunk8, workgroups x threads (members per team): 256 x 640 (%d) %5.0f | 512 x 320 (%d) %5.0f | 1024 x 192 (%d) %5.0f | 512 x 640 (%d) %5.0f\n", c.P,
             run(buf, c.m, c.rowv, c.P, 8, 1, 640, out, s, 256), 2 * c.P, run(buf, c.m, c.rowv, 2 * c.P, 8, 1, 320, out, s, 512), 4 * c.P,
             run(buf, c.m, c.rowv, 4 * c.P, 8, 1, 192, out, s, 1024), 2 * c.P, run(buf, c.m, c.rowv, 2 * c.P, 8, 1, 640, out, s, 512));
      // more loads in flight per thread (the product keeps two to three rows = 10 ... 16 vectors per thread in flight)
      printf("  chunk8 / blocked8 with U vectors per thread in flight: U5 %5.0f / %5.0f | U10 %5.0f / %5.0f | U16 %5.0f / %5.0f | slab U10 %5.0f U16 %5.0f\n",
             run<5>(buf, c.m, c.rowv, c.P, 8, 1, 640, out, s), run<5>(buf, c.m, c.rowv, c.P, 8, 2, 640, out, s), run<10>(buf, c.m, c.rowv, c.P, 8, 1, 640, out, s),
             run<10>(buf, c.m, c.rowv, c.P, 8, 2, 640, out, s), run<16>(buf, c.m, c.rowv, c.P, 8, 1, 640, out, s), run<16>(buf, c.m, c.rowv, c.P, 8, 2, 640, out, s),
             run<10>(buf, c.m, c.rowv, c.P, 1, 0, 640, out, s), run<16>(buf, c.m, c.rowv, c.P, 1, 0, 640, out, s));
      fflush(stdout);
      CK(hipFree(buf));
    }
  return 0;
}
