// slab2_probe.hip — round-2 probe for the one-pass row kernel (csrc/slab.hip): what does the gap between its 6.2 TB/s and
// the 7.2 TB/s of a pure read consist of?  Synthetic least-squares data in the shape of BASELINE config C3
// (100 000 x 10 000 fp32, rows contiguous as in the transposed copy), result checked against a host evaluation.
//
// Variants (all: one block per CU walks a slab of rows, x and the column accumulator in registers, A read once):
//   v1   the mapping of round 1: thread t holds columns t, t + BS, ... (4-byte loads, 256 B per wave-load), two groups
//        of rows that alternate, two barriers per group
//   v2   column-BLOCKED mapping: thread t holds the 16-byte vectors t, t + BS, ... of a row (1 KiB per wave-load) through
//        buffer loads (scalar row descriptor + one per-lane offset register for the whole kernel + scalar offsets: no
//        per-column address registers), G groups of H rows in flight, ONE barrier per group (every thread sums the waves'
//        partial dots itself; the partials are double-buffered)
// Build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/slab2_probe.hip -o tools/slab2_probe
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(e)                                                                          \
  do {                                                                                 \
    hipError_t _e = (e);                                                               \
    if (_e != hipSuccess) {                                                            \
      fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(_e));                          \
      exit(1);                                                                         \
    }                                                                                  \
  } while (0)

typedef float v4f __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

__host__ __device__ inline float gen(uint64_t seed, uint64_t idx) {
  uint64_t z = seed * 0x9E3779B97F4A7C15ULL + idx;
  z += 0x9E3779B97F4A7C15ULL;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  z = z ^ (z >> 31);
  return (float)(-1.0 + 2.0 * ((double)(z >> 11) * (1.0 / 9007199254740992.0)));
}

__global__ void k_fill(float* AT, int64_t ldT, int m, int n) {
  const int64_t i = blockIdx.y;
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x)
    AT[i * ldT + j] = gen(1, (uint64_t)i * (uint64_t)n + (uint64_t)j);
}

struct Args {
  const float* AT;
  int64_t ldT;
  int m, n, rows_per_block;
  const float* b;
  const float* x;
  double* part;      // [blocks][ldpart]
  int64_t ldpart;
};

// ---------------------------------------------------------------------------------------------------------------------
// v1: as csrc/slab.hip (round 1), least squares, fp32, PIPE, NT
// ---------------------------------------------------------------------------------------------------------------------
template <int C, int H, int BS>
__global__ __launch_bounds__(BS) void k_v1(Args d) {
  __shared__ double wsum[BS / 64][H];
  __shared__ double res_s[H];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  float xr[C];
  double acc[C];
  unsigned off[C];
#pragma unroll
  for (int k = 0; k < C; ++k) {
    const int j = tid + k * BS;
    off[k] = (unsigned)min(j, d.n - 1) * 4u;
    xr[k] = j < d.n ? d.x[j] : 0.f;
    acc[k] = 0.0;
  }
  double e0 = 0.0;
  const int row_begin = blockIdx.x * d.rows_per_block;
  const int row_end = min(row_begin + d.rows_per_block, d.m);
  float a0[H][C], a1[H][C];
  float b0, b1;
  auto load = [&](float(&a)[H][C], float& b_in, int r0) {
#pragma unroll
    for (int r = 0; r < H; ++r) {
      const int ri = __builtin_amdgcn_readfirstlane(min(r0 + r, row_end - 1));
      const float* row = d.AT + (int64_t)ri * d.ldT;
#pragma unroll
      for (int k = 0; k < C; ++k) a[r][k] = __builtin_nontemporal_load((const float*)((const char*)row + off[k]));
    }
    b_in = d.b[min(r0 + min(tid, H - 1), row_end - 1)];
  };
  auto group = [&](const float(&a)[H][C], float b_in, int r0) {
    double p[H];
#pragma unroll
    for (int r = 0; r < H; ++r) p[r] = 0.0;
#pragma unroll
    for (int k = 0; k < C; ++k) {
      float xv = xr[k];
      asm volatile("" : "+v"(xv));
      const double xd = (double)xv;
#pragma unroll
      for (int r = 0; r < H; ++r) p[r] = fma((double)a[r][k], xd, p[r]);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int r = 0; r < H; ++r) {
      p[r] = wave_sum(p[r]);
      if (lane == 0) wsum[wv][r] = p[r];
    }
    __syncthreads();
    if (tid < H) {
      double dot = wsum[0][tid];
#pragma unroll
      for (int w = 1; w < BS / 64; ++w) dot += wsum[w][tid];
      const int i = r0 + tid;
      double res = 0.0;
      if (i < row_end) {
        res = (double)(float)((double)(float)dot - (double)b_in);
        e0 += res * res;
      }
      res_s[tid] = res;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < H; ++r) {
      const double rr = res_s[r];
#pragma unroll
      for (int k = 0; k < C; ++k) {
        float av = a[r][k];
        asm volatile("" : "+v"(av));
        acc[k] = fma((double)av, rr, acc[k]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int k = 0; k < C; ++k) asm volatile("" : "+v"(acc[k]));
  };
  __builtin_amdgcn_sched_barrier(0);
  load(a0, b0, row_begin);
  __builtin_amdgcn_sched_barrier(0);
  load(a1, b1, row_begin + H);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
  for (int r0 = row_begin; r0 < row_end; r0 += 2 * H) {
    group(a0, b0, r0);
    __builtin_amdgcn_sched_barrier(0);
    load(a0, b0, r0 + 2 * H);
    __builtin_amdgcn_sched_barrier(0);
    group(a1, b1, r0 + H);
    __builtin_amdgcn_sched_barrier(0);
    load(a1, b1, r0 + 3 * H);
    __builtin_amdgcn_sched_barrier(0);
  }
  double* out = d.part + (int64_t)blockIdx.x * d.ldpart;
#pragma unroll
  for (int k = 0; k < C; ++k) {
    const int j = tid + k * BS;
    if (j < d.n) out[j] = acc[k];
  }
  if (wv == 0) {
    const double s = wave_sum(tid < H ? e0 : 0.0);
    if (lane == 0) out[d.n] = s;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// v2: 16-byte buffer loads, G groups of H rows in flight, one barrier per group
// ---------------------------------------------------------------------------------------------------------------------
template <int KV, int H, int G, int BS, bool NT, bool ONEBAR, bool DOT32 = false, int ACCM = 0>
__global__ __launch_bounds__(BS) void k_v2(Args d) {
  constexpr int NW = BS / 64;
  __shared__ double wsum[2][NW][H];
  __shared__ double res_s[2][H];
  __shared__ float bs[2][H];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  v4f xr[KV];
  double acc[KV][4];
#pragma unroll
  for (int k = 0; k < KV; ++k) {
    const int j = (tid + k * BS) * 4;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      xr[k][e] = (j + e) < d.n ? d.x[j + e] : 0.f;
      acc[k][e] = 0.0;
    }
  }
  const int voff = tid * 16;                              // the ONE per-lane address register of the kernel
  const int row_bytes = (int)(d.ldT * 4);                 // rows are zero-padded to ldT: whole vectors, pads weigh 0
  double e0 = 0.0;
  const int row_begin = blockIdx.x * d.rows_per_block;
  const int row_end = min(row_begin + d.rows_per_block, d.m);
  v4f a[G][H][KV];
  float b_in[G];      // b of the group's rows rides behind its loads of A (a load AFTER them would wait for all of them)
  auto load = [&](int g, int r0) {
#pragma unroll
    for (int r = 0; r < H; ++r) {
      const int ri = __builtin_amdgcn_readfirstlane(min(r0 + r, row_end - 1));
      const __amdgpu_buffer_rsrc_t rs =
          __builtin_amdgcn_make_buffer_rsrc((void*)(d.AT + (int64_t)ri * d.ldT), 0, row_bytes, 0x00020000);
#pragma unroll
      for (int k = 0; k < KV; ++k) {
        typedef unsigned v4u __attribute__((ext_vector_type(4)));
        const v4u raw = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, k * BS * 16, NT ? 2 : 0);
        a[g][r][k] = __builtin_bit_cast(v4f, raw);
      }
    }
    b_in[g] = d.b[min(r0 + min(tid, H - 1), row_end - 1)];
  };
  auto group = [&](int g, int r0, int par) {
    double p[H];
    if constexpr (DOT32) {
      // the thread's share of the dot product in Float32 (as sgemv would), the sum over the threads in Float64
      float q[H];
#pragma unroll
      for (int r = 0; r < H; ++r) q[r] = 0.f;
#pragma unroll
      for (int k = 0; k < KV; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int r = 0; r < H; ++r) q[r] = ACCM == 2 ? q[r] + a[g][r][k][e] : fmaf(a[g][r][k][e], xr[k][e], q[r]);
#pragma unroll
      for (int r = 0; r < H; ++r) p[r] = (double)q[r];
    } else {
#pragma unroll
      for (int r = 0; r < H; ++r) p[r] = 0.0;
#pragma unroll
      for (int k = 0; k < KV; ++k) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float xv = xr[k][e];
          asm volatile("" : "+v"(xv));
          const double xd = (double)xv;
#pragma unroll
          for (int r = 0; r < H; ++r) p[r] = fma((double)a[g][r][k][e], xd, p[r]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#pragma unroll
    for (int r = 0; r < H; ++r) {
      p[r] = wave_sum(p[r]);
      if (lane == 0) wsum[par][wv][r] = p[r];
    }
    if (tid < H) bs[par][tid] = b_in[g];
    __syncthreads();
    double rr[H];
    if constexpr (ONEBAR) {
      // every thread finishes the H rows itself: NW broadcast reads per row, the entry of b, the residual
#pragma unroll
      for (int r = 0; r < H; ++r) {
        double dot = wsum[par][0][r];
#pragma unroll
        for (int w = 1; w < NW; ++w) dot += wsum[par][w][r];
        const int i = r0 + r;
        const float bi = bs[par][r];
        rr[r] = i < row_end ? (double)(float)((double)(float)dot - (double)bi) : 0.0;
        if (tid == 0) e0 += rr[r] * rr[r];
      }
    } else {
      if (tid < H) {
        double dot = wsum[par][0][tid];
#pragma unroll
        for (int w = 1; w < NW; ++w) dot += wsum[par][w][tid];
        const int i = r0 + tid;
        double res = 0.0;
        if (i < row_end) {
          res = (double)(float)((double)(float)dot - (double)bs[par][tid]);
          e0 += res * res;
        }
        res_s[par][tid] = res;
      }
      __syncthreads();
#pragma unroll
      for (int r = 0; r < H; ++r) rr[r] = res_s[par][r];
    }
    if constexpr (ACCM == 0) {
#pragma unroll
      for (int r = 0; r < H; ++r) {
#pragma unroll
        for (int k = 0; k < KV; ++k) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float av = a[g][r][k][e];
            asm volatile("" : "+v"(av));
            acc[k][e] = fma((double)av, rr[r], acc[k][e]);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    } else if constexpr (ACCM == 1) {
      // the H products of a column in Float32 (the residuals ARE Float32 values), their sum added in Float64
      float rf[H];
#pragma unroll
      for (int r = 0; r < H; ++r) rf[r] = (float)rr[r];
#pragma unroll
      for (int k = 0; k < KV; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float t = a[g][0][k][e] * rf[0];
#pragma unroll
          for (int r = 1; r < H; ++r) t = fmaf(a[g][r][k][e], rf[r], t);
          acc[k][e] += (double)t;
        }
    } else {      // ACCM == 2: memory-only ceiling of this access pattern (no per-element fp64 work at all)
#pragma unroll
      for (int k = 0; k < KV; ++k) acc[k][0] += rr[0];
    }
#pragma unroll
    for (int k = 0; k < KV; ++k)
#pragma unroll
      for (int e = 0; e < 4; ++e) asm volatile("" : "+v"(acc[k][e]));
  };
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int g = 0; g < G; ++g) {
    load(g, row_begin + g * H);
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll 1
  for (int r0 = row_begin; r0 < row_end; r0 += G * H) {
#pragma unroll
    for (int g = 0; g < G; ++g) {
      group(g, r0 + g * H, g & 1);
      __builtin_amdgcn_sched_barrier(0);
      load(g, r0 + (G + g) * H);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  double* out = d.part + (int64_t)blockIdx.x * d.ldpart;
#pragma unroll
  for (int k = 0; k < KV; ++k) {
    const int j = (tid + k * BS) * 4;
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (j + e < d.n) out[j + e] = acc[k][e];
  }
  if (ONEBAR) {
    if (tid == 0) out[d.n] = e0;
  } else if (wv == 0) {
    const double s = wave_sum(tid < H ? e0 : 0.0);
    if (lane == 0) out[d.n] = s;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
struct Variant {
  const char* name;
  void (*launch)(const Args&, int blocks, hipStream_t);
  int rows_step;
};
template <int C, int H, int BS>
static void l_v1(const Args& a, int blocks, hipStream_t s) { hipLaunchKernelGGL((k_v1<C, H, BS>), dim3(blocks), dim3(BS), 0, s, a); }
template <int KV, int H, int G, int BS, bool NT, bool OB, bool D32 = false, int AM = 0>
static void l_v2(const Args& a, int blocks, hipStream_t s) {
  hipLaunchKernelGGL((k_v2<KV, H, G, BS, NT, OB, D32, AM>), dim3(blocks), dim3(BS), 0, s, a);
}

int main(int argc, char** argv) {
  const int m = argc > 1 ? atoi(argv[1]) : 100000, n = argc > 2 ? atoi(argv[2]) : 10000;
  const int reps = 30;
  const int64_t ldT = (n + 31) / 32 * 32;
  int ncu = 256;
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  ncu = prop.multiProcessorCount;
  float *AT, *b, *x;
  CK(hipMalloc(&AT, (size_t)m * ldT * 4));
  CK(hipMemset(AT, 0, (size_t)m * ldT * 4));
  hipLaunchKernelGGL(k_fill, dim3(16, m), dim3(256), 0, 0, AT, ldT, m, n);
  std::vector<float> hb(m), hx(n);
  for (int i = 0; i < m; ++i) hb[i] = gen(2, i);
  for (int j = 0; j < n; ++j) hx[j] = 0.01f * gen(3, j);
  CK(hipMalloc(&b, m * 4));
  CK(hipMalloc(&x, n * 4));
  CK(hipMemcpy(b, hb.data(), m * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(x, hx.data(), n * 4, hipMemcpyHostToDevice));
  const int64_t ldpart = (n + 2 + 7) / 8 * 8;
  double* part;
  CK(hipMalloc(&part, (size_t)1024 * ldpart * 8));
  // host reference: sum of squares and the gradient of the first 48 and last 48 columns
  std::vector<double> res(m);
  double ss_ref = 0.0;
  {
    std::vector<float> row(n);
    for (int i = 0; i < m; ++i) {
      double dot = 0.0;
      for (int j = 0; j < n; ++j) dot = fma((double)gen(1, (uint64_t)i * n + j), (double)hx[j], dot);
      const double r = (double)(float)((double)(float)dot - (double)hb[i]);
      res[i] = r;
      ss_ref += r * r;
    }
  }
  std::vector<int> cols;
  for (int j = 0; j < 48; ++j) cols.push_back(j), cols.push_back(n - 1 - j);
  std::vector<double> g_ref(cols.size(), 0.0);
  for (size_t c = 0; c < cols.size(); ++c)
    for (int i = 0; i < m; ++i) g_ref[c] = fma((double)gen(1, (uint64_t)i * n + cols[c]), res[i], g_ref[c]);

  const Variant vars[] = {
      {"v2 H2 G2 2bar dot32, acc32-pair KV4 BS768", l_v2<4, 2, 2, 768, true, false, true, 1>, 4},
      {"v2 H2 G2 2bar dot32, acc32-pair KV5 BS512", l_v2<5, 2, 2, 512, true, false, true, 1>, 4},
      {"v2 H2 G3 2bar dot32, acc32-pair KV5 BS512", l_v2<5, 2, 3, 512, true, false, true, 1>, 6},
      {"v2 H3 G2 2bar dot32, acc32-pair KV5 BS512", l_v2<5, 3, 2, 512, true, false, true, 1>, 6},
      {"v2 H4 G2 2bar dot32, acc32-pair KV5 BS512", l_v2<5, 4, 2, 512, true, false, true, 1>, 8},
      {"v2 H2 G4 2bar dot32, acc32-pair KV5 BS512", l_v2<5, 2, 4, 512, true, false, true, 1>, 8},
      {"v2 H2 G2 1bar dot32, acc32-pair KV5 BS512", l_v2<5, 2, 2, 512, true, true, true, 1>, 4},
      {"v2 H4 G2 1bar dot32, acc32-pair KV5 BS512", l_v2<5, 4, 2, 512, true, true, true, 1>, 8},
      {"v2 H2 G3 2bar loads only KV5 BS512", l_v2<5, 2, 3, 512, true, false, true, 2>, 6},
      {"v2 H2 G2 2bar dot32, acc32-pair KV4 BS640", l_v2<4, 2, 2, 640, true, false, true, 1>, 4},
      {"v2 H2 G2 2bar fp64 arithmetic KV5 BS512", l_v2<5, 2, 2, 512, true, false, false, 0>, 4},
  };
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (const Variant& v : vars) {
    int64_t B = ncu;
    int64_t rpb = (m + B - 1) / B;
    rpb = (rpb + v.rows_step - 1) / v.rows_step * v.rows_step;
    B = (m + rpb - 1) / rpb;
    Args a{AT, ldT, m, n, (int)rpb, b, x, part, ldpart};
    CK(hipMemset(part, 0, (size_t)B * ldpart * 8));
    for (int w = 0; w < 3; ++w) v.launch(a, (int)B, 0);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int r = 0; r < reps; ++r) v.launch(a, (int)B, 0);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    CK(hipGetLastError());
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    std::vector<double> hp((size_t)B * ldpart);
    CK(hipMemcpy(hp.data(), part, hp.size() * 8, hipMemcpyDeviceToHost));
    double ss = 0.0, gerr = 0.0, gmax = 0.0;
    for (int64_t bb = 0; bb < B; ++bb) ss += hp[bb * ldpart + n];
    for (size_t c = 0; c < cols.size(); ++c) {
      double g = 0.0;
      for (int64_t bb = 0; bb < B; ++bb) g += hp[bb * ldpart + cols[c]];
      gerr = fmax(gerr, fabs(g - g_ref[c]));
      gmax = fmax(gmax, fabs(g_ref[c]));
    }
    printf("%-52s blocks %3d rows/block %4d  %.4f ms  %6.0f GB/s   ss rel err %.1e  grad rel err %.1e\n", v.name, (int)B,
           (int)rpb, ms, (double)m * n * 4 / (ms * 1e-3) / 1e9, fabs(ss - ss_ref) / ss_ref, gerr / gmax);
    fflush(stdout);
  }
  return 0;
}
