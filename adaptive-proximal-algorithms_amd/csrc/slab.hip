// slab.hip — f(x) and grad f(x) of LinearLeastSquares / LogisticLoss (experiments/lasso/runme.jl:21-25,
// experiments/sparse_logreg/runme.jl:23-37) in ONE pass over A, for matrices whose rows are short enough to be held in
// the registers of a workgroup (n <= 10 752).
//
// The reference's cost model — and the generic path here — is two streaming passes per iteration: res = A x - b (or
// the logistic link), then grad = A' res (lasso/runme.jl:179-180).  The second pass needs every res[i] = <A[i,:], x>
// complete, but only of the rows it touches: a block that owns a slab of ROWS and keeps R of them at a time in
// registers (thread t holds the columns t, t + BS, ... of each, read from the transposed copy, where a row is
// contiguous) can form their R dot products, the residuals, and add res[i] * A[i,:] to a private column accumulator
// from the SAME registers.  A is read once; what is left are B partial n-vectors (20 MB at C3 against 4 GB of A),
// summed in a fixed order by a second, small kernel that also finalises f and the logistic scaling.  No atomics:
// bit-reproducible.  HBM traffic per iteration: m n sizeof(T) instead of 2 m n sizeof(T).
//
//   k_slab_rows  (one or two big blocks per CU, rows dealt in contiguous slabs): R rows in registers, as one group or as
//                two that alternate (one computes while the other's non-temporal loads arrive) ; per group two
//                barriers (the dots across the waves ; the residuals back to every thread)
//   k_slab_close (64 columns per block): the B partials of every column, the loss sums (every block, redundantly),
//                f -> Ctrl, grad (logistic: / N, the bias entry)
//
// Used by every evaluation of such an f (eval_f in solver.hip: the adaptive loops, the linesearch and momentum
// methods, apx_eval) in place of gemv_n + k_finish_rows + k_f_reduce and of the gemv_t (+ k_logistic_grad) of the
// pullback, when there is no all-reduce hook; it reads the transposed copy of A (apx_matrix_build_transpose), which a matrix of up to
// 16 GB is given on the spot.
#include <algorithm>
#include <cstdlib>
#include <memory>

#include "skinny_dev.h"
#include "small_dev.h"
#include "solver_internal.h"
#include "team_dev.h"
#include "team_rows.h"
#include "vecops_dev.h"

namespace apx {

constexpr int kSlabTeamCpt = 99;      // SlabShape::cpt of the team kernel
constexpr int kSlabMultiCpt = 98;     // ... of the H-rows-per-step kernel for short Float64 rows (csrc/team_rows.h k_multi_rows)
constexpr int kSlabRows = 4;          // rows per step (two columns per thread: 16 ; fourteen fp64 columns: 2, or 1 with the logistic link)

struct SlabDev {
  const void* AT;         // A' column-major: row i of A at AT + i * ldT, contiguous
  int64_t ldT;
  int32_t m, ncols, n, rows_per_block;
  int32_t deal_stride, deal_log;      // how the rows are dealt to the blocks: csrc/team_dev.h deal_rows / deal_row (stride 1: slabs)
  const void* b;
  double* part;           // [blocks][ldpart]: ncols column partials, then the loss sum(s)
  int64_t ldpart;
  Ctrl* c;
  double* fsc_out;        // row-sharded f: the closing kernel leaves the two LOCAL loss sums here (they are all-reduced with the
};                        // gradient, then k_f_final closes f) ; NULL: it closes f itself

// The end of a proximal-gradient iteration in TWO kernels behind the pass over A instead of four (no linear operator, no row
// shard).  The closing kernel, which has just formed the gradient of its 64 columns, also forms k_reduce_n's terms of those 64
// entries — the primal residual and the step-size sums, src/AdaProx.jl:338, :260-261 — and leaves them as wave partials, and
// block 0 leaves a SNAPSHOT of the control block behind f's value.  The step kernel (csrc/vecops.hip k_step_rule) then adds the
// partials up in EVERY block — the same totals everywhere, no grid-wide synchronisation — runs the rule and the end of the
// iteration (:340-342, :348-354) on a private copy of the snapshot, and takes the step with the new gamma ; block 0 writes the
// control block back.  Nobody reads the control block while it is written: the blocks work from the snapshot.
// The sums are formed in exactly the order of k_reduce_n + k_rule (a closing block's 64 columns are one wave of k_reduce_n's
// 256-entry block: wave partials here, the same two-level tree there), so the iterates are bit-identical to the separate kernels
// (tests/test_gpu_slab.py::test_fused_tail_is_bit_identical).
// (Round 3's first form let the LAST-ARRIVING closing block run the rule behind a fence and an arrival counter: one kernel fewer
// still, and slower than the four it replaced — the 1563 device-scope fences of C2's closing grid, an L2 write-back each, took
// the kernel from 7.4 to 53 us ; profiles/r03_tail_fence.md.)
struct SlabTail : SlabTailReq {          // (the solver's n-vectors, g, the record sink: solver_internal.h)
  double* wred;                          // [5][ldw] wave partials (entries beyond the last wave stay zero)
  Ctrl* snap;                            // the control block as the closing kernel left it (f_x final, the rule not yet run)
  int32_t ldw, enabled;
};

struct SlabState {
  DBuf part;
  SlabDev dev{};
  int B = 0, cpt = 0, bs = 0;     // blocks ; columns per thread ; threads per block
  int multi_waves = 0;            // k_multi_rows: 0 = by fit, 10 / 11 compute waves forced (APX_OPT_MULTI_WAVES)
  int kv = 0, bs2 = 768;          // long rows: 16-byte vectors per thread and row of k_slab2_rows (0: k_slab_rows), its block size
  bool v2 = true;                 // (APX_SLAB_V1=1 in the environment: the first-generation kernel for long rows too — A/B runs)
  bool pipe = true, nt = true;    // two row groups in flight ; non-temporal loads of A
  // rows of more than 10 752 Float64 entries: teams of team_p workgroups (k_team_rows), team_kv vectors per thread
  TeamPlan plan;                  // P = 0: no team kernel ; members, vectors per thread, the three-slot shape, compute waves (csrc/team_rows.h)
  TeamHost team;                  // ring, tickets, the error word and its diagnostic record, wait statistics (csrc/team_rows.h)
  DBuf tail_buf;                  // the fused end of the iteration (SlabTail): wave partials of the five reductions + the arrival counter
  std::string kname;              // the row kernel as launched (apx_ctx_kernel_name)
};
void slab_destroy(SlabState* st) { delete st; }      // (null is fine)

// H rows of A (from the transposed copy) into registers: thread t gets the columns t, t + BS, ... of each ; the rows past
// the slab are clamped to its last one (a valid address that is in the cache already ; their residual is 0)
template <typename T, int C, int H, bool NT>
__device__ __forceinline__ void slab_load(T (&a)[H][C], T& b_in, const SlabDev& d, const unsigned (&off)[C], int r0,
                                          int row_end, int tid) {
  const T* __restrict__ AT = (const T*)d.AT;
  const T* __restrict__ bv = (const T*)d.b;
#pragma unroll
  for (int r = 0; r < H; ++r) {
    // (r0, row_end: the block's OWN row numbers — deal_row gives the matrix row, clamped into the block's rows ; uniform: scalar)
    const int ri = __builtin_amdgcn_readfirstlane(min(deal_row(r0 + r, row_end, (int)blockIdx.x, d.rows_per_block, d.deal_stride, d.deal_log), d.m - 1));
    const T* __restrict__ row = AT + (int64_t)ri * d.ldT;
#pragma unroll
    for (int k = 0; k < C; ++k)                  // scalar row base + 32-bit BYTE offset: one address register per load
      a[r][k] = NT ? __builtin_nontemporal_load((const T*)((const char*)row + off[k]))   // (read once: past the caches,
                   : *(const T*)((const char*)row + off[k]);                             //  unless A fits the Infinity Cache)
  }
  b_in = bv[min(deal_row(r0 + min(tid, H - 1), row_end, (int)blockIdx.x, d.rows_per_block, d.deal_stride, d.deal_log), d.m - 1)];      // (goes out with the loads of A ; every thread, unconverted: a
                                                          //  branch or a conversion here would wait for all of them)
}

// the H rows r0 .. r0 + H - 1 held in a: their dots with x, the residuals (thread r finishes row r0 + r), and
// res[i] * A[i,:] into the column accumulator ; two barriers
template <typename T, int FK, int C, int H, int BS, bool TIGHT>
__device__ __forceinline__ void slab_group(const T (&a)[H][C], T b_in, const T (&xr)[C], double (&acc)[C], double bias,
                                           double& e0, double& e1, int r0, int row_end, double (*wsum)[H], double* res_s,
                                           int tid) {
  const int lane = tid & 63, wv = tid >> 6;
  double p[H];
#pragma unroll
  for (int r = 0; r < H; ++r) p[r] = 0.0;
#pragma unroll
  for (int k = 0; k < C; ++k) {
    T xv = xr[k];
    if constexpr (sizeof(T) == 4) asm volatile("" : "+v"(xv));        // (x stays in T: its doubles hoisted out of the row
    const double xd = (double)xv;                                     //  loop would be C more register pairs)
#pragma unroll
    for (int r = 0; r < H; ++r) p[r] = fma((double)a[r][k], xd, p[r]);
    if constexpr (TIGHT && sizeof(T) == 4) __builtin_amdgcn_sched_barrier(0);   // (no batch of conversions ahead of their use:
  }                                                                             //  the other group's rows hold the registers)
  wave_sum_multi<H>(p);                          // lane l: the wave's sum for row l / (64 / H)
  if (lane % (64 / H) == 0) wsum[wv][lane / (64 / H)] = p[0];
  __syncthreads();
  if (tid < H) {
    double dot = wsum[0][tid];
#pragma unroll
    for (int w = 1; w < BS / 64; ++w) dot += wsum[w][tid];
    const int i = r0 + tid;
    double res = 0.0;
    if (i < row_end) {
      const double bi = (double)b_in;
      const double ax = (double)(T)dot;          // A_x rounded to T like every stored A x
      if constexpr (FK == APX_F_LEAST_SQUARES) {
        res = (double)(T)(ax - bi);
        e0 += res * res;
      } else {
        const double logit = ax + bias;
        if constexpr (sizeof(T) == 4) {          // Float32 data: the link in Float32, as Julia would on Float32 arrays
          const float lf = (float)logit;         // (and the fp64 exp / log need ~50 more registers than this kernel has)
          const float ef = 1.0f + expf(-lf);
          e0 += (double)(((float)bi - 1.0f) * lf - logf(ef));
          res = (double)(1.0f / ef - (float)bi);
        } else {
          const double e = 1.0 + exp(-logit);
          e0 += (bi - 1.0) * logit - log(e);
          res = (double)(T)(1.0 / e - bi);
        }
        e1 += res;
      }
    }
    res_s[tid] = res;
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < H; ++r) {
    const double rr = res_s[r];
#pragma unroll
    for (int k = 0; k < C; ++k) {
      T av = a[r][k];
      if constexpr (sizeof(T) == 4) asm volatile("" : "+v"(av));      // convert again here: keeping the doubles of the dot
      acc[k] = fma((double)av, rr, acc[k]);                           // phase alive would double the registers of A
    }
    if constexpr (TIGHT && sizeof(T) == 4) __builtin_amdgcn_sched_barrier(0);
  }
  if constexpr (TIGHT) {                         // the sums are complete HERE: left to itself the compiler sinks these FMAs
#pragma unroll                                   // towards their next use, past the loads that re-fill a — which then need
    for (int k = 0; k < C; ++k) asm volatile("" : "+v"(acc[k]));      // registers of their own
  }
}

// BS threads, C columns per thread (BS * C >= n), R rows in registers: R x C values of A, C of x and C doubles of acc
// must fit the 128 VGPRs of a 1024-thread block.  PIPE: the R rows are two groups of R / 2 ; the loads of a group go
// out as soon as its registers are free and have the other group's dots, barriers and accumulation to arrive in
// (everything in the loop is unconditional: a branch around a load would make the compiler wait for ALL loads at the join).
template <typename T, int FK, int C, int R, int BS, bool PIPE, bool NT>
__global__ __launch_bounds__(BS) void k_slab_rows(SlabDev d, const void* xv) {
  constexpr int H = PIPE ? R / 2 : R;
  static_assert(!PIPE || R % 2 == 0, "two groups");
  __shared__ double wsum[BS / 64][H];
  __shared__ double res_s[H];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (d.c->done != 0) return;                    // written by an earlier launch: block-uniform
  const T* __restrict__ X = (const T*)xv;
  T xr[C];                                       // (kept in T: the registers are what limits the row length)
  double acc[C];
  unsigned off[C];                               // 32-bit column offsets: with a uniform row pointer every load of the
#pragma unroll                                   // step is "scalar base + vector offset", no 64-bit address per load
  for (int k = 0; k < C; ++k) {
    const int j = tid + k * BS;
    off[k] = (unsigned)min(j, d.ncols - 1) * (unsigned)sizeof(T);     // in bytes ; clamped: a valid address, weight 0
    xr[k] = j < d.ncols ? X[j] : (T)0;
    acc[k] = 0.0;
  }
  double bias = 0.0;
  if constexpr (FK == APX_F_LOGISTIC) bias = (double)X[d.n - 1];
  double e0 = 0.0, e1 = 0.0;                     // loss sums of the rows this thread finishes (tid < H)
  const int row_begin = 0;                        // (the block's own row numbers 0 .. row_end - 1 ; slab_load maps them: deal_row)
  const int row_end = deal_rows(d.m, (int)blockIdx.x, d.rows_per_block, d.deal_stride, d.deal_log);
  if constexpr (!PIPE) {
    for (int r0 = row_begin; r0 < row_end; r0 += R) {
      T a[H][C];
      T b_in;
      slab_load<T, C, H, NT>(a, b_in, d, off, r0, row_end, tid);
      slab_group<T, FK, C, H, BS, false>(a, b_in, xr, acc, bias, e0, e1, r0, row_end, wsum, res_s, tid);
    }
  } else {
    T a0[H][C], a1[H][C];
    T b0, b1;
    __builtin_amdgcn_sched_barrier(0);
    slab_load<T, C, H, NT>(a0, b0, d, off, row_begin, row_end, tid);
    __builtin_amdgcn_sched_barrier(0);           // (in this order: the waits of the loop count the loads issued after a group's)
    slab_load<T, C, H, NT>(a1, b1, d, off, row_begin + H, row_end, tid);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
    for (int r0 = row_begin; r0 < row_end; r0 += R) {
      slab_group<T, FK, C, H, BS, true>(a0, b0, xr, acc, bias, e0, e1, r0, row_end, wsum, res_s, tid);
      __builtin_amdgcn_sched_barrier(0);         // (the loads AFTER the accumulation that frees their registers, and
      slab_load<T, C, H, NT>(a0, b0, d, off, r0 + R, row_end, tid);
      __builtin_amdgcn_sched_barrier(0);         //  before the other group's dots: hoisted or sunk they cost registers or overlap)
      slab_group<T, FK, C, H, BS, true>(a1, b1, xr, acc, bias, e0, e1, r0 + H, row_end, wsum, res_s, tid);
      __builtin_amdgcn_sched_barrier(0);
      slab_load<T, C, H, NT>(a1, b1, d, off, r0 + R + H, row_end, tid);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  double* __restrict__ out = d.part + (int64_t)blockIdx.x * d.ldpart;
#pragma unroll
  for (int k = 0; k < C; ++k) {
    const int j = tid + k * BS;
    if (j < d.ncols) out[j] = acc[k];
  }
  // the loss sums of this block: the rows of every group were finished by threads 0 .. H-1 (one wave)
  if (wv == 0) {
    double ex[2] = {tid < H ? e0 : 0.0, tid < H ? e1 : 0.0};
    wave_sum_multi<2>(ex);                       // lane l: ex[l / 32]
    if ((lane & 31) == 0) out[d.ncols + (lane >> 5)] = ex[0];
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Long rows (5121 ... 10 752 columns — BASELINE config C3), second generation of the row kernel.  What round 2 measured
// on the first one (tools/slab2_probe.hip, profiles/r02_one_pass_probe.log): at C3 it is not the loads that cost — a
// kernel that only loads in this pattern reads at 6.7 - 6.9 TB/s, three rows in flight do as well as six — but (i) the
// Float64 conversions of Float32 data, two per element and one more per x entry and row: the vector ALU was the
// bottleneck (fp64 conversions and FMAs issue at a quarter of the Float32 rate); (ii) the 4-byte loads with a register
// per column for their addresses.  So:
//   * column-BLOCKED mapping: thread t holds the 16-byte vectors t, t + BS, ... of a row — 1 KiB per wave-load, through
//     buffer loads: a scalar descriptor per row, ONE per-lane offset register for the whole kernel, scalar offsets for
//     the vectors (no address registers per column; rows are zero-padded to whole vectors in the transposed copy and
//     the descriptor ends at the padded row: no clamping)
//   * Float32 data: each thread's share of a dot product (<= 16 products) is accumulated in Float32, the sum over the
//     768 threads in Float64; the H products a column receives per group are summed in Float32 (the residuals ARE
//     Float32 values) and added to the Float64 column accumulator.  Both errors are of the order of the rounding of
//     `A x` and of the gradient to Float32 that follows anyway (measured: 5e-10 on f, 4e-8 on the gradient, relative).
//     Float64 data: Float64 throughout, as before.
//   * G groups of H rows in flight (2 x 2), b of a group's rows loaded behind its rows of A (a load issued later would
//     wait for every load before it), two barriers per group as before (the link function is evaluated once per row).
// C3: 0.646 -> 0.58 ms per launch under rocprofv3 (profiles/r02_*).
// ---------------------------------------------------------------------------------------------------------------------
template <typename T> struct SlabVec;
template <> struct SlabVec<float> { typedef float type __attribute__((ext_vector_type(4))); };
template <> struct SlabVec<double> { typedef double type __attribute__((ext_vector_type(2))); };

template <typename T, int FK, int KV, int H, int G, bool NT>
__global__ __launch_bounds__(768) void k_slab2_rows(SlabDev d, const void* xv) {
  constexpr int VW = 16 / (int)sizeof(T);          // elements per 16-byte vector
  constexpr int kMaxWaves = 12;
  constexpr bool kF32 = sizeof(T) == 4;
  const int BS = (int)blockDim.x;                  // (768 ; any multiple of 64 up to it works)
  const int NW = BS / 64;
  typedef typename SlabVec<T>::type vec_t;
  typedef unsigned raw_t __attribute__((ext_vector_type(4)));
  __shared__ double wsum[kMaxWaves][H];
  __shared__ double res_s[H];
  __shared__ T bs[H];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (d.c->done != 0) return;                      // written by an earlier launch: block-uniform
  const T* __restrict__ X = (const T*)xv;
  vec_t xr[KV];
  double acc[KV][VW];
#pragma unroll
  for (int k = 0; k < KV; ++k) {
    const int j = (tid + k * BS) * VW;
#pragma unroll
    for (int e = 0; e < VW; ++e) {
      xr[k][e] = (j + e) < d.ncols ? X[j + e] : (T)0;        // (logistic: X[ncols] is the bias, not a column weight)
      acc[k][e] = 0.0;
    }
  }
  double bias = 0.0;
  if constexpr (FK == APX_F_LOGISTIC) bias = (double)X[d.n - 1];
  const int voff = tid * 16;                       // the one per-lane address register
  const int row_bytes = (int)(d.ldT * (int64_t)sizeof(T));
  double e0 = 0.0, e1 = 0.0;                       // loss sums of the rows this thread finishes (tid < H)
  const int row_begin = 0;                         // (the block's own row numbers ; the loads map them: deal_row)
  const int row_end = deal_rows(d.m, (int)blockIdx.x, d.rows_per_block, d.deal_stride, d.deal_log);
  vec_t a[G][H][KV];
  T b_in[G];
  const T* __restrict__ AT = (const T*)d.AT;
  const T* __restrict__ bv = (const T*)d.b;
  auto load = [&](int g, int r0) {                 // rows past the slab: its last row again (a valid address ; residual 0)
#pragma unroll
    for (int r = 0; r < H; ++r) {
      const int ri = __builtin_amdgcn_readfirstlane(min(deal_row(r0 + r, row_end, (int)blockIdx.x, d.rows_per_block, d.deal_stride, d.deal_log), d.m - 1));
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(AT + (int64_t)ri * d.ldT), 0, row_bytes, 0x00020000);
#pragma unroll
      for (int k = 0; k < KV; ++k) {
        const raw_t raw = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, k * BS * 16, NT ? 2 : 0);   // (aux 2: nt)
        a[g][r][k] = __builtin_bit_cast(vec_t, raw);
      }
    }
    b_in[g] = bv[min(deal_row(r0 + min(tid, H - 1), row_end, (int)blockIdx.x, d.rows_per_block, d.deal_stride, d.deal_log), d.m - 1)];
  };
  auto group = [&](int g, int r0) {
    double p[H];
    if constexpr (kF32) {
      float q[H];
#pragma unroll
      for (int r = 0; r < H; ++r) q[r] = 0.f;
#pragma unroll
      for (int k = 0; k < KV; ++k)
#pragma unroll
        for (int e = 0; e < VW; ++e)
#pragma unroll
          for (int r = 0; r < H; ++r) q[r] = fmaf(a[g][r][k][e], xr[k][e], q[r]);
#pragma unroll
      for (int r = 0; r < H; ++r) p[r] = (double)q[r];
    } else {
#pragma unroll
      for (int r = 0; r < H; ++r) p[r] = 0.0;
#pragma unroll
      for (int k = 0; k < KV; ++k)
#pragma unroll
        for (int e = 0; e < VW; ++e)
#pragma unroll
          for (int r = 0; r < H; ++r) p[r] = fma(a[g][r][k][e], xr[k][e], p[r]);
    }
    wave_sum_multi<H>(p);                          // lane l: the wave's sum for row l / (64 / H)
    if (lane % (64 / H) == 0) wsum[wv][lane / (64 / H)] = p[0];
    if (tid < H) bs[tid] = b_in[g];
    __syncthreads();
    if (tid < H) {
      double dot = wsum[0][tid];
      for (int w = 1; w < NW; ++w) dot += wsum[w][tid];
      const int i = r0 + tid;
      double res = 0.0;
      if (i < row_end) {
        const double bi = (double)bs[tid];
        const double ax = (double)(T)dot;          // A_x rounded to T like every stored A x
        if constexpr (FK == APX_F_LEAST_SQUARES) {
          res = (double)(T)(ax - bi);
          e0 += res * res;
        } else {
          const double logit = ax + bias;
          if constexpr (kF32) {                    // Float32 data: the link in Float32, as Julia would on Float32 arrays
            const float lf = (float)logit;
            const float ef = 1.0f + expf(-lf);
            e0 += (double)(((float)bi - 1.0f) * lf - logf(ef));
            res = (double)(1.0f / ef - (float)bi);
          } else {
            const double e = 1.0 + exp(-logit);
            e0 += (bi - 1.0) * logit - log(e);
            res = (double)(T)(1.0 / e - bi);
          }
          e1 += res;
        }
      }
      res_s[tid] = res;
    }
    __syncthreads();
    if constexpr (kF32) {
      float rf[H];
#pragma unroll
      for (int r = 0; r < H; ++r) rf[r] = (float)res_s[r];
#pragma unroll
      for (int k = 0; k < KV; ++k)
#pragma unroll
        for (int e = 0; e < VW; ++e) {
          float t = a[g][0][k][e] * rf[0];
#pragma unroll
          for (int r = 1; r < H; ++r) t = fmaf(a[g][r][k][e], rf[r], t);
          acc[k][e] += (double)t;
        }
    } else {
#pragma unroll
      for (int r = 0; r < H; ++r) {
        const double rr = res_s[r];
#pragma unroll
        for (int k = 0; k < KV; ++k)
#pragma unroll
          for (int e = 0; e < VW; ++e) acc[k][e] = fma(a[g][r][k][e], rr, acc[k][e]);
      }
    }
#pragma unroll                                     // the sums are complete HERE (see k_slab_rows)
    for (int k = 0; k < KV; ++k)
#pragma unroll
      for (int e = 0; e < VW; ++e) asm volatile("" : "+v"(acc[k][e]));
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
      group(g, r0 + g * H);
      __builtin_amdgcn_sched_barrier(0);           // (the loads AFTER the accumulation that frees their registers and
      load(g, r0 + (G + g) * H);                   //  before the next group's dots)
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  double* __restrict__ out = d.part + (int64_t)blockIdx.x * d.ldpart;
#pragma unroll
  for (int k = 0; k < KV; ++k) {
    const int j = (tid + k * BS) * VW;
#pragma unroll
    for (int e = 0; e < VW; ++e)
      if (j + e < d.ncols) out[j + e] = acc[k][e];
  }
  if (wv == 0) {
    double ex[2] = {tid < H ? e0 : 0.0, tid < H ? e1 : 0.0};
    wave_sum_multi<2>(ex);                         // lane l: ex[l / 32]
    if ((lane & 31) == 0) out[d.ncols + (lane >> 5)] = ex[0];
  }
}

// (k_team_rows: csrc/team_rows.h — shared with the primal-dual loop, csrc/slab_pd.hip)

// 64 columns per block (a wave reads 512 contiguous bytes of a partial row), the B partials of a column dealt to 16
// threads (b = s, s + 16, ...: with one block per CU in the row kernel that is ALL of a thread's loads in flight at
// once — 256 threads with four rounds of sixteen took 19.8 us at C3, 3 % of the iteration), then the 16 sub-sums added
// in order.
constexpr int kSlabCloseCols = 64;
constexpr int kSlabCloseBlock = 1024;
template <typename T, int FK, bool TAIL = false>
__global__ __launch_bounds__(kSlabCloseBlock) void k_slab_close(SlabDev d, int B, T* __restrict__ grad, SlabTail tl) {
  constexpr int S = kSlabCloseBlock / kSlabCloseCols;
  __shared__ double fs[2];
  __shared__ double sub[S][kSlabCloseCols];
  if (d.c->done != 0) return;
  const int tid = threadIdx.x;
  const int c = tid % kSlabCloseCols, sl = tid / kSlabCloseCols;
  const int j = blockIdx.x * kSlabCloseCols + c;
  // (TAIL: this column's entries of the solver's vectors, in flight with the partials)
  [[maybe_unused]] T tx = (T)0, txp = (T)0, tgp = (T)0, tv = (T)0;
  [[maybe_unused]] double gamma = 0.0;
  if constexpr (TAIL) {
    if (sl == 0 && j < d.n) {
      tx = ((const T*)tl.x)[j];
      txp = ((const T*)tl.xprev)[j];
      tgp = ((const T*)tl.gprev)[j];
      tv = ((const T*)tl.v)[j];
    }
    gamma = d.c->gamma;
  }
  const double part_sum = j < d.ncols ? skinny_strided_sum(d.part, d.ldpart, j, sl, S, B) : 0.0;
  // the loss sums over the B blocks, by every block in the same fixed order (waves 0 and 1: one column each)
  if (tid < 128) {
    const int which = tid >> 6, lane = tid & 63;
    double s = 0.0;
    for (int b = lane; b < B; b += 64) s += d.part[(int64_t)b * d.ldpart + d.ncols + which];
    s = wave_sum(s);
    if (lane == 0) fs[which] = s;
  }
  sub[sl][c] = part_sum;
  __syncthreads();
  const double N = d.c->n_mean;
  T gj = (T)0;
  if (sl == 0 && j < d.ncols) {
    double g = sub[0][c];
#pragma unroll
    for (int k = 1; k < S; ++k) g += sub[k][c];
    const T gs = (T)g;                                                       // A' * res
    gj = FK == APX_F_LOGISTIC ? (T)((double)gs / N) : gs;
    grad[j] = gj;
  }
  if constexpr (!TAIL) {
    if (FK == APX_F_LOGISTIC && blockIdx.x == 0 && tid == 0) grad[d.n - 1] = (T)(fs[1] / N);      // grad[end] = mean(probs - y)
    if (blockIdx.x == 0 && tid == 0) {
      if (d.fsc_out != nullptr) {                                            // row shard: local sums, closed after the all-reduce
        d.fsc_out[0] = fs[0];
        d.fsc_out[1] = fs[1];
      } else {
        f_finalize(d.c, FK, fs);                                             // f(x) (and the bias gradient) -> Ctrl
      }
    }
  } else {
    // grad[end] = mean(probs - y) by the lane whose element it is (the grid covers n = ncols + 1 entries)
    if (FK == APX_F_LOGISTIC && sl == 0 && j == d.n - 1) {
      gj = (T)(fs[1] / N);
      grad[j] = gj;
    }
    // ---- k_reduce_n's terms of this block's 64 entries: one wave of its 256-entry block ---------------------------------
    if (tid < 64) {
      double q[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
      if (j < d.n) {
        const double xi = (double)tx, gi = (double)gj;
        const double pr = ((double)tv - xi) / gamma + gi;                    // :338 (A = 0: no At_y term)
        const double dg = gi - (double)tgp;
        const double dx = xi - (double)txp;
        q[0] = fma(pr, pr, q[0]);
        q[1] = fma(dg, dg, q[1]);
        q[2] = fma(dg, dx, q[2]);
        q[3] = fma(dx, dx, q[3]);
        q[4] += gval_elem<T>(tl.g, xi, j);
      }
#pragma unroll
      for (int k = 0; k < 5; ++k) {
        const double w = wave_sum(q[k]);
        if (tid == 0) tl.wred[(int64_t)k * tl.ldw + blockIdx.x] = w;
      }
    }
    // ---- f's value, and the control block as the step kernel's blocks will read it ------------------------------------------
    if (blockIdx.x == 0 && tid == 0) {
      f_finalize(d.c, FK, fs);
      *tl.snap = *d.c;
    }
  }
}

// The same end of the iteration BEHIND THE ALL-REDUCE of a row shard (round 4): the closing kernel left this rank's A_k' r_k and
// its loss sums, the collective summed them over the ranks into `grad` / `fsc`, and what k_slab_close<TAIL> does with its own
// column sums this kernel does with the all-reduced gradient — k_reduce_n's terms as wave partials in the same layout, f's value,
// the snapshot — so that k_step_rule closes the iteration: two kernels behind the collective instead of four (vo_f_final,
// k_reduce_n, k_rule, k_step).  Blocks of 256 entries = four of the closing kernel's 64-column waves.
template <typename T, int FK>
__global__ __launch_bounds__(kBlock) void k_slab_tail(Ctrl* c, const T* __restrict__ grad, const double* __restrict__ fsc, int64_t n, SlabTail tl) {
  if (c->done != 0) return;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int64_t j = (int64_t)blockIdx.x * kBlock + tid;
  const double gamma = c->gamma;
  double q[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
  if (j < n) {
    const double xi = (double)((const T*)tl.x)[j], gi = (double)grad[j];
    const double pr = ((double)((const T*)tl.v)[j] - xi) / gamma + gi;       // :338 (A = 0: no At_y term)
    const double dg = gi - (double)((const T*)tl.gprev)[j];
    const double dx = xi - (double)((const T*)tl.xprev)[j];
    q[0] = fma(pr, pr, q[0]);
    q[1] = fma(dg, dg, q[1]);
    q[2] = fma(dg, dx, q[2]);
    q[3] = fma(dx, dx, q[3]);
    q[4] += gval_elem<T>(tl.g, xi, j);
  }
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    const double w = wave_sum(q[k]);
    if (lane == 0) tl.wred[(int64_t)k * tl.ldw + (int64_t)blockIdx.x * (kBlock / 64) + wv] = w;
  }
  if (blockIdx.x == 0 && tid == 0) {
    const double fs[2] = {fsc[0], fsc[1]};
    f_finalize(c, FK, fs);
    *tl.snap = *c;
  }
}

// ---------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------
struct SlabShape {
  int bs, cpt, rows;      // threads per block, columns per thread, rows per step (bs == 0: not eligible)
};
static SlabShape slab_shape(const apx_problem* p) {
  const int64_t cols = p->A_f->n;
  // Float64 rows of up to 5632 entries: k_multi_rows (csrc/team_rows.h) — 16-byte loads, H rows per step, the residual / the
  // link of the H rows in the lanes of a row wave that holds no row of A.  Against the shapes below (one element per lane and
  // load, the link next to the rows), tools/shape_sweep.py at 6 GB, TB/s of the bytes read: least squares 600: 4.7 -> 5.7,
  // 1100: 5.7 -> 6.4, 2100: 5.2 -> 6.3, else within +- 5 % ; logistic 400: 1.8 -> 4.6, 1000: 3.9 -> 6.2, 2100: 4.6 -> 6.3,
  // 4224: 5.5 -> 6.7.  (APX_OPT_MULTI_ROWS_MASK, bit 0 least squares / bit 1 logistic / bit 2 the primal-dual loop: an A/B
  // switch — 0 runs the first-generation shapes, which stay for Float32 ; tests/test_gpu_slab.py::test_first_generation_shapes)
  const int multi = p->ctx->multi_mask;
  if (p->dtype == APX_F64 && cols <= 2 * 4 * 704 && ((p->f_kind == APX_F_LOGISTIC ? 2 : 1) & multi)) return {768, kSlabMultiCpt, 1};
  // Float32 rows of up to 11 264 entries likewise (round 3, bit 3 of the mask): four columns per 16-byte vector, partial sums
  // in Float32, the link in Float32 in the row wave — BASELINE config C3's 10 000 columns run here
  // Measured against the first-generation shapes (tools/shape_sweep.py f32, 4 GB, profiles/r03_row_length_sweep_f32.log), TB/s of the
  // bytes read: 130: 0.8 -> 3.4, 300: 1.8 -> 5.6, 600: 3.4 -> 6.0, 2100: 3.8 -> 6.2, 3100: 5.0 -> 6.4, 5000 ... 8200: the same ;
  // 8500 ... 10 752 the fourteen-column first-generation kernel is as fast (least squares) or 1 - 4 % faster (logistic ; C3
  // itself, interleaved in one process: 0.5915 against 0.5994 ms) and stays ; beyond it the one pass is new: up to 22 528
  // entries (eight vectors per thread, one row per step)
  if (p->dtype == APX_F32 && (multi & 8) && (cols <= 8192 || (cols > 14 * 768 && cols <= 4 * 8 * 704))) return {768, kSlabMultiCpt, 1};
  if (cols <= 1024) return {1024, 1, kSlabRows};      // (one column per thread, 32 rows per step: with two, half the loads of such a row are clamped away)
  if (cols <= 2 * 768) return {768, 2, kSlabRows};    // (1025 ... 1536 entries: two columns of 768 threads keep more lanes busy than of 1024)
  if (cols <= 2 * 1024) return {1024, 2, kSlabRows};
  // (three and four columns per thread: with five, a row of 2100 entries keeps 420 of the 1024 lanes busy — 3.8 TB/s against
  //  6.5 for 2000 entries on the two-column kernel ; measured, tools/shape_bench.py)
  if (cols <= 3 * 1024) return {1024, 3, kSlabRows};
  if (cols <= 4 * 1024) return {1024, 4, kSlabRows};
  if (cols <= 5 * 1024) return {1024, 5, kSlabRows};
  // longer rows: 768 threads x 14 columns in the ~170 VGPRs that twelve waves per CU leave; fp64 takes two rows per
  // step (the same 160 KB) and one with the logistic link (see slab_launch_t).  Measured and dropped: 1024 x 10 x 2
  // (half the bytes in flight: +5 % over the two passes at C3), 512 x 20 x 4 and 1024 x 10 x 4 (spills).
  // Float64 logistic: the link function next to the rows costs the row kernels two of their three rows in flight (exp, log:
  // ~50 registers) — the communication-wave kernel with a "team" of ONE evaluates it where no row is held (k_team_rows, P = 1)
  // (a "team" of ONE hands nothing over and needs no co-residency: it stays when the team kernels are switched off)
  if (p->dtype == APX_F64 && p->f_kind == APX_F_LOGISTIC && cols > 5 * 1024 && cols <= 2 * kTeamSingleKv * 704)      // (5633 ... 11 264 by default)
    return {768, kSlabTeamCpt, 1};
  if (cols <= 14 * 768) return {768, 14, p->dtype == APX_F32 ? kSlabRows : kSlabRows / 2};
  // longer rows (BASELINE config C2: 100 000): teams of 2 / 4 / 8 / 16 workgroups, a slice of every row each (k_team_rows) —
  // Float64, on a part of 8 XCDs x 32 CUs
  if (p->dtype == APX_F64 && cols <= kTeamMaxCols && p->ctx->num_cu == 256 && p->ctx->teams && !p->teams_dead()) return {768, kSlabTeamCpt, 1};
  // Float32 rows beyond one workgroup's eight vectors (22 529 ... 225 280 entries): teams too (round 3)
  if (p->dtype == APX_F32 && cols > 4 * 8 * 704 && cols <= kTeamMaxColsF32 && p->ctx->num_cu == 256 && p->ctx->teams && !p->teams_dead() && (multi & 8))
    return {768, kSlabTeamCpt, 1};
  return {0, 0, 0};
}

static bool slab_shape_ok(const apx_problem* p) {
  if (!p->ctx->one_pass) return false;
  // (a row shard of f — communicator or callback — is as good as a whole matrix: the rows are local, only the block sums
  //  travel ; eval_f all-reduces them)
  // (a linear operator next to f — the primal-dual loops with a smooth term — does not stand in the way since round 4: f and
  //  grad f come from one read of A_f as everywhere else, the operator's half from its own kernels, csrc/slab_pd.hip ; the
  //  fused end of the iteration is the proximal-gradient loop's only, solver.hip enqueue_iteration)
  if (p->sharded() && !p->f_sharded) return false;
  if (p->f_kind != APX_F_LEAST_SQUARES && p->f_kind != APX_F_LOGISTIC) return false;
  if (p->A_f == nullptr || p->A_f->sparse || p->ctx->skip_zero_cols) return false;
  // large matrices only: every block walks its slab in steps of R rows with two barriers each, and below ~200 MB
  // (a few dozen steps per block) that latency costs more than the second pass saves — measured fp64, one pass / two
  // passes: 8300 x 1024 (65 MB) 59 / 49 us, 16 384 x 1024 (128 MB) 77 / 72, 32 768 x 1024 (256 MB) 105 / 118,
  // 8300 x 4096 (259 MB) 77 / 107, 16 384 x 8192 (1 GB) 239 / 343
  if ((double)p->A_f->m * (double)p->A_f->n * (double)p->es() < (double)p->ctx->one_pass_min_bytes) return false;
  // (very short rows: with two columns per thread and sixteen rows per step the two passes were as fast or faster below 320
  //  entries — 130: 7.7 against 7.0 ms at 6 GB ; the one-column shape with 32 rows per step reversed that: 130: 4.4, 200: 3.0,
  //  300: 2.1 ms against 7.0, 4.9, 3.6 — tools/shape_sweep.py.  Below ~125 columns the tall-skinny family takes over.)
  if (p->A_f->n < 128) return false;
  return p->A_f->m >= p->ctx->one_pass_min_rows && p->A_f->m < ((int64_t)1 << 30) && slab_shape(p).bs != 0;
}

// The kernel reads only the transposed copy.  A matrix of up to 16 GB that has none gets one here (the same second
// copy as apx_matrix_build_transpose; twice the memory, one transpose): every iteration then moves half the bytes.
// Failing to allocate it is not an error — the two passes remain.  apx_ctx_auto_transpose(ctx, 0) turns this off: the
// copy is then the caller's to build (apx_matrix_build_transpose) and to drop (apx_matrix_drop_transpose).
static bool slab_eligible(const apx_problem* p) {
  if (!slab_shape_ok(p)) return false;
  if (p->A_f->dT == nullptr && p->ctx->auto_transpose && (double)p->A_f->ld * (double)p->A_f->n_pad * (double)p->es() <= 16.0 * 1024 * 1024 * 1024) {
    if (apx_matrix_build_transpose(const_cast<apx_matrix*>(p->A_f)) != APX_OK) (void)hipGetLastError();
  }
  return p->A_f->dT != nullptr;
}

static int32_t slab_setup(apx_problem* p) {
  apx_ctx* ctx = p->ctx;
  const apx_matrix* A = p->A_f;
  std::unique_ptr<SlabState> st(new SlabState());
  const SlabShape sh = slab_shape(p);
  st->cpt = sh.cpt;
  st->bs = sh.bs;
  st->multi_waves = ctx->multi_waves;
  // as many blocks as fit the chip at once — one per CU, two of the two-columns-per-thread kind (~58 VGPRs): rows in
  // contiguous slabs, a multiple of the step ; fewer blocks = fewer partial vectors for the closing kernel
  int64_t B = (int64_t)ctx->num_cu * (st->cpt <= 2 ? 2 : 1);      // (kSlabMultiCpt, kSlabTeamCpt > 2: one block per CU)
  // rows dealt evenly: a block's last step may be partial (rows past its slab are clamped and weigh nothing), so nothing
  // is rounded up — rounding to 16 left 6 of C3's 256 CUs without a block
  const int64_t rpb = (A->m + B - 1) / B;
  B = (A->m + rpb - 1) / rpb;
  st->B = (int)B;
  int64_t rpb_team = 0;
  if (st->cpt == kSlabTeamCpt) {           // teams: the fewest members whose slice fits 5 vectors per compute thread
    const bool f32 = p->dtype == APX_F32;
    const TeamPlan pl = f32 ? team_plan(A->n, TeamTune{0, ctx->multi_waves, 0}, 4)
                            : team_plan(A->n, TeamTune{ctx->team_members, ctx->multi_waves, ctx->team_slots});      // (csrc/team_rows.h)
    if (pl.P == 0) return APX_OK;
    const int P = pl.P;
    st->plan = pl;
    B = ctx->num_cu / P;                   // slabs = teams ; ALL 256 workgroups are launched (32 / P teams per XCD)
    rpb_team = (A->m + B - 1) / B;
    st->B = (int)B;
    if (P > 1) {
      // the scheme's precondition, as far as it can be known up front: one workgroup of THIS kernel per CU, 256 of them
      const bool nt_q = true;
      const bool lg = p->f_kind == APX_F_LOGISTIC;
      const int per_cu = f32 ? (lg ? team_blocks_per_cu<APX_F_LOGISTIC, SlabDev, float>(pl, nt_q) : team_blocks_per_cu<APX_F_LEAST_SQUARES, SlabDev, float>(pl, nt_q))
                             : (lg ? team_blocks_per_cu<APX_F_LOGISTIC, SlabDev>(pl, nt_q) : team_blocks_per_cu<APX_F_LEAST_SQUARES, SlabDev>(pl, nt_q));
      if (per_cu < 1) return APX_OK;         // (it does not fit: no state, the two passes run)
    }
    APX_TRY(team_alloc(ctx, &st->team, (int)B, P, rpb_team, A->n * (int64_t)dtype_size(p->dtype)));
  }
  if (st->cpt == 14) {                     // long rows: k_slab2_rows, KV 16-byte vectors per thread and row
    const int64_t vw = 16 / (int64_t)dtype_size(p->dtype);
    st->kv = (int)((A->n + vw * 768 - 1) / (vw * 768));
    // (measured and dropped: fewer whole waves where KV vectors per thread over-cover a row — C3: 640 threads x 16 columns
    // = 10 240 for its 10 000 — ran 1 - 3 % slower than 768 threads: waves are dealt to four SIMDs, 10 or 11 of them
    // unevenly.  And Float32 rows of more than 9216 columns stay on the first-generation kernel: interleaved in one
    // process (tools/slab_ab.py, profiles/r02_one_pass_generations_ab.jsonl) it is 1 - 1.5 % faster there — its 14 x 768
    // column slots fit 10 000 columns better than 16 x 768 — while the second generation wins wherever its KV adapts:
    // 120 000 x 6000 fp32 0.47 vs 0.60 ms per iteration, 60 000 x 8000 fp64 0.60 vs 0.68, 50 000 x 10 000 fp64 0.64 vs 0.66)
    if (p->dtype == APX_F32 && st->kv == 4) st->v2 = false;
    if (ctx->slab_generation != 0) st->v2 = ctx->slab_generation == 2;      // (APX_OPT_SLAB_GENERATION: A/B runs)
  }
  SlabDev d{};
  d.AT = A->dT;
  d.ldT = A->ldT;
  d.m = (int32_t)A->m;
  d.ncols = (int32_t)A->n;
  d.n = (int32_t)p->n;
  d.rows_per_block = (int32_t)(rpb_team > 0 ? rpb_team : rpb);
  // rows dealt in chunks, the chunks cyclically over the blocks (csrc/team_dev.h ; the team kernels carry theirs in TeamDev)
  d.deal_stride = ctx->row_deal != 0 ? (int32_t)B : 1;
  d.deal_log = 0;
  if (ctx->row_deal < 0) d.deal_log = deal_chunk_log(A->n * (int64_t)dtype_size(p->dtype), d.rows_per_block);
  for (int r = ctx->row_deal; r > 1; r >>= 1) d.deal_log += 1;
  d.b = p->b.p;
  d.ldpart = round_up(d.ncols + 2, 8);
  // (k_multi_rows on very short rows leaves multi_groups(n) partial vectors per workgroup)
  const int64_t nparts = B * (st->cpt == kSlabMultiCpt ? multi_groups(A->n, p->dtype == APX_F32 ? 4 : 2) : 1);
  APX_TRY(st->part.alloc_zero((size_t)nparts * d.ldpart * sizeof(double), ctx->stream));
  d.part = (double*)st->part.p;
  d.c = p->ctrl();
  // measured (tools/slab_bench.py threshold, profiles/r01g_one_pass_variants.md): a matrix of up to 256 MB is partly
  // served by the Infinity Cache from one iteration to the next, and non-temporal loads forgo that (256 MB fp32: 81 vs
  // 87 us ; from 512 MB on they win, 121 vs 130) ; two row groups in flight pay from ~16 steps per block on (their
  // prologue and the doubled barrier count cost more than the overlap gains on a short slab)
  st->nt = (int64_t)A->m * A->n * (int64_t)dtype_size(p->dtype) >= ((int64_t)384 << 20);
  const int rows_per_step = st->cpt == 1 ? 32 : st->cpt == 2 ? 16 : kSlabRows;
  st->pipe = rpb / rows_per_step >= 16;
  st->dev = d;
  // the fused end of the iteration (SlabTail): five rows of wave partials (one per closing block, the entries beyond the last
  // wave of k_reduce_n's last block stay zero) and the snapshot of the control block
  const int64_t ldw = round_up(4 * ((p->n + kBlock - 1) / kBlock), 64) + 64;
  APX_TRY(st->tail_buf.alloc_zero((size_t)(5 * ldw) * sizeof(double) + sizeof(Ctrl) + 64, ctx->stream));
  p->slab = st.release();
  return APX_OK;
}

// the variant of the row kernel for this matrix (SlabState::pipe, ::nt, chosen at setup)
template <typename T, int FK, int C, int R, int BS, bool PIPE_OK>
static void slab_rows_launch(const SlabState* st, dim3 grid, hipStream_t q, const SlabDev& d, const void* x) {
  if constexpr (PIPE_OK) {
    if (st->pipe) {
      if (st->nt) hipLaunchKernelGGL((k_slab_rows<T, FK, C, R, BS, true, true>), grid, dim3(BS), 0, q, d, x);
      else hipLaunchKernelGGL((k_slab_rows<T, FK, C, R, BS, true, false>), grid, dim3(BS), 0, q, d, x);
      return;
    }
  }
  if (st->nt) hipLaunchKernelGGL((k_slab_rows<T, FK, C, R, BS, false, true>), grid, dim3(BS), 0, q, d, x);
  else hipLaunchKernelGGL((k_slab_rows<T, FK, C, R, BS, false, false>), grid, dim3(BS), 0, q, d, x);
}

// k_slab2_rows for this matrix: KV by the row length ; rows in flight by what the registers hold (170 per thread at 768
// threads: x and the accumulator take 2 KV (Float32: 3 KV) ... of them, a row 4 KV)
template <typename T, int FK, int KV, int H, int G>
static void slab2_rows_launch(const SlabState* st, dim3 grid, hipStream_t q, const SlabDev& d, const void* x) {
  if (st->nt) hipLaunchKernelGGL((k_slab2_rows<T, FK, KV, H, G, true>), grid, dim3((unsigned)st->bs2), 0, q, d, x);
  else hipLaunchKernelGGL((k_slab2_rows<T, FK, KV, H, G, false>), grid, dim3((unsigned)st->bs2), 0, q, d, x);
}
template <typename T, int FK>
static bool slab2_dispatch(const SlabState* st, dim3 grid, hipStream_t q, const SlabDev& d, const void* x) {
  if constexpr (sizeof(T) == 4) {
    switch (st->kv) {                      // 4 columns per vector: up to 6144 / 9216 / 12 288 columns
      case 2: slab2_rows_launch<T, FK, 2, 2, 2>(st, grid, q, d, x); return true;
      case 3: slab2_rows_launch<T, FK, 3, 2, 2>(st, grid, q, d, x); return true;
      case 4: slab2_rows_launch<T, FK, 4, 2, 2>(st, grid, q, d, x); return true;
      default: return false;
    }
  } else {
    // (Float64 logistic: exp / log in Float64 take ~50 registers of their own and every shape of this kernel spills
    // with them — such rows take k_team_rows with P = 1, whose communication wave evaluates the link: slab_shape)
    if constexpr (FK == APX_F_LOGISTIC) return false;
    else switch (st->kv) {                 // 2 columns per vector: up to 6144 / 7680 / 9216 / 10 752 columns
      case 4: slab2_rows_launch<T, FK, 4, 2, 2>(st, grid, q, d, x); return true;
      case 5: slab2_rows_launch<T, FK, 5, 2, 2>(st, grid, q, d, x); return true;
      case 6: slab2_rows_launch<T, FK, 6, 1, 3>(st, grid, q, d, x); return true;
      case 7: slab2_rows_launch<T, FK, 7, 1, 2>(st, grid, q, d, x); return true;      // (three rows in flight spill)
      default: return false;
    }
  }
}

// the closing kernel: with the end of the iteration riding in it (SlabTail) its grid covers all n entries of the solver's
// vectors — the logistic bias, entry ncols, included
template <typename T, int FK>
static void slab_close_launch(const SlabDev& d, int nparts, void* grad, const SlabTail& tl, hipStream_t q) {
  if (tl.enabled)
    hipLaunchKernelGGL((k_slab_close<T, FK, true>), dim3((unsigned)((d.n + kSlabCloseCols - 1) / kSlabCloseCols)), dim3(kSlabCloseBlock), 0, q,
                       d, nparts, (T*)grad, tl);
  else
    hipLaunchKernelGGL((k_slab_close<T, FK, false>), dim3((unsigned)((d.ncols + kSlabCloseCols - 1) / kSlabCloseCols)), dim3(kSlabCloseBlock), 0,
                       q, d, nparts, (T*)grad, tl);
}

template <typename T, int FK>
static bool slab_launch_t(SlabState* st, bool cooperative, const SlabDev& d, const void* x, void* grad, const SlabTail& tl, hipStream_t q) {
  const dim3 grid((unsigned)st->B);
  if (st->plan.P != 0) {
    const bool ok = team_launch<FK, SlabDev, T>(st->plan, st->B, st->nt, cooperative, d, st->team.dev, x, q);
    if (!ok) return false;                         // (a refused launch: nothing runs, the closing kernel included)
    slab_close_launch<T, FK>(d, st->B, grad, tl, q);
    return true;
  }
  if (st->cpt == kSlabMultiCpt) {
    if (st->nt) multi_launch<T, FK, true, SlabDev>(d.ncols, st->B, st->multi_waves, d, x, q);
    else multi_launch<T, FK, false, SlabDev>(d.ncols, st->B, st->multi_waves, d, x, q);
    slab_close_launch<T, FK>(d, st->B * multi_groups(d.ncols, 16 / (int)sizeof(T)), grad, tl, q);
    return true;
  }
  constexpr int R14 = sizeof(T) == 4 ? kSlabRows : (FK == APX_F_LOGISTIC ? 1 : kSlabRows / 2);
  constexpr bool kPipeOk = sizeof(T) == 4 || FK == APX_F_LEAST_SQUARES;       // (the others spill as two groups)
  if (st->cpt == 1) slab_rows_launch<T, FK, 1, 32, 1024, kPipeOk>(st, grid, q, d, x);
  else if (st->cpt == 2 && st->bs == 768) slab_rows_launch<T, FK, 2, 16, 768, kPipeOk>(st, grid, q, d, x);
  else if (st->cpt == 2) slab_rows_launch<T, FK, 2, 16, 1024, kPipeOk>(st, grid, q, d, x);
  else if (st->cpt == 3) slab_rows_launch<T, FK, 3, kSlabRows, 1024, kPipeOk>(st, grid, q, d, x);
  else if (st->cpt == 4) slab_rows_launch<T, FK, 4, kSlabRows, 1024, kPipeOk>(st, grid, q, d, x);
  else if (st->cpt == 5) slab_rows_launch<T, FK, 5, kSlabRows, 1024, kPipeOk>(st, grid, q, d, x);
  else if (!(st->kv != 0 && st->v2 && slab2_dispatch<T, FK>(st, grid, q, d, x)))
    slab_rows_launch<T, FK, 14, R14, 768, (sizeof(T) == 4)>(st, grid, q, d, x);
  slab_close_launch<T, FK>(d, st->B, grad, tl, q);
  return true;
}

// the row kernel of this state in words (apx_ctx_kernel_name: the bench line quotes the library, not a string of its own)
static std::string slab_kernel_name(const SlabState* st, const apx_problem* p) {
  char b[256];
  const char* fk = p->f_kind == APX_F_LOGISTIC ? "logistic" : "least_squares";
  const char* ty = p->dtype == APX_F32 ? "float" : "double";
  if (st->plan.P != 0)
    snprintf(b, sizeof b, "k_team_rows<%s%s, %d member%s%s, %d vectors per thread, %d compute waves, %d row slots, %d row%s of slack>", p->dtype == APX_F32 ? "float, " : "", fk, st->plan.P,
             st->plan.P == 1 ? "" : "s", st->team.dev.interleave ? " (shares interleaved)" : "", std::max(st->plan.kv, st->plan.P == 1 ? 4 : 2), st->plan.waves, (st->plan.big || st->plan.P == 1) ? 3 : 4,
             (st->plan.big || st->plan.P == 1) ? 1 : 2, (st->plan.big || st->plan.P == 1) ? "" : "s");
  else if (st->cpt == kSlabMultiCpt) {
    const int vw = p->dtype == APX_F32 ? 4 : 2;
    const int64_t nvec = ((int64_t)st->dev.ncols + vw - 1) / vw;
    const int gr = multi_groups(st->dev.ncols, vw), w = gr > 1 ? 0 : multi_waves(nvec, st->multi_waves);
    if (gr > 1) snprintf(b, sizeof b, "k_multi_rows<%s, %s, 1 vector per thread, %d row groups>", ty, fk, gr);
    else snprintf(b, sizeof b, "k_multi_rows<%s, %s, %d vectors per thread, %d compute waves>", ty, fk, (int)((nvec + w * 64 - 1) / (w * 64)), w);
  }
  else if (st->cpt == 14 && st->kv != 0 && st->v2 && !(p->dtype == APX_F64 && p->f_kind == APX_F_LOGISTIC))
    snprintf(b, sizeof b, "k_slab2_rows<%s, %s, %d vectors per thread>", ty, fk, st->kv);
  else
    snprintf(b, sizeof b, "k_slab_rows<%s, %s, %d columns per thread, %d threads%s%s>", ty, fk, st->cpt, st->bs, st->pipe ? ", two row groups" : "",
             st->nt ? ", non-temporal loads" : "");
  return std::string(b) + " + k_slab_close";
}

// f(x) -> Ctrl, grad f(x) -> p->grad for the device vector x: what eval_f + pullback_f leave behind.  Called by eval_f
// (solver.hip) for every least-squares / logistic evaluation: *done_here stays false when the problem is not eligible (or
// its transposed copy is gone), and the two passes run.  An evaluation whose gradient is never asked for (a rejected
// linesearch trial) costs the same one pass as before.
int32_t slab_eval_with_gradient(apx_problem* p, const void* x, bool* done_here) {
  *done_here = false;
  if (!p->slab_checked) {
    p->slab_checked = true;
    // Every rank of a row-sharded problem must take the same path — the two paths enqueue different collectives — but whether
    // THIS rank can take the one pass depends on its own shard (its row count against the thresholds, whether its transposed
    // copy could be allocated, its device): the ranks agree once, here, on the first evaluation of the problem (all of them
    // reach it together), and take the one pass only if every one of them can.
    int32_t rc = APX_OK;
    bool mine = slab_eligible(p);
    if (mine) {
      rc = slab_setup(p);
      mine = rc == APX_OK && p->slab != nullptr;
    }
    double cannot = mine ? 0.0 : 1.0;
    APX_TRY(ranks_sum(p, cannot, &cannot));
    if (cannot != 0.0 && p->slab != nullptr) {
      slab_destroy(p->slab);
      p->slab = nullptr;
    }
    APX_TRY(rc);
  }
  SlabState* st = p->slab;
  if (st == nullptr || p->A_f->dT == nullptr) return APX_OK;      // (dropped copy: A edited in place — every rank's, if any's)
  if (st->plan.P > 1 && p->teams_dead()) return APX_OK;           // (a hand-off failed earlier: the two passes from now on)
  apx_ctx* ctx = p->ctx;
  SlabDev d = st->dev;
  d.AT = p->A_f->dT;                             // (rebuilt copies may live elsewhere)
  d.ldT = p->A_f->ldT;
  d.fsc_out = p->f_sharded ? p->fsc() : nullptr;
  // the end of the iteration in the closing kernel, when the caller is the generic loop and has asked for it (solver.hip
  // enqueue_iteration ; not for a row shard: the all-reduce of the gradient sits between the two)
  SlabTail tl{};
  if (p->tail_req != nullptr && !p->f_sharded && ctx->fused_tail && p->n <= (int64_t)kMaxRedBlocks * kBlock) {
    static_cast<SlabTailReq&>(tl) = *p->tail_req;
    tl.ldw = (int32_t)(round_up(4 * ((p->n + kBlock - 1) / kBlock), 64) + 64);
    tl.wred = (double*)st->tail_buf.p;
    tl.snap = (Ctrl*)((char*)st->tail_buf.p + (size_t)5 * tl.ldw * sizeof(double));
    tl.enabled = 1;
    SlabTailReq* rq = (SlabTailReq*)p->tail_req;      // (what the step kernel needs of it)
    rq->wred = tl.wred;
    rq->ldw = tl.ldw;
    rq->snap = tl.snap;
    p->tail_done = true;
  }
  const bool ls = p->f_kind == APX_F_LEAST_SQUARES;
  if (st->kname.empty()) st->kname = slab_kernel_name(st, p);
  ctx->kname[APX_KSTAT_ONE_PASS] = st->kname;
  APX_TRY(prof_begin(ctx, APX_KSTAT_ONE_PASS, (double)p->A_f->m * (double)p->A_f->n * (double)p->es()));
  // (a cooperative launch can be refused, and a refusal is ONE rank's: on a row shard the peers would be left in the iteration's
  //  collectives — the option is not honoured there )
  const bool coop = ctx->team_cooperative && !p->sharded();
  bool launched;
  if (p->dtype == APX_F32) {
    launched = ls ? slab_launch_t<float, APX_F_LEAST_SQUARES>(st, coop, d, x, p->grad.p, tl, ctx->stream)
                  : slab_launch_t<float, APX_F_LOGISTIC>(st, coop, d, x, p->grad.p, tl, ctx->stream);
  } else {
    launched = ls ? slab_launch_t<double, APX_F_LEAST_SQUARES>(st, coop, d, x, p->grad.p, tl, ctx->stream)
                  : slab_launch_t<double, APX_F_LOGISTIC>(st, coop, d, x, p->grad.p, tl, ctx->stream);
  }
  APX_TRY(prof_end(ctx));
  if (!launched) return team_launch_refused(p, "one-pass kernel (teams)");
  APX_HIP(hipGetLastError());
  *done_here = true;
  return APX_OK;
}

// The end of the iteration behind the all-reduce of a row-sharded f (k_slab_tail): true in *done when it took the request
// (p->tail_req, the fused tail switched on, n within the partial buffer) — f's value is final in Ctrl then and the caller's
// step is vo_step_rule ; else the caller closes f and runs the separate kernels
int32_t slab_tail_after_allreduce(apx_problem* p, bool* done) {
  *done = false;
  SlabState* st = p->slab;
  apx_ctx* ctx = p->ctx;
  if (st == nullptr || p->tail_req == nullptr || !ctx->fused_tail || p->n > (int64_t)kMaxRedBlocks * kBlock) return APX_OK;
  SlabTail tl{};
  static_cast<SlabTailReq&>(tl) = *p->tail_req;
  tl.ldw = (int32_t)(round_up(4 * ((p->n + kBlock - 1) / kBlock), 64) + 64);
  tl.wred = (double*)st->tail_buf.p;
  tl.snap = (Ctrl*)((char*)st->tail_buf.p + (size_t)5 * tl.ldw * sizeof(double));
  tl.enabled = 1;
  SlabTailReq* rq = (SlabTailReq*)p->tail_req;
  rq->wred = tl.wred;
  rq->ldw = tl.ldw;
  rq->snap = tl.snap;
  const dim3 grid((unsigned)((p->n + kBlock - 1) / kBlock));
  const bool ls = p->f_kind == APX_F_LEAST_SQUARES;
  if (p->dtype == APX_F32) {
    if (ls) hipLaunchKernelGGL((k_slab_tail<float, APX_F_LEAST_SQUARES>), grid, dim3(kBlock), 0, ctx->stream, p->ctrl(), (const float*)p->grad.p, p->fsc(), p->n, tl);
    else hipLaunchKernelGGL((k_slab_tail<float, APX_F_LOGISTIC>), grid, dim3(kBlock), 0, ctx->stream, p->ctrl(), (const float*)p->grad.p, p->fsc(), p->n, tl);
  } else {
    if (ls) hipLaunchKernelGGL((k_slab_tail<double, APX_F_LEAST_SQUARES>), grid, dim3(kBlock), 0, ctx->stream, p->ctrl(), (const double*)p->grad.p, p->fsc(), p->n, tl);
    else hipLaunchKernelGGL((k_slab_tail<double, APX_F_LOGISTIC>), grid, dim3(kBlock), 0, ctx->stream, p->ctrl(), (const double*)p->grad.p, p->fsc(), p->n, tl);
  }
  APX_HIP(hipGetLastError());
  p->tail_done = true;
  *done = true;
  return APX_OK;
}

// (their launches carry a ticket base and a sequence base that advance on the host: not replayable as a captured graph)
bool slab_uses_teams(const apx_problem* p) { return p->slab != nullptr && p->slab->plan.P > 1 && !p->teams_dead(); }

// The error word of the team hand-off (0: fine), read wherever a result is about to be used.  On a row shard the word is
// SUMMED over the ranks: every rank reads it at the same points of the same control flow (the convergence poll at fixed
// iteration numbers, the result, an evaluation whose value is needed), so a failure on any rank is a failure on all of them at
// the same iteration — nobody is left inside a collective its peers have abandoned, and all of them fall back together.
int32_t slab_check(apx_problem* p) {
  SlabState* st = p->slab;
  if (st == nullptr || st->plan.P <= 1 || p->teams_dead()) return APX_OK;      // (reported once: the failure is the problem's state now)
  apx_ctx* ctx = p->ctx;
  int32_t err = 0;
  APX_TRY(team_read_err(ctx, &st->team, &err));
  double any = err != 0 ? 1.0 : 0.0;
  APX_TRY(ranks_sum(p, any, &any));
  if (any == 0.0) return APX_OK;
  p->team_failed = true;
  const std::string what = err != 0 ? team_describe(st->team.pinned->diag, "one-pass kernel (teams)")
                                    : std::string("one-pass kernel (teams): a team hand-off failed on another rank of this row-sharded problem "
                                                  "(its message has the details); the result is invalid and was discarded");
  note_incident(what);
  return fail(APX_ERR_HIP, what);
}

// hand-off statistics of the problem's team launches (apx_problem_team_waits)
int32_t slab_team_waits(apx_problem* p, int64_t out[4]) {
  out[0] = out[1] = out[2] = out[3] = 0;
  SlabState* st = p->slab;
  if (st == nullptr || st->plan.P <= 1) return APX_OK;
  APX_TRY(team_read_waits(p->ctx, &st->team, out));
  out[2] *= 10;                                  // ticks of 10 ns
  out[3] = st->team.dev.launch;
  return APX_OK;
}

// ... slot by slot (apx_problem_team_wait_slots: where and when each workgroup slot saw its longest wait)
int32_t slab_team_wait_slots(apx_problem* p, const TeamWait** slots) {
  *slots = nullptr;
  SlabState* st = p->slab;
  if (st == nullptr || st->plan.P <= 1) return APX_OK;
  return team_read_wait_slots(p->ctx, &st->team, slots);
}

}  // namespace apx
