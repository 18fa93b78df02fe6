// slab_pd.hip — the linear-operator half of the primal-dual iteration (src/AdaProx.jl:335, :344-347, :358) in ONE pass
// over A: A_x = A x ; w = y + sigma ((1 + rho) A_x - rho A_x_prev) ; y = prox_{sigma h*}(w) ; dual_res = (w - y) / sigma
// - A_x ; At_y = A' y — for a separable h (the least-absolute-deviation clamp, experiments/least_absolute_deviation/
// runme.jl:39-41) with rows of at most 10 752 entries.  The generic path streams A twice per iteration (A x, then
// A' y); but y[i] depends on row i of A only, so a block that owns a slab of rows and keeps R of them at a time in
// registers (csrc/slab.hip's scheme: thread t holds the columns t, t + BS, ..., read from the transposed copy) can
// form A_x[i], update y[i] and add y[i] * A[i,:] to a private column accumulator from the SAME registers.  The B
// partial n-vectors and the dual-residual sums are closed by a second small kernel, which also runs the end of the
// iteration (norm_res, record, termination: :348-354).  No atomics: bit-reproducible.
#include <atomic>
#include <cstdlib>
#include <memory>

#include "skinny_dev.h"
#include "solver_internal.h"
#include "team_dev.h"
#include "vecops_dev.h"

namespace apx {

struct SlabPdDev {
  const void* AT;         // A' column-major: row i of A at AT + i * ldT, contiguous
  int64_t ldT;
  int32_t m, n, rows_per_block, _pad;
  void* Y;                // y (T), updated in place
  void* AXP;              // A_x_prev (T), updated in place (:360)
  double* part;           // [blocks][ldpart]: n column partials of A' y, then sum dual_res^2, raw h(A_x)
  int64_t ldpart;
  Ctrl* c;
  ProxDev h;
};

struct SlabPdState {
  DBuf part;
  SlabPdDev dev{};
  int B = 0, cpt = 0, bs = 0, rows = 0;
  bool nt = true;         // non-temporal loads of A (a matrix beyond the Infinity Cache)
  // rows of 10 753 ... 21 504 Float64 entries: teams of two workgroups (k_teampd_rows)
  int team_kv = 0;        // 16-byte vectors per thread and row of a member's half (0: not this scheme)
  DBuf team_buf;          // ring, tickets, err
  TeamDev team{};
  int32_t* h_err = nullptr;  // pinned copy of err
  ~SlabPdState() { if (h_err) (void)hipHostFree(h_err); }
};
void slab_pd_destroy(SlabPdState* st) { delete st; }

// H rows of A into registers (csrc/slab.hip's slab_load) together with the per-row state of the rows this thread may
// finish: y, A_x_prev and the shift of h — every thread, unconverted, so that nothing waits for the loads here
template <typename T, int C, int H, bool NT>
__device__ __forceinline__ void slabpd_load(T (&a)[H][C], T& y_in, T& axp_in, T& b_in, const SlabPdDev& d,
                                            const unsigned (&off)[C], int r0, int row_end, int tid) {
  const T* __restrict__ AT = (const T*)d.AT;
#pragma unroll
  for (int r = 0; r < H; ++r) {
    const int ri = __builtin_amdgcn_readfirstlane(min(r0 + r, row_end - 1));
    const T* __restrict__ row = AT + (int64_t)ri * d.ldT;
#pragma unroll
    for (int k = 0; k < C; ++k)                  // scalar row base + 32-bit BYTE offset: one address register per load
      a[r][k] = NT ? __builtin_nontemporal_load((const T*)((const char*)row + off[k]))   // (read once: past the caches,
                   : *(const T*)((const char*)row + off[k]);                             //  unless A fits the Infinity Cache)
  }
  const int ic = min(r0 + min(tid, H - 1), row_end - 1);
  y_in = ((const T*)d.Y)[ic];
  axp_in = ((const T*)d.AXP)[ic];
  b_in = d.h.has_shift ? ((const T*)d.h.shift)[ic] : (T)0;
}

// the H rows r0 .. r0 + H - 1 held in a: A_x, the dual step of each (thread r finishes row r0 + r: :344-347, :360) and
// y[i] * A[i,:] into the column accumulator ; two barriers
template <typename T, int C, int H, int BS, bool TIGHT>
__device__ __forceinline__ void slabpd_group(const T (&a)[H][C], T y_in, T axp_in, T b_in, const T (&xr)[C], double (&acc)[C],
                                             const SlabPdDev& d, double sigma, double rho, double& e0, double& e1, int r0,
                                             int row_end, double (*wsum)[(H < 2 ? 2 : H)], double* y_s, int tid) {
  constexpr int RP = H < 2 ? 2 : H;                // (the multi-value wave sum wants at least two)
  const int lane = tid & 63, wv = tid >> 6;
  T* __restrict__ Y = (T*)d.Y;
  T* __restrict__ AXP = (T*)d.AXP;
  double p[RP];
#pragma unroll
  for (int r = 0; r < RP; ++r) p[r] = 0.0;
#pragma unroll
  for (int k = 0; k < C; ++k) {
    T xv = xr[k];
    if constexpr (sizeof(T) == 4) asm volatile("" : "+v"(xv));        // (x stays in T: see csrc/slab.hip)
    const double xd = (double)xv;
#pragma unroll
    for (int r = 0; r < H; ++r) p[r] = fma((double)a[r][k], xd, p[r]);
    if constexpr (TIGHT && sizeof(T) == 4) __builtin_amdgcn_sched_barrier(0);
  }
  wave_sum_multi<RP>(p);                           // lane l: the wave's sum for row l / (64 / RP)
  if (lane % (64 / RP) == 0) wsum[wv][lane / (64 / RP)] = p[0];
  __syncthreads();
  if (tid < H) {                                   // thread r finishes row r0 + r
    double dot = wsum[0][tid];
#pragma unroll
    for (int w = 1; w < BS / 64; ++w) dot += wsum[w][tid];
    const int i = r0 + tid;
    double yn = 0.0;
    if (i < row_end) {
      const double ax = (double)(T)dot;                                                   // :335, rounded to T like every stored A x
      const double b = (double)b_in;
      const double w = (double)y_in + sigma * ((1.0 + rho) * ax - rho * (double)axp_in);  // :344
      const T ynt = (T)prox_conj_elem_s<T>(d.h, w, sigma, i, 0.0, b);                     // :345
      yn = (double)ynt;
      const double dr = (w - yn) / sigma - ax;                                            // :347
      e0 += dr * dr;
      e1 += gval_elem_s<T>(d.h, ax, i, b);
      Y[i] = ynt;
      AXP[i] = (T)ax;                                                                     // :360
    }
    y_s[tid] = yn;
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < H; ++r) {
    const double yy = y_s[r];
#pragma unroll
    for (int k = 0; k < C; ++k) {
      T av = a[r][k];
      if constexpr (sizeof(T) == 4) asm volatile("" : "+v"(av));      // (convert again: see csrc/slab.hip)
      acc[k] = fma((double)av, yy, acc[k]);
    }
    if constexpr (TIGHT && sizeof(T) == 4) __builtin_amdgcn_sched_barrier(0);
  }
  if constexpr (TIGHT) {                           // (the sums complete HERE, not sunk past the next loads: csrc/slab.hip)
#pragma unroll
    for (int k = 0; k < C; ++k) asm volatile("" : "+v"(acc[k]));
  }
}

// PIPE: the R rows as two groups of R / 2 that alternate, one computing while the other's loads arrive (csrc/slab.hip)
template <typename T, int C, int R, int BS, bool PIPE, bool NT>
__global__ __launch_bounds__(BS) void k_slabpd_rows(SlabPdDev d, const void* xv) {
  constexpr int H = PIPE ? R / 2 : R;
  static_assert(!PIPE || R % 2 == 0, "two groups");
  constexpr int RP = H < 2 ? 2 : H;
  __shared__ double wsum[BS / 64][RP];
  __shared__ double y_s[RP];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (d.c->done != 0) return;                      // written by an earlier launch: block-uniform
  const double sigma = d.c->sigma, rho = d.c->rho; // this iteration's, set by the rule kernel before this launch
  const T* __restrict__ X = (const T*)xv;
  T xr[C];
  double acc[C];
  unsigned off[C];
#pragma unroll
  for (int k = 0; k < C; ++k) {
    const int j = tid + k * BS;
    off[k] = (unsigned)min(j, d.n - 1) * (unsigned)sizeof(T);         // in bytes ; clamped: a valid address, weight 0
    xr[k] = j < d.n ? X[j] : (T)0;
    acc[k] = 0.0;
  }
  double e0 = 0.0, e1 = 0.0;                       // sum dual_res^2, raw h(A_x) of the rows this thread finishes
  const int row_begin = blockIdx.x * d.rows_per_block;
  const int row_end = min(row_begin + d.rows_per_block, d.m);
  if constexpr (!PIPE) {
    for (int r0 = row_begin; r0 < row_end; r0 += R) {
      T a[H][C];
      T y_in, axp_in, b_in;
      slabpd_load<T, C, H, NT>(a, y_in, axp_in, b_in, d, off, r0, row_end, tid);
      slabpd_group<T, C, H, BS, false>(a, y_in, axp_in, b_in, xr, acc, d, sigma, rho, e0, e1, r0, row_end, wsum, y_s, tid);
    }
  } else {
    T a0[H][C], a1[H][C];
    T y0, y1, p0, p1, b0, b1;
    __builtin_amdgcn_sched_barrier(0);
    slabpd_load<T, C, H, NT>(a0, y0, p0, b0, d, off, row_begin, row_end, tid);
    __builtin_amdgcn_sched_barrier(0);             // (in this order: the waits of the loop count the loads issued after a group's)
    slabpd_load<T, C, H, NT>(a1, y1, p1, b1, d, off, row_begin + H, row_end, tid);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
    for (int r0 = row_begin; r0 < row_end; r0 += R) {
      slabpd_group<T, C, H, BS, true>(a0, y0, p0, b0, xr, acc, d, sigma, rho, e0, e1, r0, row_end, wsum, y_s, tid);
      __builtin_amdgcn_sched_barrier(0);
      slabpd_load<T, C, H, NT>(a0, y0, p0, b0, d, off, r0 + R, row_end, tid);
      __builtin_amdgcn_sched_barrier(0);
      slabpd_group<T, C, H, BS, true>(a1, y1, p1, b1, xr, acc, d, sigma, rho, e0, e1, r0 + H, row_end, wsum, y_s, tid);
      __builtin_amdgcn_sched_barrier(0);
      slabpd_load<T, C, H, NT>(a1, y1, p1, b1, d, off, r0 + R + H, row_end, tid);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  double* __restrict__ out = d.part + (int64_t)blockIdx.x * d.ldpart;
#pragma unroll
  for (int k = 0; k < C; ++k) {
    const int j = tid + k * BS;
    if (j < d.n) out[j] = acc[k];
  }
  if (wv == 0) {                                   // the rows were finished by threads 0 .. H-1 (one wave)
    double ex[2] = {tid < H ? e0 : 0.0, tid < H ? e1 : 0.0};
    wave_sum_multi<2>(ex);                         // lane l: ex[l / 32]
    if ((lane & 31) == 0) out[d.n + (lane >> 5)] = ex[0];
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Rows too long for one workgroup's registers (10 753 ... 21 504 Float64 entries — BASELINE config C5: 20 000, a row is
// 160 KB): TEAMS of two workgroups, one per CU, share a slab of rows.  Member s keeps columns [s n/2, (s+1) n/2) of x and
// of the column accumulator in registers and streams that half of every row (16-byte buffer loads, csrc/slab.hip's
// second-generation mapping).  Per row a member forms its partial dot product, PUBLISHES it, goes on to the dots of the
// next row, and only then takes the partner's partial of the first — the hand-off (1 - 3 us, MI355X_MICROARCH.md) is off
// the critical path.  Both members add the two partials in member order: the same `A x`, the same y+, no atomics on data.
//
// What the hand-off must not do is touch the vector-memory counter.  Loads return in order per wave, so a release
// fence, or a polling vector load, waits for every load of A the wave has in flight and the pipeline drains at every
// row: that is the 23 - 28 us per hand-off of round 1's probe.  Hence:
//   * a record is two words {value, value ^ sequence} stored by one lane, fire and forget — no fence; the check word
//     validates the pair, whatever order the words become visible in;
//   * every WAVE takes the partner's record itself with a scalar load (`s_load_dwordx4 ... glc`: its own counter, past
//     the scalar cache) requested before the dots of the next row and validated after them — no broadcast through LDS,
//     ONE barrier per row (the partial dots of the waves; each wave sums them itself);
//   * scalar loads see the L2 of the XCD, so teams are formed INSIDE an XCD at run time (XCC_ID + a ticket per XCD),
//     whatever order the dispatcher deals workgroups in;
//   * every wait is bounded by the wall clock (20 ms) and ends in an error word the host reads: the grid always drains.
// Rows past the slab are dummy rows (clamped loads, weight 0) that both members run alike, so the loop body has no
// branch between a request and its use.  tools/split_probe.hip: 8.0 GB of C5 in 1.28 ms = 6.3 TB/s, against 2 x 1.13 ms
// for the two GEMV passes.
// ---------------------------------------------------------------------------------------------------------------------
template <int KV, int G, bool NT, bool XLDS>
__global__ __launch_bounds__(768) void k_teampd_rows(SlabPdDev d, TeamDev t, const void* xv) {
  extern __shared__ double team_xs[];              // XLDS: the member's half of x in LDS (KV * 768 * 2 doubles: 84 KB at KV = 7),
  constexpr int BS = 768, NW = BS / 64;            //       where x in registers would make the row state spill
  typedef double vec_t __attribute__((ext_vector_type(2)));
  typedef unsigned raw_t __attribute__((ext_vector_type(4)));
  typedef unsigned long long rec_t __attribute__((ext_vector_type(2)));
  __shared__ double wsum[2][NW];
  __shared__ int ident[2];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);       // (provably wave-uniform: scalar branches, SGPR values)
  if (tid == 0) {                                  // (every launch takes its 32 tickets per XCD, a finished loop's too)
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u;      // HW_REG_XCC_ID
    const unsigned k = atomicAdd(t.tickets + xcc, 1u) - t.launch * 32u;
    ident[0] = k < 32u ? (int)(xcc * 16 + k / 2) : -1;
    ident[1] = (int)(k % 2);
    if (k >= 32u) atomicCAS(t.err, 0, 2);
  }
  __syncthreads();
  const int team = __builtin_amdgcn_readfirstlane(ident[0]), s = __builtin_amdgcn_readfirstlane(ident[1]);
  if (team < 0 || d.c->done != 0) return;          // (done: written by an earlier launch, uniform over the grid)
  const double sigma = d.c->sigma, rho = d.c->rho; // this iteration's, set by the rule kernel before this launch
  const double* __restrict__ X = (const double*)xv;
  const int nvs = ((d.n + 1) / 2 + 1) / 2;         // 16-byte vectors per member
  const int v0 = s * nvs;
  vec_t xr[KV];
  double acc[KV][2];
#pragma unroll
  for (int k = 0; k < KV; ++k) {
    const int lv = tid + k * BS;
    const int j = (v0 + lv) * 2;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      xr[k][e] = (lv < nvs && j + e < d.n) ? X[j + e] : 0.0;
      acc[k][e] = 0.0;
      if constexpr (XLDS) team_xs[lv * 2 + e] = xr[k][e];      // (each thread reads back only what it wrote: no barrier)
    }
  }
  const int voff = tid * 16;
  const int64_t cbase = (int64_t)v0 * 16;
  const int slice_bytes = (int)min((int64_t)nvs * 16, d.ldT * 8 - cbase);      // (rows are zero-padded to whole vectors)
  const int row_begin = team * t.rows_per_team;
  const int row_end = min(row_begin + t.rows_per_team, d.m);
  const int nrows = max(row_end - row_begin, 0);
  double e0 = 0.0, e1 = 0.0;
  vec_t a[G][KV];
  double y_in[G], axp_in[G], b_in[G];              // the row's state rides behind its loads of A (every thread: one address)
  const double* __restrict__ AT = (const double*)d.AT;
  double* __restrict__ Y = (double*)d.Y;
  double* __restrict__ AXP = (double*)d.AXP;
  auto load = [&](int g, int row) {
    const int ri = __builtin_amdgcn_readfirstlane(max(min(row, row_end - 1), 0));
    const __amdgpu_buffer_rsrc_t rs =
        __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)(AT + (int64_t)ri * d.ldT) + cbase), 0, slice_bytes, 0x00020000);
#pragma unroll
    for (int k = 0; k < KV; ++k) {
      const raw_t raw = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, k * BS * 16, NT ? 2 : 0);
      a[g][k] = __builtin_bit_cast(vec_t, raw);
    }
    y_in[g] = Y[ri];
    axp_in[g] = AXP[ri];
    b_in[g] = d.h.has_shift ? ((const double*)d.h.shift)[ri] : 0.0;
  };
  TeamRec* ring = t.ring + (int64_t)team * kTeamSlots * 2;
  // the workgroup's partial dot of row `seq` of the slab (slot g): every thread returns it ; thread 0 publishes it
  auto dots = [&](int g, long long seq) -> double {
    double p = 0.0;
#pragma unroll
    for (int k = 0; k < KV; ++k)
#pragma unroll
      for (int e = 0; e < 2; ++e) p = fma(a[g][k][e], XLDS ? team_xs[(tid + k * BS) * 2 + e] : xr[k][e], p);
    p = wave_sum(p);
    if (lane == 0) wsum[seq & 1][wv] = p;          // (double-buffered by row parity: one barrier per row)
    __syncthreads();
    double own = wsum[seq & 1][0];
#pragma unroll
    for (int w = 1; w < NW; ++w) own += wsum[seq & 1][w];
    if (tid == 0) {
      const unsigned long long bits = __builtin_bit_cast(unsigned long long, own);
      const rec_t rec = {bits, bits ^ (t.seq0 + (unsigned long long)seq + 1)};
      *(volatile rec_t*)(ring + (seq % kTeamSlots) * 2 + s) = rec;
    }
    return own;
  };
  bool dead = false;
  // row `seq` (slot g): the partner's partial (requested a phase ago), the dual step (:344-347, :360), y+[i] * A[i, half]
  auto finish = [&](int g, long long seq, double own, raw_t early) {
    const unsigned long long want = t.seq0 + (unsigned long long)seq + 1;
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(early)::"memory");
    unsigned long long v = ((unsigned long long)early[1] << 32) | early[0];
    unsigned long long chk = ((unsigned long long)early[3] << 32) | early[2];
    if ((v ^ chk) != want && !dead) {              // not there yet (or half written): poll, bounded
      const TeamRec* src = ring + (seq % kTeamSlots) * 2 + (1 - s);
      const long long t0 = wall_clock64();
      for (;;) {
        raw_t raw;
        asm volatile("s_load_dwordx4 %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(raw) : "s"(src) : "memory");
        v = ((unsigned long long)raw[1] << 32) | raw[0];
        chk = ((unsigned long long)raw[3] << 32) | raw[2];
        if ((v ^ chk) == want) break;
        if (wall_clock64() - t0 > 2000000) {       // 20 ms at 100 MHz: give up — the grid must drain
          dead = true;
          v = 0;
          if (lane == 0) atomicCAS(t.err, 0, 1);
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
    }
    const double other = __builtin_bit_cast(double, v);
    const double ax = s == 0 ? own + other : other + own;          // member order: the same sum on both members (:335)
    const int i = row_begin + (int)seq;
    double yn = 0.0;
    if (i < row_end) {
      const double b = b_in[g];
      const double w = y_in[g] + sigma * ((1.0 + rho) * ax - rho * axp_in[g]);   // :344
      yn = prox_conj_elem_s<double>(d.h, w, sigma, i, 0.0, b);                   // :345
      if (tid == 0 && s == 0) {
        const double dr = (w - yn) / sigma - ax;                                 // :347
        e0 += dr * dr;
        e1 += gval_elem_s<double>(d.h, ax, i, b);
        Y[i] = yn;
        AXP[i] = ax;                                                             // :360
      }
    }
#pragma unroll
    for (int k = 0; k < KV; ++k)
#pragma unroll
      for (int e = 0; e < 2; ++e) acc[k][e] = fma(a[g][k][e], yn, acc[k][e]);
#pragma unroll
    for (int k = 0; k < KV; ++k)
#pragma unroll
      for (int e = 0; e < 2; ++e) asm volatile("" : "+v"(acc[k][e]));
  };
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int g = 0; g < G - 1; ++g) {
    load(g, row_begin + g);
    __builtin_amdgcn_sched_barrier(0);
  }
  double own_prev = dots(0, 0);                    // row 0: nothing to finish yet
  __builtin_amdgcn_sched_barrier(0);
  load(G - 1, row_begin + G - 1);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
  for (long long q = 1; q <= nrows; q += G) {
#pragma unroll
    for (int u = 0; u < G; ++u) {
      const long long seq = q + u;                 // dots of row seq (slot (1 + u) % G) ; finishes row seq - 1 (slot u % G)
      raw_t early;
      {
        const TeamRec* src = ring + ((seq - 1) % kTeamSlots) * 2 + (1 - s);
        asm volatile("s_load_dwordx4 %0, %1, 0x0 glc" : "=s"(early) : "s"(src) : "memory");
      }
      const double own = dots((1 + u) % G, seq);
      __builtin_amdgcn_sched_barrier(0);
      finish(u % G, seq - 1, own_prev, early);
      own_prev = own;
      __builtin_amdgcn_sched_barrier(0);
      load(u % G, row_begin + (int)(seq - 1) + G);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  double* __restrict__ out = d.part + (int64_t)team * d.ldpart;
#pragma unroll
  for (int k = 0; k < KV; ++k) {
    const int lv = tid + k * BS;
    const int j = (v0 + lv) * 2;
#pragma unroll
    for (int e = 0; e < 2; ++e)
      if (lv < nvs && j + e < d.n) out[j + e] = acc[k][e];
  }
  if (tid == 0 && s == 0) {
    out[d.n] = e0;
    out[d.n + 1] = e1;
  }
}

// A'y from the block partials (64 columns per block, sixteen loads in flight, fixed order) ; the dual residual and the
// end of the iteration (:347-354) by block 0
constexpr int kSlabPdCloseCols = 64;
constexpr int kSlabPdCloseBlock = 1024;            // (sixteen threads per column: all of a thread's loads in flight at once — csrc/slab.hip)
template <typename T>
__global__ __launch_bounds__(kSlabPdCloseBlock) void k_slabpd_close(SlabPdDev d, int B, T* __restrict__ aty, apx_record* recs, int64_t rec_cap) {
  constexpr int S = kSlabPdCloseBlock / kSlabPdCloseCols;
  __shared__ double fs[2];
  __shared__ double sub[S][kSlabPdCloseCols];
  if (d.c->done != 0) return;
  const int tid = threadIdx.x;
  const int c = tid % kSlabPdCloseCols, sl = tid / kSlabPdCloseCols;
  const int j = blockIdx.x * kSlabPdCloseCols + c;
  const double part_sum = j < d.n ? skinny_strided_sum(d.part, d.ldpart, j, sl, S, B) : 0.0;
  if (blockIdx.x == 0 && tid < 128) {
    const int which = tid >> 6, lane = tid & 63;
    double s = 0.0;
    for (int b = lane; b < B; b += 64) s += d.part[(int64_t)b * d.ldpart + d.n + which];
    s = wave_sum(s);
    if (lane == 0) fs[which] = s;
  }
  sub[sl][c] = part_sum;
  __syncthreads();
  if (sl == 0 && j < d.n) {
    double g = sub[0][c];
#pragma unroll
    for (int k = 1; k < S; ++k) g += sub[k][c];
    aty[j] = (T)g;                                                                         // :358  At_y = A' * y
  }
  if (blockIdx.x == 0 && tid == 0) {
    const double ndr = sqrt(fs[0]);
    d.c->ndr2 = ndr * ndr;
    d.c->h_Ax = gval_final(d.h, fs[1]);
    finish_iteration(d.c, recs, rec_cap);                                                  // :348-354
  }
}

// ---------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------
struct SlabPdShape {
  int bs, cpt, rows;
};
static SlabPdShape slab_pd_shape(const apx_problem* p) {
  const int64_t cols = p->A_lin->n;
  const bool f32 = p->dtype == APX_F32;
  if (cols <= 2 * 1024) return {1024, 2, 4};
  if (cols <= 3 * 1024) return {1024, 3, 4};       // (three / four columns per thread: csrc/slab.hip slab_shape, measured)
  if (cols <= 4 * 1024) return {1024, 4, 4};
  if (cols <= 5 * 1024) return {1024, 5, 4};
  // (between five columns of 1024 threads and fourteen of 768: seven and ten — a row of 5200 entries kept 37 % of the
  //  fourteen-column shape's lanes busy, 3.6 TB/s ; tools/shape_sweep.py)
  if (cols <= 7 * 768) return {768, 7, 4};
  if (cols <= 10 * 768) return {768, 10, f32 ? 4 : 2};
  if (cols <= 14 * 768) return {768, 14, f32 ? 4 : 2};
  // C5's row length, 20 000 fp64, is out of reach of ONE workgroup (x, the accumulator and one row are 480 KB of a CU's
  // 512 KB of VGPRs — 1024 threads x 20 columns x 1 row spilled 432 B and ran at 3.4 ms against 2.36 ms for the two
  // passes): teams of two workgroups, half a row each (k_teampd_rows) — Float64, on a part of 8 XCDs x 32 CUs
  if (!f32 && cols <= 2 * 14 * 768 && p->ctx->num_cu == 256 && !p->teams_off) return {768, 28, 1};
  return {0, 0, 0};
}

static bool slab_pd_shape_ok(const apx_problem* p) {
  if (!p->ctx->fused_small) return false;
  if (p->A_lin == nullptr || p->sharded()) return false;
  if (p->A_lin->sparse || p->ctx->skip_zero_cols) return false;
  if (p->hd.kind == APX_G_L2) return false;        // (its conjugate prox needs a norm over all rows first)
  if ((double)p->A_lin->m * (double)p->A_lin->n * (double)p->es() < (double)p->ctx->one_pass_min_bytes) return false;   // (see slab.hip)
  return p->A_lin->m >= 4096 && p->A_lin->m < ((int64_t)1 << 30) && slab_pd_shape(p).bs != 0;
}

// (reads only the transposed copy, which a matrix of up to 16 GB is given here: see slab_eligible)
bool slab_pd_eligible(const apx_problem* p) {
  if (!slab_pd_shape_ok(p)) return false;
  const apx_matrix* A = p->A_lin;
  if (A->dT == nullptr && p->ctx->auto_transpose && (double)A->ld * (double)A->n_pad * (double)p->es() <= 16.0 * 1024 * 1024 * 1024) {
    if (apx_matrix_build_transpose(const_cast<apx_matrix*>(A)) != APX_OK) (void)hipGetLastError();
  }
  return A->dT != nullptr;
}

int32_t slab_pd_setup(apx_solver* s) {
  apx_problem* p = s->p;
  apx_ctx* ctx = p->ctx;
  const apx_matrix* A = p->A_lin;
  std::unique_ptr<SlabPdState> st(new SlabPdState());
  const SlabPdShape sh = slab_pd_shape(p);
  st->cpt = sh.cpt;
  st->bs = sh.bs;
  st->rows = sh.rows;
  int64_t B = (int64_t)ctx->num_cu * (st->cpt == 2 ? 2 : 1);
  int64_t rpb = (A->m + B - 1) / B;                // dealt evenly, nothing rounded up: a block's last step may be partial
  B = (A->m + rpb - 1) / rpb;                      // (csrc/slab.hip: rounding to 16 left CUs without a block)
  if (st->cpt == 28) {                             // teams of two: 128 slabs, ALL 256 workgroups launched (16 teams per XCD)
    B = ctx->num_cu / 2;
    rpb = (A->m + B - 1) / B;
    const int64_t nvs = ((A->n + 1) / 2 + 1) / 2;  // 16-byte vectors per member
    st->team_kv = (int)((nvs + 767) / 768);
    const size_t ring_bytes = (size_t)B * kTeamSlots * 2 * sizeof(TeamRec);
    APX_TRY(st->team_buf.alloc_zero(ring_bytes + 64, ctx->stream));
    st->team.ring = (TeamRec*)st->team_buf.p;
    st->team.tickets = (unsigned*)((char*)st->team_buf.p + ring_bytes);
    st->team.err = (int*)((char*)st->team_buf.p + ring_bytes + 32);
    st->team.seq0 = 0;
    st->team.launch = 0;
    st->team.rows_per_team = (int32_t)rpb;
    APX_HIP(hipHostMalloc((void**)&st->h_err, sizeof(int32_t), hipHostMallocDefault));
    *st->h_err = 0;
  }
  st->B = (int)B;
  SlabPdDev d{};
  d.AT = A->dT;
  d.ldT = A->ldT;
  d.m = (int32_t)A->m;
  d.n = (int32_t)p->n;
  d.rows_per_block = (int32_t)rpb;
  d.Y = s->y.p;
  d.AXP = s->axprev.p;
  d.ldpart = round_up(d.n + 2, 8);
  APX_TRY(st->part.alloc_zero((size_t)B * d.ldpart * sizeof(double), ctx->stream));
  d.part = (double*)st->part.p;
  d.c = p->ctrl();
  d.h = p->hd;
  st->nt = (int64_t)A->m * A->n * (int64_t)dtype_size(p->dtype) >= ((int64_t)384 << 20);     // (csrc/slab.hip: measured)
  st->dev = d;
  s->slab_pd = st.release();
  return APX_OK;
}

// the team kernel for this row length (KV 16-byte vectors per thread and half row ; two rows in registers: with three the
// KV = 7 shape spills, and tools/split_probe.hip measured no gain from the third — x in LDS to make room: 6.25 vs 6.27 TB/s)
template <bool NT>
static bool team_rows_launch(const SlabPdState* st, const SlabPdDev& d, const void* x, hipStream_t q) {
  const dim3 grid((unsigned)(2 * st->B));
  switch (st->team_kv) {
    case 4: hipLaunchKernelGGL((k_teampd_rows<4, 3, NT, false>), grid, dim3(768), 0, q, d, st->team, x); return true;
    case 5: hipLaunchKernelGGL((k_teampd_rows<5, 3, NT, true>), grid, dim3(768), (size_t)5 * 768 * 16, q, d, st->team, x); return true;   // (x in LDS: 60 KB)
    case 6: hipLaunchKernelGGL((k_teampd_rows<6, 2, NT, false>), grid, dim3(768), 0, q, d, st->team, x); return true;
    case 7: {                                      // (x of the half row in LDS: in registers the row state spills, 20 registers)
      constexpr int lds = 7 * 768 * 2 * (int)sizeof(double);
      // (the attribute belongs to the function ON A DEVICE: once per device of the process — the single-process group entry
      //  point drives one context per device, each from its own host thread)
      static std::atomic<unsigned long long> attr_devs{0};
      int dev = 0;
      (void)hipGetDevice(&dev);
      const unsigned long long bit = 1ull << (dev & 63);
      if ((attr_devs.load(std::memory_order_relaxed) & bit) == 0) {
        (void)hipFuncSetAttribute((const void*)k_teampd_rows<7, 2, NT, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_devs.fetch_or(bit, std::memory_order_relaxed);
      }
      hipLaunchKernelGGL((k_teampd_rows<7, 2, NT, true>), grid, dim3(768), lds, q, d, st->team, x);
      return true;
    }
    default: return false;
  }
}

template <typename T>
static void slab_pd_launch_t(SlabPdState* st, const SlabPdDev& d, const void* x, void* aty, apx_record* recs, int64_t rec_cap,
                             hipStream_t q) {
  const dim3 grid((unsigned)st->B);
  constexpr bool f32 = sizeof(T) == 4;
  if (st->team_kv != 0) {
    if constexpr (!f32) {
      if (st->nt) team_rows_launch<true>(st, d, x, q);
      else team_rows_launch<false>(st, d, x, q);
      st->team.launch += 1;                        // (tickets and sequence numbers are never reset: each launch has its own range)
      st->team.seq0 += (unsigned long long)st->team.rows_per_team + 16;
      hipLaunchKernelGGL((k_slabpd_close<T>), dim3((unsigned)((d.n + kSlabPdCloseCols - 1) / kSlabPdCloseCols)), dim3(kSlabPdCloseBlock), 0, q,
                         d, st->B, (T*)aty, recs, rec_cap);
    }
    return;
  }
  // two row groups in flight only where it measured faster here — Float32 at two columns per thread (400 000 x 2000: 0.572 ->
  // 0.552 ms) ; with the per-row state of y and A_x_prev on top of the rows, the other shapes lose (200 000 x 5000 fp64
  // 1.31 -> 1.38, 100 000 x 10 000 fp32 0.68 -> 0.71)
  // (from ~16 steps per block on, as in csrc/slab.hip ; non-temporal loads for a matrix beyond the Infinity Cache: st->nt)
  const bool pipe = f32 && d.rows_per_block / 16 >= 16;
  if (st->nt) {
    if (st->cpt == 2 && pipe) hipLaunchKernelGGL((k_slabpd_rows<T, 2, 16, 1024, f32, true>), grid, dim3(1024), 0, q, d, x);
    else if (st->cpt == 2) hipLaunchKernelGGL((k_slabpd_rows<T, 2, 16, 1024, false, true>), grid, dim3(1024), 0, q, d, x);
    else if (st->cpt == 3) hipLaunchKernelGGL((k_slabpd_rows<T, 3, 4, 1024, false, true>), grid, dim3(1024), 0, q, d, x);
    else if (st->cpt == 4) hipLaunchKernelGGL((k_slabpd_rows<T, 4, 4, 1024, false, true>), grid, dim3(1024), 0, q, d, x);
    else if (st->cpt == 5) hipLaunchKernelGGL((k_slabpd_rows<T, 5, 4, 1024, false, true>), grid, dim3(1024), 0, q, d, x);
    else if (st->cpt == 7) hipLaunchKernelGGL((k_slabpd_rows<T, 7, 4, 768, false, true>), grid, dim3(768), 0, q, d, x);
    else if (st->cpt == 10) hipLaunchKernelGGL((k_slabpd_rows<T, 10, (f32 ? 4 : 2), 768, false, true>), grid, dim3(768), 0, q, d, x);
    else hipLaunchKernelGGL((k_slabpd_rows<T, 14, (f32 ? 4 : 2), 768, false, true>), grid, dim3(768), 0, q, d, x);
  } else {
    if (st->cpt == 2 && pipe) hipLaunchKernelGGL((k_slabpd_rows<T, 2, 16, 1024, f32, false>), grid, dim3(1024), 0, q, d, x);
    else if (st->cpt == 2) hipLaunchKernelGGL((k_slabpd_rows<T, 2, 16, 1024, false, false>), grid, dim3(1024), 0, q, d, x);
    else if (st->cpt == 3) hipLaunchKernelGGL((k_slabpd_rows<T, 3, 4, 1024, false, false>), grid, dim3(1024), 0, q, d, x);
    else if (st->cpt == 4) hipLaunchKernelGGL((k_slabpd_rows<T, 4, 4, 1024, false, false>), grid, dim3(1024), 0, q, d, x);
    else if (st->cpt == 5) hipLaunchKernelGGL((k_slabpd_rows<T, 5, 4, 1024, false, false>), grid, dim3(1024), 0, q, d, x);
    else if (st->cpt == 7) hipLaunchKernelGGL((k_slabpd_rows<T, 7, 4, 768, false, false>), grid, dim3(768), 0, q, d, x);
    else if (st->cpt == 10) hipLaunchKernelGGL((k_slabpd_rows<T, 10, (f32 ? 4 : 2), 768, false, false>), grid, dim3(768), 0, q, d, x);
    else hipLaunchKernelGGL((k_slabpd_rows<T, 14, (f32 ? 4 : 2), 768, false, false>), grid, dim3(768), 0, q, d, x);
  }
  hipLaunchKernelGGL((k_slabpd_close<T>), dim3((unsigned)((d.n + kSlabPdCloseCols - 1) / kSlabPdCloseCols)), dim3(kSlabPdCloseBlock), 0, q, d,
                     st->B, (T*)aty, recs, rec_cap);
}

// what gemv_n(A, x) + the dual step + gemv_t(A, y) leave behind: y, A_x_prev updated, At_y in s->aty, the iteration closed.
// false in *done_here: the transposed copy is gone (A edited in place) — the caller runs the generic kernels
int32_t slab_pd_iteration(apx_solver* s, bool* done_here) {
  apx_problem* p = s->p;
  apx_ctx* ctx = p->ctx;
  SlabPdState* st = s->slab_pd;
  *done_here = false;
  if (p->A_lin->dT == nullptr) return APX_OK;
  SlabPdDev d = st->dev;
  d.AT = p->A_lin->dT;
  d.ldT = p->A_lin->ldT;
  APX_TRY(prof_begin(ctx, APX_KSTAT_GEMV_N, (double)p->A_lin->m * (double)p->A_lin->n * (double)p->es()));
  if (p->dtype == APX_F32) slab_pd_launch_t<float>(st, d, s->x.p, s->aty.p, (apx_record*)s->recs.p, s->rec_cap, ctx->stream);
  else slab_pd_launch_t<double>(st, d, s->x.p, s->aty.p, (apx_record*)s->recs.p, s->rec_cap, ctx->stream);
  APX_TRY(prof_end(ctx));
  APX_HIP(hipGetLastError());
  *done_here = true;
  return APX_OK;
}

bool slab_pd_uses_teams(const apx_solver* s) { return s->slab_pd != nullptr && s->slab_pd->team_kv != 0; }

// the error word of the team hand-off (0: fine), read with the convergence flag and with the result
int32_t slab_pd_check(apx_solver* s) {
  SlabPdState* st = s->slab_pd;
  if (st == nullptr || st->team_kv == 0) return APX_OK;
  apx_ctx* ctx = s->p->ctx;
  APX_HIP(hipMemcpyAsync(st->h_err, st->team.err, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
  APX_HIP(hipStreamSynchronize(ctx->stream));
  if (*st->h_err == 0) return APX_OK;
  s->p->team_failed = true;
  return fail(APX_ERR_HIP, *st->h_err == 2
                               ? "one-pass primal-dual kernel: an XCD received more than 32 workgroups of a launch (the GPU is "
                                 "shared or partitioned); the result is invalid — switch the policy off (apx_ctx_fused_small(ctx, 0))"
                               : "one-pass primal-dual kernel: a workgroup waited 20 ms for its team partner (the GPU is shared with "
                                 "other work, so the 256 workgroups were not resident together); the result is invalid — "
                                 "switch the policy off (apx_ctx_fused_small(ctx, 0))");
}

}  // namespace apx
