// split_probe.hip — round-2 probe: ONE pass over A per iteration for rows too long for one workgroup's registers
// (BASELINE config C5: 50 000 x 20 000 fp64, a row is 160 KB; csrc/slab.hip / slab_pd.hip stop at 10 752 columns).
//
// A TEAM of P workgroups (P = 2 or 4, one per CU, members 8 block indices apart = the same XCD under round-robin
// dispatch) shares a slab of rows; member s keeps columns [s n/P, (s+1) n/P) of x and of the column accumulator in
// registers and streams that slice of every row.  Per group of H rows a member forms its partial dot products, PUBLISHES
// them, and one step LATER collects the other members' partials of that group — by then they have been on their way for
// a whole step, so the hand-off latency (1 - 3 us, MI355X_MICROARCH.md) is off the critical path.
//
// What the hand-off must NOT do is touch the vector-memory counter: loads return in order per wave, so a release fence, or
// a polling vector load, waits for every load of A the wave has in flight — the pipeline drains at every step (that is
// what round 1's probe measured as 23 - 28 us per hand-off).  So: a record is ONE 16-byte store {value, value ^ sequence}
// by one lane, fire and forget (no fence; the check word validates the pair), and the consumer polls with SCALAR loads
// (`s_load_dwordx4 ... glc`: the scalar path has its own counter and bypasses the scalar cache), which see the L2 the team
// shares — teams are formed INSIDE an XCD at run time (XCC_ID + a ticket per XCD), whatever the dispatch order.  All members add the P partials in member
// order, so every member sees the same `A x`, takes the same decision and accumulates its column slice: no atomics on
// data, bit-reproducible.  Every wait is a bounded spin (wall clock) with an error exit: the grid always drains.
//
// Synthetic least-squares data, results checked against the host.  Build:
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/split_probe.hip -o tools/split_probe
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(e)                                                 \
  do {                                                        \
    hipError_t _e = (e);                                      \
    if (_e != hipSuccess) {                                   \
      fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(_e)); \
      exit(1);                                                \
    }                                                         \
  } while (0)

typedef double v2d __attribute__((ext_vector_type(2)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

__host__ __device__ inline double gen(uint64_t seed, uint64_t idx) {
  uint64_t z = seed * 0x9E3779B97F4A7C15ULL + idx;
  z += 0x9E3779B97F4A7C15ULL;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  z = z ^ (z >> 31);
  return -1.0 + 2.0 * ((double)(z >> 11) * (1.0 / 9007199254740992.0));
}

__global__ void k_fill(double* AT, int64_t ldT, int m, int n) {
  const int64_t i = blockIdx.y;
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x)
    AT[i * ldT + j] = gen(1, (uint64_t)i * (uint64_t)n + (uint64_t)j);
}

constexpr int kSlots = 8;          // ring of published groups per member (>= 4: see the header of k_split)
struct Rec {                       // one partial dot: the value and value ^ sequence, written by ONE 16-byte store
  unsigned long long v, chk;
};
struct Xch {                       // one member's publication of one group: H records
  Rec r[4];
};

struct Args {
  const double* AT;
  int64_t ldT;
  int m, n, rows_per_team, nteams;
  const double* b;
  const double* x;
  double* part;      // [teams][ldpart]: n column partials, then the sum of squares
  int64_t ldpart;
  Xch* xch;          // [teams][kSlots][P]
  int* err;          // set when a wait timed out
  unsigned long long seq0;   // sequence base of this launch (the ring is never reset)
  unsigned* tickets;         // [8] one counter per XCD, monotonic
  unsigned launch;           // launches so far: ticket base = launch * 32
};

// P members x KV 16-byte vectors (2 doubles) per thread, groups of H rows, G groups in registers (one awaiting its
// partners, one being accumulated / re-loaded, G - 2 in flight).
template <int P, int KV, int H, int G, bool NT, bool XLDS = false, bool EARLY = false>
__global__ __launch_bounds__(768) void k_split(Args d) {
  extern __shared__ double xs[];                          // XLDS: this member's slice of x (KV * 768 * 2 doubles)
  constexpr int BS = 768, NW = BS / 64;
  __shared__ double wsum[NW][H];
  __shared__ double tot_s[P][H];
  __shared__ double own_s[2][H];                           // EARLY: this member's own partials, by group parity
  __shared__ double bs[G][H];
  __shared__ int bail;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);     // (provably wave-uniform: scalar branches, SGPR values)
  // team / member of this block: formed inside an XCD (the members share its L2) from a ticket per XCD
  __shared__ int ident[2];
  if (tid == 0) {
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u;      // HW_REG_XCC_ID, bits 3:0
    const unsigned t = atomicAdd(d.tickets + xcc, 1u) - d.launch * 32u;
    ident[0] = t < 32u ? (int)(xcc * (32 / P) + t / P) : -1;
    ident[1] = (int)(t % P);
    if (t >= 32u) atomicExch(d.err, 2);                    // an XCD with more than 32 blocks of this launch: not this scheme
  }
  __syncthreads();
  const int team = __builtin_amdgcn_readfirstlane(ident[0]), s = __builtin_amdgcn_readfirstlane(ident[1]);
  if (team < 0) return;
  const int nvec = (d.n + 1) / 2;
  const int nvs = (nvec + P - 1) / P;                      // vectors per member
  const int v0 = s * nvs;                                  // this member's first vector
  v2d xr[KV];
  double acc[KV][2];
#pragma unroll
  for (int k = 0; k < KV; ++k) {
    const int lv = tid + k * BS;                           // vector within the member's slice
    const int j = (v0 + lv) * 2;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      xr[k][e] = (lv < nvs && j + e < d.n) ? d.x[j + e] : 0.0;
      acc[k][e] = 0.0;
      if constexpr (XLDS) xs[lv * 2 + e] = xr[k][e];
    }
  }
  if (tid == 0) bail = 0;
  const int voff = tid * 16;
  const int64_t cbase = (int64_t)v0 * 16;
  const int slice_bytes = (int)min((int64_t)nvs * 16, d.ldT * 8 - cbase);
  const int row_begin = team * d.rows_per_team;
  const int row_end = min(row_begin + d.rows_per_team, d.m);
  double e0 = 0.0;
  v2d a[G][H][KV];
  double b_in[G];
  auto load = [&](int g, int r0) {
#pragma unroll
    for (int r = 0; r < H; ++r) {
      const int ri = __builtin_amdgcn_readfirstlane(min(r0 + r, row_end - 1));
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
          (void*)((const char*)(d.AT + (int64_t)ri * d.ldT) + cbase), 0, slice_bytes, 0x00020000);
#pragma unroll
      for (int k = 0; k < KV; ++k) {
        const v4u raw = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, k * BS * 16, NT ? 2 : 0);
        a[g][r][k] = __builtin_bit_cast(v2d, raw);
      }
    }
    b_in[g] = d.b[min(r0 + min(tid, H - 1), row_end - 1)];
  };
  Xch* ring = d.xch + (int64_t)team * kSlots * P;
  // partial dots of group `grp` (rows r0 ...) held in slot g: block sum, then thread 0 publishes
  auto dots = [&](int g, int r0, long long grp) {
    double p[H];
#pragma unroll
    for (int r = 0; r < H; ++r) p[r] = 0.0;
#pragma unroll
    for (int k = 0; k < KV; ++k)
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int r = 0; r < H; ++r)
          p[r] = fma(a[g][r][k][e], XLDS ? xs[(tid + k * BS) * 2 + e] : xr[k][e], p[r]);
#pragma unroll
    for (int r = 0; r < H; ++r) {
      p[r] = wave_sum(p[r]);
      if (lane == 0) wsum[wv][r] = p[r];
    }
    if (tid < H) bs[g][tid] = b_in[g];
    __syncthreads();
    if (tid == 0) {
      Xch* mine = ring + (grp % kSlots) * P + s;
      const unsigned long long seq = d.seq0 + (unsigned long long)grp + 1;
#pragma unroll
      for (int r = 0; r < H; ++r) {
        double t = wsum[0][r];
#pragma unroll
        for (int w = 1; w < NW; ++w) t += wsum[w][r];
        if constexpr (EARLY) own_s[grp & 1][r] = t;
        const unsigned long long bits = __builtin_bit_cast(unsigned long long, t);
        typedef unsigned long long v2ul __attribute__((ext_vector_type(2)));
        v2ul rec = {bits, bits ^ seq};
        *(volatile v2ul*)&mine->r[r] = rec;                            // one 16-byte store, no fence, nothing waits for it
      }
    }
  };
  // collects the P partials of group `grp` (own included: read back like the others, one code path), forms the
  // residuals, accumulates slot g
  typedef unsigned v4s __attribute__((ext_vector_type(4)));
  v4s early[H];                                            // EARLY (P = 2): the partner's records, requested a phase ahead
  auto request = [&](long long grp) {                      // scalar loads, not waited for here
    if constexpr (EARLY) {
      if (wv == 0) {
#pragma unroll
        for (int r = 0; r < H; ++r) {
          const Rec* src = &(ring + (grp % kSlots) * P + (1 - s))->r[r];
          asm volatile("s_load_dwordx4 %0, %1, 0x0 glc" : "=s"(early[r]) : "s"(src) : "memory");
        }
      }
    }
  };
  auto finish = [&](int g, int r0, long long grp) {
    if (wv == 0) {                                         // wave 0 polls on the SCALAR path (wave-uniform code)
      const unsigned long long want = d.seq0 + (unsigned long long)grp + 1;
      const long long t0 = wall_clock64();
      bool ok = true;
#pragma unroll
      for (int q = 0; q < P; ++q) {
#pragma unroll
        for (int r = 0; r < H; ++r) {
          if (EARLY && q == s) {                           // (own partial: kept, not read back)
            if (lane == 0) tot_s[q][r] = own_s[grp & 1][r];
            continue;
          }
          const Rec* src = &(ring + (grp % kSlots) * P + q)->r[r];
          unsigned long long v = 0, chk = 0;
          bool first = EARLY;
          while (ok) {
            v4s raw;
            if (first) {                                   // the answer requested at the top of the step
              asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(early[r])::"memory");
              raw = early[r];
              first = false;
            } else {
              asm volatile("s_load_dwordx4 %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(raw) : "s"(src) : "memory");
            }
            v = ((unsigned long long)raw[1] << 32) | raw[0];
            chk = ((unsigned long long)raw[3] << 32) | raw[2];
            if ((v ^ chk) == want) break;
            if (wall_clock64() - t0 > 2000000) ok = false;  // 20 ms at 100 MHz: give up, the grid must drain
            __builtin_amdgcn_s_sleep(1);
          }
          if (lane == 0) tot_s[q][r] = ok ? __builtin_bit_cast(double, v) : 0.0;
        }
      }
      if (!ok && lane == 0) {
        bail = 1;
        atomicExch(d.err, 1);
      }
    }
    __syncthreads();
    double rr[H];
#pragma unroll
    for (int r = 0; r < H; ++r) {
      double dot = tot_s[0][r];
#pragma unroll
      for (int q = 1; q < P; ++q) dot += tot_s[q][r];      // member order: the same sum on every member
      const int i = r0 + r;
      rr[r] = i < row_end ? dot - bs[g][r] : 0.0;
      if (tid == 0 && s == 0) e0 += rr[r] * rr[r];
    }
#pragma unroll
    for (int r = 0; r < H; ++r)
#pragma unroll
      for (int k = 0; k < KV; ++k)
#pragma unroll
        for (int e = 0; e < 2; ++e) acc[k][e] = fma(a[g][r][k][e], rr[r], acc[k][e]);
#pragma unroll
    for (int k = 0; k < KV; ++k)
#pragma unroll
      for (int e = 0; e < 2; ++e) asm volatile("" : "+v"(acc[k][e]));
  };
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int g = 0; g < G - 1; ++g) {                        // slot G - 1 is filled inside the loop
    load(g, row_begin + g * H);
    __builtin_amdgcn_sched_barrier(0);
  }
  // step q: dots of group q (slot q % G) ; finish group q - 1 (slot (q - 1) % G) ; re-load that slot with group q + G - 1.
  // The loop body is unrolled G times so that every slot index is a constant.
  const int ngroups = (row_end - row_begin + H - 1) / H;
  long long q = 0;
#pragma unroll 1
  for (; q < ngroups + 1 && bail == 0; q += G) {
#pragma unroll
    for (int u = 0; u < G; ++u) {
      const long long grp = q + u;
      constexpr int dummy = 0;
      (void)dummy;
      if (grp >= 1 && grp - 1 < ngroups) request(grp - 1);
      if (grp < ngroups) dots(u, row_begin + (int)grp * H, grp);
      __builtin_amdgcn_sched_barrier(0);
      if (grp >= 1 && grp - 1 < ngroups) finish((u + G - 1) % G, row_begin + (int)(grp - 1) * H, grp - 1);
      __builtin_amdgcn_sched_barrier(0);
      load((u + G - 1) % G, row_begin + (int)(grp + G - 1) * H);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  double* out = d.part + (int64_t)team * d.ldpart;
#pragma unroll
  for (int k = 0; k < KV; ++k) {
    const int lv = tid + k * BS;
    const int j = (v0 + lv) * 2;
#pragma unroll
    for (int e = 0; e < 2; ++e)
      if (lv < nvs && j + e < d.n) out[j + e] = acc[k][e];
  }
  if (tid == 0 && s == 0) out[d.n] = e0;
}

// ---------------------------------------------------------------------------------------------------------------------
// Second form (P = 2, one row per group): ONE barrier per step and no broadcast through LDS — EVERY wave requests the
// partner's record itself with a scalar load at the top of the step (before the dots of the next group, which hide its
// latency), sums the block's partial dots itself after the barrier, and validates its own copy of the record.
// The loop body is branch-free between the request and its use: groups past the slab are dummy groups (clamped rows,
// residual 0) that both members run alike.
// ---------------------------------------------------------------------------------------------------------------------
template <int KV, int G, bool NT, bool XLDS = false>
__global__ __launch_bounds__(768) void k_split2(Args d) {
  extern __shared__ double xs2[];
  constexpr int BS = 768, NW = BS / 64, P = 2;
  __shared__ double wsum[2][NW];
  __shared__ int ident[2];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (tid == 0) {
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u;      // HW_REG_XCC_ID
    const unsigned t = atomicAdd(d.tickets + xcc, 1u) - d.launch * 32u;
    ident[0] = t < 32u ? (int)(xcc * (32 / P) + t / P) : -1;
    ident[1] = (int)(t % P);
    if (t >= 32u) atomicExch(d.err, 2);
  }
  __syncthreads();
  const int team = __builtin_amdgcn_readfirstlane(ident[0]), s = __builtin_amdgcn_readfirstlane(ident[1]);
  if (team < 0) return;
  const int nvec = (d.n + 1) / 2;
  const int nvs = (nvec + P - 1) / P;
  const int v0 = s * nvs;
  v2d xr[KV];
  double acc[KV][2];
#pragma unroll
  for (int k = 0; k < KV; ++k) {
    const int lv = tid + k * BS;
    const int j = (v0 + lv) * 2;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      xr[k][e] = (lv < nvs && j + e < d.n) ? d.x[j + e] : 0.0;
      acc[k][e] = 0.0;
      if constexpr (XLDS) xs2[lv * 2 + e] = xr[k][e];
    }
  }
  const int voff = tid * 16;
  const int64_t cbase = (int64_t)v0 * 16;
  const int slice_bytes = (int)min((int64_t)nvs * 16, d.ldT * 8 - cbase);
  const int row_begin = team * d.rows_per_team;
  const int row_end = min(row_begin + d.rows_per_team, d.m);
  const int ngroups = row_end - row_begin;
  double e0 = 0.0;
  v2d a[G][KV];
  double b_in[G];
  auto load = [&](int g, int row) {
    const int ri = __builtin_amdgcn_readfirstlane(min(row, row_end - 1));
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((const char*)(d.AT + (int64_t)ri * d.ldT) + cbase), 0, slice_bytes, 0x00020000);
#pragma unroll
    for (int k = 0; k < KV; ++k) {
      const v4u raw = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, k * BS * 16, NT ? 2 : 0);
      a[g][k] = __builtin_bit_cast(v2d, raw);
    }
    b_in[g] = d.b[ri];                                     // (every thread: the row's entry rides behind its row)
  };
  Xch* ring = d.xch + (int64_t)team * kSlots * P;
  typedef unsigned v4s __attribute__((ext_vector_type(4)));
  typedef unsigned long long v2ul __attribute__((ext_vector_type(2)));
  // the block's partial dot of group grp (slot g): every thread returns it ; thread 0 publishes it
  auto dots = [&](int g, long long grp) -> double {
    double p = 0.0;
#pragma unroll
    for (int k = 0; k < KV; ++k)
#pragma unroll
      for (int e = 0; e < 2; ++e) p = fma(a[g][k][e], XLDS ? xs2[(tid + k * BS) * 2 + e] : xr[k][e], p);
    p = wave_sum(p);
    if (lane == 0) wsum[grp & 1][wv] = p;
    __syncthreads();
    double own = wsum[grp & 1][0];
#pragma unroll
    for (int w = 1; w < NW; ++w) own += wsum[grp & 1][w];
    if (tid == 0) {
      const unsigned long long bits = __builtin_bit_cast(unsigned long long, own);
      v2ul rec = {bits, bits ^ (d.seq0 + (unsigned long long)grp + 1)};
      *(volatile v2ul*)&(ring + (grp % kSlots) * P + s)->r[0] = rec;
    }
    return own;
  };
  bool dead = false;
  auto finish = [&](int g, long long grp, double own, v4s early) {
    const unsigned long long want = d.seq0 + (unsigned long long)grp + 1;
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(early)::"memory");
    unsigned long long v = ((unsigned long long)early[1] << 32) | early[0];
    unsigned long long chk = ((unsigned long long)early[3] << 32) | early[2];
    if ((v ^ chk) != want && !dead) {                      // not there yet (or torn): poll, bounded
      const Rec* src = &(ring + (grp % kSlots) * P + (1 - s))->r[0];
      const long long t0 = wall_clock64();
      for (;;) {
        v4s raw;
        asm volatile("s_load_dwordx4 %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(raw) : "s"(src) : "memory");
        v = ((unsigned long long)raw[1] << 32) | raw[0];
        chk = ((unsigned long long)raw[3] << 32) | raw[2];
        if ((v ^ chk) == want) break;
        if (wall_clock64() - t0 > 2000000) {               // 20 ms: give up, the grid must drain
          dead = true;
          v = 0;
          if (tid == 0) atomicExch(d.err, 1);
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
    }
    const double other = __builtin_bit_cast(double, v);
    const double dot = s == 0 ? own + other : other + own;  // member order: the same sum on both members
    const int i = row_begin + (int)grp;
    const double rr = i < row_end ? dot - b_in[g] : 0.0;
    if (tid == 0 && s == 0) e0 += rr * rr;
#pragma unroll
    for (int k = 0; k < KV; ++k)
#pragma unroll
      for (int e = 0; e < 2; ++e) acc[k][e] = fma(a[g][k][e], rr, acc[k][e]);
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
  double own_prev = dots(0, 0);                            // step 0: nothing to finish yet
  __builtin_amdgcn_sched_barrier(0);
  load(G - 1, row_begin + G - 1);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
  for (long long q = 1; q <= ngroups; q += G) {
#pragma unroll
    for (int u = 0; u < G; ++u) {
      const long long grp = q + u;                         // slot (1 + u) % G ; finishes group grp - 1 in slot u % G
      v4s early;
      {
        const Rec* src = &(ring + ((grp - 1) % kSlots) * P + (1 - s))->r[0];
        asm volatile("s_load_dwordx4 %0, %1, 0x0 glc" : "=s"(early) : "s"(src) : "memory");
      }
      const double own = dots((1 + u) % G, grp);
      __builtin_amdgcn_sched_barrier(0);
      finish(u % G, grp - 1, own_prev, early);
      own_prev = own;
      __builtin_amdgcn_sched_barrier(0);
      load(u % G, row_begin + (int)(grp - 1) + G);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  double* out = d.part + (int64_t)team * d.ldpart;
#pragma unroll
  for (int k = 0; k < KV; ++k) {
    const int lv = tid + k * BS;
    const int j = (v0 + lv) * 2;
#pragma unroll
    for (int e = 0; e < 2; ++e)
      if (lv < nvs && j + e < d.n) out[j + e] = acc[k][e];
  }
  if (tid == 0 && s == 0) out[d.n] = e0;
}
template <int P, int KV, int G, bool NT, bool XLDS = false>
__global__ __launch_bounds__(768) void k_split3(Args d) {
  extern __shared__ double xs2[];
  constexpr int BS = 768, NW = BS / 64, TPX = 32 / P;       // teams per XCD (32 CUs each)
  __shared__ double wsum[2][NW];
  __shared__ int ident[2];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (tid == 0) {
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u;      // HW_REG_XCC_ID
    const unsigned t = atomicAdd(d.tickets + xcc, 1u) - d.launch * 32u;
    ident[0] = t < (unsigned)(TPX * P) ? (int)(xcc * TPX + t / P) : -1;     // (32 % P workgroups per XCD stay idle)
    ident[1] = (int)(t % P);
    if (t >= 32u) atomicExch(d.err, 2);
  }
  __syncthreads();
  const int team = __builtin_amdgcn_readfirstlane(ident[0]), s = __builtin_amdgcn_readfirstlane(ident[1]);
  if (team < 0) return;
  const int nvec = (d.n + 1) / 2;
  const int nvs = (nvec + P - 1) / P;
  const int v0 = s * nvs;
  v2d xr[KV];
  double acc[KV][2];
#pragma unroll
  for (int k = 0; k < KV; ++k) {
    const int lv = tid + k * BS;
    const int j = (v0 + lv) * 2;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      xr[k][e] = (lv < nvs && j + e < d.n) ? d.x[j + e] : 0.0;
      acc[k][e] = 0.0;
      if constexpr (XLDS) xs2[lv * 2 + e] = xr[k][e];
    }
  }
  const int voff = tid * 16;
  const int64_t cbase = (int64_t)v0 * 16;
  const int slice_bytes = (int)min((int64_t)nvs * 16, d.ldT * 8 - cbase);
  const int row_begin = team * d.rows_per_team;
  const int row_end = min(row_begin + d.rows_per_team, d.m);
  const int ngroups = row_end - row_begin;
  double e0 = 0.0;
  v2d a[G][KV];
  double b_in[G];
  auto load = [&](int g, int row) {
    const int ri = __builtin_amdgcn_readfirstlane(min(row, row_end - 1));
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((const char*)(d.AT + (int64_t)ri * d.ldT) + cbase), 0, slice_bytes, 0x00020000);
#pragma unroll
    for (int k = 0; k < KV; ++k) {
      const v4u raw = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, k * BS * 16, NT ? 2 : 0);
      a[g][k] = __builtin_bit_cast(v2d, raw);
    }
    b_in[g] = d.b[ri];                                     // (every thread: the row's entry rides behind its row)
  };
  Xch* ring = d.xch + (int64_t)team * kSlots * P;
  typedef unsigned v4s __attribute__((ext_vector_type(4)));
  typedef unsigned long long v2ul __attribute__((ext_vector_type(2)));
  // the block's partial dot of group grp (slot g): every thread returns it ; thread 0 publishes it
  auto dots = [&](int g, long long grp) -> double {
    double p = 0.0;
#pragma unroll
    for (int k = 0; k < KV; ++k)
#pragma unroll
      for (int e = 0; e < 2; ++e) p = fma(a[g][k][e], XLDS ? xs2[(tid + k * BS) * 2 + e] : xr[k][e], p);
    p = wave_sum(p);
    if (lane == 0) wsum[grp & 1][wv] = p;
    __syncthreads();
    double own = wsum[grp & 1][0];
#pragma unroll
    for (int w = 1; w < NW; ++w) own += wsum[grp & 1][w];
    if (tid == 0) {
      const unsigned long long bits = __builtin_bit_cast(unsigned long long, own);
      v2ul rec = {bits, bits ^ (d.seq0 + (unsigned long long)grp + 1)};
      *(volatile v2ul*)&(ring + (grp % kSlots) * P + s)->r[0] = rec;
    }
    return own;
  };
  bool dead = false;
  auto finish = [&](int g, long long grp, double own, v4s (&early)[P]) {
    const unsigned long long want = d.seq0 + (unsigned long long)grp + 1;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    double dot = 0.0;
#pragma unroll
    for (int qm = 0; qm < P; ++qm) {
    asm volatile("" : "+s"(early[qm]));
    unsigned long long v = ((unsigned long long)early[qm][1] << 32) | early[qm][0];
    unsigned long long chk = ((unsigned long long)early[qm][3] << 32) | early[qm][2];
    if (qm != s && (v ^ chk) != want && !dead) {           // not there yet (or torn): poll, bounded
      const Rec* src = &(ring + (grp % kSlots) * P + qm)->r[0];
      const long long t0 = wall_clock64();
      for (;;) {
        v4s raw;
        asm volatile("s_load_dwordx4 %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(raw) : "s"(src) : "memory");
        v = ((unsigned long long)raw[1] << 32) | raw[0];
        chk = ((unsigned long long)raw[3] << 32) | raw[2];
        if ((v ^ chk) == want) break;
        if (wall_clock64() - t0 > 2000000) {               // 20 ms: give up, the grid must drain
          dead = true;
          v = 0;
          if (tid == 0) atomicExch(d.err, 1);
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
    }
    const double term = qm == s ? own : __builtin_bit_cast(double, v);
    dot = qm == 0 ? term : dot + term;                      // member order: the same sum on every member
    }
    const int i = row_begin + (int)grp;
    const double rr = i < row_end ? dot - b_in[g] : 0.0;
    if (tid == 0 && s == 0) e0 += rr * rr;
#pragma unroll
    for (int k = 0; k < KV; ++k)
#pragma unroll
      for (int e = 0; e < 2; ++e) acc[k][e] = fma(a[g][k][e], rr, acc[k][e]);
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
  double own_prev = dots(0, 0);                            // step 0: nothing to finish yet
  __builtin_amdgcn_sched_barrier(0);
  load(G - 1, row_begin + G - 1);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
  for (long long q = 1; q <= ngroups; q += G) {
#pragma unroll
    for (int u = 0; u < G; ++u) {
      const long long grp = q + u;                         // slot (1 + u) % G ; finishes group grp - 1 in slot u % G
      v4s early[P];
#pragma unroll
      for (int qm = 0; qm < P; ++qm) {
        const Rec* src = &(ring + ((grp - 1) % kSlots) * P + qm)->r[0];
        asm volatile("s_load_dwordx4 %0, %1, 0x0 glc" : "=s"(early[qm]) : "s"(src) : "memory");
      }
      const double own = dots((1 + u) % G, grp);
      __builtin_amdgcn_sched_barrier(0);
      finish(u % G, grp - 1, own_prev, early);
      own_prev = own;
      __builtin_amdgcn_sched_barrier(0);
      load(u % G, row_begin + (int)(grp - 1) + G);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  double* out = d.part + (int64_t)team * d.ldpart;
#pragma unroll
  for (int k = 0; k < KV; ++k) {
    const int lv = tid + k * BS;
    const int j = (v0 + lv) * 2;
#pragma unroll
    for (int e = 0; e < 2; ++e)
      if (lv < nvs && j + e < d.n) out[j + e] = acc[k][e];
  }
  if (tid == 0 && s == 0) out[d.n] = e0;
}
template <int P, int KV, int G, bool NT, bool XLDS = false>
static void l_split3(const Args& a, int blocks, hipStream_t st) {
  static bool once = false;
  const size_t lds = XLDS ? (size_t)KV * 768 * 2 * 8 : 0;
  if (!once && lds > 48 * 1024) {
    CK(hipFuncSetAttribute((const void*)k_split3<P, KV, G, NT, XLDS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    once = true;
  }
  hipLaunchKernelGGL((k_split3<P, KV, G, NT, XLDS>), dim3(blocks), dim3(768), lds, st, a);
}
template <int KV, int G, bool NT, bool XLDS = false>
static void l_split2(const Args& a, int blocks, hipStream_t st) {
  static bool once = false;
  const size_t lds = XLDS ? (size_t)KV * 768 * 2 * 8 : 0;
  if (!once && lds > 48 * 1024) {
    CK(hipFuncSetAttribute((const void*)k_split2<KV, G, NT, XLDS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    once = true;
  }
  hipLaunchKernelGGL((k_split2<KV, G, NT, XLDS>), dim3(blocks), dim3(768), lds, st, a);
}

// ---------------------------------------------------------------------------------------------------------------------
// Third form, for WIDE teams (P = 8 / 16: rows of C2's length, 100 000 fp64 = 800 KB): eleven compute waves and ONE
// communication wave per workgroup.  The communication wave loads no A, so its vector-memory counter is its own: it
// publishes the workgroup's partial with an agent-scope store and collects the P records of a row with ONE vector load
// (lane q: member q's record; agent-scope atomics: correct whatever XCD the members sit on, faster inside one), sums
// them in member order and leaves the total in LDS — all of it while the compute waves form the dots of the next row(s).
// One barrier per row.  DEPTH rows between the dots of a row and its use (1: the total of row r is needed one row
// later ; 2: two rows later, more slack for the slowest member, one more row in registers).
// ---------------------------------------------------------------------------------------------------------------------
template <int P, int KV, int G, int DEPTH, bool NT, bool XLDS = false, int MODE = 0>
__global__ __launch_bounds__(768) void k_split4(Args d) {
  extern __shared__ double xs4[];
  constexpr int CW = 11, BSC = CW * 64, TPX = 32 / P;      // compute waves / threads ; teams per XCD
  static_assert(G >= DEPTH + 2, "slots: DEPTH rows awaiting their totals, one computing, one loading");
  __shared__ double wsum[2][CW];
  __shared__ double total_s[4];
  __shared__ int ident[2];
  __shared__ int bail;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (tid == 0) {
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u;
    const unsigned t = atomicAdd(d.tickets + xcc, 1u) - d.launch * 32u;
    ident[0] = t < (unsigned)(TPX * P) ? (int)(xcc * TPX + t / P) : -1;
    ident[1] = (int)(t % P);
    if (t >= 32u) atomicExch(d.err, 2);
    bail = 0;
  }
  __syncthreads();
  const int team = __builtin_amdgcn_readfirstlane(ident[0]), s = __builtin_amdgcn_readfirstlane(ident[1]);
  if (team < 0) return;
  const bool comm = wv == CW;
  const int nvec = (d.n + 1) / 2;
  const int nvs = (nvec + P - 1) / P;
  const int v0 = s * nvs;
  const int row_begin = team * d.rows_per_team;
  const int row_end = min(row_begin + d.rows_per_team, d.m);
  const int nrows = max(row_end - row_begin, 0);
  const int nsteps = nrows + DEPTH;                        // rows past the slab: dummy rows, all members alike
  Rec* ring = (Rec*)d.xch + (int64_t)team * kSlots * P;
  if (comm && MODE >= 3) return;
  if (comm) {
    // ---- the communication wave ----
    for (long long seq = 0; seq < nsteps; ++seq) {
      // before the barrier of step seq: the total of row seq - DEPTH (its partials were published DEPTH steps ago)
      const long long r = seq - DEPTH;
      if (r >= 0) {
        const unsigned long long want = d.seq0 + (unsigned long long)r + 1;
        const Rec* src = ring + (r % kSlots) * P + min(lane, P - 1);
        unsigned long long v = 0;
        const long long t0 = wall_clock64();
        for (;;) {
          v = __hip_atomic_load(&src->v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const unsigned long long chk = __hip_atomic_load(&src->chk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const bool ok = (v ^ chk) == want || lane >= P || (MODE == 2 && lane != s);
          if (__all(ok)) break;
          if (wall_clock64() - t0 > 2000000) {             // 20 ms: give up, the grid must drain
            if (lane == 0) {
              bail = 1;
              atomicExch(d.err, 1);
            }
            v = 0;
            break;
          }
          __builtin_amdgcn_s_sleep(1);
        }
        double tot = __shfl(__builtin_bit_cast(double, v), 0, 64);
#pragma unroll
        for (int q = 1; q < P; ++q) tot += __shfl(__builtin_bit_cast(double, v), q, 64);     // member order
        if (lane == 0) total_s[r & 3] = tot;
      }
      __syncthreads();                                     // the barrier of step seq
      // after it: this workgroup's partial of row seq (every member publishes dummy rows too)
      double own = lane < CW ? wsum[seq & 1][lane] : 0.0;
      double t = __shfl(own, 0, 64);
#pragma unroll
      for (int w = 1; w < CW; ++w) t += __shfl(own, w, 64);
      if (lane == 0) {
        const unsigned long long bits = __builtin_bit_cast(unsigned long long, t);
        Rec* mine = ring + (seq % kSlots) * P + s;
        __hip_atomic_store(&mine->v, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&mine->chk, bits ^ (d.seq0 + (unsigned long long)seq + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    return;
  }
  // ---- the compute waves ----
  v2d xr[KV];
  double acc[KV][2];
#pragma unroll
  for (int k = 0; k < KV; ++k) {
    const int lv = tid + k * BSC;
    const int j = (v0 + lv) * 2;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      xr[k][e] = (lv < nvs && j + e < d.n) ? d.x[j + e] : 0.0;
      acc[k][e] = 0.0;
      if constexpr (XLDS) xs4[lv * 2 + e] = xr[k][e];
    }
  }
  const int voff = tid * 16;
  const int64_t cbase = (int64_t)v0 * 16;
  const int slice_bytes = (int)min((int64_t)nvs * 16, d.ldT * 8 - cbase);
  double e0 = 0.0;
  v2d a[G][KV];
  double b_in[G];
  auto load = [&](int g, int row) {
    const int ri = __builtin_amdgcn_readfirstlane(max(min(row, row_end - 1), 0));
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        MODE == 4 ? (void*)((const char*)d.AT + ((int64_t)s * d.m + ri) * ((int64_t)nvs * 16))
                  : (void*)((const char*)(d.AT + (int64_t)ri * d.ldT) + cbase), 0, slice_bytes, 0x00020000);
#pragma unroll
    for (int k = 0; k < KV; ++k) {
      const v4u raw = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, k * BSC * 16, NT ? 2 : 0);
      a[g][k] = __builtin_bit_cast(v2d, raw);
    }
    b_in[g] = d.b[ri];
  };
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int g = 0; g < G - 1; ++g) {
    load(g, row_begin + g);
    __builtin_amdgcn_sched_barrier(0);
  }
  // step seq: dots of row seq (slot seq % G) ; barrier ; finish row seq - DEPTH (slot (seq - DEPTH) % G) and re-load it
  // with row seq - DEPTH + G.  Unrolled G times: constant slot indices.  (The first DEPTH steps finish nothing, but
  // the slot they would re-load — (seq - DEPTH) mod G — is the one the prologue left empty or has not used yet.)
#pragma unroll 1
  for (long long q = 0; q < nsteps; q += G) {
#pragma unroll
    for (int u = 0; u < G; ++u) {
      const long long seq = q + u;
      if (seq < nsteps) {                                  // (uniform ; the communication wave runs exactly nsteps barriers)
        double p = 0.0;
#pragma unroll
        for (int k = 0; k < KV; ++k)
#pragma unroll
          for (int e = 0; e < 2; ++e) p = MODE == 1 ? p + a[u][k][e] : fma(a[u][k][e], XLDS ? xs4[(tid + k * BSC) * 2 + e] : xr[k][e], p);
        p = wave_sum(p);
        if (lane == 0) wsum[seq & 1][wv] = p;
        if (MODE < 3) __syncthreads();
        constexpr int dummy = 0;
        (void)dummy;
        const long long r = seq - DEPTH;
        const int gs = (u + G - DEPTH) % G;
        if (r >= 0) {
          const double tot = MODE >= 3 ? p : total_s[r & 3];
          const int i = row_begin + (int)r;
          const double rr = i < row_end ? tot - b_in[gs] : 0.0;
          if (tid == 0 && s == 0) e0 += rr * rr;
#pragma unroll
          for (int k = 0; k < KV; ++k)
#pragma unroll
            for (int e = 0; e < 2; ++e) acc[k][e] = MODE == 1 ? acc[k][e] + rr : fma(a[gs][k][e], rr, acc[k][e]);
#pragma unroll
          for (int k = 0; k < KV; ++k)
#pragma unroll
            for (int e = 0; e < 2; ++e) asm volatile("" : "+v"(acc[k][e]));
        }
        __builtin_amdgcn_sched_barrier(0);
        if (r >= 0 || seq == DEPTH - 1 || true) {
          // slot gs is free: row r is done (or, in the first DEPTH steps, gs is the slot the prologue left empty —
          // only for seq == 0 .. DEPTH - 1 with gs == G - 1 ... G - DEPTH)
          const long long next = seq - DEPTH + G;          // the row that belongs in slot gs next
          if (next >= G - 1) load(gs, row_begin + (int)next);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  double* out = d.part + (int64_t)team * d.ldpart;
#pragma unroll
  for (int k = 0; k < KV; ++k) {
    const int lv = tid + k * BSC;
    const int j = (v0 + lv) * 2;
#pragma unroll
    for (int e = 0; e < 2; ++e)
      if (lv < nvs && j + e < d.n) out[j + e] = acc[k][e];
  }
  if (tid == 0 && s == 0) out[d.n] = e0;
}
template <int P, int KV, int G, int DEPTH, bool NT, bool XLDS = false, int MODE = 0>
static void l_split4(const Args& a, int blocks, hipStream_t st) {
  static bool once = false;
  const size_t lds = XLDS ? (size_t)KV * 704 * 2 * 8 : 0;
  if (!once && lds > 48 * 1024) {
    CK(hipFuncSetAttribute((const void*)k_split4<P, KV, G, DEPTH, NT, XLDS, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    once = true;
  }
  hipLaunchKernelGGL((k_split4<P, KV, G, DEPTH, NT, XLDS, MODE>), dim3(blocks), dim3(768), lds, st, a);
}

// ---------------------------------------------------------------------------------------------------------------------
// Fourth form: the wide teams with H = 2 rows per step (twice the bytes per barrier) — three slots of two rows, one step
// of slack ; the column accumulator optionally in LDS (next to x) to make room in the registers.
// ---------------------------------------------------------------------------------------------------------------------
template <int P, int KV, bool NT, bool ACCLDS, bool NOWAIT = false>
__global__ __launch_bounds__(768) void k_split5(Args d) {
  constexpr int CW = 11, BSC = CW * 64, TPX = 32 / P, H = 2, G = 3;
  extern __shared__ double lds5[];                         // x slice [KV * BSC * 2], then (ACCLDS) the accumulator, same shape
  double* xs = lds5;
  double* as = lds5 + KV * BSC * 2;
  __shared__ double wsum[2][H][16];
  __shared__ double res_s[4][H];
  __shared__ int ident[2];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (tid == 0) {
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u;
    const unsigned t = atomicAdd(d.tickets + xcc, 1u) - d.launch * 32u;
    ident[0] = t < (unsigned)(TPX * P) ? (int)(xcc * TPX + t / P) : -1;
    ident[1] = (int)(t % P);
    if (t >= 32u) atomicExch(d.err, 2);
  }
  __syncthreads();
  const int team = __builtin_amdgcn_readfirstlane(ident[0]), s = __builtin_amdgcn_readfirstlane(ident[1]);
  if (team < 0) return;
  const int nvec = (d.n + 1) / 2;
  const int nvs = (nvec + P - 1) / P;
  const int v0 = s * nvs;
  const int row_begin = team * d.rows_per_team;
  const int row_end = min(row_begin + d.rows_per_team, d.m);
  const int nrows = max(row_end - row_begin, 0);
  const int nsteps = 1 + ((nrows + H - 1) / H + G - 1) / G * G;      // one step of prologue, then whole rounds of G steps
  Rec* ring = (Rec*)d.xch + (int64_t)team * kSlots * H * P;
  if (wv == CW) {
    // ---- the communication wave: lane h * 16 + q handles member q's record of row h of the step ----
    const int h = lane >> 4, q = lane & 15;
    unsigned long long pv = 0, pchk = 0;
    double pb = 0.0;
    auto request = [&](long long st) {
      const Rec* src = ring + ((st % kSlots) * H + min(h, H - 1)) * P + min(q, P - 1);
      pv = __hip_atomic_load(&src->v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      pchk = __hip_atomic_load(&src->chk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      pb = d.b[max(min(row_begin + (int)st * H + min(h, H - 1), d.m - 1), 0)];
    };
    double e0 = 0.0;
    bool dead = false;
    for (long long seq = 0; seq < nsteps; ++seq) {
      const long long r = seq - 1;
      if (r >= 0) {
        const unsigned long long want = d.seq0 + (unsigned long long)r + 1;
        unsigned long long v = pv, chk = pchk;
        const bool mine = h < H && q < P;
        if (!NOWAIT && !__all((v ^ chk) == want || !mine) && !dead) {
          const Rec* src = ring + ((r % kSlots) * H + min(h, H - 1)) * P + min(q, P - 1);
          const long long t0 = wall_clock64();
          for (;;) {
            v = __hip_atomic_load(&src->v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            chk = __hip_atomic_load(&src->chk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__all((v ^ chk) == want || !mine)) break;
            if (wall_clock64() - t0 > 2000000) {
              dead = true;
              if (lane == 0) atomicExch(d.err, 1);
              break;
            }
            __builtin_amdgcn_s_sleep(1);
          }
        }
        double ax = (dead || !mine) ? 0.0 : __builtin_bit_cast(double, v);
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) ax += __shfl_xor(ax, off, 64);       // within the 16 lanes of a row
        const int i = row_begin + (int)r * H + h;
        const double res = (h < H && i < row_end) ? ax - pb : 0.0;
        if (q == 0 && h < H) {
          res_s[r & 3][h] = res;
          if (s == 0) e0 += res * res;
        }
      }
      __syncthreads();
      double own = (h < H && q < CW) ? wsum[seq & 1][h][q] : 0.0;
#pragma unroll
      for (int off = 1; off < 16; off <<= 1) own += __shfl_xor(own, off, 64);
      if (q == 0 && h < H) {
        const unsigned long long bits = __builtin_bit_cast(unsigned long long, own);
        Rec* rec = ring + ((seq % kSlots) * H + h) * P + s;
        __hip_atomic_store(&rec->v, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&rec->chk, bits ^ (d.seq0 + (unsigned long long)seq + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      request(r + 1);
    }
    // the two lanes that hold loss sums: add them
    double tot = e0 + __shfl_xor(e0, 16, 64);
    if (lane == 0 && s == 0) d.part[(int64_t)team * d.ldpart + d.n] = tot;
    return;
  }
  // ---- the compute waves ----
  double acc[ACCLDS ? 1 : KV][2];
#pragma unroll
  for (int k = 0; k < KV; ++k) {
    const int lv = tid + k * BSC;
    const int j = (v0 + lv) * 2;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      xs[lv * 2 + e] = (lv < nvs && j + e < d.n) ? d.x[j + e] : 0.0;
      if constexpr (ACCLDS) as[lv * 2 + e] = 0.0;
      else acc[k][e] = 0.0;
    }
  }
  const int voff = tid * 16;
  const int64_t cbase = (int64_t)v0 * 16;
  const int slice_bytes = (int)max((int64_t)0, min((int64_t)nvs * 16, d.ldT * 8 - cbase));
  v2d a[G][H][KV];
  auto load = [&](int g, long long st) {
#pragma unroll
    for (int h = 0; h < H; ++h) {
      const int ri = __builtin_amdgcn_readfirstlane(max(min(row_begin + (int)st * H + h, row_end - 1), 0));
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
          (void*)((const char*)(d.AT + (int64_t)ri * d.ldT) + cbase), 0, slice_bytes, 0x00020000);
#pragma unroll
      for (int k = 0; k < KV; ++k) {
        const v4u raw = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, k * BSC * 16, NT ? 2 : 0);
        a[g][h][k] = __builtin_bit_cast(v2d, raw);
      }
    }
  };
  auto dots = [&](int g, long long seq) {
    double p[H];
#pragma unroll
    for (int h = 0; h < H; ++h) p[h] = 0.0;
#pragma unroll
    for (int k = 0; k < KV; ++k) {
      const v2d xv = *(const v2d*)&xs[(tid + k * BSC) * 2];
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int h = 0; h < H; ++h) p[h] = fma(a[g][h][k][e], xv[e], p[h]);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int h = 0; h < H; ++h) {
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) p[h] += __shfl_xor(p[h], off, 64);
      wsum[seq & 1][h][wv] = p[h];
    }
    __syncthreads();
  };
  __builtin_amdgcn_sched_barrier(0);
  load(0, 0);
  __builtin_amdgcn_sched_barrier(0);
  load(1, 1);
  __builtin_amdgcn_sched_barrier(0);
  dots(0, 0);                                              // step 0: nothing to accumulate yet
  __builtin_amdgcn_sched_barrier(0);
  load(2, 2);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
  for (long long q = 1; q < nsteps; q += G) {
#pragma unroll
    for (int u = 0; u < G; ++u) {
      const long long seq = q + u, r = seq - 1;            // dots of step seq: slot (1 + u) % G ; accumulates step r: slot u
      dots((1 + u) % G, seq);
      double rr[H];
#pragma unroll
      for (int h = 0; h < H; ++h) rr[h] = res_s[r & 3][h];
#pragma unroll
      for (int k = 0; k < KV; ++k) {
        if constexpr (ACCLDS) {
          v2d av = *(v2d*)&as[(tid + k * BSC) * 2];
#pragma unroll
          for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int h = 0; h < H; ++h) av[e] = fma(a[u][h][k][e], rr[h], av[e]);
          *(v2d*)&as[(tid + k * BSC) * 2] = av;
          __builtin_amdgcn_sched_barrier(0);
        } else {
#pragma unroll
          for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int h = 0; h < H; ++h) acc[k][e] = fma(a[u][h][k][e], rr[h], acc[k][e]);
        }
      }
      if constexpr (!ACCLDS) {
#pragma unroll
        for (int k = 0; k < KV; ++k)
#pragma unroll
          for (int e = 0; e < 2; ++e) asm volatile("" : "+v"(acc[k][e]));
      }
      __builtin_amdgcn_sched_barrier(0);
      load(u, r + G);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  double* out = d.part + (int64_t)team * d.ldpart;
#pragma unroll
  for (int k = 0; k < KV; ++k) {
    const int lv = tid + k * BSC;
    const int j = (v0 + lv) * 2;
#pragma unroll
    for (int e = 0; e < 2; ++e)
      if (lv < nvs && j + e < d.n) out[j + e] = ACCLDS ? as[lv * 2 + e] : acc[k][e];
  }
}
// k_split5 with the communication wave on the SCALAR path (P = 16: 4 x s_load_dwordx16 per row of the step)
template <int KV, bool NT>
__global__ __launch_bounds__(768) void k_split6(Args d) {
  constexpr int P = 16;
  constexpr bool ACCLDS = false;
  constexpr int CW = 11, BSC = CW * 64, TPX = 32 / P, H = 2, G = 3;
  extern __shared__ double lds5[];                         // x slice [KV * BSC * 2], then (ACCLDS) the accumulator, same shape
  double* xs = lds5;
  double* as = lds5 + KV * BSC * 2;
  __shared__ double wsum[2][H][16];
  __shared__ double res_s[4][H];
  __shared__ int ident[2];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (tid == 0) {
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u;
    const unsigned t = atomicAdd(d.tickets + xcc, 1u) - d.launch * 32u;
    ident[0] = t < (unsigned)(TPX * P) ? (int)(xcc * TPX + t / P) : -1;
    ident[1] = (int)(t % P);
    if (t >= 32u) atomicExch(d.err, 2);
  }
  __syncthreads();
  const int team = __builtin_amdgcn_readfirstlane(ident[0]), s = __builtin_amdgcn_readfirstlane(ident[1]);
  if (team < 0) return;
  const int nvec = (d.n + 1) / 2;
  const int nvs = (nvec + P - 1) / P;
  const int v0 = s * nvs;
  const int row_begin = team * d.rows_per_team;
  const int row_end = min(row_begin + d.rows_per_team, d.m);
  const int nrows = max(row_end - row_begin, 0);
  const int nsteps = 1 + ((nrows + H - 1) / H + G - 1) / G * G;      // one step of prologue, then whole rounds of G steps
  Rec* ring = (Rec*)d.xch + (int64_t)team * kSlots * H * P;
  if (wv == CW) {
    // ---- the communication wave: everything wave-uniform, the records through the scalar path ----
    typedef unsigned v16u __attribute__((ext_vector_type(16)));
    typedef unsigned long long rec_t __attribute__((ext_vector_type(2)));
    double e0 = 0.0;
    bool dead = false;
    for (long long seq = 0; seq < nsteps; ++seq) {
      const long long r = seq - 1;
      if (r >= 0) {
        const unsigned long long want = d.seq0 + (unsigned long long)r + 1;
#pragma unroll
        for (int h = 0; h < H; ++h) {
          const Rec* src = ring + ((r % kSlots) * H + h) * P;
          double ax = 0.0;
          const long long t0 = wall_clock64();
          for (;;) {
            v16u q0, q1, q2, q3;
            asm volatile(
                "s_load_dwordx16 %0, %4, 0x0 glc\n\ts_load_dwordx16 %1, %4, 0x40 glc\n\ts_load_dwordx16 %2, %4, 0x80 glc\n\t"
                "s_load_dwordx16 %3, %4, 0xc0 glc\n\ts_waitcnt lgkmcnt(0)"
                : "=&s"(q0), "=&s"(q1), "=&s"(q2), "=&s"(q3)
                : "s"(src)
                : "memory");
            bool ok = true;
            double sum = 0.0;
            auto take = [&](const v16u& qq, int i) {
              const unsigned long long v = ((unsigned long long)qq[4 * i + 1] << 32) | qq[4 * i];
              const unsigned long long chk = ((unsigned long long)qq[4 * i + 3] << 32) | qq[4 * i + 2];
              ok = ok && ((v ^ chk) == want);
              sum += __builtin_bit_cast(double, v);        // member order
            };
#pragma unroll
            for (int i = 0; i < 4; ++i) take(q0, i);
#pragma unroll
            for (int i = 0; i < 4; ++i) take(q1, i);
#pragma unroll
            for (int i = 0; i < 4; ++i) take(q2, i);
#pragma unroll
            for (int i = 0; i < 4; ++i) take(q3, i);
            if (ok || dead) {
              ax = sum;
              break;
            }
            if (wall_clock64() - t0 > 2000000) {
              dead = true;
              if (lane == 0) atomicExch(d.err, 1);
            }
            __builtin_amdgcn_s_sleep(1);
          }
          const int i = row_begin + (int)r * H + h;
          const double bi = d.b[max(min(i, d.m - 1), 0)];
          const double res = i < row_end ? ax - bi : 0.0;
          if (lane == 0) {
            res_s[r & 3][h] = res;
            if (s == 0) e0 += res * res;
          }
        }
      }
      __syncthreads();
#pragma unroll
      for (int h = 0; h < H; ++h) {
        double own = lane < CW ? wsum[seq & 1][h][lane] : 0.0;
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) own += __shfl_xor(own, off, 64);
        if (lane == 0) {
          const unsigned long long bits = __builtin_bit_cast(unsigned long long, own);
          const rec_t rec = {bits, bits ^ (d.seq0 + (unsigned long long)seq + 1)};
          *(volatile rec_t*)(ring + ((seq % kSlots) * H + h) * P + s) = rec;
        }
      }
    }
    if (lane == 0 && s == 0) d.part[(int64_t)team * d.ldpart + d.n] = e0;
    return;
  }
  // ---- the compute waves ----
  double acc[ACCLDS ? 1 : KV][2];
#pragma unroll
  for (int k = 0; k < KV; ++k) {
    const int lv = tid + k * BSC;
    const int j = (v0 + lv) * 2;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      xs[lv * 2 + e] = (lv < nvs && j + e < d.n) ? d.x[j + e] : 0.0;
      if constexpr (ACCLDS) as[lv * 2 + e] = 0.0;
      else acc[k][e] = 0.0;
    }
  }
  const int voff = tid * 16;
  const int64_t cbase = (int64_t)v0 * 16;
  const int slice_bytes = (int)max((int64_t)0, min((int64_t)nvs * 16, d.ldT * 8 - cbase));
  v2d a[G][H][KV];
  auto load = [&](int g, long long st) {
#pragma unroll
    for (int h = 0; h < H; ++h) {
      const int ri = __builtin_amdgcn_readfirstlane(max(min(row_begin + (int)st * H + h, row_end - 1), 0));
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
          (void*)((const char*)(d.AT + (int64_t)ri * d.ldT) + cbase), 0, slice_bytes, 0x00020000);
#pragma unroll
      for (int k = 0; k < KV; ++k) {
        const v4u raw = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, k * BSC * 16, NT ? 2 : 0);
        a[g][h][k] = __builtin_bit_cast(v2d, raw);
      }
    }
  };
  auto dots = [&](int g, long long seq) {
    double p[H];
#pragma unroll
    for (int h = 0; h < H; ++h) p[h] = 0.0;
#pragma unroll
    for (int k = 0; k < KV; ++k) {
      const v2d xv = *(const v2d*)&xs[(tid + k * BSC) * 2];
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int h = 0; h < H; ++h) p[h] = fma(a[g][h][k][e], xv[e], p[h]);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int h = 0; h < H; ++h) {
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) p[h] += __shfl_xor(p[h], off, 64);
      wsum[seq & 1][h][wv] = p[h];
    }
    __syncthreads();
  };
  __builtin_amdgcn_sched_barrier(0);
  load(0, 0);
  __builtin_amdgcn_sched_barrier(0);
  load(1, 1);
  __builtin_amdgcn_sched_barrier(0);
  dots(0, 0);                                              // step 0: nothing to accumulate yet
  __builtin_amdgcn_sched_barrier(0);
  load(2, 2);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
  for (long long q = 1; q < nsteps; q += G) {
#pragma unroll
    for (int u = 0; u < G; ++u) {
      const long long seq = q + u, r = seq - 1;            // dots of step seq: slot (1 + u) % G ; accumulates step r: slot u
      dots((1 + u) % G, seq);
      double rr[H];
#pragma unroll
      for (int h = 0; h < H; ++h) rr[h] = res_s[r & 3][h];
#pragma unroll
      for (int k = 0; k < KV; ++k) {
        if constexpr (ACCLDS) {
          v2d av = *(v2d*)&as[(tid + k * BSC) * 2];
#pragma unroll
          for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int h = 0; h < H; ++h) av[e] = fma(a[u][h][k][e], rr[h], av[e]);
          *(v2d*)&as[(tid + k * BSC) * 2] = av;
          __builtin_amdgcn_sched_barrier(0);
        } else {
#pragma unroll
          for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int h = 0; h < H; ++h) acc[k][e] = fma(a[u][h][k][e], rr[h], acc[k][e]);
        }
      }
      if constexpr (!ACCLDS) {
#pragma unroll
        for (int k = 0; k < KV; ++k)
#pragma unroll
          for (int e = 0; e < 2; ++e) asm volatile("" : "+v"(acc[k][e]));
      }
      __builtin_amdgcn_sched_barrier(0);
      load(u, r + G);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  double* out = d.part + (int64_t)team * d.ldpart;
#pragma unroll
  for (int k = 0; k < KV; ++k) {
    const int lv = tid + k * BSC;
    const int j = (v0 + lv) * 2;
#pragma unroll
    for (int e = 0; e < 2; ++e)
      if (lv < nvs && j + e < d.n) out[j + e] = ACCLDS ? as[lv * 2 + e] : acc[k][e];
  }
}
template <int KV, bool NT>
static void l_split6(const Args& a, int blocks, hipStream_t st) {
  static bool once = false;
  const size_t lds = (size_t)KV * 704 * 2 * 8;
  if (!once && lds > 48 * 1024) {
    CK(hipFuncSetAttribute((const void*)k_split6<KV, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    once = true;
  }
  hipLaunchKernelGGL((k_split6<KV, NT>), dim3(blocks), dim3(768), lds, st, a);
}
template <int P, int KV, bool NT, bool ACCLDS, bool NOWAIT = false>
static void l_split5(const Args& a, int blocks, hipStream_t st) {
  static bool once = false;
  const size_t lds = (size_t)KV * 704 * 2 * 8 * (ACCLDS ? 2 : 1);
  if (!once && lds > 48 * 1024) {
    CK(hipFuncSetAttribute((const void*)k_split5<P, KV, NT, ACCLDS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    once = true;
  }
  hipLaunchKernelGGL((k_split5<P, KV, NT, ACCLDS, NOWAIT>), dim3(blocks), dim3(768), lds, st, a);
}

// ---------------------------------------------------------------------------------------------------------------------
// Fifth form = the product kernel's structure (csrc/slab.hip k_team_rows): branch-free compute loop (a peeled prologue,
// dummy rows), so that the compiler's waits are counted ones and the rows in the loading slots really stay in flight.
// MODE 3: no barriers, no communication wave (the access pattern alone, WITH effective prefetch).
// ---------------------------------------------------------------------------------------------------------------------
template <int P, int KV, int G, int DEPTH, int MODE, int CW = 11>
__global__ __launch_bounds__((CW + 1) * 64) void k_split7(Args d) {
  constexpr int BSC = CW * 64, TPX = 32 / P;
  extern __shared__ double xs7[];
  __shared__ double wsum[2][16];
  __shared__ double res_s[4];
  __shared__ int ident[2];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (tid == 0) {
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u;
    const unsigned t = atomicAdd(d.tickets + xcc, 1u) - d.launch * 32u;
    ident[0] = t < (unsigned)(TPX * P) ? (int)(xcc * TPX + t / P) : -1;
    ident[1] = (int)(t % P);
    if (t >= 32u) atomicExch(d.err, 2);
  }
  __syncthreads();
  const int team = __builtin_amdgcn_readfirstlane(ident[0]), s = __builtin_amdgcn_readfirstlane(ident[1]);
  if (team < 0) return;
  const int nvec = (d.n + 1) / 2;
  const int nvs = (nvec + P - 1) / P;
  const int v0 = s * nvs;
  const int row_begin = team * d.rows_per_team;
  const int row_end = min(row_begin + d.rows_per_team, d.m);
  const int nrows = max(row_end - row_begin, 0);
  const int nsteps = DEPTH + (nrows + G - 1) / G * G;
  Rec* ring = (Rec*)d.xch + (int64_t)team * kSlots * P;
  if (wv == CW) {
    if (MODE == 3) return;
    unsigned long long pv = 0, pchk = 0;
    double pb = 0.0;
    auto request = [&](long long r) {
      const Rec* src = ring + (r % kSlots) * P + min(lane, P - 1);
      pv = __hip_atomic_load(&src->v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      pchk = __hip_atomic_load(&src->chk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      pb = d.b[max(min(row_begin + (int)r, d.m - 1), 0)];
    };
    double e0 = 0.0;
    bool dead = false;
    for (long long seq = 0; seq < nsteps; ++seq) {
      const long long r = seq - DEPTH;
      if (r >= 0) {
        const unsigned long long want = d.seq0 + (unsigned long long)r + 1;
        unsigned long long v = pv, chk = pchk;
        if (!__all((v ^ chk) == want || lane >= P) && !dead) {
          const Rec* src = ring + (r % kSlots) * P + min(lane, P - 1);
          const long long t0 = wall_clock64();
          for (;;) {
            v = __hip_atomic_load(&src->v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            chk = __hip_atomic_load(&src->chk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__all((v ^ chk) == want || lane >= P)) break;
            if (wall_clock64() - t0 > 2000000) {
              dead = true;
              if (lane == 0) atomicExch(d.err, 1);
              break;
            }
            __builtin_amdgcn_s_sleep(1);
          }
        }
        double ax = (dead || lane >= P) ? 0.0 : __builtin_bit_cast(double, v);
#pragma unroll
        for (int off = 1; off < P; off <<= 1) ax += __shfl_xor(ax, off, 64);
        const int i = row_begin + (int)r;
        const double res = i < row_end ? ax - pb : 0.0;
        if (lane == 0) {
          res_s[r & 3] = res;
          if (s == 0) e0 += res * res;
        }
      }
      __syncthreads();
      double own = lane < CW ? wsum[seq & 1][lane] : 0.0;
#pragma unroll
      for (int off = 1; off < 16; off <<= 1) own += __shfl_xor(own, off, 64);
      if (lane == 0) {
        const unsigned long long bits = __builtin_bit_cast(unsigned long long, own);
        Rec* rec = ring + (seq % kSlots) * P + s;
        __hip_atomic_store(&rec->v, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&rec->chk, bits ^ (d.seq0 + (unsigned long long)seq + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (r + 1 >= 0) request(r + 1);
    }
    if (lane == 0 && s == 0) d.part[(int64_t)team * d.ldpart + d.n] = e0;
    return;
  }
  double acc[KV][2];
#pragma unroll
  for (int k = 0; k < KV; ++k) {
    const int lv = tid + k * BSC;
    const int j = (v0 + lv) * 2;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      xs7[lv * 2 + e] = (lv < nvs && j + e < d.n) ? d.x[j + e] : 0.0;
      acc[k][e] = 0.0;
    }
  }
  const int voff = tid * 16;
  const int64_t cbase = (int64_t)v0 * 16;
  const int slice_bytes = (int)max((int64_t)0, min((int64_t)nvs * 16, d.ldT * 8 - cbase));
  v2d a[G][KV];
  auto load = [&](int g, int row) {
    const int ri = __builtin_amdgcn_readfirstlane(max(min(row, row_end - 1), 0));
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((const char*)(d.AT + (int64_t)ri * d.ldT) + cbase), 0, slice_bytes, 0x00020000);
#pragma unroll
    for (int k = 0; k < KV; ++k) {
      const v4u raw = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, k * BSC * 16, 2);
      a[g][k] = __builtin_bit_cast(v2d, raw);
    }
  };
  double plast = 0.0;
  auto dots = [&](int g, long long seq) {
    double p = 0.0;
#pragma unroll
    for (int k = 0; k < KV; ++k)
#pragma unroll
      for (int e = 0; e < 2; ++e) p = fma(a[g][k][e], xs7[(tid + k * BSC) * 2 + e], p);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) p += __shfl_xor(p, off, 64);
    if (MODE == 3) {
      plast = p;
    } else {
      wsum[seq & 1][wv] = p;
      __syncthreads();
    }
  };
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int g = 0; g < G - 1; ++g) {
    load(g, row_begin + g);
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int u = 0; u < DEPTH; ++u) {
    dots(u, u);
    __builtin_amdgcn_sched_barrier(0);
    if (u == DEPTH - 1) load(G - 1, row_begin + G - 1);
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll 1
  for (long long q = DEPTH; q < nsteps; q += G) {
#pragma unroll
    for (int u = 0; u < G; ++u) {
      const long long seq = q + u;
      const long long r = seq - DEPTH;
      dots((DEPTH + u) % G, seq);
      const double rr = MODE == 3 ? plast * 1e-30 : res_s[r & 3];
#pragma unroll
      for (int k = 0; k < KV; ++k)
#pragma unroll
        for (int e = 0; e < 2; ++e) acc[k][e] = fma(a[u][k][e], rr, acc[k][e]);
#pragma unroll
      for (int k = 0; k < KV; ++k)
#pragma unroll
        for (int e = 0; e < 2; ++e) asm volatile("" : "+v"(acc[k][e]));
      __builtin_amdgcn_sched_barrier(0);
      load(u, row_begin + (int)(r + G));
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  double* out = d.part + (int64_t)team * d.ldpart;
#pragma unroll
  for (int k = 0; k < KV; ++k) {
    const int lv = tid + k * BSC;
    const int j = (v0 + lv) * 2;
#pragma unroll
    for (int e = 0; e < 2; ++e)
      if (lv < nvs && j + e < d.n) out[j + e] = acc[k][e];
  }
}
template <int P, int KV, int G, int DEPTH, int MODE, int CW = 11>
static void l_split7(const Args& a, int blocks, hipStream_t st) {
  static bool once = false;
  const size_t lds = (size_t)KV * CW * 64 * 2 * 8;
  if (!once && lds > 48 * 1024) {
    CK(hipFuncSetAttribute((const void*)k_split7<P, KV, G, DEPTH, MODE, CW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    once = true;
  }
  hipLaunchKernelGGL((k_split7<P, KV, G, DEPTH, MODE, CW>), dim3(blocks), dim3((CW + 1) * 64), lds, st, a);
}

struct Variant {
  const char* name;
  void (*launch)(const Args&, int blocks, hipStream_t);
  int P;
};
template <int P, int KV, int H, int G, bool NT, bool XLDS = false, bool EARLY = false>
static void l_split(const Args& a, int blocks, hipStream_t st) {
  static bool once = false;
  const size_t lds = XLDS ? (size_t)KV * 768 * 2 * 8 : 0;
  if (!once && lds > 48 * 1024) {
    CK(hipFuncSetAttribute((const void*)k_split<P, KV, H, G, NT, XLDS, EARLY>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    once = true;
  }
  hipLaunchKernelGGL((k_split<P, KV, H, G, NT, XLDS, EARLY>), dim3(blocks), dim3(768), lds, st, a);
}

int main(int argc, char** argv) {
  const int m = argc > 1 ? atoi(argv[1]) : 50000, n = argc > 2 ? atoi(argv[2]) : 20000;
  const int reps = 20;
  const int64_t ldT = (n + 31) / 32 * 32;
  double *AT, *b, *x;
  CK(hipMalloc(&AT, (size_t)m * ldT * 8));
  CK(hipMemset(AT, 0, (size_t)m * ldT * 8));
  hipLaunchKernelGGL(k_fill, dim3(16, m), dim3(256), 0, 0, AT, ldT, m, n);
  std::vector<double> hb(m), hx(n);
  for (int i = 0; i < m; ++i) hb[i] = gen(2, i);
  for (int j = 0; j < n; ++j) hx[j] = 0.01 * gen(3, j);
  CK(hipMalloc(&b, m * 8));
  CK(hipMalloc(&x, n * 8));
  CK(hipMemcpy(b, hb.data(), m * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(x, hx.data(), n * 8, hipMemcpyHostToDevice));
  const int64_t ldpart = (n + 2 + 7) / 8 * 8;
  double* part;
  CK(hipMalloc(&part, (size_t)256 * ldpart * 8));
  Xch* xch;
  CK(hipMalloc(&xch, sizeof(Xch) * 4096 * kSlots));
  CK(hipMemset(xch, 0, sizeof(Xch) * 4096 * kSlots));
  int* err;
  CK(hipMalloc(&err, 4));
  CK(hipMemset(err, 0, 4));
  unsigned* tickets;
  CK(hipMalloc(&tickets, 32));
  CK(hipMemset(tickets, 0, 32));
  unsigned launches = 0;
  // host reference on a sample of rows is not enough for the sum of squares: all rows, but only m * n = 1e9 MACs
  std::vector<double> res(m);
  double ss_ref = 0.0;
  for (int i = 0; i < m; ++i) {
    double dot = 0.0;
    for (int j = 0; j < n; ++j) dot = fma(gen(1, (uint64_t)i * n + j), hx[j], dot);
    res[i] = dot - hb[i];
    ss_ref += res[i] * res[i];
  }
  std::vector<int> cols;
  for (int j = 0; j < 24; ++j) cols.push_back(j), cols.push_back(n - 1 - j), cols.push_back(n / 2 - 12 + j), cols.push_back(n / 4 - 12 + j);
  std::vector<double> g_ref(cols.size(), 0.0);
  for (size_t c = 0; c < cols.size(); ++c)
    for (int i = 0; i < m; ++i) g_ref[c] = fma(gen(1, (uint64_t)i * n + cols[c]), res[i], g_ref[c]);

  const Variant vars[] = {
      {"KV5 comm wave: 11 + 1 waves, G4 D2", l_split7<16, 5, 4, 2, 0>, 16},
      {"KV4 comm wave: 15 + 1 waves (1024 threads), G4 D2", l_split7<16, 4, 4, 2, 0, 15>, 16},
      {"KV4 comm wave: 15 + 1 waves (1024 threads), G4 D2, pattern only", l_split7<16, 4, 4, 2, 3, 15>, 16},
      {"KV5 comm wave: 11 + 1 waves, G4 D2", l_split7<16, 5, 4, 2, 0>, 16},
      {"KV4 comm wave: 15 + 1 waves (1024 threads), G4 D2", l_split7<16, 4, 4, 2, 0, 15>, 16},
      {"KV4 comm wave: 15 + 1 waves (1024 threads), G4 D2, pattern only", l_split7<16, 4, 4, 2, 3, 15>, 16},
  };
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  unsigned long long seq0 = 0;
  for (const Variant& v : vars) {
    const int nv_member = ((n + 1) / 2 + v.P - 1) / v.P;
    const int kv = atoi(strstr(v.name, "KV") + 2);
    const int cap = kv * (strstr(v.name, "15 + 1") ? 960 : strstr(v.name, "comm wave") ? 704 : 768);
    if (nv_member > cap) {
      printf("%-16s skipped: %d vectors per member > %d\n", v.name, nv_member, cap);
      continue;
    }
    const int nteams = 8 * (32 / v.P);
    const int rpt = (m + nteams - 1) / nteams;
    Args a{AT, ldT, m, n, rpt, nteams, b, x, part, ldpart, xch, err, 0, tickets, 0};
    CK(hipMemset(part, 0, (size_t)256 * ldpart * 8));
    const unsigned long long groups_per_launch = (unsigned long long)rpt + 16;
    auto go = [&]() {
      a.seq0 = seq0;
      a.launch = launches++;
      seq0 += groups_per_launch + 16;
      v.launch(a, 256, 0);
    };
    for (int w = 0; w < 2; ++w) go();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int r = 0; r < reps; ++r) go();
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    CK(hipGetLastError());
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    int herr = 0;
    CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
    std::vector<double> hp((size_t)nteams * ldpart);
    CK(hipMemcpy(hp.data(), part, hp.size() * 8, hipMemcpyDeviceToHost));
    double ss = 0.0, gerr = 0.0, gmax = 0.0;
    for (int t = 0; t < nteams; ++t) ss += hp[(size_t)t * ldpart + n];
    for (size_t c = 0; c < cols.size(); ++c) {
      double g = 0.0;
      for (int t = 0; t < nteams; ++t) g += hp[(size_t)t * ldpart + cols[c]];
      gerr = fmax(gerr, fabs(g - g_ref[c]));
      gmax = fmax(gmax, fabs(g_ref[c]));
    }
    printf("%-66s teams %3d rows/team %4d  %.4f ms  %6.0f GB/s   ss rel err %.1e  grad rel err %.1e  timeouts %d\n", v.name, nteams,
           rpt, ms, (double)m * n * 8 / (ms * 1e-3) / 1e9, fabs(ss - ss_ref) / ss_ref, gerr / gmax, herr);
    fflush(stdout);
    if (herr) CK(hipMemset(err, 0, 4));
  }
  return 0;
}
