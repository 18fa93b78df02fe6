// pattern_probe.hip — what a CU-per-column-slice READ PATTERN alone streams at, by workgroup size, slice width and rows in
// flight (no barriers, no hand-off, one fma per element).  Question behind it (DESIGN §4 "Round 2"): k_team_rows reads at
// 6.0 TB/s where the streaming GEMV reads at 7.1 — is it the number of waves per CU, the width of a slice, or the stride?
//
//   hipcc -O3 --offload-arch=gfx950 -o tools/pattern_probe tools/pattern_probe.hip && tools/pattern_probe [m] [n]
//
// Block b: column slice b % PS (of PS slices of a row), row slab b / PS.  Per row the slice is nvs 16-byte vectors,
// KV per thread ; G rows in flight (a slot is re-loaded right after it has been consumed).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                      \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

typedef double vec_t __attribute__((ext_vector_type(2)));
typedef unsigned raw_t __attribute__((ext_vector_type(4)));

template <class T>
__device__ __forceinline__ const T* uniform_ptr(const T* p) {
  const unsigned long long v = (unsigned long long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return (const T*)(((unsigned long long)hi << 32) | lo);
}

struct Args {
  const double* A;
  int64_t ld;
  int m, nvec, PS, nvs, rows_per_slab;
  double* out;
};

template <int BS, int KV, int G, bool NT, int S = BS, bool XL = false, int STAGE = 0>
__global__ __launch_bounds__(BS) void k_pat(Args a) {
  // STAGE: what of the product's row arithmetic is added to the bare pattern —
  //   1: the dot with x from LDS   2: + the wave butterfly   3: + the store of the wave's sum and the barrier per row
  //   4: + the residual read from LDS (written by wave 0 after the barrier from the sums of the row two back)
  __shared__ double xs[STAGE > 0 ? BS * KV * 2 : 2];
  __shared__ double wsum[2][16];
  __shared__ double res_s[4];
  const int tid = threadIdx.x;
  // XL: the slices of a slab on ONE XCD (workgroups are dealt to the XCDs round robin: b % 8), as the product forms its teams
  const int s = XL ? (int)(blockIdx.x >> 3) % a.PS : blockIdx.x % a.PS;
  const int slab = XL ? (int)(blockIdx.x & 7) * (32 / a.PS) + (int)(blockIdx.x >> 3) / a.PS : blockIdx.x / a.PS;
  const int row_begin = slab * a.rows_per_slab;
  const int row_end = min(row_begin + a.rows_per_slab, a.m);
  if (row_begin >= row_end) return;
  const int64_t cbase = (int64_t)s * a.nvs * 16;
  const int slice_bytes = (int)max((int64_t)0, min((int64_t)a.nvs * 16, a.ld * 8 - cbase));
  vec_t v[G][KV];
  double acc[KV][2];
#pragma unroll
  for (int k = 0; k < KV; ++k) acc[k][0] = acc[k][1] = 0.0;
  if (STAGE > 0) {
#pragma unroll
    for (int k = 0; k < KV; ++k) xs[(tid + k * BS) * 2] = 0.5, xs[(tid + k * BS) * 2 + 1] = 0.25;
    if (tid < 4) res_s[tid] = 1.0;
    __syncthreads();
  }
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  double keep = 0.0;
  auto dots = [&](int g, int seq) {
    if (STAGE == 0) return;
    double p = 0.0;
#pragma unroll
    for (int k = 0; k < KV; ++k)
#pragma unroll
      for (int e = 0; e < 2; ++e) p = fma(v[g][k][e], xs[(tid + k * BS) * 2 + e], p);
    if (STAGE >= 2) {
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) p += __shfl_xor(p, off, 64);
    }
    if (STAGE >= 3) {
      wsum[seq & 1][wv] = p;
      __syncthreads();
      if (STAGE >= 4 && wv == 0) {
        double own = (tid & 63) < BS / 64 ? wsum[seq & 1][tid & 63] : 0.0;
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) own += __shfl_xor(own, off, 64);
        if (tid == 0) res_s[(seq + 2) & 3] = own * 1e-30 + 1.0;
      }
    } else {
      keep += p;
    }
  };
  auto load = [&](int g, int row) {
    const int ri = __builtin_amdgcn_readfirstlane(max(min(row, row_end - 1), 0));
    const __amdgpu_buffer_rsrc_t rs =
        __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)(a.A + (int64_t)ri * a.ld) + cbase), 0, slice_bytes, 0x00020000);
#pragma unroll
    for (int k = 0; k < KV; ++k)
      v[g][k] = __builtin_bit_cast(vec_t, __builtin_amdgcn_raw_buffer_load_b128(rs, ((tid / S) * KV * S + tid % S) * 16, k * S * 16, NT ? 2 : 0));
  };
#pragma unroll
  for (int g = 0; g < G; ++g) {
    load(g, row_begin + g);
    __builtin_amdgcn_sched_barrier(0);
  }
  const int nrows = row_end - row_begin;
  const int nsteps = (nrows + G - 1) / G * G;
#pragma unroll 1
  for (int q = 0; q < nsteps; q += G) {
#pragma unroll
    for (int u = 0; u < G; ++u) {
      dots((u + 2) % G, q + u);
      const double w = STAGE >= 4 ? res_s[(q + u) & 3] : ((q + u) < nrows ? 1.0 : 0.0);
#pragma unroll
      for (int k = 0; k < KV; ++k)
#pragma unroll
        for (int e = 0; e < 2; ++e) acc[k][e] = fma(v[u][k][e], w, acc[k][e]);
#pragma unroll
      for (int k = 0; k < KV; ++k)
#pragma unroll
        for (int e = 0; e < 2; ++e) asm volatile("" : "+v"(acc[k][e]));
      __builtin_amdgcn_sched_barrier(0);
      load(u, row_begin + q + u + G);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  double t = 0.0;
#pragma unroll
  for (int k = 0; k < KV; ++k) t += acc[k][0] + acc[k][1];
  if (t + keep == 12345.678) a.out[0] = t;      // (never: keeps the loads)
}


// The product's structure with its hand-off (csrc/slab.hip k_team_rows): 11 compute waves + a communication wave, teams
// formed inside an XCD (XCC_ID + tickets).  What the hand-off is made of:
//   ST  0: records stored with agent-scope atomics (sc1)   1: plain stores (they stay in the XCD's L2)   2: none
//   LD  0: agent-scope atomic loads (sc1), lane q member q   1: workgroup-scope (sc0: past the L1, from the XCD's L2)
//       2: scalar loads, glc (4 x s_load_dwordx16 for sixteen records)   3: none
//   WAIT: the records are validated and waited for (else: the same traffic, no coupling between the members)
struct Rec5 {
  unsigned long long v, chk;
};
struct Args5 {
  Args a;
  Rec5* ring;
  unsigned long long seq0;
  int* err;
  unsigned* tickets;
  unsigned launch;
};
typedef unsigned raw16_t __attribute__((ext_vector_type(16)));
template <int KV, int G, int DEPTH, int ST, int LD, bool WAIT, int P = 16>
__global__ __launch_bounds__(768) void k_pat5(Args5 d) {
  constexpr int CW = 11, BSC = CW * 64, SLOTS = 8, TPX = 32 / P;
  const Args& a = d.a;
  __shared__ double xs[BSC * KV * 2];
  __shared__ double wsum[2][16];
  __shared__ double res_s[4];
  __shared__ int ident[2];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (tid == 0) {
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u;
    const unsigned k = atomicAdd(d.tickets + xcc, 1u) - d.launch * 32u;
    ident[0] = k < (unsigned)(TPX * P) ? (int)(xcc * TPX + k / P) : -1;
    ident[1] = (int)(k % P);
    if (k >= 32u) atomicCAS(d.err, 0, 2);
  }
  if (tid < 4) res_s[tid] = 1.0;
  __syncthreads();
  const int team = __builtin_amdgcn_readfirstlane(ident[0]), s = __builtin_amdgcn_readfirstlane(ident[1]);
  if (team < 0) return;
  const int row_begin = team * a.rows_per_slab;
  const int row_end = min(row_begin + a.rows_per_slab, a.m);
  const int nrows = max(row_end - row_begin, 0);
  const int nsteps = DEPTH + (nrows + G - 1) / G * G;
  Rec5* ring = d.ring + (int64_t)team * SLOTS * P;
  if (wv == CW) {
    __syncthreads();
    unsigned long long pv = 0, pchk = 0;
    raw16_t q0 = {}, q1 = {}, q2 = {}, q3 = {};
    auto fetch = [&](long long r) {              // the P records of row r -> pv / pchk (lane q: member q)
      const Rec5* row = ring + (r % SLOTS) * P;
      if (LD == 0) {
        pv = __hip_atomic_load(&row[min(lane, P - 1)].v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        pchk = __hip_atomic_load(&row[min(lane, P - 1)].chk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else if (LD == 1) {
        pv = __hip_atomic_load(&row[min(lane, P - 1)].v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        pchk = __hip_atomic_load(&row[min(lane, P - 1)].chk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      } else if (LD == 2 && P == 4) {
        asm volatile("s_load_dwordx16 %0, %1, 0x0 glc" : "=&s"(q0) : "s"(row) : "memory");
      } else if (LD == 2 && P == 8) {
        asm volatile("s_load_dwordx16 %0, %2, 0x0 glc\n\ts_load_dwordx16 %1, %2, 0x40 glc" : "=&s"(q0), "=&s"(q1) : "s"(row) : "memory");
      } else if (LD == 2 && P == 2) {
        typedef unsigned raw8_t __attribute__((ext_vector_type(8)));
        raw8_t h;
        asm volatile("s_load_dwordx8 %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=&s"(h) : "s"(row) : "memory");
#pragma unroll
        for (int c = 0; c < 8; ++c) q0[c] = h[c];
      } else if (LD == 2) {
        asm volatile("s_load_dwordx16 %0, %4, 0x0 glc\n\ts_load_dwordx16 %1, %4, 0x40 glc\n\t"
                     "s_load_dwordx16 %2, %4, 0x80 glc\n\ts_load_dwordx16 %3, %4, 0xc0 glc"
                     : "=&s"(q0), "=&s"(q1), "=&s"(q2), "=&s"(q3)
                     : "s"(row)
                     : "memory");
      }
    };
    auto unpack = [&]() {                         // (scalar path: the sixteen records from SGPRs to lanes 0 .. 15)
      if (LD != 2) return;
      asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(q0), "+s"(q1), "+s"(q2), "+s"(q3)::"memory");
      unsigned w0 = 0, w1 = 0, w2 = 0, w3 = 0;
#pragma unroll
      for (int m = 0; m < P; ++m) {
        const raw16_t& q = m < 4 ? q0 : m < 8 ? q1 : m < 12 ? q2 : q3;
        const int o = (m & 3) * 4;
        if (lane == m) w0 = q[o], w1 = q[o + 1], w2 = q[o + 2], w3 = q[o + 3];
      }
      pv = ((unsigned long long)w1 << 32) | w0;
      pchk = ((unsigned long long)w3 << 32) | w2;
    };
    bool dead = false;
    double keepc = 0.0;
    for (long long seq = 0; seq < nsteps; ++seq) {
      const long long r = seq - DEPTH;
      if (r >= 0) {
        double ax = 0.0;
        if (LD != 3) {
          const unsigned long long want = d.seq0 + (unsigned long long)r + 1;
          unpack();
          if (WAIT && !__all((pv ^ pchk) == want || lane >= P) && !dead) {
            const long long t0 = wall_clock64();
            for (;;) {
              fetch(r);
              unpack();
              if (__all((pv ^ pchk) == want || lane >= P)) break;
              if (wall_clock64() - t0 > 2000000) {
                dead = true;
                if (lane == 0) atomicCAS(d.err, 0, 1);
                break;
              }
              __builtin_amdgcn_s_sleep(1);
            }
          }
          ax = (dead || lane >= P) ? 0.0 : __builtin_bit_cast(double, pv);
#pragma unroll
          for (int off = 1; off < P; off <<= 1) ax += __shfl_xor(ax, off, 64);
        }
        keepc += ax;
        if (lane == 0) res_s[r & 3] = ax * 1e-300 + 1.0;
      }
      __syncthreads();
      double own = lane < CW ? wsum[seq & 1][lane] : 0.0;
#pragma unroll
      for (int off = 1; off < 16; off <<= 1) own += __shfl_xor(own, off, 64);
      keepc += own;
      if (ST != 2 && lane == 0) {
        const unsigned long long bits = __builtin_bit_cast(unsigned long long, own);
        Rec5* rec = ring + (seq % SLOTS) * P + s;
        const unsigned long long chk = bits ^ (d.seq0 + (unsigned long long)seq + 1);
        if (ST == 0) {
          __hip_atomic_store(&rec->v, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(&rec->chk, chk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
          typedef unsigned long long u2_t __attribute__((ext_vector_type(2)));
          const u2_t both = {bits, chk};
          asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(rec), "v"(both) : "memory");
        }
      }
      if (r + 1 >= 0) fetch(r + 1);
    }
    if (keepc == 12345.678) a.out[1] = keepc;
    return;
  }
  const int64_t cbase = (int64_t)s * a.nvs * 16;
  const int slice_bytes = (int)max((int64_t)0, min((int64_t)a.nvs * 16, a.ld * 8 - cbase));
  vec_t v[G][KV];
  double acc[KV][2];
#pragma unroll
  for (int k = 0; k < KV; ++k) {
    acc[k][0] = acc[k][1] = 0.0;
    xs[(tid + k * BSC) * 2] = 0.5, xs[(tid + k * BSC) * 2 + 1] = 0.25;
  }
  __syncthreads();
  auto load = [&](int g, int row) {
    const int ri = __builtin_amdgcn_readfirstlane(max(min(row, row_end - 1), 0));
    const __amdgpu_buffer_rsrc_t rs =
        __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)(a.A + (int64_t)ri * a.ld) + cbase), 0, slice_bytes, 0x00020000);
#pragma unroll
    for (int k = 0; k < KV; ++k)
      v[g][k] = __builtin_bit_cast(vec_t, __builtin_amdgcn_raw_buffer_load_b128(rs, tid * 16, k * BSC * 16, 2));
  };
  auto dots = [&](int g, long long seq) {
    double p = 0.0;
#pragma unroll
    for (int k = 0; k < KV; ++k)
#pragma unroll
      for (int e = 0; e < 2; ++e) p = fma(v[g][k][e], xs[(tid + k * BSC) * 2 + e], p);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) p += __shfl_xor(p, off, 64);
    wsum[seq & 1][wv] = p;
    __syncthreads();
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
      const double rr = res_s[r & 3];
#pragma unroll
      for (int k = 0; k < KV; ++k)
#pragma unroll
        for (int e = 0; e < 2; ++e) acc[k][e] = fma(v[u][k][e], rr, acc[k][e]);
#pragma unroll
      for (int k = 0; k < KV; ++k)
#pragma unroll
        for (int e = 0; e < 2; ++e) asm volatile("" : "+v"(acc[k][e]));
      __builtin_amdgcn_sched_barrier(0);
      load(u, row_begin + (int)(r + G));
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  double t = 0.0;
#pragma unroll
  for (int k = 0; k < KV; ++k) t += acc[k][0] + acc[k][1];
  if (t == 12345.678) a.out[0] = t;
}

// A two-level hand-off: member 0 of a team (the leader) collects the sixteen partials of a row, adds them and publishes ONE
// record with the sum, padded to a 64-byte line of its own ; every member takes that record with one s_load_dwordx4.  A row's
// residual is then due three steps after its dots (DEPTH = 3, five row slots).  80 scalar loads of 64 bytes per team and row
// become 4 + 16 of 16.
template <int KV, int G, int DEPTH, int P = 16>
__global__ __launch_bounds__(768) void k_pat6(Args5 d) {
  constexpr int ST = 0, LD = 2;
  constexpr bool WAIT = true;
  constexpr int CW = 11, BSC = CW * 64, SLOTS = 8, TPX = 32 / P;
  const Args& a = d.a;
  __shared__ double xs[BSC * KV * 2];
  __shared__ double wsum[2][16];
  __shared__ double res_s[4];
  __shared__ int ident[2];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (tid == 0) {
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u;
    const unsigned k = atomicAdd(d.tickets + xcc, 1u) - d.launch * 32u;
    ident[0] = k < (unsigned)(TPX * P) ? (int)(xcc * TPX + k / P) : -1;
    ident[1] = (int)(k % P);
    if (k >= 32u) atomicCAS(d.err, 0, 2);
  }
  if (tid < 4) res_s[tid] = 1.0;
  __syncthreads();
  const int team = __builtin_amdgcn_readfirstlane(ident[0]), s = __builtin_amdgcn_readfirstlane(ident[1]);
  if (team < 0) return;
  const int row_begin = team * a.rows_per_slab;
  const int row_end = min(row_begin + a.rows_per_slab, a.m);
  const int nrows = max(row_end - row_begin, 0);
  const int nsteps = DEPTH + (nrows + G - 1) / G * G;
  Rec5* ring = d.ring + (int64_t)team * SLOTS * P;
  if (wv == CW) {
    static_assert(DEPTH == 3 && P == 16, "leader tree: sixteen members, three steps");
    typedef unsigned raw4_t __attribute__((ext_vector_type(4)));
    __syncthreads();
    Rec5* bring = d.ring + 8 * 256 + (int64_t)team * SLOTS * 4;          // (the sums: one 64-byte line per row slot)
    raw16_t q0 = {}, q1 = {}, q2 = {}, q3 = {};
    raw4_t bq = {};
    auto fetch16 = [&](long long r) {
      const Rec5* row = uniform_ptr(ring + (r % SLOTS) * P);
      asm volatile("s_load_dwordx16 %0, %4, 0x0 glc\n\ts_load_dwordx16 %1, %4, 0x40 glc\n\t"
                   "s_load_dwordx16 %2, %4, 0x80 glc\n\ts_load_dwordx16 %3, %4, 0xc0 glc\n\ts_waitcnt lgkmcnt(0)"
                   : "=&s"(q0), "=&s"(q1), "=&s"(q2), "=&s"(q3)
                   : "s"(row)
                   : "memory");
    };
    auto fetch1 = [&](long long r) {
      const Rec5* src = uniform_ptr(bring + (r % SLOTS) * 4);
      asm volatile("s_load_dwordx4 %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(bq) : "s"(src) : "memory");
    };
    auto word = [&](int m, int w) -> unsigned long long {
      const raw16_t& c = m < 4 ? q0 : m < 8 ? q1 : m < 12 ? q2 : q3;
      const int o = (m & 3) * 4 + w * 2;
      return ((unsigned long long)c[o + 1] << 32) | c[o];
    };
    auto all16 = [&](unsigned long long want) {
      bool ok = true;
#pragma unroll
      for (int m = 0; m < P; ++m) ok &= (word(m, 0) ^ word(m, 1)) == want;
      return ok;
    };
    bool dead = false;
    double keepc = 0.0;
    for (long long seq = 0; seq < nsteps; ++seq) {
      if (s == 0 && seq - 2 >= 0) {                 // the leader: the sixteen partials of row seq - 2 -> their sum
        const long long r1 = seq - 2;
        const unsigned long long want = d.seq0 + (unsigned long long)r1 + 1;
        if (!all16(want) && !dead) {
          const long long t0 = wall_clock64();
          for (;;) {
            fetch16(r1);
            if (all16(want)) break;
            if (wall_clock64() - t0 > 2000000) {
              dead = true;
              if (lane == 0) atomicCAS(d.err, 0, 1);
              break;
            }
            __builtin_amdgcn_s_sleep(1);
          }
        }
        double part[P];
#pragma unroll
        for (int m = 0; m < P; ++m) part[m] = dead ? 0.0 : __builtin_bit_cast(double, word(m, 0));
#pragma unroll
        for (int w = 1; w < P; w <<= 1)
#pragma unroll
          for (int m = 0; m < P; m += 2 * w) part[m] += part[m + w];
        if (lane == 0) {
          const unsigned long long bits = __builtin_bit_cast(unsigned long long, part[0]);
          Rec5* rec = bring + (r1 % SLOTS) * 4;
          __hip_atomic_store(&rec->v, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(&rec->chk, bits ^ want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      const long long r = seq - DEPTH;
      if (r >= 0) {                                 // every member: the sum of row seq - 3
        const unsigned long long want = d.seq0 + (unsigned long long)r + 1;
        auto got = [&]() { return ((((unsigned long long)bq[1] << 32) | bq[0]) ^ (((unsigned long long)bq[3] << 32) | bq[2])) == want; };
        if (!got() && !dead) {
          const long long t0 = wall_clock64();
          for (;;) {
            fetch1(r);
            if (got()) break;
            if (wall_clock64() - t0 > 2000000) {
              dead = true;
              if (lane == 0) atomicCAS(d.err, 0, 1);
              break;
            }
            __builtin_amdgcn_s_sleep(1);
          }
        }
        const double ax = dead ? 0.0 : __builtin_bit_cast(double, ((unsigned long long)bq[1] << 32) | bq[0]);
        keepc += ax;
        if (lane == 0) res_s[r & 3] = ax * 1e-300 + 1.0;
      }
      __syncthreads();
      double own = lane < CW ? wsum[seq & 1][lane] : 0.0;
#pragma unroll
      for (int off = 1; off < 16; off <<= 1) own += __shfl_xor(own, off, 64);
      keepc += own;
      if (lane == 0) {
        const unsigned long long bits = __builtin_bit_cast(unsigned long long, own);
        Rec5* rec = ring + (seq % SLOTS) * P + s;
        __hip_atomic_store(&rec->v, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&rec->chk, bits ^ (d.seq0 + (unsigned long long)seq + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (s == 0 && seq - 1 >= 0) fetch16(seq - 1);        // (published a step ago)
      if (seq - 2 >= 0) fetch1(seq - 2);                   // (the leader stored it before this step's barrier)
    }
    if (keepc == 12345.678) a.out[1] = keepc;
    return;
  }
  const int64_t cbase = (int64_t)s * a.nvs * 16;
  const int slice_bytes = (int)max((int64_t)0, min((int64_t)a.nvs * 16, a.ld * 8 - cbase));
  vec_t v[G][KV];
  double acc[KV][2];
#pragma unroll
  for (int k = 0; k < KV; ++k) {
    acc[k][0] = acc[k][1] = 0.0;
    xs[(tid + k * BSC) * 2] = 0.5, xs[(tid + k * BSC) * 2 + 1] = 0.25;
  }
  __syncthreads();
  auto load = [&](int g, int row) {
    const int ri = __builtin_amdgcn_readfirstlane(max(min(row, row_end - 1), 0));
    const __amdgpu_buffer_rsrc_t rs =
        __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)(a.A + (int64_t)ri * a.ld) + cbase), 0, slice_bytes, 0x00020000);
#pragma unroll
    for (int k = 0; k < KV; ++k)
      v[g][k] = __builtin_bit_cast(vec_t, __builtin_amdgcn_raw_buffer_load_b128(rs, tid * 16, k * BSC * 16, 2));
  };
  auto dots = [&](int g, long long seq) {
    double p = 0.0;
#pragma unroll
    for (int k = 0; k < KV; ++k)
#pragma unroll
      for (int e = 0; e < 2; ++e) p = fma(v[g][k][e], xs[(tid + k * BSC) * 2 + e], p);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) p += __shfl_xor(p, off, 64);
    wsum[seq & 1][wv] = p;
    __syncthreads();
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
      const double rr = res_s[r & 3];
#pragma unroll
      for (int k = 0; k < KV; ++k)
#pragma unroll
        for (int e = 0; e < 2; ++e) acc[k][e] = fma(v[u][k][e], rr, acc[k][e]);
#pragma unroll
      for (int k = 0; k < KV; ++k)
#pragma unroll
        for (int e = 0; e < 2; ++e) asm volatile("" : "+v"(acc[k][e]));
      __builtin_amdgcn_sched_barrier(0);
      load(u, row_begin + (int)(r + G));
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  double t = 0.0;
#pragma unroll
  for (int k = 0; k < KV; ++k) t += acc[k][0] + acc[k][1];
  if (t == 12345.678) a.out[0] = t;
}

static Rec5* g_ring = nullptr;
static int* g_err = nullptr;
static unsigned* g_tickets = nullptr;
static unsigned g_launch = 0;
static unsigned long long g_seq0 = 0;
template <int KV, int G, int DEPTH, int ST, int LD, bool WAIT, int P = 16>
static void l_pat5(const Args& a, int nb, hipStream_t q) {
  Args5 d{a, g_ring, g_seq0, g_err, g_tickets, g_launch++};
  g_seq0 += (unsigned long long)a.rows_per_slab + 64;
  hipLaunchKernelGGL((k_pat5<KV, G, DEPTH, ST, LD, WAIT, P>), dim3(nb), dim3(768), 0, q, d);
}

template <int KV, int G, int DEPTH>
static void l_pat6(const Args& a, int nb, hipStream_t q) {
  Args5 d{a, g_ring, g_seq0, g_err, g_tickets, g_launch++};
  g_seq0 += (unsigned long long)a.rows_per_slab + 64;
  hipLaunchKernelGGL((k_pat6<KV, G, DEPTH>), dim3(nb), dim3(768), 0, q, d);
}

// Teams of TWO (csrc/slab_pd.hip k_teampd_rows, BASELINE config C5: rows of 20 000): every wave takes the partner's record
// itself on the scalar path, one barrier per row.
//   EXTRA: the row's y, A_x_prev and shift ride behind its loads of A (three broadcast vector loads per wave and row)
//   HS 0: no hand-off   1: the store and every wave's scalar load, nothing waited for   2: the full hand-off
//      3: the full hand-off, but ONLY wave 0 loads the record (and leaves the sum in LDS: a second barrier per row)
template <int KV, int G, bool EXTRA, int HS, bool FIRST = false, bool ASMST = false>
__global__ __launch_bounds__(768) void k_pat2(Args5 d) {
  constexpr int BS = 768, NW = 12, SLOTS = 8;
  typedef unsigned raw4_t __attribute__((ext_vector_type(4)));
  typedef unsigned long long rec_t __attribute__((ext_vector_type(2)));
  const Args& a = d.a;
  extern __shared__ double xs2[];
  __shared__ double wsum[2][NW];
  __shared__ double other_s[2];
  __shared__ int ident[2];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (tid == 0) {
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u;
    const unsigned k = atomicAdd(d.tickets + xcc, 1u) - d.launch * 32u;
    ident[0] = k < 32u ? (int)(xcc * 16 + k / 2) : -1;
    ident[1] = (int)(k % 2);
    if (k >= 32u) atomicCAS(d.err, 0, 2);
  }
  __syncthreads();
  const int team = __builtin_amdgcn_readfirstlane(ident[0]), s = __builtin_amdgcn_readfirstlane(ident[1]);
  if (team < 0) return;
  const int row_begin = team * a.rows_per_slab;
  const int row_end = min(row_begin + a.rows_per_slab, a.m);
  const int nrows = max(row_end - row_begin, 0);
  Rec5* ring = d.ring + (int64_t)team * SLOTS * 2;
  const int64_t cbase = (int64_t)s * a.nvs * 16;
  const int slice_bytes = (int)max((int64_t)0, min((int64_t)a.nvs * 16, a.ld * 8 - cbase));
  vec_t v[G][KV];
  double y_in[G], axp_in[G], b_in[G];
  double acc[KV][2];
#pragma unroll
  for (int k = 0; k < KV; ++k) {
    acc[k][0] = acc[k][1] = 0.0;
    xs2[(tid + k * BS) * 2] = 0.5, xs2[(tid + k * BS) * 2 + 1] = 0.25;
  }
  const double* __restrict__ Y = a.out + 1024;         // (three vectors of m entries behind the out word)
  const double* __restrict__ AXP = Y + a.m;
  const double* __restrict__ SH = AXP + a.m;
  auto load = [&](int g, int row) {
    const int ri = __builtin_amdgcn_readfirstlane(max(min(row, row_end - 1), 0));
    const __amdgpu_buffer_rsrc_t rs =
        __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)(a.A + (int64_t)ri * a.ld) + cbase), 0, slice_bytes, 0x00020000);
#pragma unroll
    for (int k = 0; k < KV; ++k)
      v[g][k] = __builtin_bit_cast(vec_t, __builtin_amdgcn_raw_buffer_load_b128(rs, tid * 16, k * BS * 16, 2));
    if (EXTRA) {
      y_in[g] = Y[ri];
      axp_in[g] = AXP[ri];
      b_in[g] = SH[ri];
    } else {
      y_in[g] = axp_in[g] = b_in[g] = 0.0;
    }
  };
  auto dots = [&](int g, long long seq) -> double {
    double p = 0.0;
#pragma unroll
    for (int k = 0; k < KV; ++k)
#pragma unroll
      for (int e = 0; e < 2; ++e) p = fma(v[g][k][e], xs2[(tid + k * BS) * 2 + e], p);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) p += __shfl_xor(p, off, 64);
    wsum[seq & 1][wv] = p;
    __syncthreads();
    double own = wsum[seq & 1][0];
#pragma unroll
    for (int w = 1; w < NW; ++w) own += wsum[seq & 1][w];
    if (HS != 0 && tid == 0) {
      const unsigned long long bits = __builtin_bit_cast(unsigned long long, own);
      const rec_t rec = {bits, bits ^ (d.seq0 + (unsigned long long)seq + 1)};
      // (a volatile store is followed by `s_waitcnt vmcnt(0)`: wave 0 then waits for the store's acknowledgement before it goes
      //  on to finish the previous row and to re-load its slot — ASMST: the same store, fire and forget)
      if (ASMST) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(ring + (seq % SLOTS) * 2 + s), "v"(rec) : "memory");
      else *(volatile rec_t*)(ring + (seq % SLOTS) * 2 + s) = rec;
    }
    return own;
  };
  bool dead = false;
  double keep = 0.0;
  auto finish = [&](int g, long long seq, double own, raw4_t early) {
    double other = 0.0;
    if (HS == 1 || HS == 2) {
      const unsigned long long want = d.seq0 + (unsigned long long)seq + 1;
      asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(early)::"memory");
      unsigned long long vv = ((unsigned long long)early[1] << 32) | early[0];
      unsigned long long chk = ((unsigned long long)early[3] << 32) | early[2];
      if (HS == 2 && (vv ^ chk) != want && !dead) {
        const Rec5* src = uniform_ptr(ring + (seq % SLOTS) * 2 + (1 - s));
        const long long t0 = wall_clock64();
        for (;;) {
          raw4_t raw;
          asm volatile("s_load_dwordx4 %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(raw) : "s"(src) : "memory");
          vv = ((unsigned long long)raw[1] << 32) | raw[0];
          chk = ((unsigned long long)raw[3] << 32) | raw[2];
          if ((vv ^ chk) == want) break;
          if (wall_clock64() - t0 > 2000000) {
            dead = true;
            vv = 0;
            if (lane == 0) atomicCAS(d.err, 0, 1);
            break;
          }
          __builtin_amdgcn_s_sleep(1);
        }
      }
      other = __builtin_bit_cast(double, vv);
    } else if (HS == 3) {
      if (wv == 0) {
        const unsigned long long want = d.seq0 + (unsigned long long)seq + 1;
        const Rec5* src = uniform_ptr(ring + (seq % SLOTS) * 2 + (1 - s));
        unsigned long long vv = 0, chk = 0;
        const long long t0 = wall_clock64();
        for (;;) {
          raw4_t raw;
          asm volatile("s_load_dwordx4 %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(raw) : "s"(src) : "memory");
          vv = ((unsigned long long)raw[1] << 32) | raw[0];
          chk = ((unsigned long long)raw[3] << 32) | raw[2];
          if ((vv ^ chk) == want || dead) break;
          if (wall_clock64() - t0 > 2000000) {
            dead = true;
            vv = 0;
            if (lane == 0) atomicCAS(d.err, 0, 1);
            break;
          }
          __builtin_amdgcn_s_sleep(1);
        }
        if (lane == 0) other_s[seq & 1] = __builtin_bit_cast(double, vv);
      }
      __syncthreads();
      other = other_s[seq & 1];
    }
    const double ax = s == 0 ? own + other : other + own;
    const double w = y_in[g] + 0.5 * (1.5 * ax - 0.5 * axp_in[g]);
    const double yn = fmin(fmax(w - 0.5 * b_in[g], -1.0), 1.0) * 1e-30 + 1.0;
    keep += yn;
#pragma unroll
    for (int k = 0; k < KV; ++k)
#pragma unroll
      for (int e = 0; e < 2; ++e) acc[k][e] = fma(v[g][k][e], yn, acc[k][e]);
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
  double own_prev = dots(0, 0);
  __builtin_amdgcn_sched_barrier(0);
  load(G - 1, row_begin + G - 1);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
  for (long long q = 1; q <= nrows; q += G) {
#pragma unroll
    for (int u = 0; u < G; ++u) {
      const long long seq = q + u;
      raw4_t early = {};
      if (HS == 1 || HS == 2) {
        const Rec5* src = uniform_ptr(ring + ((seq - 1) % SLOTS) * 2 + (1 - s));
        asm volatile("s_load_dwordx4 %0, %1, 0x0 glc" : "=s"(early) : "s"(src) : "memory");
      }
      if (FIRST) {
        // FIRST: row seq - 1 is finished and its slot re-loaded BEFORE the dots of row seq — the load of row seq + 1 then has a
        // whole step to arrive (two slots, yet a row always in flight) ; the hand-off is on the critical path instead
        finish(u % G, seq - 1, own_prev, early);
        __builtin_amdgcn_sched_barrier(0);
        load(u % G, row_begin + (int)(seq - 1) + G);
        __builtin_amdgcn_sched_barrier(0);
        own_prev = dots((1 + u) % G, seq);
        __builtin_amdgcn_sched_barrier(0);
      } else {
        const double own = dots((1 + u) % G, seq);
        __builtin_amdgcn_sched_barrier(0);
        finish(u % G, seq - 1, own_prev, early);
        own_prev = own;
        __builtin_amdgcn_sched_barrier(0);
        load(u % G, row_begin + (int)(seq - 1) + G);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  double t = keep;
#pragma unroll
  for (int k = 0; k < KV; ++k) t += acc[k][0] + acc[k][1];
  if (t == 12345.678) a.out[0] = t;
}

template <int KV, int G, bool EXTRA, int HS, bool FIRST = false, bool ASMST = false>
static void l_pat2(const Args& a, int nb, hipStream_t q) {
  static bool once = false;
  const int lds = KV * 768 * 16;
  if (!once) {
    CK(hipFuncSetAttribute((const void*)k_pat2<KV, G, EXTRA, HS, FIRST, ASMST>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    once = true;
  }
  Args5 d{a, g_ring, g_seq0, g_err, g_tickets, g_launch++};
  g_seq0 += (unsigned long long)a.rows_per_slab + 64;
  hipLaunchKernelGGL((k_pat2<KV, G, EXTRA, HS, FIRST, ASMST>), dim3(nb), dim3(768), lds, q, d);
}

// contiguous stream of the same bytes, the shape of the two-pass kernels: block b walks a contiguous chunk
template <int BS, int U, bool NT>
__global__ __launch_bounds__(BS) void k_stream(const double* __restrict__ A, int64_t nvec_total, int64_t vec_per_block, double* out) {
  const int64_t v0 = (int64_t)blockIdx.x * vec_per_block;
  const int64_t v1 = min(v0 + vec_per_block, nvec_total);
  double acc0 = 0.0, acc1 = 0.0;
  const vec_t* p = (const vec_t*)A;
  int64_t i = v0 + threadIdx.x;
  for (; i + (int64_t)(U - 1) * BS < v1; i += (int64_t)U * BS) {
    vec_t x[U];
#pragma unroll
    for (int u = 0; u < U; ++u) x[u] = NT ? __builtin_nontemporal_load(p + i + (int64_t)u * BS) : p[i + (int64_t)u * BS];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      acc0 += x[u][0];
      acc1 += x[u][1];
    }
  }
  for (; i < v1; i += BS) {
    const vec_t x = p[i];
    acc0 += x[0];
    acc1 += x[1];
  }
  if (acc0 + acc1 == 12345.678) out[0] = acc0;
}

// mode 0: ten significant bits per entry ; 1: uniform in (-1, 1) with all 52 bits of the mantissa random (what a generated
// design matrix looks like to the data buses) ; 2: zeros
__global__ void k_fill(double* A, int64_t n, int mode) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    unsigned long long z = (unsigned long long)i + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    A[i] = mode == 0 ? (double)((i * 2654435761u) & 1023) * (1.0 / 1024.0)
                     : mode == 1 ? ((double)(z >> 11) * (1.0 / 9007199254740992.0)) * 2.0 - 1.0 : 0.0;
  }
}

struct Case {
  const char* name;
  void (*launch)(const Args&, int nblocks, hipStream_t);
  int bs, kv, wg_per_cu;      // the geometry: PS = ceil(nvec / (bs * kv)) ; slabs so that ~wg_per_cu * 256 blocks exist
  int ps = 0;                 // (given: that many slices, of ceil(nvec / ps) vectors, the rest of a thread's loads clamped away)
};

template <int BS, int KV, int G, bool NT, int S = BS, bool XL = false, int STAGE = 0>
static void l_pat(const Args& a, int nb, hipStream_t q) {
  hipLaunchKernelGGL((k_pat<BS, KV, G, NT, S, XL, STAGE>), dim3(nb), dim3(BS), 0, q, a);
}

int main(int argc, char** argv) {
  const int m = argc > 1 ? atoi(argv[1]) : 50000;
  const int n = argc > 2 ? atoi(argv[2]) : 100000;
  const int64_t ld = ((int64_t)n + 15) / 16 * 16;
  const int64_t total = (int64_t)m * ld;
  double *A, *out;
  CK(hipMalloc(&A, total * 8));
  CK(hipMalloc(&out, (1024 + 3 * (size_t)m) * 8));
  CK(hipMemset(out, 0, (1024 + 3 * (size_t)m) * 8));
  const int fill_mode = argc > 3 ? atoi(argv[3]) : 0;
  const int ncases = argc > 4 ? atoi(argv[4]) : 1000;
  hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, A, total, fill_mode);
  printf("fill mode %d\n", fill_mode);
  CK(hipDeviceSynchronize());
  CK(hipMalloc(&g_ring, sizeof(Rec5) * (8 * 256 + 64 * 8 * 4)));
  CK(hipMemset(g_ring, 0, sizeof(Rec5) * (8 * 256 + 64 * 8 * 4)));
  CK(hipMalloc(&g_err, 4));
  CK(hipMemset(g_err, 0, 4));
  CK(hipMalloc(&g_tickets, 32));
  CK(hipMemset(g_tickets, 0, 32));
  const double gb = (double)m * n * 8 / 1e9;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int nvec = (n + 1) / 2;
  const std::vector<Case> cases = {
      {"n = 90000: teams of EIGHT, KV8 G3 D1 (one step of slack)", l_pat5<8, 3, 1, 0, 2, true, 8>, 704, 8, 1, 8},
      {"n = 90000: teams of SIXTEEN, KV4 G4 D2 (product)", l_pat5<4, 4, 2, 0, 2, true, 16>, 704, 4, 1, 16},
      {"n = 90000: teams of EIGHT, no hand-off traffic", l_pat5<8, 3, 1, 2, 3, false, 8>, 704, 8, 1, 8},
      {"n = 180000: teams of SIXTEEN, KV8 G3 D1 (one step of slack)", l_pat5<8, 3, 1, 0, 2, true, 16>, 704, 8, 1, 16},
      {"n = 180000: teams of SIXTEEN, KV8 G3 D1, no hand-off traffic", l_pat5<8, 3, 1, 2, 3, false, 16>, 704, 8, 1, 16},
      {"contiguous stream reference     ", nullptr, 0, 0, 0, -1},
      {"n = 40000: teams of FOUR, KV8 G3 D1 (one step of slack)", l_pat5<8, 3, 1, 0, 2, true, 4>, 704, 8, 1, 4},
      {"n = 40000: teams of FOUR, KV8 G3 D1, no hand-off traffic", l_pat5<8, 3, 1, 2, 3, false, 4>, 704, 8, 1, 4},
      {"contiguous stream reference    ", nullptr, 0, 0, 0, -1},
      {"teams of TWO with a communication wave, KV8 G3 D1 (C5)", l_pat5<8, 3, 1, 0, 2, true, 2>, 704, 8, 1, 2},
      {"   the same, no hand-off traffic", l_pat5<8, 3, 1, 2, 3, false, 2>, 704, 8, 1, 2},
      {"teams of two, every wave on the scalar path, KV7 G2 (product)", l_pat2<7, 2, true, 2>, 768, 7, 1, 2},
      {"contiguous stream reference   ", nullptr, 0, 0, 0, -1},
      {"wide teams, scalar path, G4 D2 (product)", l_pat5<5, 4, 2, 0, 2, true>, 704, 5, 1, 16},
      {"wide teams, scalar path, G5 D3", l_pat5<5, 5, 3, 0, 2, true>, 704, 5, 1, 16},
      {"wide teams, LEADER tree (one sum record per row), G5 D3", l_pat6<5, 5, 3>, 704, 5, 1, 16},
      {"wide teams, no hand-off traffic", l_pat5<5, 4, 2, 2, 3, false>, 704, 5, 1, 16},
      {"contiguous stream reference  ", nullptr, 0, 0, 0, -1},
      {"teams of two KV7 G2 (product): volatile store of the record", l_pat2<7, 2, true, 2>, 768, 7, 1, 2},
      {"   the same store, fire and forget (no wait behind it)", l_pat2<7, 2, true, 2, false, true>, 768, 7, 1, 2},
      {"   no hand-off", l_pat2<7, 2, true, 0>, 768, 7, 1, 2},
      {"teams of two KV7 G2 (product order): dots, finish, load", l_pat2<7, 2, true, 2>, 768, 7, 1, 2},
      {"teams of two KV7 G2, finish and re-load BEFORE the dots", l_pat2<7, 2, true, 2, true>, 768, 7, 1, 2},
      {"   the same, no hand-off", l_pat2<7, 2, true, 0, true>, 768, 7, 1, 2},
      {"teams of two KV7 G3 no row state (spills with it)", l_pat2<7, 3, false, 2>, 768, 7, 1, 2},
      {"teams of FOUR, comm wave, KV4 G4 D2, scalar path (C5 shape)", l_pat5<4, 4, 2, 0, 2, true, 4>, 704, 4, 1, 4},
      {"teams of FOUR, comm wave, KV4 G5 D3, scalar path", l_pat5<4, 5, 3, 0, 2, true, 4>, 704, 4, 1, 4},
      {"teams of FOUR, comm wave, KV4 G4 D2, no hand-off", l_pat5<4, 4, 2, 2, 3, false, 4>, 704, 4, 1, 4},
      {"teams of two KV7 G2: no hand-off, no row state", l_pat2<7, 2, false, 0>, 768, 7, 1, 2},
      {"   + y, A_x_prev, shift behind the row", l_pat2<7, 2, true, 0>, 768, 7, 1, 2},
      {"   + store, every wave's scalar load, not waited for", l_pat2<7, 2, true, 1>, 768, 7, 1, 2},
      {"   + waited for (product)", l_pat2<7, 2, true, 2>, 768, 7, 1, 2},
      {"   wave 0 alone loads the record, second barrier", l_pat2<7, 2, true, 3>, 768, 7, 1, 2},
      {"teams of two KV7 G3: no hand-off, no row state", l_pat2<7, 3, false, 0>, 768, 7, 1, 2},
      {"   G3, full hand-off", l_pat2<7, 3, true, 2>, 768, 7, 1, 2},
      {"   G3, full hand-off, no row state", l_pat2<7, 3, false, 2>, 768, 7, 1, 2},
      {"   G2, full hand-off, no row state", l_pat2<7, 2, false, 2>, 768, 7, 1, 2},
      {"contiguous stream reference ", nullptr, 0, 0, 0, -1},
      {"hand-off: sc1 stores, sc1 loads, waited for (product)", l_pat5<5, 4, 2, 0, 0, true>, 704, 5, 1, 16},
      {"   sc1 stores, scalar glc loads, waited for, G4 D2", l_pat5<5, 4, 2, 0, 2, true>, 704, 5, 1, 16},
      {"   sc1 stores, scalar glc loads, waited for, G5 D3", l_pat5<5, 5, 3, 0, 2, true>, 704, 5, 1, 16},
      {"   sc1 stores, scalar glc loads, waited for, G5 D2", l_pat5<5, 5, 2, 0, 2, true>, 704, 5, 1, 16},
      {"   sc1 stores, scalar glc loads, waited for, G3 D1", l_pat5<5, 3, 1, 0, 2, true>, 704, 5, 1, 16},
      {"   sc1 stores, scalar glc loads, NOT waited for, G4 D2", l_pat5<5, 4, 2, 0, 2, false>, 704, 5, 1, 16},
      {"   plain stores, scalar glc loads, waited for, G4 D2", l_pat5<5, 4, 2, 1, 2, true>, 704, 5, 1, 16},
      {"   no hand-off traffic", l_pat5<5, 4, 2, 2, 3, false>, 704, 5, 1, 16},
      {"contiguous stream reference", nullptr, 0, 0, 0, -1},
      {"768 thr x KV5 G4, 16 slices: bare pattern", l_pat<768, 5, 4, true, 768, true, 0>, 768, 5, 1, 16},
      {"   + dot with x from LDS", l_pat<768, 5, 4, true, 768, true, 1>, 768, 5, 1, 16},
      {"   + wave butterfly", l_pat<768, 5, 4, true, 768, true, 2>, 768, 5, 1, 16},
      {"   + store of the wave sum, barrier per row", l_pat<768, 5, 4, true, 768, true, 3>, 768, 5, 1, 16},
      {"   + residual through LDS", l_pat<768, 5, 4, true, 768, true, 4>, 768, 5, 1, 16},
      {"704 thr x KV5 G4, 16 slices: bare pattern", l_pat<704, 5, 4, true, 704, true, 0>, 704, 5, 1, 16},
      {"   + dot with x from LDS", l_pat<704, 5, 4, true, 704, true, 1>, 704, 5, 1, 16},
      {"   + wave butterfly", l_pat<704, 5, 4, true, 704, true, 2>, 704, 5, 1, 16},
      {"   + store of the wave sum, barrier per row", l_pat<704, 5, 4, true, 704, true, 3>, 704, 5, 1, 16},
      {"   + residual through LDS", l_pat<704, 5, 4, true, 704, true, 4>, 704, 5, 1, 16},
      {"768 thr x KV5 G4, 16 slices, slab's slices on ONE XCD", l_pat<768, 5, 4, true, 768, true>, 768, 5, 1, 16},
      {"768 thr x KV5 G4, 16 slices, one XCD, per wave contiguous", l_pat<768, 5, 4, true, 64, true>, 768, 5, 1, 16},
      {"704 thr x KV5 G4, 16 slices, slab's slices on ONE XCD", l_pat<704, 5, 4, true, 704, true>, 704, 5, 1, 16},
      {"704 thr x KV5 G4, 16 slices, spread over the XCDs", l_pat<704, 5, 4, true>, 704, 5, 1, 16},
      {"768 thr x KV7 G3, 8 slices, one XCD", l_pat<768, 7, 3, true, 768, true>, 768, 7, 1, 8},
      {"768 thr x KV7 G3, 8 slices, spread", l_pat<768, 7, 3, true>, 768, 7, 1, 8},
      {"768 thr x KV5 G4, 32 slices of half the row... (n/a)", nullptr, 0, 0, 0, 0},
      // exactly 256 workgroups, sixteen slices of 3125 vectors (C2 by teams of sixteen), by the thread -> column mapping:
      {"768 thr x KV5 G4, 16 slices, k-th loads 12 KB apart (product)", l_pat<768, 5, 4, true>, 768, 5, 1, 16},
      {"768 thr x KV5 G4, 16 slices, sub-groups of 256 (4 KB apart)", l_pat<768, 5, 4, true, 256>, 768, 5, 1, 16},
      {"768 thr x KV5 G4, 16 slices, sub-groups of 128 (2 KB apart)", l_pat<768, 5, 4, true, 128>, 768, 5, 1, 16},
      {"768 thr x KV5 G4, 16 slices, per wave contiguous (1 KB apart)", l_pat<768, 5, 4, true, 64>, 768, 5, 1, 16},
      {"1024 thr x KV4 G4, 16 slices, 16 KB apart", l_pat<1024, 4, 4, true>, 1024, 4, 1, 16},
      {"1024 thr x KV4 G4, 16 slices, sub-groups of 256", l_pat<1024, 4, 4, true, 256>, 1024, 4, 1, 16},
      // the same 256 CUs x 50 KB per step as two / four workgroups per CU (32 / 64 slices)
      {"512 thr x KV4 G4, 32 slices, 2 wg/CU", l_pat<512, 4, 4, true>, 512, 4, 2, 32},
      {"512 thr x KV4 G4, 32 slices, 2 wg/CU, sub-groups of 256", l_pat<512, 4, 4, true, 256>, 512, 4, 2, 32},
      {"256 thr x KV4 G4, 64 slices, 4 wg/CU", l_pat<256, 4, 4, true>, 256, 4, 4, 64},
      {"256 thr x KV7 G3, 32 slices, 2 wg/CU", l_pat<256, 7, 3, true>, 256, 7, 2, 32},
      {"256 thr x KV7 G3, 32 slices, 4 wg/CU", l_pat<256, 7, 3, true>, 256, 7, 4, 32},
      // the product's geometry: 704 threads x 5 vectors = a sixteenth of C2's row, one workgroup per CU
      {"704 thr x KV5, G1, 1 wg/CU (the team kernel's slice)", l_pat<704, 5, 1, true>, 704, 5, 1},
      {"704 thr x KV5, G2", l_pat<704, 5, 2, true>, 704, 5, 1},
      {"704 thr x KV5, G4", l_pat<704, 5, 4, true>, 704, 5, 1},
      {"704 thr x KV5, G4, plain loads", l_pat<704, 5, 4, false>, 704, 5, 1},
      // the same bytes per CU and step, more and lighter waves: two / four workgroups per CU
      {"1024 thr x KV4, G4, 1 wg/CU", l_pat<1024, 4, 4, true>, 1024, 4, 1},
      {"512 thr x KV4, G4, 2 wg/CU (slices of 32 KB)", l_pat<512, 4, 4, true>, 512, 4, 2},
      {"768 thr x KV3, G4, 2 wg/CU", l_pat<768, 3, 4, true>, 768, 3, 2},
      {"1024 thr x KV2, G4, 2 wg/CU", l_pat<1024, 2, 4, true>, 1024, 2, 2},
      {"512 thr x KV2, G4, 4 wg/CU (slices of 16 KB)", l_pat<512, 2, 4, true>, 512, 2, 4},
      {"256 thr x KV4, G4, 8 wg/CU (slices of 16 KB)", l_pat<256, 4, 4, true>, 256, 4, 8},
      {"256 thr x KV4, G2, 8 wg/CU", l_pat<256, 4, 2, true>, 256, 4, 8},
      {"256 thr x KV2, G4, 8 wg/CU (slices of 8 KB)", l_pat<256, 2, 4, true>, 256, 2, 8},
      {"256 thr x KV8, G2, 4 wg/CU (slices of 32 KB)", l_pat<256, 8, 2, true>, 256, 8, 4},
      // whole rows per workgroup are not possible at this length; WIDE slices: 1024 x KV7 = 112 KB (teams of 7)
      {"1024 thr x KV7, G3, 1 wg/CU (slices of 112 KB)", l_pat<1024, 7, 3, true>, 1024, 7, 1},
      {"768 thr x KV10, G3, 1 wg/CU (slices of 120 KB)", l_pat<768, 10, 3, true>, 768, 10, 1},
  };
  auto stream_ref = [&]() {
    const int64_t nv = total / 2;
    const int nb = 256 * 8;
    const int64_t vpb = (nv + nb - 1) / nb;
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k_stream<256, 8, true>), dim3(nb), dim3(256), 0, 0, A, nv, vpb, out);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= 5;
    printf("%-58s %39s %7.3f ms  %6.0f GB/s\n", "contiguous stream, 256 thr x 8 loads, 8 wg/CU", "", ms, (double)total * 8 / 1e9 / (ms * 1e-3));
  };
  const int nrounds = argc > 5 ? atoi(argv[5]) : 2;
  for (int rep = 0; rep < nrounds; ++rep) {
    int done_cases = 0;
    for (const Case& c : cases) {
      if (c.launch == nullptr && c.ps != -1) continue;
      if (done_cases++ >= ncases) break;
      if (c.ps == -1) {
        stream_ref();
        continue;
      }
      Args a{};
      a.A = A;
      a.ld = ld;
      a.m = m;
      a.nvec = nvec;
      a.nvs = c.bs * c.kv;
      a.PS = (nvec + a.nvs - 1) / a.nvs;
      if (c.ps > 0) {
        a.PS = c.ps;
        a.nvs = (nvec + c.ps - 1) / c.ps;
      }
      int slabs = c.ps > 0 ? 256 * c.wg_per_cu / a.PS : (256 * c.wg_per_cu + a.PS - 1) / a.PS;
      a.rows_per_slab = (m + slabs - 1) / slabs;
      slabs = (m + a.rows_per_slab - 1) / a.rows_per_slab;
      a.out = out;
      const int nb = slabs * a.PS;
      c.launch(a, nb, 0);
      CK(hipDeviceSynchronize());
      const int reps = 5;
      CK(hipEventRecord(e0, 0));
      for (int i = 0; i < reps; ++i) c.launch(a, nb, 0);
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms = 0;
      CK(hipEventElapsedTime(&ms, e0, e1));
      ms /= reps;
      int herr = 0;
      CK(hipMemcpy(&herr, g_err, 4, hipMemcpyDeviceToHost));
      CK(hipMemset(g_err, 0, 4));
      printf("%-58s PS %3d slabs %4d blocks %5d  %7.3f ms  %6.0f GB/s%s\n", c.name, a.PS, slabs, nb, ms, gb / (ms * 1e-3), herr ? "  TIMED OUT" : "");
      fflush(stdout);
    }
    // contiguous streams
    if (ncases >= 1000) {
      const int64_t nv = total / 2;
      const int nb = 256 * 8;
      const int64_t vpb = (nv + nb - 1) / nb;
      hipLaunchKernelGGL((k_stream<256, 8, true>), dim3(nb), dim3(256), 0, 0, A, nv, vpb, out);
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0, 0));
      for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k_stream<256, 8, true>), dim3(nb), dim3(256), 0, 0, A, nv, vpb, out);
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms = 0;
      CK(hipEventElapsedTime(&ms, e0, e1));
      ms /= 5;
      printf("%-58s %7.3f ms  %6.0f GB/s\n", "contiguous stream, 256 thr x 8 loads, 8 wg/CU", ms, (double)total * 8 / 1e9 / (ms * 1e-3));
      const int nb2 = 256;
      const int64_t vpb2 = (nv + nb2 - 1) / nb2;
      hipLaunchKernelGGL((k_stream<768, 16, true>), dim3(nb2), dim3(768), 0, 0, A, nv, vpb2, out);
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0, 0));
      for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k_stream<768, 16, true>), dim3(nb2), dim3(768), 0, 0, A, nv, vpb2, out);
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms, e0, e1));
      ms /= 5;
      printf("%-58s %7.3f ms  %6.0f GB/s\n", "contiguous stream, 768 thr x 16 loads, 1 wg/CU", ms, (double)total * 8 / 1e9 / (ms * 1e-3));
    }
  }
  int herr = 0;
  CK(hipMemcpy(&herr, g_err, 4, hipMemcpyDeviceToHost));
  printf("hand-off error word %d\n", herr);
  return 0;
}
