// team_rows.h — k_team_rows: f and grad f (csrc/slab.hip), or A x, the dual step and A' y of the primal-dual loop
// (csrc/slab_pd.hip), from ONE read of A for Float64 rows of up to 180 224 entries — one workgroup, or teams of workgroups.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <string>

#include "skinny_dev.h"
#include "solver_internal.h"
#include "team_dev.h"
#include "vecops_dev.h"

namespace apx {

struct SlabDev;

// ---- what the communication wave does with a row ---------------------------------------------------------------------
constexpr int kTeamFkPd = 100;        // FK of the primal-dual loop: A x, the dual step, y+ * A[i, :]  (csrc/slab_pd.hip)
struct TeamRowState {                 // b[i]  /  y[i], A_x_prev[i], shift[i]
  double a = 0.0, b = 0.0, c = 0.0;
};
template <int FK>
struct TeamRowConsts {
  double bias = 0.0, sigma = 0.0, rho = 0.0;
  template <class DEV>
  __device__ TeamRowConsts(const DEV& d, const void* xv, bool f32 = false) {
    if constexpr (FK == APX_F_LOGISTIC) bias = f32 ? (double)((const float*)xv)[d.n - 1] : ((const double*)xv)[d.n - 1];
    if constexpr (FK == kTeamFkPd) {
      sigma = d.trial ? d.sigma_arg : d.c->sigma;      // this iteration's, set by the rule kernel before this launch — or a
      rho = d.trial ? d.rho_arg : d.c->rho;            // linesearch trial's, from the host (csrc/slab_pd.hip slab_pd_trial)
    }
  }
};
template <int FK, class DEV, typename T = double>
__device__ __forceinline__ TeamRowState team_row_state(const DEV& d, int i) {
  TeamRowState s;
  if constexpr (FK == kTeamFkPd) {
    s.a = (double)((const T*)d.Y)[i];
    s.b = (double)((const T*)d.AXP)[i];
    s.c = d.h.has_shift ? (double)((const T*)d.h.shift)[i] : 0.0;
  } else {
    s.a = (double)((const T*)d.b)[i];
  }
  return s;
}
template <int FK>
__device__ __forceinline__ void team_row_state_landed(TeamRowState& s) {
  if constexpr (FK == kTeamFkPd) asm volatile("" : "+v"(s.a), "+v"(s.b), "+v"(s.c));      // (a use: the compiler waits for the loads)
  else asm volatile("" : "+v"(s.a));
}
// lane l's state, on every lane (l uniform)
__device__ __forceinline__ double team_lane_value(double v, int l) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}
template <int FK>
__device__ __forceinline__ TeamRowState team_row_state_of_lane(const TeamRowState& s, int l) {
  TeamRowState o;
  o.a = team_lane_value(s.a, l);
  if constexpr (FK == kTeamFkPd) {
    o.b = team_lane_value(s.b, l);
    o.c = team_lane_value(s.c, l);
  }
  return o;
}
// The dual step of row i of the primal-dual loop in two parts: y+[i] — what the compute waves wait for — and the row's terms of
// the residual norm and of h(A x), which nothing waits for (a division and the value of h: k_team_rows forms them BEHIND the
// publication of its next partial, off the path between two barriers).
//
// h = NormL2 (the square-root lasso's l2 ball, experiments/square_root_lasso/runme.jl:39-41) is not separable — its conjugate prox
// scales by a factor that depends on the norm of ALL rows' z[i] = w[i] / sigma + shift[i] (csrc/vecops_dev.h prox_core:
// sc = max(0, 1 - lambda / (sigma ||z||))) — but it is LINEAR in z once that factor is known: y+ = sigma (1 - sc) z, hence
// A' y+ = sigma (1 - sc) A' z.  So one pass over A does for it too (round 4): the row's multiplier is z[i], the launch leaves
// A' z and sum z^2, the closing kernel forms the factor and scales ; w[i] waits in y[i]'s place for the small kernel behind it
// that writes y+[i] and the dual residual in the two passes' own arithmetic (csrc/vecops.hip k_dual_post).
template <class DEV>
__device__ __forceinline__ bool team_pd_l2(const DEV& d) { return d.h.kind == APX_G_L2; }
template <class DEV, typename T>
__device__ __forceinline__ double team_pd_dual(const DEV& d, const TeamRowConsts<kTeamFkPd>& kc, const TeamRowState& st, double dot, int i, double& w) {
  const double ax = (double)(T)dot;
  w = st.a + kc.sigma * ((1.0 + kc.rho) * ax - kc.rho * st.b);                     // src/AdaProx.jl:344
  if (team_pd_l2(d)) return w / kc.sigma + st.c;                                   // z[i] (k_dual_pre's expression)
  return (double)(T)prox_conj_elem_s<T>(d.h, w, kc.sigma, i, 0.0, st.c);           // :345
}
template <class DEV, typename T>
__device__ __forceinline__ void team_pd_sums(const DEV& d, const TeamRowConsts<kTeamFkPd>& kc, double shift, double ax, double w, double yn, int i,
                                             double& e0, double& e1) {
  const double dr = team_pd_l2(d) ? yn : (w - yn) / kc.sigma - ax;                 // :347 (l2: sum z^2 — the residual comes later)
  e0 += dr * dr;
  e1 += gval_elem_s<T>(d.h, ax, i, shift);
}
// what goes to y[i] behind the row: y+[i], or (l2) w[i], from which k_dual_post forms y+[i] once the norm is known
template <class DEV>
__device__ __forceinline__ double team_pd_stored(const DEV& d, double w, double yn) { return team_pd_l2(d) ? w : yn; }
// the value the compute waves multiply row i by ; the loss sums (every member forms them — member 0's are the ones stored).
// `dot` is the row's <A[i, :], x> in Float64 ; T is the storage type: A x is rounded to T like every stored A x, and for Float32
// data the logistic link runs in Float32, as Julia does on Float32 arrays (csrc/slab.hip slab_group) — for T = double nothing
// is rounded.
template <int FK, class DEV, typename T = double>
__device__ __forceinline__ double team_row_finish(const DEV& d, const TeamRowConsts<FK>& kc, const TeamRowState& st, double dot, int i,
                                                  bool writer, double& e0, double& e1) {
  const double ax = (double)(T)dot;
  if constexpr (FK == APX_F_LEAST_SQUARES) {
    const double res = (double)(T)(ax - st.a);                        // experiments/lasso/runme.jl:22
    e0 += res * res;
    return res;
  } else if constexpr (FK == APX_F_LOGISTIC) {
    const double logit = ax + kc.bias;                                // experiments/sparse_logreg/runme.jl:24-36
    double res;
    if constexpr (sizeof(T) == 4) {
      const float lf = (float)logit;
      const float ef = 1.0f + expf(-lf);
      e0 += (double)(((float)st.a - 1.0f) * lf - logf(ef));
      res = (double)(1.0f / ef - (float)st.a);
    } else {
      const double e = 1.0 + exp(-logit);
      e0 += (st.a - 1.0) * logit - log(e);
      res = 1.0 / e - st.a;
    }
    e1 += res;
    return res;
  } else {
    double w;
    const double yn = team_pd_dual<DEV, T>(d, kc, st, dot, i, w);
    const T ynt = (T)team_pd_stored(d, w, yn);
    team_pd_sums<DEV, T>(d, kc, st.c, ax, w, yn, i, e0, e1);
    if (writer) {
      ((T*)d.YN)[i] = ynt;
      ((T*)d.AXN)[i] = (T)ax;                                                      // :360
    }
    return yn;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Rows too long for one workgroup's registers (10 753 ... 180 224 Float64 entries — BASELINE config C2: 100 000, a row is
// 800 KB): TEAMS of P = 2 / 4 / 8 / 16 workgroups, one per CU, formed inside an XCD at run time, share chunks of rows.  Member
// s keeps its share of the columns — the s-th contiguous P-th of the row, or (teams of four or more) the row's 10 KB pieces s,
// s + P, s + 2 P, ... — of x (in LDS) and of the column accumulator (in registers) and streams that share of every row.  A
// workgroup is ten or eleven COMPUTE waves and one COMMUNICATION wave:
//   * the compute waves form the partial dot product of a row (one barrier per row) and, DEPTH = 2 rows later, take the
//     row's residual from LDS and accumulate res[i] * A[i, slice] from the registers the row still sits in;
//   * the communication wave loads no A, so its vector-memory counter is its own (loads return in order per wave: a
//     poll or a fence in a wave that streams A waits for every load in flight and drains the pipeline — that is what
//     round 1's probe measured as 23 - 28 us per hand-off).  It sums the workgroup's partial, publishes it (two words
//     {value, value ^ sequence}, agent scope, no fence: the check word validates the pair), collects the P records of
//     the row two rows back on the SCALAR path (P / 4 `s_load_dwordx16 ... glc`, from the XCD's L2, a step ahead), adds
//     them in a fixed tree — the same `A x` on every member, no atomics on data —, evaluates the residual / the logistic
//     link ONCE and leaves it in LDS; member 0's accumulates the loss sums.  Two rows of slack absorb the hand-off
//     (1 - 3 us) and the slowest member.
// Every wait is bounded (TeamDev::wait_ticks of wall clock — one second by default — AND 4096 polls) and ends in an error
// word the host reads: the grid always drains.  The word is STICKY: a launch that finds it set returns at once, a wave that
// finds it set while it polls stops waiting — one failure costs one bound, not one per launch — and the wave that set it
// leaves a TeamDiag record (csrc/team_dev.h) from which the host's message is built.
// tools/split_probe.hip, tools/pattern_probe.hip (profiles/r02_team_probe_*.log, r02_team_handoff_traffic.log): 6.6 - 6.7
// TB/s on rows of C2's length — A read once per evaluation of f and its gradient instead of twice at 7.1 TB/s.
// ---------------------------------------------------------------------------------------------------------------------
// The P records of one row of a team's ring (P x 16 bytes, aligned to their size up to 64) into SGPRs: P / 4 scalar loads
// of 64 bytes (teams of two: one of 32), glc (past
// the scalar cache), waited for here — nothing may copy the registers before the data is in them.
typedef unsigned raw16_t __attribute__((ext_vector_type(16)));
template <int P>
__device__ __forceinline__ void team_fetch(const TeamRec* row, raw16_t (&q)[(P + 3) / 4]) {
  static_assert(P == 2 || P == 4 || P == 8 || P == 16, "teams of 2, 4, 8 or 16");
  if constexpr (P == 2) {
    typedef unsigned raw8_t __attribute__((ext_vector_type(8)));
    raw8_t h;
    asm volatile("s_load_dwordx8 %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(h) : "s"(row) : "memory");
#pragma unroll
    for (int c = 0; c < 8; ++c) q[0][c] = h[c];
  } else if constexpr (P == 4)
    asm volatile("s_load_dwordx16 %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(q[0]) : "s"(row) : "memory");
  else if constexpr (P == 8)
    asm volatile("s_load_dwordx16 %0, %2, 0x0 glc\n\ts_load_dwordx16 %1, %2, 0x40 glc\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(q[0]), "=&s"(q[1])
                 : "s"(row)
                 : "memory");
  else
    asm volatile("s_load_dwordx16 %0, %4, 0x0 glc\n\ts_load_dwordx16 %1, %4, 0x40 glc\n\t"
                 "s_load_dwordx16 %2, %4, 0x80 glc\n\ts_load_dwordx16 %3, %4, 0xc0 glc\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(q[0]), "=&s"(q[1]), "=&s"(q[2]), "=&s"(q[3])
                 : "s"(row)
                 : "memory");
}

template <typename T> struct TeamVec;
template <> struct TeamVec<float> { typedef float type __attribute__((ext_vector_type(4))); };
template <> struct TeamVec<double> { typedef double type __attribute__((ext_vector_type(2))); };

// CW: compute waves (11, or 10 where ten waves take no more vectors per thread than eleven: fewer idle lanes for the same
// registers — C2's slice of 3125 vectors is 5 x 640 x 0.98 or 5 x 704 x 0.89) ; the wave behind them is the communication wave
// T: the storage type.  Float32 (round 3: rows of 22 529 ... 225 280 entries as teams of 2 ... 16, two to five vectors per thread)
// is k_multi_rows' Float32 arithmetic on this kernel's structure: four columns per vector, a thread's share of a dot product and
// its products with the residual in Float32, the sums across threads, members and rows in Float64 — the hand-off carries Float64
// partials exactly as for Float64 data.
template <int FK, int P, int KV, bool NT, int G = 4, int DEPTH = 2, class DEV = SlabDev, int CW = 11, typename T = double>
__global__ __launch_bounds__(768) void k_team_rows(DEV d, TeamDev t, const void* xv) {
  constexpr int BSC = CW * 64, TPX = 32 / P;
  constexpr int VW = 16 / (int)sizeof(T);          // columns per 16-byte vector
  constexpr bool kF32 = sizeof(T) == 4;
  static_assert(CW == 10 || CW == 11, "ten or eleven compute waves and the communication wave");
  static_assert(G >= DEPTH + 2 && DEPTH <= 3, "slots: DEPTH rows awaiting their residuals, one computing, one loading ; res_s has 4 entries");
  static_assert(P > 1 || DEPTH == 1, "a workgroup on its own has its row sum one step later");
  typedef typename TeamVec<T>::type vec_t;
  typedef unsigned raw_t __attribute__((ext_vector_type(4)));
  extern __shared__ double team_x_raw[];           // the member's slice of x: KV * BSC vectors of 16 bytes
  T* team_x = (T*)team_x_raw;
  __shared__ double wsum[2][CW];
  __shared__ double res_s[4];
  __shared__ int ident[5];                         // team, member, the sticky error word, XCC_ID, ticket
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (P > 1 && tid == 0) {                         // (every launch takes its 32 tickets per XCD, a finished loop's too)
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u;      // HW_REG_XCC_ID
    const unsigned k = atomicAdd(t.tickets + xcc, 1u) - t.launch * 32u;
    ident[0] = k < (unsigned)(TPX * P) ? (int)(xcc * TPX + k / P) : -1;
    ident[1] = (int)(k % P);
    ident[3] = (int)xcc;
    ident[4] = (int)k;
    if (k >= 32u && atomicCAS(t.err, 0, kTeamErrTickets) == 0) {      // an XCD with more than 32 workgroups of a launch
      TeamDiag* g = t.diag;
      g->code = kTeamErrTickets;
      g->members = P;
      g->team = g->member = -1;
      g->xcc = (int)xcc;
      g->ticket = (int)k;
      g->block = (int)blockIdx.x;
      g->depth = DEPTH;
      g->launch = t.launch;
      g->seq0 = t.seq0;
#pragma unroll
      for (int c = 0; c < 8; ++c) g->tickets[c] = __hip_atomic_load(t.tickets + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // the error word of an EARLIER launch (or of a workgroup of this one that has given up already): nothing is waited for
    // any more — the host reads the word with its next poll and the result is discarded
    ident[2] = __hip_atomic_load(t.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (P == 1 && tid == 0) {                        // P = 1: a workgroup on its own — no team to form, nothing to hand over
    ident[0] = (int)blockIdx.x;
    ident[1] = 0;
    ident[2] = 0;
  }
  __syncthreads();
  const int team = __builtin_amdgcn_readfirstlane(ident[0]), s = __builtin_amdgcn_readfirstlane(ident[1]);
  if (team < 0 || ident[2] != 0 || d.c->done != 0) return;          // (done: written by an earlier launch, uniform over the grid)
  const int nvec = (d.ncols + VW - 1) / VW;
  // The member's share of a row (and of x, and of the column accumulator): piece k of a thread is vector v0 + k * kstep + tid.
  // Round 2 gave member s the s-th contiguous P-th of the row (v0 = s nvs, kstep = BSC): the sixteen members of C2's teams then
  // stream sixteen places of a row 50 KB apart.  INTERLEAVED (round 3, t.interleave): the pieces of BSC vectors — one load
  // instruction of a workgroup, 10 KB — are dealt to the members in turn (v0 = s BSC, kstep = P BSC): with every load instruction
  // a team covers P x 10 KB CONTIGUOUS bytes of the row.  Pure reads of C2's rows in the two patterns: 6.37 against 6.78 TB/s
  // (tools/stream_pattern_probe.hip, profiles/r03_stream_pattern_probe2.log) ; teams of two: no difference.
  const bool il = t.interleave != 0;
  const int nvs = il ? nvec : (nvec + P - 1) / P;  // 16-byte vectors a member's share may reach (contiguous: its length)
  const int v0 = il ? s * BSC : s * nvs;
  const int kstep = il ? P * BSC : BSC;
  // The team's rows: a SLAB of consecutive rows (row_stride 1), or every T-th row of the matrix (the cyclic deal, row_stride = T
  // teams): with slabs the 256 workgroups stream 256 places of the matrix a fixed 60 MB apart for the whole launch, and how those
  // fall on the HBM channels is a matter of the allocation — the same binary on the same box ran C5 at 1.20 or at 1.28 ms from
  // one process to the next ; dealt cyclically the chip reads ONE window of T consecutive rows at a time, spread over the channels
  // like any contiguous stream.  rowi(k): the matrix row of the team's k-th row, clamped into its rows.
  // In between: CHUNKS of 2^chunk_log consecutive rows dealt cyclically (consecutive rows keep a team on the same pages, the
  // cyclic deal of the chunks keeps the chip on one window).
  const int nrows = deal_rows(d.m, team, t.rows_per_team, t.row_stride, t.chunk_log);
  // (a block WITHOUT rows — a forced deal on a small matrix — would be pointed past the matrix: the last row then, like every dummy row)
  auto rowi = [&](long long k) { return min(deal_row((int)max(min(k, (long long)nrows), -1LL), nrows, team, t.rows_per_team, t.row_stride, t.chunk_log), d.m - 1); };
  // rows past the slab are dummy rows (clamped loads, residual 0) that every member runs alike: DEPTH steps of prologue,
  // then whole rounds of G steps — the loop body of the compute waves has no branch between a load and its use, so the
  // waits the compiler places are counted ones and the rows in the loading slots really stay in flight
  const int nsteps = DEPTH + (nrows + G - 1) / G * G;
  TeamRec* ring = t.ring + (int64_t)team * kTeamSlots * P;
  const T* __restrict__ X = (const T*)xv;
  if (wv == CW) {
    // ---- the communication wave -------------------------------------------------------------------------------------
    // It issues AHEAD of the compute waves of its SIMD: everything it does between two barriers is on the workgroup's path whenever
    // the workgroup is catching up (DESIGN section 4 "What the communication wave waits for"), and in the arbiter's round-robin every
    // one of its few hundred dependent instructions stood in line behind two waves of FMAs.  C2's kernel 5.80 / 5.84 -> 5.66 / 5.77
    // ms, C5's shape of round 2 (teams of two, one row of slack) 1.246 -> 1.167, least squares on that matrix 1.165 -> 1.13 (two
    // builds alternated on one box, profiles/r03_ab_priority.jsonl) ; a workgroup on its own: no difference.
    __builtin_amdgcn_s_setprio(3);
    // What it needs of a row besides `A x` — b[i] (least squares, logistic) or y[i], A_x_prev[i] and the shift of h (the
    // primal-dual loop) — must be read BEFORE this workgroup's partial of row i is published: in the primal-dual loop member 0
    // overwrites y[i] and A_x_prev[i] once it has all partials of row i, so a member that read them after publishing its own
    // could see the new values.  Earlier is always safe, and they are read a batch of rows ahead (below).  The rows between
    // their step and their residual wait in h[0 .. DEPTH - 1].
    TeamRowState h[DEPTH];
#pragma unroll
    for (int j = 0; j < DEPTH; ++j) h[j] = TeamRowState{};
    const TeamRowConsts<FK> kc(d, xv, kF32);
    double e0 = 0.0, e1 = 0.0;
    bool dead = false;
    [[maybe_unused]] unsigned w_waits = 0, w_polls = 0;              // hand-offs that were not there at the first look ; the longest
    [[maybe_unused]] unsigned long long w_ticks = 0;                 // of them, in polls and in 10 ns ticks (TeamWait: statistics)
    [[maybe_unused]] unsigned w_long = 0;                            // ... those of more than 100 us
    [[maybe_unused]] long long w_row = -1, w_clock = 0;              // ... the row the longest one waited for, the clock when it ended
    [[maybe_unused]] double own_prev = 0.0;        // P = 1: the workgroup's own row sum, one step later
    // The P records of a row are taken on the SCALAR path: P / 4 loads of 64 bytes (`s_load_dwordx16 ... glc`: past the
    // scalar cache, from the XCD's L2 — the teams are formed inside an XCD, and the members' agent-scope stores pass
    // through that L2).  Measured (tools/pattern_probe.hip, profiles/r02_team_handoff_traffic.log): agent-scope VECTOR
    // loads of lines that other CUs keep storing to are served by the memory side, and that small read-after-write
    // traffic alone — nothing waited for — takes the stream of A from 7.1 to 5.8 TB/s; stores alone, or loads alone, cost
    // nothing; the same hand-off with scalar loads runs at 6.6 - 6.7.  The records of the row whose residual is due NEXT
    // are taken a step ahead, right after this workgroup's own partial has been published.
    raw16_t q[(P + 3) / 4];
#pragma unroll
    for (int c = 0; c < (P + 3) / 4; ++c) q[c] = raw16_t{};
    auto fetch = [&](long long r) {
      if constexpr (P > 1) team_fetch<P>(ring + (r % kTeamSlots) * P, q);
    };
    auto rec_word = [&](int m, int w) -> unsigned long long {        // word w (0: value, 1: check) of member m's record
      const raw16_t& c = q[m / 4];
      const int o = (m & 3) * 4 + w * 2;
      return ((unsigned long long)c[o + 1] << 32) | c[o];
    };
    auto all_there = [&](unsigned long long want) {
      bool ok = true;
#pragma unroll
      for (int m = 0; m < P; ++m) ok &= (rec_word(m, 0) ^ rec_word(m, 1)) == want;
      return ok;
    };
    constexpr bool late = FK == kTeamFkPd && P > 1;
    // What the wave needs of a row — b[i], or y[i], A_x_prev[i] and the shift — comes in BATCHES of kBatch rows, lane l of the wave
    // taking row base + l, a whole batch AHEAD: with the chip's memory pipes full of A a load takes several steps to return (the
    // compute waves keep two to three rows — 4 to 6 us — in flight for that reason), and a wait for one falls between two barriers
    // of the workgroup.  (Round 2 took a row's state at the top of its step — the primal-dual loop — or a step ahead: the wait
    // stood in the critical path of C5's kernel, 6.3 TB/s against 6.95 for least squares on the same matrix.)  Reading y and
    // A_x_prev early is safe: member 0 overwrites row i's only after every member has published row i's partial.
    constexpr int kBatch = 16;
    // b[i] of least squares / the logistic loss the same way where the hand-off has two rows of slack: C2's kernel 5.88 -> 5.78 ms,
    // teams of eight on 100 000 x 50 000 5.91 -> 5.79 (two builds alternated on one box, profiles/r03_ab_state_batches.jsonl) ;
    // NOT with one row of slack (teams of two with eight vectors, 50 000 x 20 000: 1.153 -> 1.169 — there the lane reads, on
    // the path between two barriers, cost more than the wait they replace ; nor for a workgroup on its own: Float64 logistic
    // 100 000 x 10 000 1.171 / 1.180 -> 1.186 / 1.198).
    constexpr bool kBatched = FK == kTeamFkPd || (P > 1 && DEPTH == 2);
    auto batch = [&](long long base) { return team_row_state<FK, DEV, T>(d, rowi(base + min(lane, kBatch - 1))); };
    [[maybe_unused]] TeamRowState bcur{}, bnext{};
    if constexpr (kBatched) {
      bcur = batch(0);
      team_row_state_landed<FK>(bcur);             // (a wait is for everything asked for before it: each batch is waited for BEFORE the next is asked for)
      bnext = batch(kBatch);
    }
    // (round 4) What nothing waits for — a row's terms of the residual norm and of h(A x): a division and the value of h ; member
    // 0's stores of y+ and A x — is kept in the LANE of the row's number and worked off sixteen rows at a time by sixteen lanes at
    // once: one division sequence per sixteen rows instead of one per row on the path between two hand-offs (t.pd_defer ; e0 / e1
    // are then per-lane sums, added up in the wave's fixed order at the end).
    [[maybe_unused]] double df_w = 0.0, df_y = 0.0, df_ax = 0.0, df_c = 0.0;
    [[maybe_unused]] int df_i = -1;
    const bool defer = late && t.pd_defer != 0;
    auto flush_deferred = [&]() {
      if constexpr (late) {
        if (df_i >= 0) {
          team_pd_sums<DEV, T>(d, kc, df_c, df_ax, df_w, df_y, df_i, e0, e1);
          if (s == 0) {
            ((T*)d.YN)[df_i] = (T)team_pd_stored(d, df_w, df_y);
            ((T*)d.AXN)[df_i] = (T)df_ax;                                          // :360
          }
        }
        df_i = -1;
      }
    };
    for (long long seq = 0; seq < nsteps; ++seq) {
      [[maybe_unused]] double st_y = 0.0, st_ax = 0.0, st_w = 0.0, st_c = 0.0;      // (the primal-dual loop: row r's y+, A x, w and shift, for what
      [[maybe_unused]] int st_i = -1;                                                //  is done behind the publication of the next partial)
      [[maybe_unused]] TeamRowState snew{};
      if constexpr (kBatched) {
        const int bl = (int)(seq % kBatch);
        if (bl == 0 && seq > 0) {
          team_row_state_landed<FK>(bnext);        // (asked for sixteen steps ago)
          bcur = bnext;
          bnext = batch(seq + kBatch);
        }
        snew = team_row_state_of_lane<FK>(bcur, bl);
      }
      const long long r = seq - DEPTH;             // before the barrier of step seq: the residual of row seq - DEPTH
      if (r >= 0) {
        const int i = rowi(r);
        double ax;
        if constexpr (P == 1) {
          // a workgroup on its own: the sum of its eleven waves IS `A x` of the row — formed after the barrier of step
          // seq, used before the barrier of step seq + 1 (DEPTH = 1)
          ax = own_prev;
        } else {
          const unsigned long long want = t.seq0 + (unsigned long long)r + 1;
          if (!all_there(want) && !dead) {         // not all there yet (or half written): poll, bounded
            // Bounded twice over: t.wait_ticks of wall clock (one second unless the host set another bound) AND 4096 polls.
            // The clock alone would also expire in a wave that was merely descheduled in the middle of a poll — a queue
            // preempted for another process's work has all its waves saved and restored, the clock running on — and a
            // partner that is only late would pass for a missing one ; a partner that IS late (the same preemption, seen from
            // the other side) gets the whole bound, three orders of magnitude above a step.  Every 64 polls the wave also looks
            // at the error word: once any wave of the problem has given up the launch is lost, and waiting on is pointless.
            const long long t0 = wall_clock64();
            int polls = 0;
            for (;;) {
              fetch(r);
              if (all_there(want)) break;
              if ((++polls & 63) == 0) {
                if (__hip_atomic_load(t.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
                  dead = true;
                  break;
                }
                const long long waited = wall_clock64() - t0;
                if (polls > 4096 && waited > (long long)t.wait_ticks) {       // give up — the grid must drain
                  dead = true;
                  if (lane == 0 && atomicCAS(t.err, 0, kTeamErrTimeout) == 0) {
                    TeamDiag* g = t.diag;                                       // (the first failure of the problem: what this wave saw)
                    g->code = kTeamErrTimeout;
                    g->members = P;
                    g->team = team;
                    g->member = s;
                    g->xcc = ident[3];
                    g->ticket = ident[4];
                    g->block = (int)blockIdx.x;
                    g->depth = DEPTH;
                    g->launch = t.launch;
                    g->polls = polls;
                    g->row = r;
                    g->nsteps = nsteps;
                    g->want = want;
                    g->seq0 = t.seq0;
                    g->ticks = (unsigned long long)waited;
                    g->limit = t.wait_ticks;
#pragma unroll
                    for (int c = 0; c < 8; ++c) g->tickets[c] = __hip_atomic_load(t.tickets + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                    for (int m = 0; m < P; ++m) g->seen[m] = TeamRec{rec_word(m, 0), rec_word(m, 1)};
                  }
                  break;
                }
              }
              __builtin_amdgcn_s_sleep(1);
            }
            const long long t1 = wall_clock64();
            const unsigned long long waited = (unsigned long long)(t1 - t0);
            w_waits += 1;
            w_long += waited > 10000ull ? 1u : 0u;
            w_polls = max(w_polls, (unsigned)polls);
            if (waited > w_ticks) w_ticks = waited, w_row = r, w_clock = t1;
          }
          // the P partials added in a fixed pairwise tree, from the same P records on every member: the same `A x` bit for bit
          double part[P];
#pragma unroll
          for (int m = 0; m < P; ++m) part[m] = dead ? 0.0 : __builtin_bit_cast(double, rec_word(m, 0));
#pragma unroll
          for (int w = 1; w < P; w <<= 1)
#pragma unroll
            for (int m = 0; m < P; m += 2 * w) part[m] += part[m + w];
          ax = part[0];
        }
        // the residual / the logistic link / the dual step, ONCE per row, where no row of A is held (exp and log in Float64
        // take ~50 registers: next to the rows they cost the row kernels two of their three rows in flight)
        double res = 0.0;
        if constexpr (late) {
          if (r < nrows) {
            res = team_pd_dual<DEV, T>(d, kc, h[0], ax, i, st_w);
            st_i = i, st_y = res, st_ax = (double)(T)ax, st_c = h[0].c;
          }
        } else {
          res = r < nrows ? team_row_finish<FK, DEV, T>(d, kc, h[0], ax, i, s == 0 && lane == 0, e0, e1) : 0.0;
        }
        if (lane == 0) res_s[r & 3] = res;
      }
      __syncthreads();                             // the barrier of step seq
      double own = lane < CW ? wsum[seq & 1][lane] : 0.0;            // after it: this workgroup's partial of row seq,
#pragma unroll                                                        // the eleven waves' sums in a fixed tree
      for (int off = 1; off < 16; off <<= 1) own += __shfl_xor(own, off, 64);
      if constexpr (P == 1) {
        own_prev = own;
      } else {
        if (lane == 0) {
          const unsigned long long bits = __builtin_bit_cast(unsigned long long, own);
          TeamRec* rec = ring + (seq % kTeamSlots) * P + s;
          __hip_atomic_store(&rec->v, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(&rec->chk, bits ^ (t.seq0 + (unsigned long long)seq + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      if constexpr (late) {
        if (defer) {
          // (sixteen rows at a time, one step BEHIND the rotation of the row-state batches: that rotation waits for everything the
          //  wave has in flight — a burst of stores right in front of it stood on the hand-off's path, teams of four 6.80 -> 6.36
          //  TB/s with 64 rows flushed at step 63 — and these stores have fifteen steps to drain before the next one)
          if (st_i >= 0 && lane == (int)(seq & 15)) df_w = st_w, df_y = st_y, df_ax = st_ax, df_c = st_c, df_i = st_i;
          if ((seq & 15) == 1) flush_deferred();
        } else if (st_i >= 0) {
          team_pd_sums<DEV, T>(d, kc, st_c, st_ax, st_w, st_y, st_i, e0, e1);
        }
      }
      if constexpr (kBatched) {
#pragma unroll
        for (int j = 0; j + 1 < DEPTH; ++j) h[j] = h[j + 1];
        h[DEPTH - 1] = snew;
      } else {
        // (one row of slack, or a workgroup on its own) b[i] is taken HERE, behind the publication of the partial and a step
        // ahead of its use, for the row whose residual is due next.  (Taken at the top of the step the same load cost C2 3 % of
        // its iterations in round 2: the wait for it, which the in-order return per wave extends to the acknowledgement of the
        // stores in front of it, then falls between a step's records and its barrier.)
        if (r + 1 >= 0) h[0] = team_row_state<FK, DEV, T>(d, rowi(r + 1));
      }
      if constexpr (late) {
        if (!defer && st_i >= 0 && s == 0 && lane == 0) {
          ((T*)d.YN)[st_i] = (T)team_pd_stored(d, st_w, st_y);
          ((T*)d.AXN)[st_i] = (T)st_ax;                                            // :360
        }
      }
      if constexpr (P > 1) {
        if (r + 1 >= 0) fetch(r + 1);
      }
    }
    if constexpr (late) {
      if (defer) {                                 // the rows still waiting, then the lanes' sums in the wave's fixed order
        flush_deferred();
        e0 = wave_sum(e0);
        e1 = wave_sum(e1);
      }
    }
    if (lane == 0 && s == 0) {
      double* __restrict__ out = d.part + (int64_t)team * d.ldpart;
      out[d.ncols] = e0;
      out[d.ncols + 1] = e1;
    }
    if constexpr (P > 1) {
      if (lane == 0 && w_waits != 0 && blockIdx.x < 256u) {      // (this launch's only writer of the slot ; launches are serialised ;
        TeamWait* w = t.wait + blockIdx.x;         //  256 slots: a test hook's over-sized grid keeps no statistics beyond them)
        w->waits += w_waits;
        w->long_waits += w_long;
        w->max_polls = max(w->max_polls, w_polls);
        if (w_ticks > w->max_ticks) {
          w->max_ticks = w_ticks;
          w->at_clock = (unsigned long long)w_clock;
          w->at_launch = t.launch;
          w->at_step = (int32_t)w_row;
          w->team = team;
          w->member = s;
          w->xcc = ident[3];
        }
      }
    }
    return;
  }
  // ---- the compute waves ----------------------------------------------------------------------------------------------
  // The member's slice of x goes to LDS — its KV 16-byte loads per thread all in flight at once, issued AHEAD of the first rows of
  // A and written behind them.  (Round 2 took it entry by entry, every entry behind a branch of its own with a wait for the
  // load in it: sixteen dependent trips to L2 before the first row of A was asked for — 10 us of every launch, 1.5 % of C3's.)
  vec_t xin[KV];
#pragma unroll
  for (int k = 0; k < KV; ++k) {
    const int src = max(min(v0 + tid + k * kstep, nvec - 1), 0);     // (a valid address ; what lies past the slice or the row is zeroed below)
    xin[k] = *(const vec_t*)(X + (int64_t)src * VW);
  }
  __builtin_amdgcn_sched_barrier(0);
  double acc[KV][VW];
  const int voff = tid * 16;
  const int kstep16 = __builtin_amdgcn_readfirstlane(kstep * 16);
  const int64_t cbase = (int64_t)v0 * 16;
  // (rows are zero-padded to whole vectors ; interleaved, the descriptor runs from the member's first piece to the END OF THE ROW)
  const int slice_bytes = il ? (int)max((int64_t)0, min((int64_t)nvec * 16, d.ldT * (int64_t)sizeof(T)) - cbase)
                             : (int)max((int64_t)0, min((int64_t)nvs * 16, d.ldT * (int64_t)sizeof(T) - cbase));
  const T* __restrict__ AT = (const T*)d.AT;
  vec_t a[G][KV];
  auto load = [&](int g, long long k) {            // (k: the team's k-th row)
    const int ri = __builtin_amdgcn_readfirstlane(rowi(k));
    // (a dummy row past the slab — the steps that fill the last round — gets a descriptor of ZERO bytes: every lane is out of
    //  range and reads 0 without a memory request ; clamped to the slab's last row these steps re-read it from HBM, the loads
    //  being non-temporal: 3 of C3's 199 steps)
    const int nb = __builtin_amdgcn_readfirstlane(k < nrows ? slice_bytes : 0);
    const __amdgpu_buffer_rsrc_t rs =
        __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)(AT + (int64_t)ri * d.ldT) + cbase), 0, nb, 0x00020000);
#pragma unroll
    for (int k = 0; k < KV; ++k) {
      const raw_t raw = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, k * kstep16, NT ? 2 : 0);
      a[g][k] = __builtin_bit_cast(vec_t, raw);
    }
  };
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int g = 0; g < G - 1; ++g) {
    load(g, g);
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int k = 0; k < KV; ++k) {
    const int lv = tid + k * BSC;
    const int gv = tid + k * kstep;                // (its place in the member's share of the row)
    const int j = (v0 + gv) * VW;
#pragma unroll
    for (int e = 0; e < VW; ++e) {
      team_x[lv * VW + e] = (gv < nvs && j + e < d.ncols) ? xin[k][e] : (T)0;     // (each thread reads back only what it wrote)
      acc[k][e] = 0.0;
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  // the dots of the row in slot g (step seq) and the barrier behind them
  auto dots = [&](int g, long long seq) {
    double p = 0.0;
    if constexpr (kF32) {
      float q = 0.f;
#pragma unroll
      for (int k = 0; k < KV; ++k)
#pragma unroll
        for (int e = 0; e < VW; ++e) q = fmaf(a[g][k][e], team_x[(tid + k * BSC) * VW + e], q);
      p = (double)q;
    } else {
#pragma unroll
      for (int k = 0; k < KV; ++k)
#pragma unroll
        for (int e = 0; e < VW; ++e) p = fma(a[g][k][e], team_x[(tid + k * BSC) * VW + e], p);
    }
#pragma unroll                                     // (a butterfly: EVERY lane ends with the wave's sum, so all of them store
    for (int off = 32; off > 0; off >>= 1) p += __shfl_xor(p, off, 64);      //  it and the loop body has no branch)
    wsum[seq & 1][wv] = p;                         // (double-buffered by row parity: one barrier per row)
    __syncthreads();
  };
  // steps 0 .. DEPTH - 1: nothing to accumulate yet ; the last of them fills slot G - 1, which the prologue left empty
#pragma unroll
  for (int u = 0; u < DEPTH; ++u) {
    dots(u, u);
    __builtin_amdgcn_sched_barrier(0);
    if (u == DEPTH - 1) load(G - 1, G - 1);
    __builtin_amdgcn_sched_barrier(0);
  }
  // step seq >= DEPTH: the dots of row seq (slot seq % G) ; the barrier ; row seq - DEPTH (slot (seq - DEPTH) % G)
  // accumulated and its slot re-loaded with row seq - DEPTH + G.  Unrolled G times: every slot index is a constant.
#pragma unroll 1
  for (long long q = DEPTH; q < nsteps; q += G) {
#pragma unroll
    for (int u = 0; u < G; ++u) {
      const long long seq = q + u;
      const long long r = seq - DEPTH;
      dots((DEPTH + u) % G, seq);
      const double rr = res_s[r & 3];
      if constexpr (kF32) {
        const float rf = (float)rr;                // (the residual IS a Float32 value)
#pragma unroll
        for (int k = 0; k < KV; ++k)
#pragma unroll
          for (int e = 0; e < VW; ++e) acc[k][e] += (double)(a[u][k][e] * rf);      // (slot of row r: (seq - DEPTH) % G = u)
      } else {
#pragma unroll
        for (int k = 0; k < KV; ++k)
#pragma unroll
          for (int e = 0; e < VW; ++e) acc[k][e] = fma(a[u][k][e], rr, acc[k][e]);
      }
#pragma unroll                                     // (the sums are complete HERE: see k_slab_rows)
      for (int k = 0; k < KV; ++k)
#pragma unroll
        for (int e = 0; e < VW; ++e) asm volatile("" : "+v"(acc[k][e]));
      __builtin_amdgcn_sched_barrier(0);
      load(u, r + G);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  double* __restrict__ out = d.part + (int64_t)team * d.ldpart;
#pragma unroll
  for (int k = 0; k < KV; ++k) {
    const int gv = tid + k * kstep;
    const int j = (v0 + gv) * VW;
#pragma unroll
    for (int e = 0; e < VW; ++e)
      if (gv < nvs && j + e < d.ncols) out[j + e] = acc[k][e];
  }
}


// ---------------------------------------------------------------------------------------------------------------------
// SHORT rows (a few hundred ... 5632 Float64 entries) by the same division of labour, H rows per step: a workgroup on its
// own, eleven compute waves that hold G = 3 steps of H rows in registers (KV 16-byte vectors per thread and row, H KV <= 8),
// and a ROW wave whose lane h takes row h of the step — it adds the eleven waves' partial dots, evaluates the residual /
// the logistic link / the dual step for H rows AT ONCE (exp and log across the lanes, where no row of A is held) and leaves
// the H multipliers in LDS.  One barrier per step of H rows.  The first-generation row kernels (csrc/slab.hip k_slab_rows:
// one element per lane and load, the link next to the rows) stay for Float32 and wherever they measure faster.
// ---------------------------------------------------------------------------------------------------------------------
// GR > 1 (rows of fewer than ~700 entries, KV = 1): the compute waves form GR groups of WG waves, each group its own H rows of
// a step — GR H rows per step, one lane of the row wave each — and its own column accumulator, stored as a block partial of
// its own (the closing kernel sums GR partials per workgroup).  GR = 1 with WG < 11: the row is dealt to WG waves only (a row of
// 2500 vectors — C3's — fills four vectors of 640 threads to 98 %, of 704 to 89 %) ; the waves beyond keep the barriers.
//
// T = float (round 3): the same kernel for Float32 data — four columns per 16-byte vector ; a thread's share of a dot product
// (<= 16 products per row) and the H products a column receives per step are summed in Float32, everything across threads and
// steps in Float64 (csrc/slab.hip k_slab2_rows has the error analysis: 5e-10 on f, 4e-8 on the gradient, relative — the size of
// the rounding of `A x` and of the gradient to Float32 that follows anyway) ; the row wave evaluates the logistic link in
// Float32, as Julia does on Float32 arrays.  What it buys over the first-generation Float32 shapes is what it bought Float64:
// the residual / the link of a row is evaluated where no row of A is held and OFF the compute waves' critical path — one
// barrier per step instead of two with a serial section between them.

template <typename T, int FK, int KV, int H, bool NT, class DEV = SlabDev, int GR = 1, int WG = 11, int G = 3>
__global__ __launch_bounds__(768) void k_multi_rows(DEV d, const void* xv) {
  constexpr int CW = 11, BSC = WG * 64, RS = GR * H;                 // (BSC: threads of a group ; RS: rows per step ; G: steps held in registers)
  static_assert(G == 2 || G == 3, "two or three steps in registers");
  constexpr int VW = 16 / (int)sizeof(T);                            // columns per 16-byte vector
  constexpr bool kF32 = sizeof(T) == 4;
  static_assert(H * KV <= 8 && (H == 1 || H == 2 || H == 4 || H == 8), "rows per step x vectors per row: the registers of three steps");
  static_assert(GR * WG <= CW && RS <= 64 && (GR == 1 || KV == 1), "groups of whole waves ; a lane of the row wave per row");
  typedef typename TeamVec<T>::type vec_t;
  typedef unsigned raw_t __attribute__((ext_vector_type(4)));
  constexpr int HP = H < 2 ? 2 : H;                // (the multi-value wave sum wants at least two)
  extern __shared__ double multi_x_raw[];          // x: KV * BSC vectors of 16 bytes
  T* multi_x = (T*)multi_x_raw;
  __shared__ double wsum[2][CW][H];
  __shared__ double res_s[4][RS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (d.c->done != 0) return;                      // (written by an earlier launch: uniform over the grid)
  // (the block's rows: dealt in chunks, csrc/team_dev.h ; k = its own row number, rowi(k) the matrix row, clamped into its rows)
  const int nrows = deal_rows(d.m, (int)blockIdx.x, d.rows_per_block, d.deal_stride, d.deal_log);
  auto rowi = [&](int k) { return min(deal_row(k, nrows, (int)blockIdx.x, d.rows_per_block, d.deal_stride, d.deal_log), d.m - 1); };
  const int nsteps = 1 + ((nrows + RS - 1) / RS + G - 1) / G * G;    // one step of prologue, then whole rounds of G steps
  if (wv == CW) {
    // ---- the row wave: lane L < RS is row L of a step (group L / H, its row L % H) ---------------------------------------
    const TeamRowConsts<FK> kc(d, xv, kF32);
    double e0 = 0.0, e1 = 0.0, own_prev = 0.0;
    TeamRowState st_prev{};
    const int ll = min(lane, RS - 1), lg = ll / H, lh = ll % H;
    for (int step = 0; step < nsteps; ++step) {
      // (the state of THIS step's rows, taken before the barrier behind which their dots complete ; the same wave writes
      //  y+ and A_x a step later: no other reader)
      const TeamRowState snew = team_row_state<FK, DEV, T>(d, rowi(step * RS + ll));
      const int r = step - 1;
      if (r >= 0) {
        const int k = r * RS + ll;
        double res = 0.0;
        if (lane < RS && k < nrows) res = team_row_finish<FK, DEV, T>(d, kc, st_prev, own_prev, rowi(k), true, e0, e1);
        if (lane < RS) res_s[r & 3][lane] = res;
      }
      __syncthreads();                             // the barrier of the step
      double own = 0.0;
#pragma unroll
      for (int w = 0; w < WG; ++w) own += wsum[step & 1][lg * WG + w][lh];      // the group's waves' partials of the row, in order
      own_prev = own;
      st_prev = snew;
    }
    if (lane >= RS) e0 = e1 = 0.0;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {       // (RS <= 64 lanes hold sums, in a fixed tree)
      e0 += __shfl_xor(e0, off, 64);
      e1 += __shfl_xor(e1, off, 64);
    }
    if (lane < GR) {                               // the loss sums in the first of the workgroup's GR partials, zeros in the others
      double* __restrict__ out = d.part + ((int64_t)blockIdx.x * GR + lane) * d.ldpart;
      out[d.ncols] = lane == 0 ? e0 : 0.0;
      out[d.ncols + 1] = lane == 0 ? e1 : 0.0;
    }
    return;
  }
  // ---- the compute waves ----------------------------------------------------------------------------------------------
  const int grp = __builtin_amdgcn_readfirstlane(GR * WG == CW && GR == 1 ? 0 : wv / WG);      // (waves beyond GR WG: no group — they only keep the barriers)
  const bool idle = grp >= GR;
  const int gt = tid - grp * BSC;                  // the thread within its group (an idle wave: a valid index too)
  const T* __restrict__ X = (const T*)xv;
  const int nvec = (d.ncols + VW - 1) / VW;
  // x to LDS: all of a thread's 16-byte loads in flight at once, issued ahead of the first rows of A, written behind them
  // (see k_team_rows ; logistic: X[ncols] is the bias, not a column weight — zeroed with everything past the row)
  vec_t xin[KV];
#pragma unroll
  for (int k = 0; k < KV; ++k) {
    const int src = max(min(gt + k * BSC, nvec - 1), 0);
    xin[k] = *(const vec_t*)(X + (int64_t)src * VW);
  }
  __builtin_amdgcn_sched_barrier(0);
  double acc[KV][VW];
  const int voff = gt * 16;
  const int row_bytes = idle ? 0 : (int)min((int64_t)nvec * 16, d.ldT * (int64_t)sizeof(T));      // (rows are zero-padded to whole vectors ; idle waves load nothing)
  const T* __restrict__ AT = (const T*)d.AT;
  vec_t a[G][H][KV];
  auto load = [&](int g, int step) {
#pragma unroll
    for (int h = 0; h < H; ++h) {
      const int k = step * RS + (idle ? 0 : grp) * H + h;
      const int ri = __builtin_amdgcn_readfirstlane(rowi(k));
      const int nb = __builtin_amdgcn_readfirstlane(k < nrows ? row_bytes : 0);          // (a dummy row: zero bytes, no memory request — see k_team_rows)
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(AT + (int64_t)ri * d.ldT), 0, nb, 0x00020000);
#pragma unroll
      for (int k = 0; k < KV; ++k) {
        const raw_t raw = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, k * BSC * 16, NT ? 2 : 0);
        a[g][h][k] = __builtin_bit_cast(vec_t, raw);
      }
    }
  };
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int g = 0; g < G - 1; ++g) {
    load(g, g);
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int k = 0; k < KV; ++k) {
    const int lv = gt + k * BSC;
    const int j = lv * VW;
#pragma unroll
    for (int e = 0; e < VW; ++e) {
      // (each thread reads back only the entries it wrote — with GR > 1 the same thread of every group writes them, the same
      //  values: no barrier)
      const T xe = (lv < nvec && j + e < d.ncols) ? xin[k][e] : (T)0;
      if (!idle) multi_x[lv * VW + e] = xe;
      acc[k][e] = 0.0;
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  auto dots = [&](int g, int step) {
    double p[HP];
#pragma unroll
    for (int h = 0; h < HP; ++h) p[h] = 0.0;
    if constexpr (kF32) {
      float q[H];
#pragma unroll
      for (int h = 0; h < H; ++h) q[h] = 0.f;
#pragma unroll
      for (int k = 0; k < KV; ++k)
#pragma unroll
        for (int e = 0; e < VW; ++e) {
          const float xe = multi_x[(gt + k * BSC) * VW + e];
#pragma unroll
          for (int h = 0; h < H; ++h) q[h] = fmaf(a[g][h][k][e], xe, q[h]);
        }
#pragma unroll
      for (int h = 0; h < H; ++h) p[h] = (double)q[h];
    } else {
#pragma unroll
      for (int k = 0; k < KV; ++k)
#pragma unroll
        for (int e = 0; e < VW; ++e) {
          const double xe = multi_x[(gt + k * BSC) * VW + e];
#pragma unroll
          for (int h = 0; h < H; ++h) p[h] = fma(a[g][h][k][e], xe, p[h]);
        }
    }
    if constexpr (H == 1) {
#pragma unroll                                     // (a butterfly: every lane ends with the wave's sum and stores it — no branch)
      for (int off = 32; off > 0; off >>= 1) p[0] += __shfl_xor(p[0], off, 64);
      wsum[step & 1][wv][0] = p[0];
    } else {
      wave_sum_multi<HP>(p);                       // lane l: the wave's sum for row l / (64 / HP), the same in all its lanes
      wsum[step & 1][wv][lane / (64 / H)] = p[0];  // (every lane stores: no branch in the loop body)
    }
    __syncthreads();
  };
  dots(0, 0);                                      // step 0: nothing to accumulate yet ; fills the slot the prologue left empty
  __builtin_amdgcn_sched_barrier(0);
  load(G - 1, G - 1);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
  for (int q = 1; q < nsteps; q += G) {
#pragma unroll
    for (int u = 0; u < G; ++u) {
      const int step = q + u, r = step - 1;
      dots((1 + u) % G, step);
      if constexpr (kF32) {
        float rf[H];
#pragma unroll
        for (int h = 0; h < H; ++h) rf[h] = (float)res_s[r & 3][(idle ? 0 : grp) * H + h];      // (the residuals ARE Float32 values)
#pragma unroll
        for (int k = 0; k < KV; ++k)
#pragma unroll
          for (int e = 0; e < VW; ++e) {
            float t = a[u][0][k][e] * rf[0];       // (slot of step r: (step - 1) % G = u)
#pragma unroll
            for (int h = 1; h < H; ++h) t = fmaf(a[u][h][k][e], rf[h], t);
            acc[k][e] += (double)t;
          }
      } else {
#pragma unroll
        for (int h = 0; h < H; ++h) {
          const double rr = res_s[r & 3][(idle ? 0 : grp) * H + h];
#pragma unroll
          for (int k = 0; k < KV; ++k)
#pragma unroll
            for (int e = 0; e < VW; ++e) acc[k][e] = fma(a[u][h][k][e], rr, acc[k][e]);      // (slot of step r: (step - 1) % G = u)
        }
      }
#pragma unroll
      for (int k = 0; k < KV; ++k)
#pragma unroll
        for (int e = 0; e < VW; ++e) asm volatile("" : "+v"(acc[k][e]));
      __builtin_amdgcn_sched_barrier(0);
      load(u, r + G);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  if (idle) return;
  double* __restrict__ out = d.part + ((int64_t)blockIdx.x * GR + grp) * d.ldpart;
#pragma unroll
  for (int k = 0; k < KV; ++k) {
    const int lv = gt + k * BSC;
    const int j = lv * VW;
#pragma unroll
    for (int e = 0; e < VW; ++e)
      if (lv < nvec && j + e < d.ncols) out[j + e] = acc[k][e];
  }
}

// the launch for a row length: KV = vectors per compute thread and row, H = 8 / KV rows per step (H KV <= 8)
// rows per workgroup partial: the workgroup's GR groups each leave one (the closing kernels sum nblocks * multi_groups(ncols))
inline int multi_groups(int64_t ncols, int vw = 2) {
  const int64_t nvec = (ncols + vw - 1) / vw;
  return nvec <= 128 ? 5 : nvec <= 192 ? 3 : nvec <= 320 ? 2 : 1;
}
// ten compute waves instead of eleven where that takes no more vectors per thread: fewer idle lanes for the same registers
// (C3: 2500 vectors per row = 4 x 640 x 0.98 against 4 x 704 x 0.89) ; `waves` = 10 / 11 forces one (APX_OPT_MULTI_WAVES: A/B)
inline int multi_waves(int64_t nvec, int waves) {
  if (waves == 10 || waves == 11) return waves;
  const int64_t kv11 = (nvec + 703) / 704, kv10 = (nvec + 639) / 640;
  return kv10 == kv11 && kv11 >= 2 ? 10 : 11;
}
constexpr int64_t kMultiMaxVec = 4 * 704;          // rows per step >= 2: up to four vectors per thread
template <typename T, int FK, bool NT, class DEV>
inline bool multi_launch(int64_t ncols, int nblocks, int waves, const DEV& d, const void* x, hipStream_t q) {
  constexpr int VW = 16 / (int)sizeof(T);
  const int64_t nvec = (ncols + VW - 1) / VW;
  const dim3 grid((unsigned)nblocks), block(768);
  // very short rows: GR groups of WG waves, eight rows each per step (40 / 24 / 16 rows per step)
  if (nvec <= 128) { hipLaunchKernelGGL((k_multi_rows<T, FK, 1, 8, NT, DEV, 5, 2>), grid, block, (size_t)128 * 16, q, d, x); return true; }
  if (nvec <= 192) { hipLaunchKernelGGL((k_multi_rows<T, FK, 1, 8, NT, DEV, 3, 3>), grid, block, (size_t)192 * 16, q, d, x); return true; }
  if (nvec <= 320) { hipLaunchKernelGGL((k_multi_rows<T, FK, 1, 8, NT, DEV, 2, 5>), grid, block, (size_t)320 * 16, q, d, x); return true; }
  if (multi_waves(nvec, waves) == 10) {
    switch ((int)((nvec + 639) / 640)) {
      case 2: hipLaunchKernelGGL((k_multi_rows<T, FK, 2, 4, NT, DEV, 1, 10>), grid, block, (size_t)2 * 640 * 16, q, d, x); return true;
      case 3: hipLaunchKernelGGL((k_multi_rows<T, FK, 3, 2, NT, DEV, 1, 10>), grid, block, (size_t)3 * 640 * 16, q, d, x); return true;
      case 4: hipLaunchKernelGGL((k_multi_rows<T, FK, 4, 2, NT, DEV, 1, 10>), grid, block, (size_t)4 * 640 * 16, q, d, x); return true;
      default: break;
    }
  }
  switch ((int)((nvec + 703) / 704)) {
    case 1: hipLaunchKernelGGL((k_multi_rows<T, FK, 1, 8, NT, DEV>), grid, block, (size_t)1 * 704 * 16, q, d, x); return true;
    case 2: hipLaunchKernelGGL((k_multi_rows<T, FK, 2, 4, NT, DEV>), grid, block, (size_t)2 * 704 * 16, q, d, x); return true;
    case 3: hipLaunchKernelGGL((k_multi_rows<T, FK, 3, 2, NT, DEV>), grid, block, (size_t)3 * 704 * 16, q, d, x); return true;
    case 4: hipLaunchKernelGGL((k_multi_rows<T, FK, 4, 2, NT, DEV>), grid, block, (size_t)4 * 704 * 16, q, d, x); return true;
    default: break;
  }
  // one row per step, five to eight vectors per thread — Float32 rows of 11 265 ... 22 528 entries (Float64 rows of this many
  // vectors take k_team_rows with one workgroup, the same structure) ; seven and eight vectors hold TWO steps in registers (three
  // spill: the Float64 accumulator of a Float32 row is twice the row)
  if constexpr (sizeof(T) == 4) {
    if (multi_waves(nvec, waves) == 10) {
      switch ((int)((nvec + 639) / 640)) {
        case 5: hipLaunchKernelGGL((k_multi_rows<T, FK, 5, 1, NT, DEV, 1, 10>), grid, block, (size_t)5 * 640 * 16, q, d, x); return true;
        case 6: hipLaunchKernelGGL((k_multi_rows<T, FK, 6, 1, NT, DEV, 1, 10>), grid, block, (size_t)6 * 640 * 16, q, d, x); return true;
        case 7: hipLaunchKernelGGL((k_multi_rows<T, FK, 7, 1, NT, DEV, 1, 10, 2>), grid, block, (size_t)7 * 640 * 16, q, d, x); return true;
        case 8: hipLaunchKernelGGL((k_multi_rows<T, FK, 8, 1, NT, DEV, 1, 10, 2>), grid, block, (size_t)8 * 640 * 16, q, d, x); return true;
        default: break;
      }
    }
    switch ((int)((nvec + 703) / 704)) {
      case 5: hipLaunchKernelGGL((k_multi_rows<T, FK, 5, 1, NT, DEV>), grid, block, (size_t)5 * 704 * 16, q, d, x); return true;
      case 6: hipLaunchKernelGGL((k_multi_rows<T, FK, 6, 1, NT, DEV>), grid, block, (size_t)6 * 704 * 16, q, d, x); return true;
      case 7: hipLaunchKernelGGL((k_multi_rows<T, FK, 7, 1, NT, DEV, 1, 11, 2>), grid, block, (size_t)7 * 704 * 16, q, d, x); return true;
      case 8: hipLaunchKernelGGL((k_multi_rows<T, FK, 8, 1, NT, DEV, 1, 11, 2>), grid, block, (size_t)8 * 704 * 16, q, d, x); return true;
      default: break;
    }
  }
  return false;
}

// ---- host side: the plan for a row length, and the launch ----------------------------------------------------------------
constexpr int kTeamSingleKv = 8;      // one workgroup / the three-slot shape: up to 8 vectors per compute thread
constexpr int64_t kTeamMaxCols = (int64_t)16 * kTeamSingleKv * 704 * 2;      // 180 224 Float64 columns
constexpr int64_t kTeamMaxColsF32 = (int64_t)16 * 5 * 704 * 4;               // 225 280 Float32 columns (teams of sixteen, five vectors)

struct TeamPlan {
  int P = 0, kv = 0;                  // members ; 16-byte vectors per compute thread and row of a member's slice
  bool big = false;                   // seven or eight vectors: three row slots, ONE row of slack
  int waves = 11;                     // compute waves: 10 where that takes no more vectors per thread (k_team_rows CW)
  int alt = 0;                        // experimental slot shapes, G * 10 + DEPTH (APX_OPT_TEAM_SLOTS ; teams of four, KV = 4, the primal-dual form only)
};
struct TeamTune {                     // A/B overrides (apx_ctx_set_option): 0 = the plan below decides
  int members = 0, waves = 0, alt = 0;
  bool pd = false;                    // the primal-dual form (csrc/slab_pd.hip)
};
// One workgroup where the row fits its eight vectors per thread (three row slots, nothing to hand over).  Else the members'
// slices as large as the registers allow, because a step costs a barrier and a hand-off whatever its bytes:
//   up to six vectors per thread with four row slots and TWO rows of slack for the hand-off (168 registers), or
//   seven or eight with three slots and ONE row of slack — which teams of two and four absorb (a step is 60 - 90 KB per
//   member: tools/pattern_probe.hip, 6.6 and 6.9 TB/s) but teams of eight and sixteen do not (6.2 and 5.5 against 6.6 with
//   two rows of slack) ; sixteen members take it only beyond 135 168 columns, where the alternative is two passes.
inline TeamPlan team_plan(int64_t ncols, const TeamTune& tune = TeamTune{}, int vw = 2) {
  const int64_t nvec = (ncols + vw - 1) / vw, v6 = 6 * 704, v8 = (int64_t)kTeamSingleKv * 704;
  TeamPlan pl;
  if (vw == 4) {      // Float32: the four-slot shapes of real teams only, two to FIVE vectors (the Float64 accumulator of a Float32 row
    const int64_t v5 = 5 * 704;      // is twice the row: six vectors spill 44 bytes)
    pl.P = nvec <= 2 * v5 ? 2 : nvec <= 4 * v5 ? 4 : nvec <= 8 * v5 ? 8 : nvec <= 16 * v5 ? 16 : 0;
    if (pl.P == 0) return pl;
    const int64_t nvs = (nvec + pl.P - 1) / pl.P;
    pl.kv = (int)((nvs + 703) / 704);
    const int kv10 = (int)((nvs + 639) / 640);
    pl.waves = tune.waves == 10 || tune.waves == 11 ? tune.waves : (kv10 == pl.kv ? 10 : 11);
    if (pl.waves == 10 && kv10 > 5) pl.waves = 11;
    if (pl.waves == 10) pl.kv = kv10;
    return pl;
  }
  pl.P = 16;
  if (nvec <= v8) pl.P = 1;
  else if (nvec <= 2 * v6) pl.P = 2;
  else if (nvec <= 2 * v8) pl.P = 2, pl.big = true;
  else if (nvec <= 4 * v6) pl.P = 4;
  else if (nvec <= 4 * v8) pl.P = 4, pl.big = true;
  else if (nvec <= 8 * v6) pl.P = 8;
  // (round 4: teams of eight in the three-slot shape, where sixteen members would hold three or four vectors each — their step is
  //  the shortest of all, ~ 35 KB per member, and the communication wave's chain does not fit it: 70 000 columns, 6 GB, two builds of
  //  the plan alternated in one process, least squares 6.36 -> 6.81 TB/s, the primal-dual form 5.11 -> 6.60 ; 85 000: 6.68 -> 6.90,
  //  5.67 -> 6.83 — profiles/r04i_ab_teams_of_eight_three_slots.jsonl)
  else if (nvec <= 8 * v8) pl.P = 8, pl.big = true;
  else if (nvec <= 16 * v6) pl.P = 16;
  else pl.big = true;
  // (The primal-dual form in the three-slot shape has the whole dual step of a row — a division, the conjugate prox — between two
  // barriers a single row behind the hand-off.  Before the communication wave had its priority, twice the members with half the
  // vectors and two rows of slack were the better shape wherever they filled their lanes 6 % better — 20 000 columns: 1.239 ->
  // 1.202 ms — and `team_plan` chose by lane fill ; with the priority the wide shape wins at every length measured: 17 000 columns
  // 1.000 against 1.130 ms, 20 000: 1.154 against 1.214, 22 528: 1.309 against 1.372, 40 000: 2.298 against 2.439 —
  // profiles/r03_ab_priority.jsonl.  `tune.pd` is still passed: APX_OPT_TEAM_SLOTS applies to the primal-dual form only.)
  // (A/B: a forced team size, where the row still fits its eight vectors per thread and is long enough to be split)
  if (pl.P > 1 && (tune.members == 2 || tune.members == 4 || tune.members == 8 || tune.members == 16) && nvec <= tune.members * v8 &&
      nvec > (int64_t)tune.members * 704) {
    pl.P = tune.members;
    pl.big = nvec > (int64_t)tune.members * v6;
  }
  const int64_t nvs = (nvec + pl.P - 1) / pl.P;
  pl.kv = (int)((nvs + 703) / 704);
  const int kv10 = (int)((nvs + 639) / 640);
  // Ten compute waves where they take no more vectors per thread than eleven (fewer idle lanes, the same registers) — measured,
  // interleaved in one process (tools/ab.py, profiles/r03_ab_*.jsonl): C2 (sixteen members, five vectors) 6.134 against 6.168 ms,
  // Float64 logistic 100 000 x 10 000 (one workgroup, eight vectors) 1.199 against 1.252, C5's length as teams of four 1.197 against
  // 1.235 — but NOT the three-slot shape of real teams, where the hand-off sits a single row behind: C5 (two members, eight
  // vectors) 1.194 ms with eleven waves against 1.221 with ten.
  pl.waves = tune.waves == 10 || tune.waves == 11 ? tune.waves : (kv10 == pl.kv && !(pl.P > 1 && pl.big) ? 10 : 11);
  if (pl.waves == 10 && kv10 > (pl.big || pl.P == 1 ? 8 : 6)) pl.waves = 11;      // (a forced 10 that would not fit the shape)
  if (pl.waves == 10) pl.kv = kv10;
  if (pl.P == 4 && pl.kv == 4 && !pl.big && tune.pd) pl.alt = tune.alt;
  // (sixteen members in the three-slot shape — C2's five vectors with one row of slack — measured again in round 4 with the
  //  communication wave's priority: 8.02 against 5.76 ms, profiles/r04l_ab_c2_three_slots.jsonl ; they need the second row: removed)
  // The primal-dual form of teams of two with FIVE vectors takes the three-slot shape as well (round 4: rows of 11 265 ... 14 080
  // entries, 6 GB, alternated in one process — 11 300: 5.74 -> 6.27 TB/s, 12 000: 6.05 -> 6.39, 14 000: 6.39 -> 6.68 ; with six
  // vectors, 16 000: 6.94 against 6.90, left alone — profiles/r04j_ab_teams_of_two_three_slots.jsonl).  APX_OPT_TEAM_SLOTS = 42 keeps
  // the four slots (A/B), 31 forces the three-slot shape for six vectors too.
  if (tune.pd && pl.P == 2 && !pl.big && ((pl.kv == 5 && tune.alt != 42) || (pl.kv == 6 && tune.alt == 31))) pl.big = true, pl.waves = 11;
  return pl;
}

// (the dynamic-LDS attribute belongs to the function ON A DEVICE: once per device of the process — csrc/small_dev.h)
template <auto KERNEL>
inline void team_allow_lds() {
  static std::atomic<unsigned long long> done_devs{0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  const unsigned long long bit = 1ull << (dev & 63);
  if ((done_devs.load(std::memory_order_relaxed) & bit) == 0) {
    if (hipFuncSetAttribute((const void*)KERNEL, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024) != hipSuccess) (void)hipGetLastError();
    done_devs.fetch_or(bit, std::memory_order_relaxed);
  }
}
// One launch of a plan, or — `query` — the question whether it CAN run: the scheme needs all of a launch's workgroups resident
// together, one per CU, and that is a property of the kernel's registers and LDS that the runtime can be asked about before
// anything is launched (hipOccupancyMaxActiveBlocksPerMultiprocessor).  What it cannot answer is whether the CUs are FREE — a
// GPU shared with another process's kernels: that is what the bounded waits and the error word remain for.
struct TeamCall {
  bool query = false;       // do not launch: fill blocks_per_cu
  bool cooperative = false; // hipLaunchCooperativeKernel (APX_OPT_TEAM_COOPERATIVE: the runtime refuses a grid that cannot be co-resident)
  int blocks_per_cu = -1;
  hipError_t launch_error = hipSuccess;
};
// (TeamDev::fault_grid: a TEST hook — launches that many workgroups instead of all of them, so that some teams miss members:
//  what a GPU shared with other work does to this scheme.  tests/test_gpu_team_fault.py checks that the waits time out,
//  the grid drains and the error comes back)
//  (more than 256, with APX_OPT_TEAM_COOPERATIVE only: a grid that cannot be co-resident — the runtime refuses the launch)
inline unsigned team_grid(const TeamDev& t, int nteams, int P, bool cooperative) {
  const bool hook = t.fault_grid > 0 && P > 1 && (t.fault_grid <= 256 || cooperative);
  return (unsigned)(hook ? t.fault_grid : nteams * P);
}
template <int FK, int P, int KV, bool NT, int G, int DEPTH, class DEV, int CW, typename T = double>
inline void team_launch_one(TeamCall& call, int nteams, const DEV& d, const TeamDev& t, const void* x, hipStream_t q) {
  constexpr size_t lds = (size_t)KV * CW * 64 * 2 * sizeof(double);      // the slice of x
  constexpr int kThreads = (CW + 1) * 64;
  auto kernel = k_team_rows<FK, P, KV, NT, G, DEPTH, DEV, CW, T>;
  if constexpr (lds > 64 * 1024) team_allow_lds<k_team_rows<FK, P, KV, NT, G, DEPTH, DEV, CW, T>>();
  if (call.query) {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, kThreads, lds) != hipSuccess) {
      (void)hipGetLastError();
      nb = 0;
    }
    call.blocks_per_cu = nb;
    return;
  }
  const dim3 grid(team_grid(t, nteams, P, call.cooperative)), block(kThreads);
  if (call.cooperative && P > 1) {
    DEV dd = d;
    TeamDev tt = t;
    const void* xx = x;
    void* args[3] = {&dd, &tt, &xx};
    call.launch_error = hipLaunchCooperativeKernel((const void*)kernel, grid, block, args, (unsigned)lds, q);
    return;
  }
  hipLaunchKernelGGL(kernel, grid, block, lds, q, d, t, x);
}
// four row slots, two rows of slack: 2 ... 6 vectors
// (measured and dropped at C2, interleaved runs of bench.py: five and six slots, two or three rows of slack, one to three rows
//  really in flight — nothing ; profiles/r02_team_handoff_traffic.log has what the gap to the streaming rate was instead)
template <int FK, int P, class DEV, int CW, typename T = double>
inline bool team_launch_d2(TeamCall& call, int kv, int alt, int nteams, const DEV& d, const TeamDev& t, const void* x, hipStream_t q) {
  if constexpr (sizeof(T) == 4) {                  // Float32: two to six vectors, four slots, two rows of slack
    switch (kv) {
      case 1:
      case 2: team_launch_one<FK, P, 2, true, 4, 2, DEV, CW, T>(call, nteams, d, t, x, q); return true;
      case 3: team_launch_one<FK, P, 3, true, 4, 2, DEV, CW, T>(call, nteams, d, t, x, q); return true;
      case 4: team_launch_one<FK, P, 4, true, 4, 2, DEV, CW, T>(call, nteams, d, t, x, q); return true;
      case 5: team_launch_one<FK, P, 5, true, 4, 2, DEV, CW, T>(call, nteams, d, t, x, q); return true;
      default: return false;
    }
  }
  // (APX_OPT_TEAM_SLOTS: more row slots / rows of slack for C5's length as teams of four — an A/B hook, the primal-dual form only)
  if constexpr (P == 4 && FK == kTeamFkPd) {
    if (kv == 4) {
      switch (alt) {
        case 52: team_launch_one<FK, P, 4, true, 5, 2, DEV, CW>(call, nteams, d, t, x, q); return true;
        case 62: team_launch_one<FK, P, 4, true, 6, 2, DEV, CW>(call, nteams, d, t, x, q); return true;
        case 53: team_launch_one<FK, P, 4, true, 5, 3, DEV, CW>(call, nteams, d, t, x, q); return true;
        case 63: team_launch_one<FK, P, 4, true, 6, 3, DEV, CW>(call, nteams, d, t, x, q); return true;
        default: break;
      }
    }
  }
  switch (kv) {
    case 1:
    case 2: team_launch_one<FK, P, 2, true, 4, 2, DEV, CW>(call, nteams, d, t, x, q); return true;
    case 3: team_launch_one<FK, P, 3, true, 4, 2, DEV, CW>(call, nteams, d, t, x, q); return true;
    case 4: team_launch_one<FK, P, 4, true, 4, 2, DEV, CW>(call, nteams, d, t, x, q); return true;
    case 5: team_launch_one<FK, P, 5, true, 4, 2, DEV, CW>(call, nteams, d, t, x, q); return true;
    case 6: team_launch_one<FK, P, 6, true, 4, 2, DEV, CW>(call, nteams, d, t, x, q); return true;
    default: return false;
  }
}
// three row slots, one row of slack: 7 or 8 vectors (a workgroup on its own: 4 ... 8, non-temporal loads by the matrix size)
template <int FK, int P, bool NT, class DEV, int CW>
inline bool team_launch_d1(TeamCall& call, int kv, int nteams, const DEV& d, const TeamDev& t, const void* x, hipStream_t q) {
  switch (kv) {
    case 7: team_launch_one<FK, P, 7, NT, 3, 1, DEV, CW>(call, nteams, d, t, x, q); return true;
    case 8: team_launch_one<FK, P, 8, NT, 3, 1, DEV, CW>(call, nteams, d, t, x, q); return true;
    default: break;
  }
  if constexpr (P == 2 && FK == kTeamFkPd) {       // (the primal-dual form of teams of two: five or six vectors in the three-slot shape)
    switch (kv) {
      case 5: team_launch_one<FK, 2, 5, NT, 3, 1, DEV, CW>(call, nteams, d, t, x, q); return true;
      case 6: team_launch_one<FK, 2, 6, NT, 3, 1, DEV, CW>(call, nteams, d, t, x, q); return true;
      default: break;
    }
  }
  if constexpr (P == 1) {
    switch (kv) {
      case 1:
      case 2:
      case 3:
      case 4: team_launch_one<FK, 1, 4, NT, 3, 1, DEV, CW>(call, nteams, d, t, x, q); return true;
      case 5: team_launch_one<FK, 1, 5, NT, 3, 1, DEV, CW>(call, nteams, d, t, x, q); return true;
      case 6: team_launch_one<FK, 1, 6, NT, 3, 1, DEV, CW>(call, nteams, d, t, x, q); return true;
      default: break;
    }
  }
  return false;
}
// the launch of a plan (or the query whether it can be resident: TeamCall) ; a launch advances the ranges of tickets and
// sequence numbers (never reset)
template <int FK, class DEV, int CW, typename T = double>
inline bool team_dispatch_w(TeamCall& call, const TeamPlan& pl, int nteams, bool nt, const DEV& d, const TeamDev& t, const void* x, hipStream_t q) {
  bool ok = false;
  if constexpr (sizeof(T) == 4) {
    if (pl.big || pl.P < 2) return false;
    if (pl.P == 2) ok = team_launch_d2<FK, 2, DEV, CW, T>(call, pl.kv, 0, nteams, d, t, x, q);
    else if (pl.P == 4) ok = team_launch_d2<FK, 4, DEV, CW, T>(call, pl.kv, 0, nteams, d, t, x, q);
    else if (pl.P == 8) ok = team_launch_d2<FK, 8, DEV, CW, T>(call, pl.kv, 0, nteams, d, t, x, q);
    else if (pl.P == 16) ok = team_launch_d2<FK, 16, DEV, CW, T>(call, pl.kv, 0, nteams, d, t, x, q);
    return ok;
  } else {
    if (pl.P == 1) ok = nt ? team_launch_d1<FK, 1, true, DEV, CW>(call, pl.kv, nteams, d, t, x, q) : team_launch_d1<FK, 1, false, DEV, CW>(call, pl.kv, nteams, d, t, x, q);
    else if (pl.big) {
      if (pl.P == 2) ok = team_launch_d1<FK, 2, true, DEV, CW>(call, pl.kv, nteams, d, t, x, q);
      else if (pl.P == 4) ok = team_launch_d1<FK, 4, true, DEV, CW>(call, pl.kv, nteams, d, t, x, q);
      else if (pl.P == 8) ok = team_launch_d1<FK, 8, true, DEV, CW>(call, pl.kv, nteams, d, t, x, q);
      else if (pl.P == 16) ok = team_launch_d1<FK, 16, true, DEV, CW>(call, pl.kv, nteams, d, t, x, q);
    } else if (pl.P == 2) ok = team_launch_d2<FK, 2, DEV, CW>(call, pl.kv, pl.alt, nteams, d, t, x, q);
    else if (pl.P == 4) ok = team_launch_d2<FK, 4, DEV, CW>(call, pl.kv, pl.alt, nteams, d, t, x, q);
    else if (pl.P == 8) ok = team_launch_d2<FK, 8, DEV, CW>(call, pl.kv, pl.alt, nteams, d, t, x, q);
    else if (pl.P == 16) ok = team_launch_d2<FK, 16, DEV, CW>(call, pl.kv, pl.alt, nteams, d, t, x, q);
    return ok;
  }
}
template <int FK, class DEV, typename T = double>
inline bool team_dispatch(TeamCall& call, const TeamPlan& pl, int nteams, bool nt, const DEV& d, const TeamDev& t, const void* x, hipStream_t q) {
  return pl.waves == 10 ? team_dispatch_w<FK, DEV, 10, T>(call, pl, nteams, nt, d, t, x, q) : team_dispatch_w<FK, DEV, 11, T>(call, pl, nteams, nt, d, t, x, q);
}
template <int FK, class DEV, typename T = double>
inline bool team_launch(const TeamPlan& pl, int nteams, bool nt, bool cooperative, const DEV& d, TeamDev& t, const void* x, hipStream_t q) {
  TeamCall call;
  call.cooperative = cooperative;
  const bool ok = team_dispatch<FK, DEV, T>(call, pl, nteams, nt, d, t, x, q) && call.launch_error == hipSuccess;
  if (!ok) return false;    // (nothing ran: no tickets were drawn, no sequence word used — the ranges stay where they are)
  t.launch += 1;
  t.seq0 += (unsigned long long)t.rows_per_team + (1u << t.chunk_log) + 16;      // (> the steps of a launch: the sequence words never repeat)
  return true;
}
// workgroups of this plan's kernel that fit one CU at once (0: the runtime could not say, or none)
template <int FK, class DEV, typename T = double>
inline int team_blocks_per_cu(const TeamPlan& pl, bool nt) {
  TeamCall call;
  call.query = true;
  const DEV d{};
  const TeamDev t{};
  if (!team_dispatch<FK, DEV, T>(call, pl, 1, nt, d, t, nullptr, nullptr)) return 0;
  return call.blocks_per_cu;
}

// ---- host side: the hand-off state of a problem's teams, and what a failure looked like --------------------------------------
struct TeamHost {
  DBuf buf;                 // ring | tickets | err | diag | wait
  TeamDev dev{};
  struct Pinned { int32_t err; int32_t _pad; TeamDiag diag; TeamWait wait[256]; }* pinned = nullptr;
  ~TeamHost() { if (pinned) (void)hipHostFree(pinned); }
};
inline int32_t team_alloc(apx_ctx* ctx, TeamHost* th, int nteams, int P, int64_t rows_per_team, int64_t row_bytes) {
  const size_t ring_bytes = (size_t)nteams * kTeamSlots * P * sizeof(TeamRec);
  const size_t off_tickets = ring_bytes, off_err = off_tickets + 32, off_diag = off_err + 32;
  const size_t off_wait = off_diag + (sizeof(TeamDiag) + 63) / 64 * 64;
  APX_TRY(th->buf.alloc_zero(off_wait + 256 * sizeof(TeamWait), ctx->stream));
  char* base = (char*)th->buf.p;
  th->dev.ring = (TeamRec*)base;
  th->dev.tickets = (unsigned*)(base + off_tickets);
  th->dev.err = (int*)(base + off_err);
  th->dev.diag = (TeamDiag*)(base + off_diag);
  th->dev.wait = (TeamWait*)(base + off_wait);
  th->dev.seq0 = 0;
  th->dev.launch = 0;
  th->dev.rows_per_team = (int32_t)rows_per_team;
  // APX_OPT_ROW_DEAL: -1 = chunks by the row length (default) ; 0 = slabs ; R = 1, 2, 4, ... = chunks of R consecutive rows
  th->dev.row_stride = ctx->row_deal != 0 ? nteams : 1;
  th->dev.chunk_log = 0;
  if (ctx->row_deal < 0) th->dev.chunk_log = deal_chunk_log(row_bytes, rows_per_team);
  for (int r = ctx->row_deal; r > 1; r >>= 1) th->dev.chunk_log += 1;
  th->dev.wait_ticks = (unsigned long long)ctx->team_wait_ms * 100000ull;      // (s_memrealtime: 100 MHz)
  th->dev.fault_grid = ctx->team_fault_grid;
  th->dev.pd_defer = ctx->team_pd_defer;
  // (teams of two gain nothing in the pure-read probe and LOSE 1.7 % in the product — C5: 1.289 -> 1.312 ms — : from four members on)
  th->dev.interleave = (ctx->team_interleave == 2 && P > 1) || (ctx->team_interleave == 1 && P >= 4) ? 1 : 0;
  APX_HIP(hipHostMalloc((void**)&th->pinned, sizeof(TeamHost::Pinned), hipHostMallocDefault));
  memset(th->pinned, 0, sizeof(TeamHost::Pinned));
  return APX_OK;
}
// the error word (0: fine) ; on a failure the diagnostic record comes with it
inline int32_t team_read_err(apx_ctx* ctx, TeamHost* th, int32_t* err) {
  APX_HIP(hipMemcpyAsync(&th->pinned->err, th->dev.err, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
  APX_HIP(hipStreamSynchronize(ctx->stream));
  *err = th->pinned->err;
  if (*err != 0) {
    APX_HIP(hipMemcpyAsync(&th->pinned->diag, th->dev.diag, sizeof(TeamDiag), hipMemcpyDeviceToHost, ctx->stream));
    APX_HIP(hipStreamSynchronize(ctx->stream));
  }
  return APX_OK;
}
// the slots' records as they are (apx_problem_team_wait_slots) ; valid until the next read
inline int32_t team_read_wait_slots(apx_ctx* ctx, TeamHost* th, const TeamWait** slots) {
  APX_HIP(hipMemcpyAsync(th->pinned->wait, th->dev.wait, sizeof(th->pinned->wait), hipMemcpyDeviceToHost, ctx->stream));
  APX_HIP(hipStreamSynchronize(ctx->stream));
  *slots = th->pinned->wait;
  return APX_OK;
}
// the longest hand-off wait any workgroup slot has seen so far (polls, 10 ns ticks) and the number of waits
inline int32_t team_read_waits(apx_ctx* ctx, TeamHost* th, int64_t out[3]) {
  APX_HIP(hipMemcpyAsync(th->pinned->wait, th->dev.wait, sizeof(th->pinned->wait), hipMemcpyDeviceToHost, ctx->stream));
  APX_HIP(hipStreamSynchronize(ctx->stream));
  out[0] = out[1] = out[2] = 0;
  for (const TeamWait& w : th->pinned->wait) {
    out[0] += w.waits;
    out[1] = std::max<int64_t>(out[1], w.max_polls);
    out[2] = std::max<int64_t>(out[2], (int64_t)w.max_ticks);
  }
  return APX_OK;
}
// the failure in words: which wave gave up, on what, and what it saw
inline std::string team_describe(const TeamDiag& g, const char* which) {
  char b[512];
  std::string s = std::string(which) + ": ";
  if (g.code == kTeamErrTickets) {
    snprintf(b, sizeof b, "XCD %d handed out ticket %d of launch %u (workgroup %d): more than 32 workgroups of one launch on an XCD — the "
             "GPU is partitioned, or an earlier launch of this problem did not run all its workgroups. Tickets per XCD now:",
             g.xcc, g.ticket, g.launch, g.block);
    s += b;
  } else {
    snprintf(b, sizeof b, "a hand-off wait expired after %.1f ms and %d polls (bound %.0f ms): launch %u, team %d member %d of %d "
             "(workgroup %d, XCD %d, ticket %d), its row number %lld (step %lld of %lld, %d rows of slack), expected sequence word %llu. "
             "Records of that row as last read — member:state —",
             (double)g.ticks * 1e-5, g.polls, (double)g.limit * 1e-5, g.launch, g.team, g.member, g.members, g.block, g.xcc, g.ticket,
             g.row, g.row + g.depth, g.nsteps, g.depth, g.want);
    s += b;
    for (int m = 0; m < g.members && m < kTeamMaxMembers; ++m) {
      const unsigned long long tag = g.seen[m].v ^ g.seen[m].chk;
      if (tag == g.want) snprintf(b, sizeof b, " %d:ok", m);
      else if (tag > g.seq0 && tag <= g.seq0 + (unsigned long long)g.nsteps) snprintf(b, sizeof b, " %d:row %lld", m, (long long)(tag - g.seq0 - 1));
      else snprintf(b, sizeof b, " %d:none(tag %llx)", m, tag);
      s += b;
    }
    s += ". Tickets per XCD:";
  }
  for (int c = 0; c < 8; ++c) {
    snprintf(b, sizeof b, " %u", g.tickets[c] - g.launch * 32u);
    s += b;
  }
  s += g.code == kTeamErrTickets
           ? " (32 each is a complete launch)."
           : " (32 each: every workgroup of the launch had started ; fewer: some never became resident — the GPU is shared with other "
             "work, or partitioned).";
  s += " The result of this launch is invalid and was discarded ; the problem now runs on the two GEMV passes (apx_ctx_one_pass(ctx, 2) "
       "keeps every new problem off the team kernels).";
  return s;
}

}  // namespace apx
