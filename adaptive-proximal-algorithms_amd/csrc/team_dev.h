// team_dev.h — hand-off state of the workgroup TEAMS of the one-pass kernels for rows too long for one workgroup's
// registers (csrc/team_rows.h k_team_rows: 2 / 4 / 8 / 16 members — f and its gradient, csrc/slab.hip ; the primal-dual loop,
// csrc/slab_pd.hip).
#pragma once
#include <stdint.h>

namespace apx {

struct TeamRec {          // one partial dot product: the value and value ^ sequence (the check word validates the pair,
  unsigned long long v, chk;   // whatever order the two words become visible in — nothing fences them)
};
// Ring of published rows per member.  The bound it must satisfy ("the ring cannot be lapped"): a member
// publishes row q only after it has validated row q - DEPTH of EVERY member, so all members have published row q - DEPTH,
// hence passed the top of step q - DEPTH, where each validated row q - 2 DEPTH: the oldest row anyone can still need when
// row q is written is q - 2 DEPTH + 1, and the slot row q overwrites held row q - kTeamSlots.  Safe while
// kTeamSlots >= 2 DEPTH ; DEPTH <= 3 is asserted in the kernel.
constexpr int kTeamSlots = 8;
constexpr int kTeamMaxMembers = 16;

enum { kTeamErrTimeout = 1, kTeamErrTickets = 2 };

// What the FIRST failing wave of a problem's team launches saw — written once, by the wave whose compare-and-swap set
// `err`, read by the host into the APX_ERR_HIP message (team_describe, csrc/team_rows.h).
struct TeamDiag {
  int32_t code;           // kTeamErr*
  int32_t members;        // P
  int32_t team, member;   // of the wave that gave up (-1: it never got a place: code 2)
  int32_t xcc;            // HW_REG_XCC_ID of its workgroup
  int32_t ticket;         // the ticket it drew in its XCD, relative to the launch's base
  int32_t block;          // blockIdx.x
  int32_t depth;          // rows of slack of the kernel
  uint32_t launch;        // team launches of this problem before this one
  int32_t polls;          // polls of the wait that expired
  long long row;          // the row (relative to the team's slab) whose records were awaited
  long long nsteps;       // steps of the launch
  unsigned long long want;      // the sequence word a valid record of that row carries: seq0 + row + 1
  unsigned long long seq0;
  unsigned long long ticks;     // wall-clock ticks (100 MHz) the wait lasted
  unsigned long long limit;     // ... and the bound it ran into
  uint32_t tickets[8];          // the eight XCDs' ticket counters as the wave saw them when it gave up
  TeamRec seen[kTeamMaxMembers];  // the members' records as last fetched: v ^ chk == want marks a valid one
};

// Longest hand-off wait a workgroup slot has seen (statistics: how close to the bound normal operation comes)
// ... and WHERE and WHEN it saw it (round 4): whether the long waits sit at a launch's first rows or in mid-stream, and whether
// many slots have theirs at the same moment (the whole queue was not running: every polling wave sees the gap) or only the
// partners of one member (that member's wave was late) is what tells a preempted queue from a slow workgroup — tools/soak.py.
struct TeamWait {
  unsigned long long max_ticks;
  unsigned long long at_clock;  // s_memrealtime (10 ns ticks) when the longest wait ended
  unsigned max_polls, waits;    // waits: hand-offs that were not there at the first look
  unsigned long_waits;          // of them, those that took more than 100 us
  unsigned at_launch;           // the problem's team launch in which the longest wait happened
  int32_t at_step;              // ... the row of the team's deal it waited for
  int32_t team, member, xcc;    // ... and who the slot's workgroup was in that launch
};

// How the rows of A are dealt to the workgroups (teams) of a one-pass launch.  Round 2 gave each a SLAB of consecutive rows: the
// 256 workgroups then stream 256 places of the matrix a fixed distance apart for the whole launch, and how those fall on the HBM
// channels depends on where the allocation landed — the same binary on the same box ran C5's kernel at 1.20 or at 1.31 ms from
// one process to the next, C2's at 6.13 or 6.32 (profiles/r03_row_deal.md).  Dealing single rows cyclically (the chip reads one
// window of consecutive rows) cured that and cost C2 5 %: every step of a team then starts on new pages.  CHUNKS of 2^chunk_log
// consecutive rows dealt cyclically have both — a team stays on its pages for a chunk, the teams stay within one window of the
// matrix: C2 6.09 ms in every process.  deal_rows / deal_row: the number of rows of a block, and the matrix row of its k-th.
__host__ __device__ inline int deal_rows(int m, int block, int rows_per_block, int stride, int lr) {
  if (stride == 1) {
    const int b = block * rows_per_block;
    const int e = b + rows_per_block < m ? b + rows_per_block : m;
    return e > b ? e - b : 0;
  }
  const int R = 1 << lr, chunks = (m + R - 1) >> lr;
  const int mine = block < chunks ? (chunks - block + stride - 1) / stride : 0;
  if (mine <= 0) return 0;
  const int last = (mine - 1) * stride + block;                      // its last chunk (the matrix's last one may be short)
  const int tail = m - (last << lr);
  return ((mine - 1) << lr) + (tail < R ? tail : R);
}
__host__ __device__ inline int deal_row(int k, int nrows, int block, int rows_per_block, int stride, int lr) {
  const int kk = k < nrows ? (k > 0 ? k : 0) : (nrows > 0 ? nrows - 1 : 0);      // (clamped into the block's rows)
  if (stride == 1) return block * rows_per_block + kk;
  return (block << lr) + (((kk >> lr) * stride) << lr) + (kk & ((1 << lr) - 1));
}
// The chunk for a row length: about 4 MB of consecutive rows (a 2 MB page and some), but at least 48 chunks per block — the blocks'
// row counts differ by up to one chunk, and that imbalance is paid in full: C3's kernel (391 rows per block) 0.580 - 0.594 ms with
// chunks of 8 rows, 0.598 - 0.600 with slabs, 0.608 - 0.628 with chunks of 32, 0.643 - 0.657 with 64 ; C5's (391 rows per team)
// 1.22 with 4 or 8, 1.34 with 32 on the box where slabs gave 1.28 - 1.31 ; C2's (3125 rows per team) 6.06 - 6.10 ms with 4, 8,
// 16 or 64 against 6.13 - 6.36 for slabs and 6.53 for single rows (tools/ab.py, profiles/r03_row_deal.md).
inline int deal_chunk_log(int64_t row_bytes, int64_t rows_per_block) {
  int lr = 0;
  while (((int64_t)1 << lr) * row_bytes < ((int64_t)4 << 20) && ((int64_t)1 << (lr + 1)) * 48 <= rows_per_block) lr += 1;
  return lr;
}

struct TeamDev {
  TeamRec* ring;          // [teams][kTeamSlots][members]
  unsigned* tickets;      // [8] one counter per XCD, monotonic over the launches
  int* err;               // != 0, the FIRST failure (kTeamErr*) ; sticky: every later launch returns at once
  TeamDiag* diag;
  TeamWait* wait;         // [256] by blockIdx.x
  unsigned long long seq0;   // sequence base of this launch (the ring is never reset)
  unsigned long long wait_ticks;   // bound of a hand-off wait, wall-clock ticks of 10 ns
  unsigned launch;        // launches so far: the ticket base is launch * 32
  int32_t rows_per_team;
  int32_t row_stride;     // 1: a team's rows are a slab of rows_per_team consecutive rows ; T (= teams): chunks dealt cyclically,
  int32_t chunk_log;      //    2^chunk_log consecutive rows each (0: every T-th row)
  int32_t interleave;     // 1: the members' slices of a row are INTERLEAVED in pieces of one workgroup's load (member s holds pieces
                          //    s, s + P, s + 2 P, ... of BSC vectors) ; 0: member s holds the s-th contiguous P-th of the row (round 2)
  int32_t fault_grid;     // TEST hook (APX_OPT_TEAM_FAULT_GRID): launch this many workgroups instead of all 256
  int32_t pd_defer;       // the primal-dual form: residual sums and member 0's stores 64 rows at a time, lane-parallel (APX_OPT_TEAM_PD_DEFER)
};

}  // namespace apx
