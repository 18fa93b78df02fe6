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
struct TeamWait {
  unsigned long long max_ticks;
  unsigned max_polls, waits;    // waits: hand-offs that were not there at the first look
};

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
  int32_t fault_grid;     // TEST hook (APX_OPT_TEAM_FAULT_GRID): launch this many workgroups instead of all 256
};

}  // namespace apx
