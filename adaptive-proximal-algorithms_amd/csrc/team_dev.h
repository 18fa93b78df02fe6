// team_dev.h — hand-off state of the workgroup TEAMS of the one-pass kernels for rows too long for one workgroup's
// registers (csrc/team_rows.h k_team_rows: 2 / 4 / 8 / 16 members — f and its gradient, csrc/slab.hip ; the primal-dual loop,
// csrc/slab_pd.hip).
#pragma once
#include <stdint.h>

namespace apx {

struct TeamRec {          // one partial dot product: the value and value ^ sequence (the check word validates the pair,
  unsigned long long v, chk;   // whatever order the two words become visible in — nothing fences them)
};
constexpr int kTeamSlots = 8;      // ring of published rows per member (a member is at most DEPTH + 1 rows ahead of another)
struct TeamDev {
  TeamRec* ring;          // [teams][kTeamSlots][members]
  unsigned* tickets;      // [8] one counter per XCD, monotonic over the launches
  int* err;               // != 0, the FIRST failure: a wait timed out (1) / an XCD received more than 32 workgroups of a launch (2)
  unsigned long long seq0;   // sequence base of this launch (the ring is never reset)
  unsigned launch;        // launches so far: the ticket base is launch * 32
  int32_t rows_per_team;
};

}  // namespace apx
