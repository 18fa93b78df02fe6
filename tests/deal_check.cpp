// deal_check.cpp — CPU check of how the rows of A are dealt to the workgroups (teams) of a one-pass launch
// (csrc/team_dev.h deal_rows / deal_row / deal_chunk_log: the same functions the kernels call).  Built with g++ by
// tests/test_row_deal.py: every row of the matrix must be dealt to exactly one block, in increasing order within a block,
// and a block's row numbers past its last row must clamp into its own rows (the kernels' dummy steps load those addresses
// with zero-byte descriptors, but compute them).
#include <cstdio>
#include <cstdlib>
#include <vector>

#define __host__
#define __device__
#include "../adaptive-proximal-algorithms_amd/csrc/team_dev.h"

using namespace apx;

static int check(int m, int blocks, int stride_mode, int lr) {
  const int rpb = (m + blocks - 1) / blocks;
  const int stride = stride_mode ? blocks : 1;
  std::vector<int> owner((size_t)m, -1);
  long long total = 0;
  int max_rows = 0, min_rows = 1 << 30;
  for (int b = 0; b < blocks; ++b) {
    const int n = deal_rows(m, b, rpb, stride, lr);
    total += n;
    if (n > max_rows) max_rows = n;
    if (n < min_rows) min_rows = n;
    int prev = -1;
    for (int k = 0; k < n; ++k) {
      const int r = deal_row(k, n, b, rpb, stride, lr);
      if (r < 0 || r >= m) { printf("FAIL m=%d blocks=%d stride=%d lr=%d: block %d row %d -> %d out of range\n", m, blocks, stride, lr, b, k, r); return 1; }
      if (owner[(size_t)r] != -1) { printf("FAIL m=%d blocks=%d stride=%d lr=%d: row %d dealt twice (blocks %d and %d)\n", m, blocks, stride, lr, r, owner[(size_t)r], b); return 1; }
      if (r <= prev) { printf("FAIL m=%d blocks=%d stride=%d lr=%d: block %d not increasing at %d\n", m, blocks, stride, lr, b, k); return 1; }
      owner[(size_t)r] = b;
      prev = r;
    }
    if (n > 0) {      // clamping: negative and past-the-end row numbers stay inside the block's rows
      const int lo = deal_row(-3, n, b, rpb, stride, lr), hi = deal_row(n + 5, n, b, rpb, stride, lr);
      if (lo != deal_row(0, n, b, rpb, stride, lr) || hi != deal_row(n - 1, n, b, rpb, stride, lr)) { printf("FAIL clamp m=%d blocks=%d lr=%d block %d\n", m, blocks, lr, b); return 1; }
    }
  }
  if (total != m) { printf("FAIL m=%d blocks=%d stride=%d lr=%d: %lld rows dealt\n", m, blocks, stride, lr, total); return 1; }
  // the chunked deal's imbalance is at most one chunk
  if (stride_mode && max_rows - min_rows > (1 << lr)) { printf("FAIL imbalance m=%d blocks=%d lr=%d: %d .. %d\n", m, blocks, lr, min_rows, max_rows); return 1; }
  return 0;
}

int main() {
  int cases = 0;
  const int ms[] = {4096, 4097, 4101, 9800, 50000, 100000, 140001, 400000};
  const int bs[] = {16, 32, 64, 128, 255, 256, 512};
  for (int m : ms)
    for (int b : bs)
      for (int lr = 0; lr <= 7; ++lr) {
        if (((m + b - 1) / b) < (1 << lr)) continue;      // (a chunk larger than a block's share is never chosen: deal_chunk_log)
        if (check(m, b, 1, lr)) return 1;
        cases++;
      }
  for (int m : ms)
    for (int b : bs) {
      if (check(m, b, 0, 0)) return 1;
      cases++;
    }
  // the rule: ~4 MB of rows, at least 48 chunks per block
  struct { long long row_bytes, rpb; int want; } rule[] = {{800000, 3125, 3}, {160000, 391, 3}, {40000, 391, 3}, {800, 27344, 9}, {4 << 20, 4096, 0}, {100000, 40, 0}};
  for (auto& r : rule)
    if (deal_chunk_log(r.row_bytes, r.rpb) != r.want) { printf("FAIL deal_chunk_log(%lld, %lld) = %d, expected %d\n", r.row_bytes, r.rpb, deal_chunk_log(r.row_bytes, r.rpb), r.want); return 1; }
  printf("OK %d cases\n", cases);
  return 0;
}
