// sparse.hip — compressed sparse storage for the design matrix: the
// reference's sparse_logreg / dual_svm / LAD experiments load LIBSVM files into a
// SparseMatrixCSC (experiments/libsvm.jl:3-61) and multiply with it (`f.X * w`, `f.X' * r`,
// experiments/sparse_logreg/runme.jl:24,31).
//
// Both orientations are stored as compressed ROWS (A and A'), built once at upload by a
// stable counting sort, so that both products are gather-style row dot products, summed in a
// fixed order.  No scatter, no atomics => bit-reproducible like the dense path.  Output conventions
// are those of the dense kernels, so every fused epilogue (residual, logistic link, ...) is
// shared: spmv_n writes fp64 "splits" into the partial buffer, spmv_t writes T.
// Four kernels: spmv_tiles (round 5, matrices of a million entries or more: the operand's windows staged in LDS, the entries tile by
// tile, the rows' sums in LDS — 4.7 - 5.1 TB/s of the entries' bytes), spmv_stream (round 5, smaller matrices whose rows have a few
// entries each: a block of rows' entries as one run, products through LDS, eight lanes per row — with the gathered operand taken
// in L2 WINDOWS where it is large, launch_side), spmv_rows (round 1: G lanes per row ; short rows, unsorted input, the A/B option)
// and spmv_long (few rows of thousands of entries: chunks).
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <vector>

#include "apx_internal.h"
#include "gemv_kernels.h"

#include "small_dev.h"
#include "solver_internal.h"

namespace apx {

// A group of G lanes per row; every group works on R rows at once (row, row + groups, ...): their pointer
// pairs are loaded together, then the first U strides of all R rows (R*U independent index / value loads
// per lane), then the R*U gathers; rows longer than U*G entries finish in a 4-way unrolled remainder loop.
// The summation order per row is fixed (stride-major, then the width-G shuffle tree): bit-reproducible.
// Measured (tools/sparse_bench.py, profiles/r01f_sparse_bench.jsonl): LIBSVM-sized matrices are cache-resident and
// the iteration is launch-bound (26 - 35 us; the smallest ones run on a dense copy instead, small_dense_view below);
// rows of thousands of entries take the chunked form (spmv_long); a 60 M-entry matrix with a 100 000-long gathered
// operand runs at
// 1.1 - 1.7 TB/s of value + index bytes whatever the loads in flight — every random 8-byte gather moves a
// whole cache line from L2, which is what bounds it; an operand that fits LDS / a row-blocked layout is the
// next step for that regime, not this kernel's schedule (round 5: spmv_tiles below).
// (p0 / p1: where a row's entries begin and end — the row pointers `ptr` and `ptr + 1`, or two consecutive slabs of a window
//  table: the entries of the row whose index falls into one L2 window of the operand, see launch_windows)
template <typename T, typename OUT, int G, int U = 2, int RU = 4>
__global__ __launch_bounds__(kBlock) void spmv_rows(const int64_t* __restrict__ p0, const int64_t* __restrict__ p1,
                                                    const int32_t* __restrict__ ind,
                                                    const T* __restrict__ val, const T* __restrict__ x,
                                                    OUT* __restrict__ out, int64_t nrows, int64_t nrows_pad,
                                                    const int* __restrict__ done) {
  if (done != nullptr && *done != 0) return;
  constexpr int R = 4;
  const int lane = threadIdx.x % G;
  const int64_t groups = (int64_t)gridDim.x * (kBlock / G);
  // (the pointer pairs of the NEXT batch of rows are requested while this batch's entries are in flight: one dependent trip to
  //  memory less per batch — round 5)
  int64_t n0[R], n1[R];
  auto ptrs = [&](int64_t b, int64_t (&a0)[R], int64_t (&a1)[R]) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int64_t row = b + r * groups;
      const bool ok = row < nrows;
      a0[r] = ok ? p0[row] : 0;
      a1[r] = ok ? p1[row] : 0;
    }
  };
  const int64_t base0 = (int64_t)blockIdx.x * (kBlock / G) + threadIdx.x / G;
  if (base0 < nrows_pad) ptrs(base0, n0, n1);
  for (int64_t base = base0; base < nrows_pad; base += R * groups) {
    int64_t k0[R], k1[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      k0[r] = n0[r];
      k1[r] = n1[r];
    }
    int32_t j[R][U];
    double v[R][U], xv[R][U], s[R];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t kk = k0[r] + lane + (int64_t)u * G;
        const bool ok = kk < k1[r];
        j[r][u] = ok ? ind[kk] : 0;
        v[r][u] = ok ? (double)val[kk] : 0.0;
      }
    if (base + R * groups < nrows_pad) ptrs(base + R * groups, n0, n1);
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int u = 0; u < U; ++u) xv[r][u] = (double)x[j[r][u]];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      s[r] = 0.0;
#pragma unroll
      for (int u = 0; u < U; ++u) s[r] = fma(v[r][u], xv[r][u], s[r]);
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      for (int64_t k = k0[r] + lane + (int64_t)U * G; k < k1[r]; k += RU * G) {     // long rows
        int32_t jj[RU];
        double vv[RU], xx[RU];
#pragma unroll
        for (int u = 0; u < RU; ++u) {
          const int64_t kk = k + (int64_t)u * G;
          const bool ok = kk < k1[r];
          jj[u] = ok ? ind[kk] : 0;
          vv[u] = ok ? (double)val[kk] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < RU; ++u) xx[u] = (double)x[jj[u]];
#pragma unroll
        for (int u = 0; u < RU; ++u) s[r] = fma(vv[u], xx[u], s[r]);
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      double t = s[r];
#pragma unroll
      for (int off = G / 2; off > 0; off >>= 1) t += __shfl_down(t, off, G);
      const int64_t row = base + r * groups;
      if (lane == 0 && row < nrows_pad) out[row] = (OUT)t;     // rows in [nrows, nrows_pad) get exact zeros
    }
  }
}

// (U: strides of G entries a group takes of its rows before the remainder loop — 2 by default ; 3 where the rows are long enough for
//  64 lanes and their lengths spread beyond 2 G (avg + 3 sqrt(avg) entries: a Poisson row of 150 entries in a window, `avg` given),
//  so that almost no row enters the four-way unrolled remainder, most of whose lanes would be masked)
template <typename T, typename OUT>
static void launch_rows_pp(apx_ctx* ctx, int G, const int64_t* p0, const int64_t* p1, const int32_t* ind, const void* val, const void* x,
                           OUT* out, int64_t nrows, int64_t nrows_pad, const int* done, double avg = 0.0) {
  const int64_t want = (nrows_pad * G + kBlock - 1) / kBlock;
  const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(want, 16 * (int64_t)ctx->num_cu));
  const bool u3 = ctx->sparse_windows != 0 && G >= 32 && avg + 3.0 * std::sqrt(std::max(avg, 0.0)) > 2.0 * G;
  // (measured and not taken, round 5: half the lanes with four strides and a remainder that advances one stride at a time — 94 % of the
  //  lane slots filled instead of 62 %: X w 429 -> 419 us, X' r 462 -> 449, Float32 X' r 377 -> 399: the kernel is not bound by its lane slots)
#define APX_SPMV(GG)                                                                                      \
  if (u3 && GG >= 32)                                                                                     \
    hipLaunchKernelGGL((spmv_rows<T, OUT, (GG >= 32 ? GG : 32), 3>), dim3(grid), dim3(kBlock), 0, ctx->stream, p0, p1, ind, (const T*)val, \
                       (const T*)x, out, nrows, nrows_pad, done);                                          \
  else                                                                                                    \
    hipLaunchKernelGGL((spmv_rows<T, OUT, GG>), dim3(grid), dim3(kBlock), 0, ctx->stream, p0, p1, ind, (const T*)val, \
                       (const T*)x, out, nrows, nrows_pad, done)
  switch (G) {
    case 1: APX_SPMV(1); break;
    case 2: APX_SPMV(2); break;
    case 4: APX_SPMV(4); break;
    case 8: APX_SPMV(8); break;
    case 16: APX_SPMV(16); break;
    case 32: APX_SPMV(32); break;
    default: APX_SPMV(64); break;
  }
#undef APX_SPMV
}
template <typename T, typename OUT>
static void launch_rows(apx_ctx* ctx, int G, const int64_t* ptr, const int32_t* ind, const void* val, const void* x,
                        OUT* out, int64_t nrows, int64_t nrows_pad, const int* done, double avg = 0.0) {
  launch_rows_pp<T, OUT>(ctx, G, ptr, ptr + 1, ind, val, x, out, nrows, nrows_pad, done, avg);
}

// The STREAMING form (round 5): the entries of a block of consecutive rows — up to kStreamTile of them, whole rows only — are
// loaded without regard to the rows (every lane of every load instruction busy, the loads perfectly coalesced), multiplied with
// the gathered operand and left in LDS ; then eight lanes per row add each row's run of products in a fixed order.  One barrier
// per tile, no pointer-dependent load in front of the stream.  For matrices whose rows all fit a tile (row blocks built at upload).
// (the tile: 4096 entries for Float64 — 32 KB of products, sixteen loads per lane in flight: X w 332 -> 298 us, X' r 369 -> 343 against
//  2048, and 6144 no better: 299 / 375 ; Float32 keeps 2048: with 4096 its products 300 -> 360 and 320 -> 375 us — measured, one build each)
template <typename T>
struct StreamU { static constexpr int value = sizeof(T) == 8 ? 16 : 8; };
static int64_t stream_tile(int32_t dtype) { return (int64_t)kBlock * (dtype == APX_F32 ? StreamU<float>::value : StreamU<double>::value); }
template <typename T, typename OUT>
__global__ __launch_bounds__(kBlock) void spmv_stream(const int32_t* __restrict__ blk, const int64_t* __restrict__ ptr,
                                                      const int32_t* __restrict__ ind, const T* __restrict__ val,
                                                      const T* __restrict__ x, OUT* __restrict__ out, int64_t nrows, int64_t nrows_pad,
                                                      const int* __restrict__ done) {
  constexpr int kStreamU = StreamU<T>::value, kStreamTile = kBlock * kStreamU;
  __shared__ double prod[kStreamTile];
  if (done != nullptr && *done != 0) return;
  const int r0 = blk[blockIdx.x], r1 = blk[blockIdx.x + 1];
  const int64_t k0 = ptr[r0], k1 = ptr[r1];
  if (k1 - k0 > kStreamTile) {                       // ONE row longer than a tile (the host makes it a block of its own): the whole
    double s = 0.0;                                  // workgroup strides over it, then a fixed tree over the 256 partial sums
    for (int64_t k = k0 + threadIdx.x; k < k1; k += kBlock) s = fma((double)val[k], (double)x[ind[k]], s);
    prod[threadIdx.x] = s;
    __syncthreads();
    for (int w = kBlock / 2; w > 0; w >>= 1) {
      if ((int)threadIdx.x < w) prod[threadIdx.x] += prod[threadIdx.x + w];
      __syncthreads();
    }
    if (threadIdx.x == 0) out[r0] = (OUT)prod[0];
    if (blockIdx.x == gridDim.x - 1)
      for (int64_t row = nrows + threadIdx.x; row < nrows_pad; row += kBlock) out[row] = (OUT)0;
    return;
  }
  int32_t j[kStreamU];
  double v[kStreamU];
#pragma unroll
  for (int u = 0; u < kStreamU; ++u) {
    const int64_t kk = k0 + threadIdx.x + (int64_t)u * kBlock;
    const bool ok = kk < k1;
    j[u] = ok ? ind[kk] : 0;
    v[u] = ok ? (double)val[kk] : 0.0;
  }
  // (the rows' bounds for the second phase, asked for while the entries are in flight: eight lanes per row, 32 rows per round)
  const int sub = threadIdx.x & 7, grp = threadIdx.x >> 3;
  int64_t a0 = 0, a1 = 0;
  if (r0 + grp < r1) {
    a0 = ptr[r0 + grp];
    a1 = ptr[r0 + grp + 1];
  }
#pragma unroll
  for (int u = 0; u < kStreamU; ++u) prod[threadIdx.x + u * kBlock] = v[u] * (double)x[j[u]];
  __syncthreads();
  for (int row = r0 + grp; row < r1; row += kBlock / 8) {
    if (row != r0 + grp) {
      a0 = ptr[row];
      a1 = ptr[row + 1];
    }
    double s = 0.0;
    for (int64_t i = a0 - k0 + sub; i < a1 - k0; i += 8) s += prod[i];
    s += __shfl_down(s, 4, 8);
    s += __shfl_down(s, 2, 8);
    s += __shfl_down(s, 1, 8);
    if (sub == 0) out[row] = (OUT)s;
  }
  if (blockIdx.x == gridDim.x - 1)                  // rows in [nrows, nrows_pad) get exact zeros
    for (int64_t row = nrows + threadIdx.x; row < nrows_pad; row += kBlock) out[row] = (OUT)0;
}

// L2 WINDOWS of the gathered operand (round 5).  tools/gather_probe.hip: one 8-byte gather per stored entry runs at 2.0 TB/s of the
// entries' 12 bytes while the operand (0.8 MB) stays in an XCD's L2 and at 1.1 TB/s when it does not (8 MB: X' r of a
// 1 000 000-row X).  The entries of a compressed row are sorted by index, so the product can be run WINDOW BY WINDOW of the
// operand: launch w takes, of every row, the run of entries whose index lies in window w (two consecutive slabs of the window
// table are its begin / end pointers) and leaves an fp64 partial per row — one "split" per window, summed in window order by
// the consumers that sum the column splits of the dense path (the row-finishing kernels, sum_partials_rows): fixed order, no
// atomics.  The launches follow each other on the stream, so the chip gathers from ONE window at a time.
constexpr int64_t kWindowBytes = (int64_t)1 << 21;       // 2 MB of operand per window: half an XCD's L2
static int64_t window_elems(int32_t dtype) { return kWindowBytes / (int64_t)dtype_size(dtype); }

// one orientation's product: window by window (W >= 1 slabs of nrows + 1 pointers), or the whole rows (win == nullptr: ptr) ;
// each window by the streaming form where it has row blocks, else by the lanes-per-row kernel.  OUT: double partials into `out`
// ([W][nrows_pad], one split per window) — or, without windows, whatever the caller's output type is.
template <typename T, typename OUT>
static void launch_side(apx_ctx* ctx, int W, const int64_t* win, const int64_t* ptr, const int32_t* blk, const std::vector<int32_t>& blk_off,
                        const int32_t* ind, const void* val, const void* x, OUT* out, int64_t nnz, int64_t nrows, int64_t nrows_pad,
                        const int* done) {
  const int nw = win != nullptr ? W : 1;
  const double avg = nrows > 0 ? (double)nnz / (double)nrows / (double)nw : 0.0;
  int g = 1;
  while (g < 64 && g * 2 <= avg) g *= 2;
  for (int w = 0; w < nw; ++w) {
    const int64_t* p = win != nullptr ? win + (int64_t)w * (nrows + 1) : ptr;
    OUT* o = out + (int64_t)w * nrows_pad;
    const int nb = blk != nullptr && (int)blk_off.size() > w + 1 ? blk_off[w + 1] - blk_off[w] - 1 : 0;
    if (nb > 0)
      hipLaunchKernelGGL((spmv_stream<T, OUT>), dim3((unsigned)nb), dim3(kBlock), 0, ctx->stream, blk + blk_off[w], p, ind, (const T*)val,
                         (const T*)x, o, nrows, nrows_pad, done);
    else
      launch_rows_pp<T, OUT>(ctx, g, p, p + 1, ind, val, x, o, nrows, nrows_pad, done, avg);
  }
}

// LDS TILES (round 5, after tools/gather_probe.hip's last rows: with the operand in LDS and 16 waves on the CU the entries stream at
// 5.4 - 5.9 TB/s of their 12 bytes, against 2.0 when every gather is an L2 transaction).  The orientation is cut into tiles of
// (up to) kTileRows rows x one window of the gathered operand (64 KB or somewhat less) ; a tile's entries lie together, sorted by row, each with its row and index
// LOCAL to the tile in one 32-bit word, padded to whole chunks of 512.  A workgroup of 1024 lanes owns a row block and a run of its
// windows (a work item): per window it stages the operand's window in LDS — asked for a tile ahead, held in registers meanwhile — and
// its 16 waves take the tile's chunks in turn (wave v: chunks v, v + 16, ... — the workgroup reads one contiguous run ; with a run per
// wave, 4000 streams on the chip, the same kernel was 10 % slower), a lane EIGHT consecutive entries per chunk (six 16-byte loads, the
// next chunk's in flight).  The rows' sums go to an accumulator in LDS that lives across the windows:
//  * a row that begins and ends inside one lane's entries is added by that lane ;
//  * rows that cross lanes are summed by a segmented scan over the wave (data-parallel-primitive moves: four row shifts, two row
//    broadcasts — no trip through LDS ; the segments from one ballot) ;
//  * the row a chunk ends in may go on in the next chunk, another wave's: it is left in a table, an entry per chunk, and after the
//    tile's barrier the first entry of every run of equal rows adds the run in chunk order.
// Every row's sum has ONE owner and a fixed shape: bit-reproducible (the LDS adds are to addresses no other lane touches between
// two barriers).  Output: fp64 splits [splits][nrows_pad] like the windows' — the runs of windows of a row block are its splits, the
// consumers sum them in order.
// Measured (tools/sparse_kernels.py, 1 M x 100 k, 60 M entries): X w 298 -> 137 us, X' r 343 -> 145 (5.0 - 5.2 TB/s of the entries'
// bytes) ; Float32 306 -> 105, 328 -> 110.  Along the way: a run of chunks per wave with the crossing rows in registers 160 us ; the lane's
// runs with branches and a guarded product for the padding 148 / 147 (now: selects, and padding that points at a zero) ; four entries
// per lane 144 / 149 but Float32 130 ; two register sets taken in turn instead of copying the loaded chunk 152 / 154 (128 registers and
// a spill) ; the scan through ds_bpermute, read-modify-write adds instead of ds_add_f64: no difference.  What is left below the probe's
// 5.4 - 5.9 TB/s is the tile's barrier, which waits for the last round of chunks — hence the window's width (plan_tiles).
constexpr int kTileLanes = 1024, kTileWaves = kTileLanes / 64, kTileE = 8, kTileChunk = 64 * kTileE;
constexpr int kTileRows = 8192, kTileWindowBytes = 65536;
// an entry's word: its local row << 17 | the BYTE offset of its operand entry in the window (0 ... 65536: one slot past the window holds a
// zero — the padding entries, whose values are zero too, point there: no product to guard, whatever the operand holds)
constexpr int kTileRowShift = 17, kTileXwBytes = kTileWindowBytes + 16;
template <typename T>
struct TileCols { static constexpr int value = kTileWindowBytes / (int)sizeof(T); };
constexpr int kTileItemInts = 8;         // a work item: row block, split, first / one past the last window, splits to zero [z0, z1), first row, rows
constexpr int kTileTable = kTileLanes;   // chunks of a tile between two barriers: every chunk leaves its last row in a table, an entry per lane
constexpr size_t kTileLds = (size_t)kTileXwBytes + (size_t)kTileRows * 8 + (size_t)kTileTable * 12;

// (an address has one owner between two barriers, and a wave's LDS operations execute in order: the add needs no return and no wait)
__device__ __forceinline__ void tile_add(double* acc, int row, double v) { unsafeAtomicAdd(acc + row, v); }

// data-parallel-primitive moves (VALU, no trip through the LDS crossbar): lanes without a source keep `old`
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ int dpp_i(int old, int v) { return __builtin_amdgcn_update_dpp(old, v, CTRL, ROW_MASK, 0xf, false); }
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ double dpp_d(double old, double v) {
  const long long o = __double_as_longlong(old), q = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_update_dpp((int)o, (int)q, CTRL, ROW_MASK, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp((int)(o >> 32), (int)(q >> 32), CTRL, ROW_MASK, 0xf, false);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
constexpr int kDppRowShr1 = 0x111, kDppRowShr2 = 0x112, kDppRowShr4 = 0x114, kDppRowShr8 = 0x118, kDppWaveShl1 = 0x130, kDppWaveShr1 = 0x138,
              kDppRowBcast15 = 0x142, kDppRowBcast31 = 0x143;

template <typename T>
__global__ __launch_bounds__(kTileLanes) void spmv_tiles(const int32_t* __restrict__ items, const int32_t* __restrict__ off,
                                                         const int32_t* __restrict__ ind, const T* __restrict__ val,
                                                         const T* __restrict__ x, double* __restrict__ part, int64_t ncols_pad,
                                                         int64_t nrows_pad, int nw, int wcols, const int* __restrict__ done) {
  extern __shared__ __align__(16) unsigned char tile_lds[];
  if (done != nullptr && *done != 0) return;
  constexpr int kCols = TileCols<T>::value, kPer = 16 / (int)sizeof(T), kVecs = kCols / kPer / kTileLanes;      // 4 vectors per lane
  typedef T V __attribute__((ext_vector_type(16 / sizeof(T))));
  T* xw = (T*)tile_lds;
  double* acc = (double*)(tile_lds + kTileXwBytes);
  double* tsum = acc + kTileRows;
  int* trow = (int*)(tsum + kTileTable);
  const int32_t* it = items + (int64_t)blockIdx.x * kTileItemInts;
  const int rb = it[0], split = it[1], w1 = it[3], z0 = it[4], z1 = it[5];
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int32_t* toff = off + (int64_t)rb * nw;
  auto next_tile = [&](int w) {                        // (uniform: the first tile at or after w that has entries)
    while (w < w1 && toff[w] == toff[w + 1]) ++w;
    return w;
  };
  // the operand's next window: asked for a tile ahead, kept in registers, stored between the tile's two barriers
  V xr[kVecs];
  auto window_load = [&](int w) {
    const int64_t base = (int64_t)w * wcols;
#pragma unroll
    for (int q = 0; q < kVecs; ++q) {
      const int i = tid + q * kTileLanes;
      V v = {};
      if (base + (int64_t)(i + 1) * kPer <= ncols_pad) v = *(const V*)(x + base + (int64_t)i * kPer);
      xr[q] = v;
    }
  };
  auto window_store = [&]() {
#pragma unroll
    for (int q = 0; q < kVecs; ++q) *(V*)(xw + (tid + q * kTileLanes) * kPer) = xr[q];
  };
  // a wave's loads of one chunk (a lane: kTileE consecutive entries) — every load instruction 64 lanes x 16 consecutive bytes, the
  // host stores a chunk so
  typedef int J4 __attribute__((ext_vector_type(4)));
  int ja[kTileE] = {};
  double va[kTileE] = {};
  auto load_into = [&](int (&jn)[kTileE], double (&vn)[kTileE], int64_t c) {
#pragma unroll
    for (int q = 0; q < kTileE / 4; ++q) {
      const J4 t = *((const J4*)(ind + c * kTileChunk) + q * 64 + lane);
      jn[4 * q] = t.x; jn[4 * q + 1] = t.y; jn[4 * q + 2] = t.z; jn[4 * q + 3] = t.w;
    }
    if (sizeof(T) == 8) {
      typedef double D2 __attribute__((ext_vector_type(2)));
#pragma unroll
      for (int q = 0; q < kTileE / 2; ++q) {
        const D2 t = *((const D2*)(val + c * kTileChunk) + q * 64 + lane);
        vn[2 * q] = t.x; vn[2 * q + 1] = t.y;
      }
    } else {
      typedef float F4 __attribute__((ext_vector_type(4)));
#pragma unroll
      for (int q = 0; q < kTileE / 4; ++q) {
        const F4 t = *((const F4*)(val + c * kTileChunk) + q * 64 + lane);
        vn[4 * q] = t.x; vn[4 * q + 1] = t.y; vn[4 * q + 2] = t.z; vn[4 * q + 3] = t.w;
      }
    }
  };
  auto load = [&](int64_t c) { load_into(ja, va, c); };
  int w = next_tile(it[2]);
  int64_t cb = 0, c1 = 0;                              // the batch of chunks at hand begins at cb ; the tile ends at c1
  if (w < w1) {
    window_load(w);
    cb = toff[w];
    c1 = toff[w + 1];
    if (cb + wave < c1) load(cb + wave);
  }
  for (int i = tid; i < kTileRows; i += kTileLanes) acc[i] = 0.0;
  if (tid < 4) ((int*)(tile_lds + kTileWindowBytes))[tid] = 0;      // the zero past the window
  if (w < w1) window_store();
  __syncthreads();
  bool fresh = true;                                   // the tile's first batch: ask for the window after it
  int wn = w;
  while (w < w1) {
    if (fresh) {
      wn = next_tile(w + 1);
      if (wn < w1) window_load(wn);
      fresh = false;
    }
    const int64_t ce = std::min<int64_t>(c1, cb + kTileTable);
    for (int64_t c = cb + wave; c < ce; c += kTileWaves) {
      int j[kTileE];
      double p[kTileE];
#pragma unroll
      for (int e = 0; e < kTileE; ++e) {
        j[e] = ja[e];
        p[e] = va[e];
      }
      if (c + kTileWaves < ce) load(c + kTileWaves);
      int r[kTileE];
#pragma unroll
      for (int e = 0; e < kTileE; ++e) {
        r[e] = (int)((unsigned)j[e] >> kTileRowShift);
        p[e] = p[e] * (double)*(const T*)(tile_lds + (j[e] & ((1 << kTileRowShift) - 1)));
      }
      // the lane's own runs, without branches: run[e] the sum of the run entry e ends ; the first run (head) and the last (tail) may go on
      // in the neighbours, the ones between are whole rows — added here where they end
      double run = p[0], hs = p[0];
      bool head = true;
#pragma unroll
      for (int e = 1; e < kTileE; ++e) {
        const bool same = r[e] == r[e - 1];
        if (!same && !head) tile_add(acc, r[e - 1], run);
        run = same ? run + p[e] : p[e];
        head = head && same;
        hs = head ? run : hs;
      }
      const bool single = head;
      const int hr = r[0], tr = r[kTileE - 1];
      const double cs = run;
      const int prev_tr = dpp_i<kDppWaveShr1>(-1, tr);        // (lane 0 keeps -1: the chunk's first row is added as if it began here)
      const int next_fr = dpp_i<kDppWaveShl1>(-1, r[0]);
      const bool start = !single || prev_tr != tr;
      const unsigned long long m = __ballot(start);
      const int seg = 63 - __clzll(m & (~0ull >> (63 - lane)));      // the lane this lane's last run began in
      // segmented inclusive scan of the lanes' last runs: within rows of 16 lanes by row shifts, then lane 15 / 47 to the row above, then
      // lane 31 to the upper half — a fixed tree
      double S = cs;
      {
        double up = dpp_d<kDppRowShr1>(0.0, S);
        if (lane - 1 >= seg && (lane & 15) >= 1) S = up + S;
        up = dpp_d<kDppRowShr2>(0.0, S);
        if (lane - 2 >= seg && (lane & 15) >= 2) S = up + S;
        up = dpp_d<kDppRowShr4>(0.0, S);
        if (lane - 4 >= seg && (lane & 15) >= 4) S = up + S;
        up = dpp_d<kDppRowShr8>(0.0, S);
        if (lane - 8 >= seg && (lane & 15) >= 8) S = up + S;
        up = dpp_d<kDppRowBcast15, 0xa>(0.0, S);
        if ((lane & 16) != 0 && seg < (lane & ~15)) S = up + S;
        up = dpp_d<kDppRowBcast31, 0xc>(0.0, S);
        if (lane >= 32 && seg < 32) S = up + S;
      }
      const double prevS = dpp_d<kDppWaveShr1>(0.0, S);
      if (!single) tile_add(acc, hr, prev_tr == hr ? prevS + hs : hs);
      if (lane < 63) {
        if (next_fr != tr) tile_add(acc, tr, S);
      } else {                                         // the chunk's last row may go on in the next chunk — another wave's: to the table
        trow[c - cb] = tr;
        tsum[c - cb] = S;
      }
    }
    const int n = (int)(ce - cb);
    const bool tile_done = ce == c1;
    // the first chunk of the next batch: in flight across the barriers
    int64_t nb = ce, n1 = c1;
    if (tile_done && wn < w1) {
      nb = toff[wn];
      n1 = toff[wn + 1];
    }
    if ((!tile_done || wn < w1) && nb + wave < n1) load(nb + wave);
    __syncthreads();                                   // the batch's gathers are done ; its chunks' last rows are in the table
    if (tile_done && wn < w1) window_store();
    if (tid < n) {                                     // equal rows are neighbours: the first of a run adds the run in chunk order
      const int row = trow[tid];
      if (tid == 0 || trow[tid - 1] != row) {
        double t = tsum[tid];
        for (int u = tid + 1; u < n && trow[u] == row; ++u) t += tsum[u];
        tile_add(acc, row, t);
      }
    }
    __syncthreads();
    cb = nb;
    c1 = n1;
    if (tile_done) {
      w = wn;
      fresh = true;
    }
  }
  const int64_t row0 = it[6];
  const int nr = it[7];
  for (int i = tid; i < nr; i += kTileLanes) {
    part[(int64_t)split * nrows_pad + row0 + i] = acc[i];
    for (int z = z0; z < z1; ++z) part[(int64_t)z * nrows_pad + row0 + i] = 0.0;
  }
}

template <typename T>
static int32_t launch_tiles(apx_ctx* ctx, const apx_matrix::Tiles& t, const void* x, int64_t ncols_pad, int64_t nrows_pad, const int* done) {
  hipLaunchKernelGGL((spmv_tiles<T>), dim3((unsigned)t.n_items), dim3(kTileLanes), kTileLds, ctx->stream, t.items, t.off, t.ind, (const T*)t.val,
                     (const T*)x, ctx->part, ncols_pad, nrows_pad, t.nw, t.wcols, done);
  return APX_OK;
}

// LONG rows (few rows, thousands of entries each: A' of a tall LIBSVM matrix — covtype is 581 012 x 54 — or A of a
// wide one): a group of lanes per row leaves the chip idle (54 rows = 54 waves: 390 GB/s), so a row is cut into
// chunks of kLongChunk entries, one block per (chunk, row), coalesced index / value loads, eight gathers in flight
// per lane; the chunk totals are fp64 partials summed in a fixed order by the consumer (the row-finishing kernels
// for A x, `sum_partials_rows` for A' r) — still no atomics, still bit-reproducible.
constexpr int kLongU = 8;
constexpr int kLongChunk = kBlock * kLongU;
template <typename T>
__global__ __launch_bounds__(kBlock) void spmv_long(const int64_t* __restrict__ ptr, const int32_t* __restrict__ ind,
                                                    const T* __restrict__ val, const T* __restrict__ x,
                                                    double* __restrict__ part, int64_t nrows, int64_t ld,
                                                    const int* __restrict__ done) {
  if (done != nullptr && *done != 0) return;
  const int64_t row = blockIdx.y;
  double s[1] = {0.0};
  if (row < nrows) {                                    // (rows in [nrows, gridDim.y) get exact zeros)
    const int64_t k0 = ptr[row] + (int64_t)blockIdx.x * kLongChunk, kend = ptr[row + 1];
    if (k0 < kend) {
      int32_t j[kLongU];
      double v[kLongU], xv[kLongU];
#pragma unroll
      for (int u = 0; u < kLongU; ++u) {
        const int64_t kk = k0 + threadIdx.x + (int64_t)u * kBlock;
        const bool ok = kk < kend;
        j[u] = ok ? ind[kk] : 0;
        v[u] = ok ? (double)val[kk] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < kLongU; ++u) xv[u] = (double)x[j[u]];
#pragma unroll
      for (int u = 0; u < kLongU; ++u) s[0] = fma(v[u], xv[u], s[0]);
    }
  }
  block_sums_t0<1>(s);
  if (threadIdx.x == 0) part[(int64_t)blockIdx.x * ld + row] = s[0];
}

// long-row form when the average row has more than 512 entries and the rows are few
static bool long_rows(int64_t nnz, int64_t nrows, int64_t nrows_pad, int64_t maxlen, int* chunks) {
  *chunks = (int)((maxlen + kLongChunk - 1) / kLongChunk);
  return nrows > 0 && nnz / nrows > 512 && nrows_pad <= 8192 && *chunks >= 1 && *chunks <= 4096;
}

template <typename T>
static void launch_long(apx_ctx* ctx, int chunks, const int64_t* ptr, const int32_t* ind, const void* val, const void* x,
                        int64_t nrows, int64_t nrows_pad, const int* done) {
  hipLaunchKernelGGL((spmv_long<T>), dim3((unsigned)chunks, (unsigned)nrows_pad), dim3(kBlock), 0, ctx->stream, ptr, ind,
                     (const T*)val, (const T*)x, ctx->part, nrows, nrows_pad, done);
}

static int group_for(int64_t nnz, int64_t nrows) {
  const double avg = nrows > 0 ? (double)nnz / (double)nrows : 0.0;
  int g = 1;
  while (g < 64 && g * 2 <= avg) g *= 2;      // ~1-2 entries per lane and step
  return g;
}

// (apx_ctx_profile: the algorithmic bytes of a sparse product are its values and indices, nnz * (sizeof(T) + 4) — the gathered
//  operand and the row pointers are excluded, SURVEY 8d's convention for the dense path carried over)
static double spmv_bytes(const apx_matrix* A) { return (double)A->nnz * (double)(dtype_size(A->dtype) + 4); }
static void spmv_name(apx_ctx* ctx, int kind, const apx_matrix* A, bool lng, int g_or_chunks) {
  if (!ctx->profile) return;
  char b[128];
  if (lng) snprintf(b, sizeof b, "spmv_long<%s, %d chunk(s) of %d entries per row>", A->dtype == APX_F32 ? "float" : "double", g_or_chunks, kLongChunk);
  else if (g_or_chunks < 0) snprintf(b, sizeof b, "spmv_rows<%s> in %d L2 windows of the gathered operand (%d launches)", A->dtype == APX_F32 ? "float" : "double", -g_or_chunks, -g_or_chunks);
  else snprintf(b, sizeof b, "spmv_rows<%s, %d lane(s) per row>", A->dtype == APX_F32 ? "float" : "double", g_or_chunks);
  ctx->kname[kind] = b;
}

static void side_name(apx_ctx* ctx, int kind, const apx_matrix* A, int W, const int32_t* blk) {
  if (!ctx->profile) return;
  char b[160];
  const char* t = A->dtype == APX_F32 ? "float" : "double";
  if (W > 0) snprintf(b, sizeof b, "%s<%s> in %d L2 windows of the gathered operand (%d launches)", blk ? "spmv_stream" : "spmv_rows", t, W, W);
  else snprintf(b, sizeof b, "spmv_stream<%s>", t);
  ctx->kname[kind] = b;
}

static void tiles_name(apx_ctx* ctx, int kind, const apx_matrix* A, const apx_matrix::Tiles& t) {
  if (!ctx->profile) return;
  char b[200];
  snprintf(b, sizeof b, "spmv_tiles<%s>: %d workgroups of %d lanes, the operand in LDS windows (%d of %d), row blocks of up to %d, %d split(s)",
           A->dtype == APX_F32 ? "float" : "double", t.n_items, kTileLanes, t.nw, t.wcols, t.rows, t.splits);
  ctx->kname[kind] = b;
}

int32_t launch_spmv_n(apx_ctx* ctx, const apx_matrix* A, const void* x, int* splits_out, const int* done) {
  if (A->rtile.items != nullptr) {                                  // LDS tiles: the runs of windows of a row block are its splits
    APX_TRY(ensure_part(ctx, (size_t)A->rtile.splits * A->ld * sizeof(double)));
    tiles_name(ctx, APX_KSTAT_GEMV_N, A, A->rtile);
    APX_TRY(prof_begin(ctx, APX_KSTAT_GEMV_N, spmv_bytes(A)));
    if (A->dtype == APX_F32) APX_TRY(launch_tiles<float>(ctx, A->rtile, x, A->n_pad, A->ld, done));
    else APX_TRY(launch_tiles<double>(ctx, A->rtile, x, A->n_pad, A->ld, done));
    APX_TRY(prof_end(ctx));
    APX_HIP(hipGetLastError());
    *splits_out = A->rtile.splits;
    return APX_OK;
  }
  int chunks = 1;
  if (long_rows(A->nnz, A->m, A->ld, A->r_maxlen, &chunks)) {      // every chunk is one fp64 split of A x
    APX_TRY(ensure_part(ctx, (size_t)chunks * A->ld * sizeof(double)));
    spmv_name(ctx, APX_KSTAT_GEMV_N, A, true, chunks);
    APX_TRY(prof_begin(ctx, APX_KSTAT_GEMV_N, spmv_bytes(A)));
    if (A->dtype == APX_F32) launch_long<float>(ctx, chunks, A->rptr, A->rind, A->rval, x, A->m, A->ld, done);
    else launch_long<double>(ctx, chunks, A->rptr, A->rind, A->rval, x, A->m, A->ld, done);
    APX_TRY(prof_end(ctx));
    APX_HIP(hipGetLastError());
    *splits_out = chunks;
    return APX_OK;
  }
  if (A->rwin != nullptr || A->rblk != nullptr) {                   // windows of x (one split each) and / or the streaming form
    const int W = A->rwin != nullptr ? A->rwin_count : 0;
    APX_TRY(ensure_part(ctx, (size_t)std::max(W, 1) * A->ld * sizeof(double)));
    side_name(ctx, APX_KSTAT_GEMV_N, A, W, A->rblk);
    APX_TRY(prof_begin(ctx, APX_KSTAT_GEMV_N, spmv_bytes(A)));
    if (A->dtype == APX_F32) launch_side<float, double>(ctx, W, A->rwin, A->rptr, A->rblk, A->rblk_off, A->rind, A->rval, x, ctx->part, A->nnz, A->m, A->ld, done);
    else launch_side<double, double>(ctx, W, A->rwin, A->rptr, A->rblk, A->rblk_off, A->rind, A->rval, x, ctx->part, A->nnz, A->m, A->ld, done);
    APX_TRY(prof_end(ctx));
    APX_HIP(hipGetLastError());
    *splits_out = std::max(W, 1);
    return APX_OK;
  }
  APX_TRY(ensure_part(ctx, (size_t)A->ld * sizeof(double)));
  const int G = group_for(A->nnz, A->m);
  spmv_name(ctx, APX_KSTAT_GEMV_N, A, false, G);
  APX_TRY(prof_begin(ctx, APX_KSTAT_GEMV_N, spmv_bytes(A)));
  if (A->dtype == APX_F32)
    launch_rows<float, double>(ctx, G, A->rptr, A->rind, A->rval, x, ctx->part, A->m, A->ld, done, (double)A->nnz / std::max<int64_t>(A->m, 1));
  else
    launch_rows<double, double>(ctx, G, A->rptr, A->rind, A->rval, x, ctx->part, A->m, A->ld, done, (double)A->nnz / std::max<int64_t>(A->m, 1));
  APX_TRY(prof_end(ctx));
  APX_HIP(hipGetLastError());
  *splits_out = 1;
  return APX_OK;
}

int32_t launch_spmv_t(apx_ctx* ctx, const apx_matrix* A, const void* r, void* out, const int* done) {
  if (A->ttile.items != nullptr) {
    APX_TRY(ensure_part(ctx, (size_t)A->ttile.splits * A->n_pad * sizeof(double)));
    tiles_name(ctx, APX_KSTAT_GEMV_T, A, A->ttile);
    APX_TRY(prof_begin(ctx, APX_KSTAT_GEMV_T, spmv_bytes(A)));
    if (A->dtype == APX_F32) APX_TRY(launch_tiles<float>(ctx, A->ttile, r, A->ld, A->n_pad, done));
    else APX_TRY(launch_tiles<double>(ctx, A->ttile, r, A->ld, A->n_pad, done));
    APX_TRY(prof_end(ctx));
    APX_HIP(hipGetLastError());
    return launch_sum_partials_rows(ctx, A->dtype, A->ttile.splits, A->n_pad, A->n_pad, out, done);
  }
  int chunks = 1;
  if (long_rows(A->nnz, A->n, A->n_pad, A->t_maxlen, &chunks)) {
    APX_TRY(ensure_part(ctx, (size_t)chunks * A->n_pad * sizeof(double)));
    spmv_name(ctx, APX_KSTAT_GEMV_T, A, true, chunks);
    APX_TRY(prof_begin(ctx, APX_KSTAT_GEMV_T, spmv_bytes(A)));
    if (A->dtype == APX_F32) launch_long<float>(ctx, chunks, A->tptr, A->tind, A->tval, r, A->n, A->n_pad, done);
    else launch_long<double>(ctx, chunks, A->tptr, A->tind, A->tval, r, A->n, A->n_pad, done);
    APX_TRY(prof_end(ctx));
    APX_HIP(hipGetLastError());
    return launch_sum_partials_rows(ctx, A->dtype, chunks, A->n_pad, A->n_pad, out, done);
  }
  if (A->twin != nullptr) {                                         // the operand r does not fit an L2 window: a partial per window
    const int W = A->twin_count;
    APX_TRY(ensure_part(ctx, (size_t)W * A->n_pad * sizeof(double)));
    side_name(ctx, APX_KSTAT_GEMV_T, A, W, A->tblk);
    APX_TRY(prof_begin(ctx, APX_KSTAT_GEMV_T, spmv_bytes(A)));
    if (A->dtype == APX_F32) launch_side<float, double>(ctx, W, A->twin, A->tptr, A->tblk, A->tblk_off, A->tind, A->tval, r, ctx->part, A->nnz, A->n, A->n_pad, done);
    else launch_side<double, double>(ctx, W, A->twin, A->tptr, A->tblk, A->tblk_off, A->tind, A->tval, r, ctx->part, A->nnz, A->n, A->n_pad, done);
    APX_TRY(prof_end(ctx));
    APX_HIP(hipGetLastError());
    return launch_sum_partials_rows(ctx, A->dtype, W, A->n_pad, A->n_pad, out, done);
  }
  if (A->tblk != nullptr) {                                         // the streaming form over whole rows, straight into the output
    side_name(ctx, APX_KSTAT_GEMV_T, A, 0, A->tblk);
    APX_TRY(prof_begin(ctx, APX_KSTAT_GEMV_T, spmv_bytes(A)));
    if (A->dtype == APX_F32) launch_side<float, float>(ctx, 0, nullptr, A->tptr, A->tblk, A->tblk_off, A->tind, A->tval, r, (float*)out, A->nnz, A->n, A->n_pad, done);
    else launch_side<double, double>(ctx, 0, nullptr, A->tptr, A->tblk, A->tblk_off, A->tind, A->tval, r, (double*)out, A->nnz, A->n, A->n_pad, done);
    APX_TRY(prof_end(ctx));
    APX_HIP(hipGetLastError());
    return APX_OK;
  }
  const int G = group_for(A->nnz, A->n);
  spmv_name(ctx, APX_KSTAT_GEMV_T, A, false, G);
  APX_TRY(prof_begin(ctx, APX_KSTAT_GEMV_T, spmv_bytes(A)));
  if (A->dtype == APX_F32)
    launch_rows<float, float>(ctx, G, A->tptr, A->tind, A->tval, r, (float*)out, A->n, A->n_pad, done, (double)A->nnz / std::max<int64_t>(A->n, 1));
  else
    launch_rows<double, double>(ctx, G, A->tptr, A->tind, A->tval, r, (double*)out, A->n, A->n_pad, done, (double)A->nnz / std::max<int64_t>(A->n, 1));
  APX_TRY(prof_end(ctx));
  APX_HIP(hipGetLastError());
  return APX_OK;
}

// stable transpose of compressed rows (counting sort): entries of each output row keep the
// order of the input rows => a fixed summation order for A'*y
template <typename T>
static void transpose_rows(int64_t nrows, int64_t ncols, const int64_t* ptr, const int32_t* ind, const T* val,
                           std::vector<int64_t>& tptr, std::vector<int32_t>& tind, std::vector<T>& tval) {
  const int64_t nnz = ptr[nrows];
  tptr.assign((size_t)ncols + 1, 0);
  for (int64_t k = 0; k < nnz; ++k) tptr[(size_t)ind[k] + 1]++;
  for (int64_t j = 0; j < ncols; ++j) tptr[(size_t)j + 1] += tptr[(size_t)j];
  tind.resize((size_t)nnz);
  tval.resize((size_t)nnz);
  std::vector<int64_t> cur(tptr.begin(), tptr.end() - 1);
  for (int64_t i = 0; i < nrows; ++i)
    for (int64_t k = ptr[i]; k < ptr[i + 1]; ++k) {
      const int64_t dst = cur[(size_t)ind[k]]++;
      tind[(size_t)dst] = (int32_t)i;
      tval[(size_t)dst] = val[k];
    }
}

// The LDS-tile form of one orientation (spmv_tiles), built on the host from its compressed rows: the tiles in row-block-major order,
// each padded to whole chunks ; a chunk stored so that every one of a lane's 16-byte loads is one of 64 consecutive ones (lane L's
// entries 8L .. 8L+7: its indices at 4L .. 4L+3 and 256+4L .. , its Float64 values at 2L, 2L+1, 128+2L, ...) ; the work items — a row
// block and a run of its windows each, one per CU, as many per row block as its share of the cost, the largest first — and per row
// block its number of splits.
template <typename T>
struct TilePlan {
  std::vector<int32_t> items, off, ind;
  std::vector<T> val;
  int nw = 0, splits = 0, rows = 0, wcols = 0;
};
template <typename T>
static bool plan_tiles(apx_ctx* ctx, const int64_t* ptr_h, const int32_t* ind_h, const T* val_h, int64_t nrows, int64_t nrows_pad, int64_t ncols,
                       int64_t nnz, bool force, TilePlan<T>& tp) {
  // the window: 64 KB of the operand — or somewhat less, so that the average tile of a full row block is WHOLE rounds of chunks for the
  // 16 waves (the tile's barrier waits for its last round: 74 chunks = 4.6 rounds leave 7.5 % of the wave slots empty, 63 leave 1.5 %)
  int64_t wc = TileCols<T>::value;
  if (ncols > wc && nrows_pad > 0) {
    const double c = (double)nnz * ((double)std::min<int64_t>(kTileRows, nrows_pad) / (double)nrows_pad) * ((double)wc / (double)ncols) / (double)kTileChunk;
    const double k = std::floor((c + 1.2) / kTileWaves);
    if (k >= 1.0 && c > kTileWaves) wc = std::max<int64_t>(wc / 2, (int64_t)((double)wc * (kTileWaves * k - 1.2) / c) / 32 * 32);
  }
  const int64_t NW = std::max<int64_t>(1, (ncols + wc - 1) / wc);
  // row blocks: the accumulator's 8192 rows — fewer where blocks x windows would not give every CU an item (400 000 x 2000: one
  // window, 49 blocks of 8192 rows ; with 1568 rows a block there are 256), and a block ends early where its entries exceed what its
  // windows can be divided into (a block makes at most min(windows, 64) items: the head of a power-law X' — a few thousand features
  // holding half of the entries — becomes several blocks)
  if (nnz <= 0 || nrows <= 0) return false;
  int64_t rbs = kTileRows;
  const int64_t per_block = std::min<int64_t>(NW, 64), cus = std::max(ctx->num_cu, 1);
  const int64_t full = (nrows_pad + rbs - 1) / rbs;
  if (full * per_block < cus) {
    const int64_t want = (cus + per_block - 1) / per_block;
    rbs = std::max<int64_t>(32, std::min<int64_t>(kTileRows, round_up((nrows_pad + want - 1) / want, 32)));
  } else if (full > cus) {      // more blocks than CUs (2 400 000 rows: 293): whole rounds of equal blocks (512 of 4704 rows: X w 96 -> 78 us)
    const int64_t want = (full + cus - 1) / cus * cus;
    rbs = std::max<int64_t>(32, std::min<int64_t>(kTileRows, round_up((nrows_pad + want - 1) / want, 32)));
  }
  // (three quarters of what min(windows, 64) items of average size hold ; a block of one window is not divided at all — its rows are)
  const double cap = per_block >= 2 ? 0.75 * std::max(1.0, (double)nnz * (double)per_block / (double)cus) : 4.0e18;
  std::vector<int64_t> first(1, 0);
  {
    int64_t in_block = 0;
    for (int64_t i = 0; i < nrows_pad; ++i) {
      const int64_t len = i < nrows ? ptr_h[i + 1] - ptr_h[i] : 0;
      if (i > first.back() && (i - first.back() == rbs || (double)(in_block + len) > cap)) {
        first.push_back(i);
        in_block = 0;
      }
      in_block += len;
    }
    first.push_back(nrows_pad);
  }
  const int64_t RB = (int64_t)first.size() - 1;
  if (RB * NW > ((int64_t)1 << 24)) return false;
  // where they pay: a tile streams its entries and stages 64 KB of the operand (from L2) — half a window's worth of entries per tile on
  // average, and a matrix whose products are more than launches.  Measured against the streaming form (X w / X' r, us): 72 309 x 20 958
  // with 3.7 M entries 17 / 17 against 26 / 27 ; 20 242 x 47 236 with 1.5 M (5800 per tile) 13 / 13 against 16 / 16 ; 49 749 x 300
  // with 0.58 M 8 / 11 against 11 / 11 ; 1 000 000 x 1 000 000 with 4 M (265 per tile) 136 / 137 against 69 / 69
  if (!force && (nnz < ((int64_t)1 << 20) || nnz / (RB * NW) < 4096)) return false;
  std::vector<int64_t> cnt((size_t)(RB * NW), 0);
  for (int64_t rb = 0; rb < RB; ++rb) {
    int64_t* c = cnt.data() + rb * NW;
    for (int64_t i = first[(size_t)rb]; i < std::min(nrows, first[(size_t)rb + 1]); ++i)
      for (int64_t k = ptr_h[i]; k < ptr_h[i + 1]; ++k) c[ind_h[k] / wc]++;
  }
  std::vector<int64_t> off((size_t)(RB * NW) + 1, 0);
  for (int64_t t = 0; t < RB * NW; ++t) off[(size_t)t + 1] = off[(size_t)t] + (cnt[(size_t)t] + kTileChunk - 1) / kTileChunk;
  const int64_t chunks = off[(size_t)(RB * NW)];
  if (chunks >= 2147483647LL / 2) return false;
  tp.off.assign(off.begin(), off.end());
  tp.ind.assign((size_t)chunks * kTileChunk, 0);
  tp.val.assign((size_t)chunks * kTileChunk, (T)0);
  std::vector<int64_t> cur((size_t)NW);
  std::vector<int32_t> last((size_t)NW);
  for (int64_t rb = 0; rb < RB; ++rb) {
    std::fill(cur.begin(), cur.end(), 0);
    std::fill(last.begin(), last.end(), 0);
    const int64_t i1 = std::min<int64_t>(nrows, first[(size_t)rb + 1]);
    for (int64_t i = first[(size_t)rb]; i < i1; ++i)
      for (int64_t k = ptr_h[i]; k < ptr_h[i + 1]; ++k) {
        const int64_t w = ind_h[k] / wc, e = cur[(size_t)w]++;
        const int64_t base = (off[(size_t)(rb * NW + w)] + e / kTileChunk) * kTileChunk;
        const int q = (int)(e % kTileChunk), L = q / kTileE, u = q % kTileE;
        const int32_t rl = (int32_t)(i - first[(size_t)rb]);
        const int at4 = (u / 4) * 256 + 4 * L + u % 4, at2 = (u / 2) * 128 + 2 * L + u % 2;      // 16-byte pieces of 4-byte / of 8-byte items
        tp.ind[(size_t)(base + at4)] = (int32_t)(((uint32_t)rl << kTileRowShift) | (uint32_t)((ind_h[k] - w * wc) * (int64_t)sizeof(T)));
        tp.val[(size_t)(base + (sizeof(T) == 8 ? at2 : at4))] = val_h[k];
        last[(size_t)w] = rl;
      }
    for (int64_t w = 0; w < NW; ++w) {                 // padding: zero entries of the tile's last row that point at the zero past the window
      const int64_t t = rb * NW + w, n = cnt[(size_t)t], cap = (off[(size_t)t + 1] - off[(size_t)t]) * kTileChunk;
      for (int64_t e = n; e < cap; ++e) {
        const int q = (int)(e % kTileChunk), L = q / kTileE, u = q % kTileE;
        tp.ind[(size_t)((off[(size_t)t] + e / kTileChunk) * kTileChunk + (u / 4) * 256 + 4 * L + u % 4)] =
            (int32_t)(((uint32_t)last[(size_t)w] << kTileRowShift) | (uint32_t)kTileWindowBytes);
      }
    }
  }
  // work items
  // one item per CU: a workgroup fills a CU (1024 lanes, 140 KB of LDS), so a second round would begin when the first has drained —
  // measured on 1 M x 100 k, 60 M entries: 256 items 148 / 147 us for X w / X' r, 512 items 146 / 151 (Float32 118 / 117 and 110 / 106)
  const int64_t budget = std::max(ctx->num_cu, 1);
  struct Item { int32_t rb, split, w0, w1, z0, z1; int64_t cost; };
  std::vector<Item> its;
  std::vector<int> ngroups((size_t)RB, 1);
  int splits = 1;
  // a tile costs its entries and its two barriers (measured: a row block of 1696 rows against one of 8192 — about 3000 entries' time)
  auto cost = [&](int64_t t) { return cnt[(size_t)t] > 0 ? cnt[(size_t)t] + 3000 : 0; };
  int64_t all = 0;
  for (int64_t t = 0; t < RB * NW; ++t) all += cost(t);
  // the smallest item size with sum over the row blocks of ceil(cost / size) <= budget: one item per CU at most where the row blocks allow
  std::vector<int64_t> rbcost((size_t)RB, 0);
  for (int64_t rb = 0; rb < RB; ++rb)
    for (int64_t w = 0; w < NW; ++w) rbcost[(size_t)rb] += cost(rb * NW + w);
  const int64_t gmax = std::min<int64_t>(NW, 64);
  auto groups_of = [&](int64_t c, int64_t size) { return std::max<int64_t>(1, std::min<int64_t>((c + size - 1) / size, gmax)); };
  int64_t lo = 1, hi = std::max<int64_t>(all, 1);
  while (lo < hi) {
    const int64_t size = lo + (hi - lo) / 2;
    int64_t n = 0;
    for (int64_t rb = 0; rb < RB; ++rb) n += groups_of(rbcost[(size_t)rb], size);
    if (n <= std::max<int64_t>(budget, RB)) hi = size;
    else lo = size + 1;
  }
  for (int64_t rb = 0; rb < RB; ++rb) {
    const int64_t tot = rbcost[(size_t)rb];
    const int64_t G = groups_of(tot, lo);
    int g = 0;
    int64_t cum = 0, w0 = 0, in_group = 0;
    for (int64_t w = 0; w < NW; ++w) {
      cum += cost(rb * NW + w);
      in_group += cost(rb * NW + w);
      const bool last_w = w + 1 == NW;
      if (last_w || (g + 1 < G && (double)cum >= (double)tot * (double)(g + 1) / (double)G)) {
        its.push_back({(int32_t)rb, (int32_t)g, (int32_t)w0, (int32_t)(w + 1), 0, 0, in_group});
        ++g;
        w0 = w + 1;
        in_group = 0;
      }
    }
    ngroups[(size_t)rb] = g;
    splits = std::max(splits, g);
  }
  for (auto& q : its)
    if (q.split + 1 == ngroups[(size_t)q.rb]) {
      q.z0 = ngroups[(size_t)q.rb];
      q.z1 = splits;
    }
  std::stable_sort(its.begin(), its.end(), [](const Item& a, const Item& b) { return a.cost > b.cost; });
  tp.items.reserve(its.size() * kTileItemInts);
  for (const auto& q : its)
    for (int32_t v : {q.rb, q.split, q.w0, q.w1, q.z0, q.z1, (int32_t)first[(size_t)q.rb], (int32_t)(first[(size_t)q.rb + 1] - first[(size_t)q.rb])})
      tp.items.push_back(v);
  tp.nw = (int)NW;
  tp.wcols = (int)wc;
  tp.splits = splits;
  tp.rows = 0;
  for (int64_t rb = 0; rb < RB; ++rb) tp.rows = std::max<int>(tp.rows, (int)(first[(size_t)rb + 1] - first[(size_t)rb]));
  return true;
}

template <typename T>
static int32_t upload_sparse_t(apx_ctx* ctx, apx_matrix* A, int32_t layout, const int64_t* ptr, const int32_t* idx,
                               const T* vals) {
  // `layout` 0: ptr/idx/vals are the rows of A (CSR); 1: the columns of A (CSC = rows of A')
  const int64_t nrows_in = layout == 0 ? A->m : A->n, ncols_in = layout == 0 ? A->n : A->m;
  std::vector<int64_t> tptr;
  std::vector<int32_t> tind;
  std::vector<T> tval;
  transpose_rows<T>(nrows_in, ncols_in, ptr, idx, vals, tptr, tind, tval);
  const int64_t nnz = A->nnz;
  auto up = [&](void** dst, const void* src, size_t bytes) -> int32_t {
    APX_HIP(hipMalloc(dst, std::max<size_t>(bytes, 16)));
    if (bytes) APX_HIP(hipMemcpyAsync(*dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    return APX_OK;
  };
  const void *p0 = ptr, *i0 = idx, *v0 = vals, *p1 = tptr.data(), *i1 = tind.data(), *v1 = tval.data();
  if (layout == 1) { std::swap(p0, p1); std::swap(i0, i1); std::swap(v0, v1); }
  const int64_t* rp = (const int64_t*)p0;
  const int64_t* tp = (const int64_t*)p1;
  for (int64_t i = 0; i < A->m; ++i) A->r_maxlen = std::max(A->r_maxlen, rp[i + 1] - rp[i]);
  for (int64_t j = 0; j < A->n; ++j) A->t_maxlen = std::max(A->t_maxlen, tp[j + 1] - tp[j]);
  // What an orientation gets besides its rows (launch_side): the WINDOW-MAJOR order of its entries with a table of W CSR slabs, where
  // its gathered operand is longer than a window and a half, its rows are not the long-row form's and the indices are sorted within
  // every row (the transposed side always is: a stable counting sort over rows in order ; the caller's side is checked) ; and the ROW
  // BLOCKS of the streaming form, per window, where every row (of that window) fits a tile and the rows are not too short for it.
  struct Side {
    std::vector<int64_t> win;
    std::vector<int32_t> ind, blk, blk_off;
    std::vector<T> val;
    int W = 0;
  };
  auto plan = [&](const int64_t* ptr_h, const int32_t* ind_h, const T* val_h, int64_t nrows, int64_t nrows_pad, int64_t ncols, int64_t maxlen,
                  Side& sd) {
    int chunks = 0;
    if (ctx->sparse_windows == 0 || nnz == 0 || nrows == 0 || nrows >= 2147483647LL || long_rows(nnz, nrows, nrows_pad, maxlen, &chunks)) return;
    const int64_t we = window_elems(A->dtype);
    int W = ncols > we + we / 2 ? (int)((ncols + we - 1) / we) : 0;
    if (W > 64) W = 0;
    if (W > 0)
      for (int64_t i = 0; i < nrows && W > 0; ++i)
        for (int64_t k = ptr_h[i] + 1; k < ptr_h[i + 1]; ++k)
          if (ind_h[k] < ind_h[k - 1]) { W = 0; break; }               // an unsorted row: whole rows, in the caller's order
    const int64_t stride = nrows + 1;
    if (W > 0) {
      // cut[w * stride + i]: first entry of row i with index >= w * we (relative to the row) ; then the window-major offsets
      sd.win.assign((size_t)W * (size_t)stride, 0);
      std::vector<int64_t> cut((size_t)(W + 1) * (size_t)nrows);
      for (int64_t i = 0; i < nrows; ++i) {
        const int32_t* b = ind_h + ptr_h[i];
        const int32_t* e = ind_h + ptr_h[i + 1];
        cut[(size_t)i] = 0;
        for (int w = 1; w < W; ++w) cut[(size_t)w * nrows + i] = std::lower_bound(b, e, (int32_t)std::min<int64_t>((int64_t)w * we, 2147483647LL)) - b;
        cut[(size_t)W * nrows + i] = e - b;
      }
      sd.ind.resize((size_t)nnz);
      sd.val.resize((size_t)nnz);
      int64_t pos = 0;
      for (int w = 0; w < W; ++w) {
        int64_t* slab = sd.win.data() + (size_t)w * stride;
        for (int64_t i = 0; i < nrows; ++i) {
          slab[i] = pos;
          const int64_t a = ptr_h[i] + cut[(size_t)w * nrows + i], e = ptr_h[i] + cut[(size_t)(w + 1) * nrows + i];
          for (int64_t k = a; k < e; ++k, ++pos) {
            sd.ind[(size_t)pos] = ind_h[k];
            sd.val[(size_t)pos] = val_h[k];
          }
        }
        slab[nrows] = pos;
      }
      sd.W = W;
    }
    // row blocks, per window (or over whole rows): <= kStreamTile entries and <= 256 rows each
    const int nw = std::max(W, 1);
    if (nnz / nrows / nw < 4) return;                                   // (rows too short for eight lanes each: the lanes-per-row kernel)
    sd.blk_off.assign(1, 0);
    const int64_t tile = stream_tile(A->dtype);
    for (int w = 0; w < nw; ++w) {
      const int64_t* q = W > 0 ? sd.win.data() + (size_t)w * stride : ptr_h;
      sd.blk.push_back(0);                             // (a row longer than a tile ends up as a block of its own: the kernel strides over it)
      int64_t r0 = 0;
      for (int64_t r = 1; r < nrows; ++r)
        if (q[r + 1] - q[r0] > tile || r - r0 >= 256) {
          sd.blk.push_back((int32_t)r);
          r0 = r;
        }
      sd.blk.push_back((int32_t)nrows);
      sd.blk_off.push_back((int32_t)sd.blk.size());
    }
    if (sd.blk.empty()) sd.blk_off.clear();
  };
  Side sr, st;
  // LDS tiles first (APX_OPT_SPARSE_WINDOWS 2: where they pay ; 3: wherever they can be built) ;
  // a side without them gets the L2 windows / row blocks
  TilePlan<T> tr, tt;
  bool has_tr = false, has_tt = false;
  if (ctx->sparse_windows >= 2) {
    const bool force = ctx->sparse_windows == 3;
    // (also where the rows are the long-row form's — A' of covtype's 581 012 x 54: 41 -> 22 us ; of 2 400 000 x 3000: 444 -> 81)
    has_tr = plan_tiles<T>(ctx, (const int64_t*)p0, (const int32_t*)i0, (const T*)v0, A->m, A->ld, A->n, nnz, force, tr);
    has_tt = plan_tiles<T>(ctx, (const int64_t*)p1, (const int32_t*)i1, (const T*)v1, A->n, A->n_pad, A->m, nnz, force, tt);
  }
  if (!has_tr) plan((const int64_t*)p0, (const int32_t*)i0, (const T*)v0, A->m, A->ld, A->n, A->r_maxlen, sr);
  if (!has_tt) plan((const int64_t*)p1, (const int32_t*)i1, (const T*)v1, A->n, A->n_pad, A->m, A->t_maxlen, st);
  APX_TRY(up((void**)&A->rptr, p0, (size_t)(A->m + 1) * sizeof(int64_t)));
  APX_TRY(up((void**)&A->rind, sr.W ? (const void*)sr.ind.data() : i0, (size_t)nnz * sizeof(int32_t)));
  APX_TRY(up(&A->rval, sr.W ? (const void*)sr.val.data() : v0, (size_t)nnz * sizeof(T)));
  APX_TRY(up((void**)&A->tptr, p1, (size_t)(A->n + 1) * sizeof(int64_t)));
  APX_TRY(up((void**)&A->tind, st.W ? (const void*)st.ind.data() : i1, (size_t)nnz * sizeof(int32_t)));
  APX_TRY(up(&A->tval, st.W ? (const void*)st.val.data() : v1, (size_t)nnz * sizeof(T)));
  if (sr.W) APX_TRY(up((void**)&A->rwin, sr.win.data(), sr.win.size() * sizeof(int64_t)));
  if (st.W) APX_TRY(up((void**)&A->twin, st.win.data(), st.win.size() * sizeof(int64_t)));
  A->rwin_count = sr.W;
  A->twin_count = st.W;
  if (!sr.blk.empty()) APX_TRY(up((void**)&A->rblk, sr.blk.data(), sr.blk.size() * sizeof(int32_t)));
  if (!st.blk.empty()) APX_TRY(up((void**)&A->tblk, st.blk.data(), st.blk.size() * sizeof(int32_t)));
  A->rblk_off = sr.blk_off;
  A->tblk_off = st.blk_off;
  auto up_tiles = [&](apx_matrix::Tiles& d, const TilePlan<T>& tp) -> int32_t {
    APX_TRY(up((void**)&d.items, tp.items.data(), tp.items.size() * sizeof(int32_t)));
    APX_TRY(up((void**)&d.off, tp.off.data(), tp.off.size() * sizeof(int32_t)));
    APX_TRY(up((void**)&d.ind, tp.ind.data(), tp.ind.size() * sizeof(int32_t)));
    APX_TRY(up(&d.val, tp.val.data(), tp.val.size() * sizeof(T)));
    d.n_items = (int32_t)(tp.items.size() / kTileItemInts);
    d.nw = tp.nw;
    d.wcols = tp.wcols;
    d.splits = tp.splits;
    d.rows = tp.rows;
    return APX_OK;
  };
  if (has_tr) APX_TRY(up_tiles(A->rtile, tr));
  if (has_tt) APX_TRY(up_tiles(A->ttile, tt));
  if (has_tr || has_tt)
    APX_HIP(hipFuncSetAttribute((const void*)spmv_tiles<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kTileLds));
  APX_HIP(hipStreamSynchronize(ctx->stream));   // the host staging vectors die with this frame
  return APX_OK;
}

// one thread per row of A: D[i, j] += v for the row's entries, in storage order (duplicates add up, as they do in
// the products)
template <typename T>
__global__ __launch_bounds__(256) void k_densify(const int64_t* __restrict__ rptr, const int32_t* __restrict__ rind,
                                                 const T* __restrict__ rval, int64_t m, T* __restrict__ D, int64_t ld) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= m) return;
  for (int64_t k = rptr[i]; k < rptr[i + 1]; ++k) D[(int64_t)rind[k] * ld + i] += rval[k];
}

// The reference's LIBSVM datasets are tiny as dense matrices (mushrooms 8124 x 112, a5a 6414 x 123, phishing
// 11 055 x 68: experiments/sparse_logreg/runme.jl:177-204): a problem built on such a sparse matrix runs on a
// dense copy, where the small-size paths apply (2 launches per iteration instead of the SpMV path's 8).
int32_t small_dense_view(const apx_matrix* A, const apx_matrix** out) {
  *out = A;
  if (A == nullptr || !A->sparse || !A->ctx->fused_small) return APX_OK;
  if (A->rwin != nullptr) return APX_OK;      // (window-major storage: rptr no longer walks rind / rval row by row — and such a matrix is not small)
  const int64_t m = A->m, n = A->n;
  const bool tall_skinny = n + 1 <= kSkinnyMaxN && m <= 256 * 1024;
  if (!(tall_skinny || small_shape_ok(m, n + 1)) || m <= 0 || n <= 0) return APX_OK;
  if ((double)m * (double)n * (double)dtype_size(A->dtype) > 256.0 * 1024 * 1024) return APX_OK;
  if (A->dense_view == nullptr) {
    apx_matrix* D = nullptr;
    if (apx_matrix_create(A->ctx, m, n, A->dtype, &D) != APX_OK) {     // zero-filled ; no memory for it: stay sparse
      (void)hipGetLastError();
      return APX_OK;
    }
    const unsigned grid = (unsigned)((m + 255) / 256);
    if (A->dtype == APX_F32)
      hipLaunchKernelGGL((k_densify<float>), dim3(grid), dim3(256), 0, A->ctx->stream, A->rptr, A->rind, (const float*)A->rval, m,
                         (float*)D->d, D->ld);
    else
      hipLaunchKernelGGL((k_densify<double>), dim3(grid), dim3(256), 0, A->ctx->stream, A->rptr, A->rind, (const double*)A->rval, m,
                         (double*)D->d, D->ld);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
      apx_matrix_free(D);
      return fail(APX_ERR_HIP, std::string("small_dense_view: ") + hipGetErrorString(e));
    }
    A->dense_view = D;
  }
  *out = A->dense_view;
  return APX_OK;
}

}  // namespace apx

using namespace apx;

extern "C" {

int32_t apx_matrix_upload_sparse(apx_ctx* ctx, int64_t m, int64_t n, int64_t nnz, int32_t layout, const int64_t* ptr,
                                 const int32_t* idx, const void* vals, int32_t dtype, apx_matrix** out) {
  APX_REQUIRE(ctx != nullptr && out != nullptr && ptr != nullptr, "apx_matrix_upload_sparse: NULL argument");
  APX_REQUIRE(m >= 0 && n >= 0 && nnz >= 0, "apx_matrix_upload_sparse: negative size");
  APX_REQUIRE(layout == 0 || layout == 1, "apx_matrix_upload_sparse: layout must be 0 (CSR) or 1 (CSC)");
  APX_REQUIRE(dtype == APX_F64 || dtype == APX_F32, "apx_matrix_upload_sparse: bad dtype");
  APX_REQUIRE(nnz == 0 || (idx != nullptr && vals != nullptr), "apx_matrix_upload_sparse: NULL index / value array");
  APX_REQUIRE(m < 2147483647LL && n < 2147483647LL, "apx_matrix_upload_sparse: indices are 32-bit");
  *out = nullptr;
  const int64_t nrows_in = layout == 0 ? m : n, ncols_in = layout == 0 ? n : m;
  APX_REQUIRE(ptr[0] == 0 && ptr[nrows_in] == nnz, "apx_matrix_upload_sparse: pointer array must start at 0 and end at nnz");
  for (int64_t i = 0; i < nrows_in; ++i) APX_REQUIRE(ptr[i] <= ptr[i + 1], "apx_matrix_upload_sparse: pointer array not monotone");
  for (int64_t k = 0; k < nnz; ++k) APX_REQUIRE(idx[k] >= 0 && idx[k] < ncols_in, "apx_matrix_upload_sparse: index out of range");
  APX_HIP(hipSetDevice(ctx->device));
  apx_matrix* A = new apx_matrix();
  A->ctx = ctx;
  A->m = m;
  A->n = n;
  A->dtype = dtype;
  A->ld = round_up(std::max<int64_t>(m, 1), kPad);
  A->n_pad = round_up(std::max<int64_t>(n, 1), kPad);
  A->sparse = true;
  A->nnz = nnz;
  ctx_retain(ctx);
  const int32_t rc = dtype == APX_F32 ? upload_sparse_t<float>(ctx, A, layout, ptr, idx, (const float*)vals)
                                      : upload_sparse_t<double>(ctx, A, layout, ptr, idx, (const double*)vals);
  if (rc != APX_OK) {
    apx_matrix_free(A);
    return rc;
  }
  *out = A;
  return APX_OK;
}

}  // extern "C"
