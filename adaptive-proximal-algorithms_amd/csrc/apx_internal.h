// apx_internal.h — private structs shared by the translation units of libadaprox_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>

#include "../../include/adaprox_hip.h"

namespace apx {

// ---- error plumbing: nothing throws across the C boundary ------------------------
void set_error(const std::string& msg);
int32_t fail(int32_t code, const std::string& msg);
void note_incident(const std::string& msg);      // apx_last_incident(): what the last team-kernel failure looked like

#define APX_HIP(expr)                                                                       \
  do {                                                                                      \
    hipError_t _e = (expr);                                                                 \
    if (_e != hipSuccess)                                                                   \
      return ::apx::fail(APX_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));  \
  } while (0)

#define APX_TRY(expr)                  \
  do {                                 \
    int32_t _s = (expr);               \
    if (_s != APX_OK) return _s;       \
  } while (0)

#define APX_REQUIRE(cond, msg)                                   \
  do {                                                           \
    if (!(cond)) return ::apx::fail(APX_ERR_INVALID, (msg));     \
  } while (0)

inline int64_t round_up(int64_t v, int64_t q) { return (v + q - 1) / q * q; }
inline size_t dtype_size(int32_t dtype) { return dtype == APX_F32 ? 4 : 8; }

constexpr int64_t kPad = 32;          // every vector / leading dimension is padded to 32 elements (>=128 B)
constexpr int kMaxRedBlocks = 1024;   // grid cap of the fused vector kernels (= partials per quantity)
constexpr int kRedQ = 8;              // quantities per partial buffer

// ---- device control block: every scalar of the loop lives here --------------------
struct Ctrl {
  double gamma, sigma, rho, gamma_prev;
  double st0, st1;                 // rule state: Ours/OursPlus (gamma1, gamma0); MM (gamma_prev, rho)
  double f_x, bias_grad;           // smooth term value; logistic: mean(probs - y)
  double g_x, h_Ax, objective;
  double npr2, ndr2, norm_res;     // norm(primal_res)^2, norm(dual_res)^2, residual
  double h_pre;                    // ||w/sigma + shift|| for the l2 conjugate prox
  double tol;
  double n_mean;                   // logistic: N of the mean
  int64_t it;
  int32_t done;
  int32_t flags;
  // rule
  int32_t rule_kind, _pad;
  double r_gamma, r_t, r_normA, r_delta, r_Theta, r_xi, r_nu, r_r;
};

// prox descriptor as the kernels see it (device pointers)
struct ProxDev {
  int32_t kind;
  int32_t has_shift;
  double lambda, lb, ub;
  const void* lb_vec;
  const void* ub_vec;
  const void* shift;
};

struct GemvConfig {
  int n_variant;       // index into the compiled gemv_n variants
  int t_variant;
  int64_t n_target_blocks;   // column splits are chosen so that grid ~ this
  int64_t t_target_blocks;
  bool n_auto;               // cap the column splits by the partial-sum traffic (off once a sweep pins a config)
  int n_variant_narrow;      // RPT=1 variant used when the cap leaves too few blocks
};

}  // namespace apx

struct apx_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  int num_cu = 256;
  // matrices / problems created on this context keep it alive: apx_ctx_destroy on a context
  // that still has children only marks it, and the last child to go frees it
  int64_t children = 0;
  bool zombie = false;
  // scratch
  double* part = nullptr;        // GEMV partial sums
  size_t part_bytes = 0;
  double* red = nullptr;         // [kRedQ][kMaxRedBlocks] block partials of the fused vector kernels
  double* scal = nullptr;        // small device scalar scratch (16 doubles)
  double* h_scal = nullptr;      // pinned host scalars (64 doubles): read-backs of scal / Ctrl fields
  // opt-in: A*x streams only the columns with x_j != 0 (apx_ctx_exploit_sparsity)
  bool fused_small = true;       // small dense proximal-gradient problems run as two kernels per iteration (small.hip)
  bool one_pass = true;          // large dense matrices: one pass over A per iteration (apx_ctx_one_pass: slab.hip, slab_pd.hip)
  bool teams = true;             // ... and its team kernels for rows too long for one workgroup (team_rows.h ; mode 2 switches them off)
  // options by key (apx_ctx_set_option)
  int multi_mask = 15;           // APX_OPT_MULTI_ROWS_MASK
  int multi_waves = 0;           // APX_OPT_MULTI_WAVES
  int team_members = 0;          // APX_OPT_TEAM_MEMBERS
  int row_deal = -1;             // APX_OPT_ROW_DEAL: -1 chunks by the row length, 0 slabs of consecutive rows, R chunks of R rows
  int team_slots = 0;            // APX_OPT_TEAM_SLOTS
  int slab_generation = 0;       // APX_OPT_SLAB_GENERATION
  int team_wait_ms = 1000;       // APX_OPT_TEAM_WAIT_MS
  bool team_cooperative = false; // APX_OPT_TEAM_COOPERATIVE
  int team_fault_grid = 0;       // APX_OPT_TEAM_FAULT_GRID (test hook)
  int fault_setup = 0;           // APX_OPT_FAULT_SETUP (test hook)
  int64_t one_pass_min_rows = 4096;   // APX_OPT_ONE_PASS_MIN_ROWS (test hook: the parity tests of the widest rows use matrices of ~1000 rows)
  int sparse_windows = 2;        // APX_OPT_SPARSE_WINDOWS: what a sparse matrix uploaded on this context gets besides its rows — 0 nothing,
                                 // 1 L2 window tables + row blocks, 2 (default) LDS tiles where they pay and 1's forms elsewhere, 3 tiles always
  int team_interleave = 1;       // APX_OPT_TEAM_INTERLEAVE
  int team_pd_defer = 1;         // APX_OPT_TEAM_PD_DEFER
  bool fused_tail = true;        // APX_OPT_FUSED_TAIL
  int64_t one_pass_min_bytes = 192ll << 20;   // slab.hip / slab_pd.hip take matrices from this size on (apx_ctx_one_pass_threshold)
  bool auto_transpose = true;    // a one-pass-eligible problem may give its matrix the transposed copy (apx_ctx_auto_transpose)
  bool skip_zero_cols = false;
  int32_t* nz_idx = nullptr;     // compacted column indices + [nz_cap] = count
  int64_t nz_cap = 0;
  // profiling
  bool profile = false;
  struct Ev { hipEvent_t a, b; int kind; double bytes; };
  std::vector<Ev> ev_pending;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_free;
  apx_kernel_stat stats[APX_KSTAT_COUNT] = {};
  std::string kname[APX_KSTAT_COUNT];   // the kernels last booked under a slot (apx_ctx_kernel_name)
  apx::GemvConfig cfg;
};

struct apx_matrix {
  apx_ctx* ctx = nullptr;
  int64_t m = 0, n = 0, ld = 0, n_pad = 0;
  int32_t dtype = APX_F64;
  void* d = nullptr;           // dense: column-major ld x n_pad (Julia's layout) — nullptr for a ROWS-ONLY matrix (below)
  // the ROW storage, A' column-major (ldT = round_up(n,32) rows x ld columns: row i of A contiguous at dT + i * ldT).  Next to d
  // it is the optional second copy (apx_matrix_build_transpose: `A*x` then streams whole rows like `A'*r` streams whole
  // columns) ; it is what the one-pass kernels read, and a matrix may live in it ALONE (APX_LAYOUT_ROWS, apx_matrix_drop_primary):
  // one resident copy for a loop that reads one — the two GEMV orientations are served from it by the same kernel pair with the
  // roles swapped (gemv.hip), the small-size kernels (which read d) do not apply
  void* dT = nullptr;
  int64_t ldT = 0;
  // sparse storage: BOTH orientations are kept, each as compressed rows, so
  // `A*x` and `A'*y` are gather-style row products with a fixed summation order (no atomics)
  bool sparse = false;
  int64_t nnz = 0;
  int64_t* rptr = nullptr;     // rows of A      : m + 1
  int32_t* rind = nullptr;     //                  nnz column indices
  void* rval = nullptr;
  int64_t* tptr = nullptr;     // rows of A'     : n + 1
  int32_t* tind = nullptr;     //                  nnz row indices
  void* tval = nullptr;
  // a small sparse matrix also keeps a dense copy, built on first use (apx::small_dense_view): the small-size
  // execution policy (apx_ctx_fused_small) runs on dense storage
  int64_t r_maxlen = 0, t_maxlen = 0;   // longest row of A / of A' (entries)
  // Round 5 (sparse.hip), per orientation (r: rows of A, operand x ; t: rows of A', operand r):
  //  * L2 WINDOWS of the gathered operand: where it is larger than an XCD's L2 keeps next to the streams (X' r with a million rows:
  //    8 MB) the entries are stored WINDOW-MAJOR — all rows' entries whose index falls into window 0, then window 1, ... — and the
  //    product runs window by window.  win: W slabs of nrows + 1 pointers into ind / val (each slab a CSR of its own) ; nullptr where
  //    the operand is small, the rows are the long-row form's or the caller's indices were not sorted.  With it, ptr no longer
  //    describes ind / val (only the row lengths).
  //  * ROW BLOCKS of the streaming form (spmv_stream): consecutive rows whose entries together fill a workgroup's LDS tile ;
  //    blk: first rows, concatenated per window, blk_off[w] .. blk_off[w + 1] the entries of window w (one "window" without win) ;
  //    a window whose rows do not fit a tile has no blocks and runs on the lanes-per-row kernel.
  int64_t* rwin = nullptr;
  int64_t* twin = nullptr;
  int32_t rwin_count = 0, twin_count = 0;
  int32_t* rblk = nullptr;
  int32_t* tblk = nullptr;
  std::vector<int32_t> rblk_off, tblk_off;
  //  * LDS TILES (spmv_tiles): the orientation cut into tiles of (up to) 8192 rows x one 64 KB window of the gathered operand, the entries of a
  //    tile stored together (row-major within it) with their row and index LOCAL to the tile packed into 32 bits — the window of the
  //    operand is staged in LDS and every gather is an LDS read, the rows' sums are kept in LDS across the windows.  A side that has
  //    tiles keeps its plain compressed rows (ptr / ind / val) next to them and has neither windows nor row blocks.
  struct Tiles {
    int32_t* items = nullptr;   // n_items x 8: row block, split, first / one past the last window, splits to fill with zeros [z0, z1), first row, rows
    int32_t* off = nullptr;     // [row blocks x windows + 1]: where the tiles begin, in chunks of 256 entries
    int32_t* ind = nullptr;     // packed entries (local row << 17 | byte offset in the window), every tile padded to whole chunks
    void* val = nullptr;
    int32_t wcols = 0;          // entries of the operand per window (<= 64 KB worth: chosen so that a tile is whole rounds of chunks)
    int32_t n_items = 0, nw = 0, splits = 0, rows = 0;   // rows: of the largest row block (8192 ; fewer where blocks x windows would not fill the chip)
  };
  Tiles rtile, ttile;
  // problems created on this matrix and not yet destroyed: a solver of theirs may hold pointers into the column-major storage (the
  // small-size kernels), so apx_matrix_drop_primary refuses while there are any
  mutable int64_t users = 0;
  mutable apx_matrix* dense_view = nullptr;
};

namespace apx {

// the small-size / tall-skinny kernel families read the column-major storage: not for sparse or rows-only matrices
inline bool no_column_storage(const apx_matrix* A) { return A->sparse || A->d == nullptr; }

void ctx_retain(apx_ctx* ctx);
void ctx_release(apx_ctx* ctx);
// true once exit() has begun to run this library's exit handlers: the destroy entry points then leave everything to the
// dying process (the HIP runtime may be torn down already — matrix.hip)
bool process_exiting();
// comm.hip: in-place sum over the ranks of `count` elements at the device pointer buf, on the context's stream
int32_t comm_allreduce(apx_comm* c, void* buf, int64_t count, int32_t dtype);
apx_ctx* comm_ctx(const apx_comm* c);
void comm_retain(apx_comm* c);      // a problem on the communicator keeps it alive (apx_comm_destroy is deferred to the last one)
void comm_release(apx_comm* c);

// ---- launchers implemented in gemv.hip -------------------------------------------
int32_t ensure_part(apx_ctx* ctx, size_t bytes);
// part_out[(split) * ld + row] ; returns number of splits used
int32_t launch_gemv_n(apx_ctx* ctx, const apx_matrix* A, const void* x, int* splits_out, const int* done);
// out (T, length n_pad)
int32_t launch_gemv_t(apx_ctx* ctx, const apx_matrix* A, const void* r, void* out, const int* done);
// sums the gemv_n partials into y (T, length ld): `A*x` with no epilogue
int32_t launch_sum_partials_rows(apx_ctx* ctx, int32_t dtype, int splits, int64_t m, int64_t ld, void* y,
                                 const int* done);
// sparse.hip: the same two products for compressed storage (same output conventions)
// HIP events around one launch (apx_ctx_profile): kind = APX_KSTAT_*, bytes = its algorithmic bytes
int32_t prof_begin(apx_ctx* ctx, int kind, double bytes);
int32_t prof_end(apx_ctx* ctx);
int32_t launch_spmv_n(apx_ctx* ctx, const apx_matrix* A, const void* x, int* splits_out, const int* done);
// A itself, or — for a sparse A small enough for the small-size paths — its dense copy (owned by A)
int32_t small_dense_view(const apx_matrix* A, const apx_matrix** out);
int32_t launch_spmv_t(apx_ctx* ctx, const apx_matrix* A, const void* r, void* out, const int* done);
int gemv_n_variant_count();
int gemv_t_variant_count();
const char* gemv_n_variant_name(int i);
const char* gemv_t_variant_name(int i);

// ---- device buffers --------------------------------------------------------------
struct DBuf {
  void* p = nullptr;
  size_t bytes = 0;
  ~DBuf() { if (p) (void)hipFree(p); }
  DBuf() = default;
  DBuf(const DBuf&) = delete;
  DBuf& operator=(const DBuf&) = delete;
  int32_t alloc_zero(size_t nbytes, hipStream_t s);
};

}  // namespace apx
