// matrix.hip — context, dense column-major matrices resident in HBM, and the `A*x` / `A'*y`
// operator entry points of the C-ABI.
//
// HBM layout: one allocation of ld * n_pad elements, ld = round_up(m, 32), n_pad =
// round_up(n, 32), every pad entry zero.  Column j starts at a 128-B boundary (fp32) /
// 256-B boundary (fp64), so every 16-B lane access of the GEMV kernels is aligned and
// whole-wave loads are 1 KiB contiguous.  A 50 000 x 100 000 fp64 matrix is 40.0 GB of the
// 288 GB; nothing is tiled or duplicated.
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <ctime>

#include "apx_internal.h"
#include "gemv_kernels.h"

namespace apx {

static thread_local std::string g_last_error;
static thread_local std::string g_last_incident;
void set_error(const std::string& msg) { g_last_error = msg; }
void note_incident(const std::string& msg) { g_last_incident = msg; }
int32_t fail(int32_t code, const std::string& msg) {
  g_last_error = msg;
  return code;
}

// Process exit.  Finalizers of a host language may call the destroy entry points while the process is going down
// (Julia runs every registered finalizer at exit; Python `__del__`s run during interpreter teardown), possibly after the
// HIP runtime's own exit handlers: a HIP call then throws from inside the runtime (seen in round 1 as
// `terminate called after throwing 'std::bad_variant_access'` at the end of a sweep — DESIGN.md §9).  This object is
// constructed when the library is loaded, i.e. after libamdhip64 it depends on, so its destructor runs BEFORE the HIP
// runtime's teardown: from then on every destroy is a no-op — the dying process gives the memory back anyway.
static std::atomic<bool> g_exiting{false};
struct ExitMark {
  ~ExitMark() { g_exiting.store(true); }
};
static ExitMark g_exit_mark;
bool process_exiting() { return g_exiting.load(); }

void ctx_retain(apx_ctx* ctx) { ctx->children += 1; }
void ctx_release(apx_ctx* ctx) {
  ctx->children -= 1;
  if (ctx->children <= 0 && ctx->zombie) apx_ctx_destroy(ctx);
}

int32_t DBuf::alloc_zero(size_t nbytes, hipStream_t s) {
  if (p) (void)hipFree(p);
  p = nullptr;
  bytes = 0;
  if (nbytes == 0) nbytes = 256;
  APX_HIP(hipMalloc(&p, nbytes));
  bytes = nbytes;
  APX_HIP(hipMemsetAsync(p, 0, nbytes, s));
  return APX_OK;
}

// splitmix64 finaliser — same function as oracle/problems.py::hash_uniform
__device__ __forceinline__ double hash_uniform(uint64_t seed, uint64_t idx) {
  uint64_t z = seed * 0x9E3779B97F4A7C15ULL + idx;
  z += 0x9E3779B97F4A7C15ULL;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  z = z ^ (z >> 31);
  return (double)(z >> 11) * (1.0 / 9007199254740992.0);
}

template <typename T>
__global__ __launch_bounds__(kBlock) void k_fill_uniform(T* __restrict__ A, int64_t ld, int64_t m, int64_t n,
                                                         uint64_t seed, double lo, double hi, int64_t row_offset,
                                                         int64_t m_total) {
  const int64_t j = blockIdx.y;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < m; i += (int64_t)gridDim.x * kBlock) {
    const uint64_t idx = (uint64_t)j * (uint64_t)m_total + (uint64_t)(row_offset + i);
    A[j * ld + i] = (T)(lo + (hi - lo) * hash_uniform(seed, idx));
  }
}

template <typename T>
__global__ __launch_bounds__(kBlock) void k_scale_columns(T* __restrict__ A, int64_t ld, int64_t m, int64_t n,
                                                          const double* __restrict__ alpha) {
  const int64_t j = blockIdx.y;
  const double a = alpha[j];
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < m; i += (int64_t)gridDim.x * kBlock)
    A[j * ld + i] = (T)((double)A[j * ld + i] * a);
}

template <typename T>
__global__ __launch_bounds__(kBlock) void k_fill_column(T* __restrict__ col, int64_t m, double value) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < m; i += (int64_t)gridDim.x * kBlock)
    col[i] = (T)value;
}

// The same three edits on the ROW storage (A' column-major: element (i, j) of A at AT[i * ldT + j]) — a row of A per blockIdx.y,
// the threads along it (coalesced) ; the values are those of the column-major kernels, element for element.
template <typename T>
__global__ __launch_bounds__(kBlock) void k_fill_uniform_rows(T* __restrict__ AT, int64_t ldT, int64_t n, uint64_t seed, double lo,
                                                              double hi, int64_t row_offset, int64_t m_total) {
  const int64_t i = blockIdx.y;
  for (int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x; j < n; j += (int64_t)gridDim.x * kBlock) {
    const uint64_t idx = (uint64_t)j * (uint64_t)m_total + (uint64_t)(row_offset + i);
    AT[i * ldT + j] = (T)(lo + (hi - lo) * hash_uniform(seed, idx));
  }
}

template <typename T>
__global__ __launch_bounds__(kBlock) void k_scale_columns_rows(T* __restrict__ AT, int64_t ldT, int64_t n,
                                                               const double* __restrict__ alpha) {
  const int64_t i = blockIdx.y;
  for (int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x; j < n; j += (int64_t)gridDim.x * kBlock)
    AT[i * ldT + j] = (T)((double)AT[i * ldT + j] * alpha[j]);
}

template <typename T>
__global__ __launch_bounds__(kBlock) void k_fill_column_rows(T* __restrict__ AT, int64_t ldT, int64_t m, int64_t j, double value) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < m; i += (int64_t)gridDim.x * kBlock)
    AT[i * ldT + j] = (T)value;
}

// sum of squares over the whole padded buffer (pads are zero); block partials in a fixed order
template <typename T>
__global__ __launch_bounds__(kBlock) void k_sumsq(const T* __restrict__ A, int64_t total, double* __restrict__ red) {
  constexpr int V = V16<T>::N;
  double s = 0.0;
  const int64_t nvec = total / V;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * kBlock) {
    const auto v = ld16<T, false>(A + i * V);
#pragma unroll
    for (int e = 0; e < V; ++e) s = fma((double)v[e], (double)v[e], s);
  }
  __shared__ double sh[kBlock / 64];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = sh[0];
    for (int w = 1; w < kBlock / 64; ++w) t += sh[w];
    red[blockIdx.x] = t;
  }
}

// out-of-place transpose through a padded LDS tile: B (ldb x ...) = A' ; both sides coalesced
template <typename T>
__global__ __launch_bounds__(kBlock) void k_transpose(const T* __restrict__ A, int64_t lda, T* __restrict__ B,
                                                      int64_t ldb, int64_t m, int64_t n) {
  __shared__ T tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 32 x 8 threads
  const int64_t i0 = (int64_t)blockIdx.x * 32, j0 = (int64_t)blockIdx.y * 32;
#pragma unroll
  for (int k = 0; k < 32; k += 8) {
    const int64_t i = i0 + tx, j = j0 + ty + k;
    tile[ty + k][tx] = (i < m && j < n) ? A[j * lda + i] : (T)0;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 32; k += 8) {
    const int64_t j = j0 + tx, i = i0 + ty + k;
    if (j < n && i < m) B[i * ldb + j] = tile[tx][ty + k];
  }
}

__global__ void k_device_clock(unsigned long long* out) { *out = (unsigned long long)wall_clock64(); }

// the same over a plain array of `count` elements (sparse value arrays)
template <typename T>
__global__ __launch_bounds__(kBlock) void k_sumsq_elems(const T* __restrict__ v, int64_t count, double* __restrict__ red) {
  double s = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < count; i += (int64_t)gridDim.x * kBlock)
    s = fma((double)v[i], (double)v[i], s);
  __shared__ double sh[kBlock / 64];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = sh[0];
    for (int w = 1; w < kBlock / 64; ++w) t += sh[w];
    red[blockIdx.x] = t;
  }
}

template <typename T>
static void transpose_block(hipStream_t q, const void* src, int64_t ld_src, void* dst, int64_t ld_dst, int64_t m, int64_t n) {
  // dst (n x m column-major, ld_dst) = src' (src m x n column-major, ld_src) ; gridDim.y <= 65535 tiles of 32 columns
  const int64_t slab = 65535LL * 32;
  for (int64_t j0 = 0; j0 < n; j0 += slab) {
    const int64_t nj = std::min<int64_t>(slab, n - j0);
    hipLaunchKernelGGL(k_transpose<T>, dim3((unsigned)((m + 31) / 32), (unsigned)((nj + 31) / 32)), dim3(kBlock), 0, q,
                       (const T*)src + (size_t)j0 * ld_src, ld_src, (T*)dst + j0, ld_dst, m, nj);
  }
}

}  // namespace apx

using namespace apx;

extern "C" {

int32_t apx_abi_version(void) { return APX_ABI_VERSION; }
const char* apx_last_error(void) { return g_last_error.c_str(); }
const char* apx_last_incident(void) { return g_last_incident.c_str(); }

int32_t apx_ctx_create(int32_t device, void* stream, apx_ctx** out) {
  APX_REQUIRE(out != nullptr, "apx_ctx_create: out is NULL");
  *out = nullptr;
  int ndev = 0;
  APX_HIP(hipGetDeviceCount(&ndev));
  if (ndev <= 0) return fail(APX_ERR_HIP, "apx_ctx_create: no HIP device visible (the HIP path has no CPU fallback)");
  APX_REQUIRE(device >= 0 && device < ndev, "apx_ctx_create: bad device index");
  APX_HIP(hipSetDevice(device));
  apx_ctx* ctx = new apx_ctx();
  ctx->device = device;
  if (stream != nullptr) {
    ctx->stream = (hipStream_t)stream;
    ctx->own_stream = false;
  } else {
    hipError_t e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete ctx; return fail(APX_ERR_HIP, std::string("hipStreamCreate: ") + hipGetErrorString(e)); }
    ctx->own_stream = true;
  }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) == hipSuccess) ctx->num_cu = prop.multiProcessorCount;
  auto bail = [&](hipError_t e, const char* what) {
    apx_ctx_destroy(ctx);
    return fail(APX_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
  };
  hipError_t e;
  if ((e = hipMalloc((void**)&ctx->red, sizeof(double) * kRedQ * kMaxRedBlocks)) != hipSuccess) return bail(e, "hipMalloc red");
  if ((e = hipMalloc((void**)&ctx->scal, sizeof(double) * 16)) != hipSuccess) return bail(e, "hipMalloc scal");
  if ((e = hipHostMalloc((void**)&ctx->h_scal, sizeof(double) * 64, hipHostMallocDefault)) != hipSuccess) return bail(e, "hipHostMalloc");
  // defaults: measured best on MI355X; overridable for sweeps
  // sweep of round 1 (gpurun_out/sweep2.log, C2 matrix): every NT variant is within 1.5 %;
  // n_rpt2_unr16_nt @ 4096 blocks 6.89 TB/s, t_cpb4_unr8_nt 7.17 TB/s were the fastest
  ctx->cfg.n_variant = 4;
  ctx->cfg.t_variant = 2;
  ctx->cfg.n_target_blocks = 16 * (int64_t)ctx->num_cu;
  ctx->cfg.n_auto = true;
  ctx->cfg.n_variant_narrow = 0;   // n_rpt1_unr8_nt
  ctx->cfg.t_target_blocks = 8 * (int64_t)ctx->num_cu;
  // (no environment variables: sweeps use apx_ctx_set_gemv_config / apx_ctx_set_option)
  *out = ctx;
  return APX_OK;
}

static void ctx_free(apx_ctx* ctx);

int32_t apx_ctx_destroy(apx_ctx* ctx) {
  if (ctx == nullptr || process_exiting()) return APX_OK;
  if (ctx->children > 0) {   // matrices / problems still alive: defer
    ctx->zombie = true;
    return APX_OK;
  }
  try {
    ctx_free(ctx);
  } catch (...) {            // (nothing throws across the boundary, whatever state the runtime is in)
  }
  return APX_OK;
}

static void ctx_free(apx_ctx* ctx) {
  (void)hipSetDevice(ctx->device);
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  for (auto& ev : ctx->ev_pending) { (void)hipEventDestroy(ev.a); (void)hipEventDestroy(ev.b); }
  for (auto& ev : ctx->ev_free) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
  if (ctx->part) (void)hipFree(ctx->part);
  if (ctx->nz_idx) (void)hipFree(ctx->nz_idx);
  if (ctx->red) (void)hipFree(ctx->red);
  if (ctx->scal) (void)hipFree(ctx->scal);
  if (ctx->h_scal) (void)hipHostFree(ctx->h_scal);
  if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

int32_t apx_ctx_sync(apx_ctx* ctx) {
  APX_REQUIRE(ctx != nullptr, "apx_ctx_sync: ctx is NULL");
  APX_HIP(hipStreamSynchronize(ctx->stream));
  return APX_OK;
}

int32_t apx_ctx_stream(apx_ctx* ctx, void** stream_out) {
  APX_REQUIRE(ctx != nullptr && stream_out != nullptr, "apx_ctx_stream: NULL argument");
  *stream_out = (void*)ctx->stream;
  return APX_OK;
}

int32_t apx_ctx_profile(apx_ctx* ctx, int32_t enable) {
  APX_REQUIRE(ctx != nullptr, "apx_ctx_profile: ctx is NULL");
  ctx->profile = enable != 0;
  return APX_OK;
}

int32_t apx_ctx_kernel_stats(apx_ctx* ctx, apx_kernel_stat* stats, int32_t reset) {
  APX_REQUIRE(ctx != nullptr, "apx_ctx_kernel_stats: ctx is NULL");
  APX_HIP(hipStreamSynchronize(ctx->stream));
  for (auto& ev : ctx->ev_pending) {
    float ms = 0.f;
    APX_HIP(hipEventElapsedTime(&ms, ev.a, ev.b));
    ctx->stats[ev.kind].launches += 1;
    ctx->stats[ev.kind].total_ms += (double)ms;
    ctx->stats[ev.kind].bytes += ev.bytes;
    ctx->ev_free.push_back({ev.a, ev.b});
  }
  ctx->ev_pending.clear();
  if (stats != nullptr) memcpy(stats, ctx->stats, sizeof(ctx->stats));
  if (reset) memset(ctx->stats, 0, sizeof(ctx->stats));
  return APX_OK;
}

int32_t apx_ctx_device_clock(apx_ctx* ctx, int64_t* device_ns, int64_t* host_realtime_ns, int64_t* uncertainty_ns) {
  APX_REQUIRE(ctx != nullptr && device_ns != nullptr && host_realtime_ns != nullptr, "apx_ctx_device_clock: NULL argument");
  APX_HIP(hipSetDevice(ctx->device));
  APX_HIP(hipStreamSynchronize(ctx->stream));
  auto now = []() -> int64_t {
    timespec ts;
    clock_gettime(CLOCK_REALTIME, &ts);
    return (int64_t)ts.tv_sec * 1000000000ll + ts.tv_nsec;
  };
  // (the pinned scalars' slot 58: a spin-free hand-shake would need a mapped flag ; the launch + synchronise round trip is ~10 us)
  const int64_t t0 = now();
  hipLaunchKernelGGL(k_device_clock, dim3(1), dim3(1), 0, ctx->stream, (unsigned long long*)(ctx->scal + 15));
  APX_HIP(hipGetLastError());
  APX_HIP(hipMemcpyAsync(&ctx->h_scal[58], ctx->scal + 15, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  APX_HIP(hipStreamSynchronize(ctx->stream));
  const int64_t t1 = now();
  unsigned long long ticks;
  memcpy(&ticks, &ctx->h_scal[58], sizeof ticks);
  *device_ns = (int64_t)(ticks * 10ull);                     // s_memrealtime: 100 MHz
  *host_realtime_ns = t0 + (t1 - t0) / 2;
  if (uncertainty_ns != nullptr) *uncertainty_ns = (t1 - t0) / 2;
  return APX_OK;
}

int32_t apx_ctx_exploit_sparsity(apx_ctx* ctx, int32_t enable) {
  APX_REQUIRE(ctx != nullptr, "apx_ctx_exploit_sparsity: ctx is NULL");
  ctx->skip_zero_cols = enable != 0;
  return APX_OK;
}

int32_t apx_ctx_one_pass_threshold(apx_ctx* ctx, int64_t min_bytes) {
  APX_REQUIRE(ctx != nullptr && min_bytes >= 0, "apx_ctx_one_pass_threshold: bad argument");
  ctx->one_pass_min_bytes = min_bytes;
  return APX_OK;
}

int32_t apx_ctx_auto_transpose(apx_ctx* ctx, int32_t enable) {
  APX_REQUIRE(ctx != nullptr, "apx_ctx_auto_transpose: ctx is NULL");
  ctx->auto_transpose = enable != 0;
  return APX_OK;
}

int32_t apx_ctx_fused_small(apx_ctx* ctx, int32_t enable) {
  APX_REQUIRE(ctx != nullptr, "apx_ctx_fused_small: ctx is NULL");
  ctx->fused_small = enable != 0;
  return APX_OK;
}

int32_t apx_ctx_one_pass(apx_ctx* ctx, int32_t mode) {
  APX_REQUIRE(ctx != nullptr, "apx_ctx_one_pass: ctx is NULL");
  APX_REQUIRE(mode >= 0 && mode <= 2, "apx_ctx_one_pass: mode 0 (off), 1 (on) or 2 (on, without the team kernels)");
  ctx->one_pass = mode != 0;
  ctx->teams = mode == 1;
  return APX_OK;
}

const char* apx_ctx_kernel_name(apx_ctx* ctx, int32_t slot) {
  if (ctx == nullptr || slot < 0 || slot >= APX_KSTAT_COUNT) return "";
  return ctx->kname[slot].c_str();
}

int32_t apx_ctx_set_option(apx_ctx* ctx, int32_t key, int64_t value) {
  APX_REQUIRE(ctx != nullptr, "apx_ctx_set_option: ctx is NULL");
  switch (key) {
    case APX_OPT_MULTI_ROWS_MASK:
      APX_REQUIRE(value >= 0 && value <= 15, "APX_OPT_MULTI_ROWS_MASK: a mask of four bits");
      ctx->multi_mask = (int)value;
      return APX_OK;
    case APX_OPT_SLAB_GENERATION:
      APX_REQUIRE(value >= 0 && value <= 2, "APX_OPT_SLAB_GENERATION: 0 (by measurement), 1 or 2");
      ctx->slab_generation = (int)value;
      return APX_OK;
    case APX_OPT_TEAM_WAIT_MS:
      APX_REQUIRE(value >= 1 && value <= 60000, "APX_OPT_TEAM_WAIT_MS: 1 ... 60000 ms");
      ctx->team_wait_ms = (int)value;
      return APX_OK;
    case APX_OPT_TEAM_COOPERATIVE:
      ctx->team_cooperative = value != 0;
      return APX_OK;
    case APX_OPT_TEAM_FAULT_GRID:
      APX_REQUIRE(value >= 0 && value <= 512, "APX_OPT_TEAM_FAULT_GRID: 0 (off) ... 256 workgroups (257 ... 512: with APX_OPT_TEAM_COOPERATIVE only)");
      ctx->team_fault_grid = (int)value;
      return APX_OK;
    case APX_OPT_ONE_PASS_MIN_ROWS:
      APX_REQUIRE(value >= 256 && value <= 4096, "APX_OPT_ONE_PASS_MIN_ROWS: 256 ... 4096");
      ctx->one_pass_min_rows = value;
      return APX_OK;
    case APX_OPT_SPARSE_WINDOWS:
      APX_REQUIRE(value >= 0 && value <= 3, "APX_OPT_SPARSE_WINDOWS: 0 ... 3");
      ctx->sparse_windows = (int)value;
      return APX_OK;
    case APX_OPT_FAULT_SETUP:
      APX_REQUIRE(value >= 0 && value <= 1, "APX_OPT_FAULT_SETUP: 0 or 1");
      ctx->fault_setup = (int)value;
      return APX_OK;
    case APX_OPT_GEMV_N_VARIANT: return apx_ctx_set_gemv_config(ctx, (int32_t)value, -1, 0, 0);
    case APX_OPT_GEMV_T_VARIANT: return apx_ctx_set_gemv_config(ctx, -1, (int32_t)value, 0, 0);
    case APX_OPT_GEMV_N_BLOCKS: return apx_ctx_set_gemv_config(ctx, -1, -1, value, 0);
    case APX_OPT_GEMV_T_BLOCKS: return apx_ctx_set_gemv_config(ctx, -1, -1, 0, value);
    case APX_OPT_FUSED_TAIL:
      ctx->fused_tail = value != 0;
      return APX_OK;
    case APX_OPT_MULTI_WAVES:
      APX_REQUIRE(value == 0 || value == 10 || value == 11, "APX_OPT_MULTI_WAVES: 0 (by fit), 10 or 11");
      ctx->multi_waves = (int)value;
      return APX_OK;
    case APX_OPT_TEAM_MEMBERS:
      APX_REQUIRE(value == 0 || value == 2 || value == 4 || value == 8 || value == 16, "APX_OPT_TEAM_MEMBERS: 0 (by the row length), 2, 4, 8 or 16");
      ctx->team_members = (int)value;
      return APX_OK;
    case APX_OPT_ROW_DEAL:
      APX_REQUIRE(value >= -1 && value <= 1024 && (value <= 0 || (value & (value - 1)) == 0),
                  "APX_OPT_ROW_DEAL: -1 (chunks by the row length), 0 (slabs of consecutive rows) or a power of two (rows per chunk of the cyclic deal)");
      ctx->row_deal = (int)value;
      return APX_OK;
    case APX_OPT_TEAM_INTERLEAVE:
      APX_REQUIRE(value >= 0 && value <= 2, "APX_OPT_TEAM_INTERLEAVE: 0 (off), 1 (teams of four or more members) or 2 (every team)");
      ctx->team_interleave = (int)value;
      return APX_OK;
    case APX_OPT_TEAM_PD_DEFER:
      APX_REQUIRE(value == 0 || value == 1, "APX_OPT_TEAM_PD_DEFER: 0 or 1");
      ctx->team_pd_defer = (int)value;
      return APX_OK;
    case APX_OPT_TEAM_SLOTS:
      APX_REQUIRE(value == 0 || value == 52 || value == 62 || value == 53 || value == 63 || value == 31 || value == 42, "APX_OPT_TEAM_SLOTS: 0, 52, 62, 53, 63, 31 or 42");
      ctx->team_slots = (int)value;
      return APX_OK;
    default:
      return fail(APX_ERR_INVALID, "apx_ctx_set_option: unknown key");
  }
}

int32_t apx_ctx_get_option(apx_ctx* ctx, int32_t key, int64_t* value) {
  APX_REQUIRE(ctx != nullptr && value != nullptr, "apx_ctx_get_option: NULL argument");
  switch (key) {
    case APX_OPT_MULTI_ROWS_MASK: *value = ctx->multi_mask; return APX_OK;
    case APX_OPT_SLAB_GENERATION: *value = ctx->slab_generation; return APX_OK;
    case APX_OPT_TEAM_WAIT_MS: *value = ctx->team_wait_ms; return APX_OK;
    case APX_OPT_TEAM_COOPERATIVE: *value = ctx->team_cooperative ? 1 : 0; return APX_OK;
    case APX_OPT_TEAM_FAULT_GRID: *value = ctx->team_fault_grid; return APX_OK;
    case APX_OPT_FAULT_SETUP: *value = ctx->fault_setup; return APX_OK;
    case APX_OPT_SPARSE_WINDOWS: *value = ctx->sparse_windows; return APX_OK;
    case APX_OPT_ONE_PASS_MIN_ROWS: *value = ctx->one_pass_min_rows; return APX_OK;
    case APX_OPT_GEMV_N_VARIANT: *value = ctx->cfg.n_variant; return APX_OK;
    case APX_OPT_GEMV_T_VARIANT: *value = ctx->cfg.t_variant; return APX_OK;
    case APX_OPT_GEMV_N_BLOCKS: *value = ctx->cfg.n_target_blocks; return APX_OK;
    case APX_OPT_GEMV_T_BLOCKS: *value = ctx->cfg.t_target_blocks; return APX_OK;
    case APX_OPT_FUSED_TAIL: *value = ctx->fused_tail ? 1 : 0; return APX_OK;
    case APX_OPT_MULTI_WAVES: *value = ctx->multi_waves; return APX_OK;
    case APX_OPT_TEAM_MEMBERS: *value = ctx->team_members; return APX_OK;
    case APX_OPT_TEAM_SLOTS: *value = ctx->team_slots; return APX_OK;
    case APX_OPT_ROW_DEAL: *value = ctx->row_deal; return APX_OK;
    case APX_OPT_TEAM_INTERLEAVE: *value = ctx->team_interleave; return APX_OK;
    case APX_OPT_TEAM_PD_DEFER: *value = ctx->team_pd_defer; return APX_OK;
    default:
      return fail(APX_ERR_INVALID, "apx_ctx_get_option: unknown key");
  }
}

int32_t apx_ctx_set_gemv_config(apx_ctx* ctx, int32_t n_variant, int32_t t_variant, int64_t n_blocks,
                                int64_t t_blocks) {
  APX_REQUIRE(ctx != nullptr, "apx_ctx_set_gemv_config: ctx is NULL");
  if (n_variant >= 0) {
    APX_REQUIRE(n_variant < gemv_n_variant_count(), "gemv_n variant out of range");
    ctx->cfg.n_variant = n_variant;
    ctx->cfg.n_auto = false;   // an explicit choice (sweeps) is taken literally
  }
  if (n_blocks > 0) ctx->cfg.n_auto = false;
  if (t_variant >= 0) {
    APX_REQUIRE(t_variant < gemv_t_variant_count(), "gemv_t variant out of range");
    ctx->cfg.t_variant = t_variant;
  }
  if (n_blocks > 0) ctx->cfg.n_target_blocks = n_blocks;
  if (t_blocks > 0) ctx->cfg.t_target_blocks = t_blocks;
  return APX_OK;
}

int32_t apx_gemv_variant_count(int32_t transposed) {
  return transposed ? gemv_t_variant_count() : gemv_n_variant_count();
}

const char* apx_gemv_variant_name(int32_t transposed, int32_t i) {
  if (i < 0 || i >= apx_gemv_variant_count(transposed)) return "";
  return transposed ? gemv_t_variant_name(i) : gemv_n_variant_name(i);
}

// ---- matrices ----------------------------------------------------------------------
int32_t apx_matrix_create_layout(apx_ctx* ctx, int64_t m, int64_t n, int32_t dtype, int32_t layout, apx_matrix** out) {
  APX_REQUIRE(ctx != nullptr && out != nullptr, "apx_matrix_create: NULL argument");
  APX_REQUIRE(m >= 0 && n >= 0, "apx_matrix_create: negative dimension");
  APX_REQUIRE(dtype == APX_F64 || dtype == APX_F32, "apx_matrix_create: bad dtype");
  APX_REQUIRE(layout == APX_LAYOUT_COLS || layout == APX_LAYOUT_ROWS, "apx_matrix_create_layout: layout must be APX_LAYOUT_COLS or APX_LAYOUT_ROWS");
  *out = nullptr;
  APX_HIP(hipSetDevice(ctx->device));
  apx_matrix* A = new apx_matrix();
  A->ctx = ctx;
  A->m = m;
  A->n = n;
  A->dtype = dtype;
  A->ld = round_up(std::max<int64_t>(m, 1), kPad);
  A->n_pad = round_up(std::max<int64_t>(n, 1), kPad);
  // (the two storages have the same size: ld x n_pad column-major, or n_pad (= ldT) x ld for the rows)
  const size_t bytes = (size_t)A->ld * (size_t)A->n_pad * dtype_size(dtype);
  void* buf = nullptr;
  hipError_t e = hipMalloc(&buf, bytes);
  if (e != hipSuccess) {
    delete A;
    return fail(APX_ERR_HIP, std::string("apx_matrix_create: hipMalloc of ") + std::to_string(bytes) + " B: " +
                                 hipGetErrorString(e));
  }
  e = hipMemsetAsync(buf, 0, bytes, ctx->stream);
  if (e != hipSuccess) {
    (void)hipFree(buf);
    delete A;
    return fail(APX_ERR_HIP, std::string("apx_matrix_create: memset: ") + hipGetErrorString(e));
  }
  if (layout == APX_LAYOUT_ROWS) {
    A->dT = buf;
    A->ldT = A->n_pad;
  } else {
    A->d = buf;
  }
  ctx_retain(ctx);
  *out = A;
  return APX_OK;
}

int32_t apx_matrix_create(apx_ctx* ctx, int64_t m, int64_t n, int32_t dtype, apx_matrix** out) {
  return apx_matrix_create_layout(ctx, m, n, dtype, APX_LAYOUT_COLS, out);
}

// the budget of the staging slabs of a rows-only matrix's upload / download (bytes)
static constexpr size_t kStageBytes = (size_t)256 << 20;

int32_t apx_matrix_upload_layout(apx_ctx* ctx, const void* host, int64_t m, int64_t n, int64_t lda, int32_t dtype, int32_t layout,
                                 apx_matrix** out) {
  APX_REQUIRE(host != nullptr || m * n == 0, "apx_matrix_upload: host is NULL");
  APX_REQUIRE(lda >= m, "apx_matrix_upload: lda < m");
  APX_TRY(apx_matrix_create_layout(ctx, m, n, dtype, layout, out));
  apx_matrix* A = *out;
  if (m > 0 && n > 0) {
    const size_t es = dtype_size(dtype);
    hipError_t e = hipSuccess;
    if (layout == APX_LAYOUT_COLS) {
      e = hipMemcpy2DAsync(A->d, (size_t)A->ld * es, host, (size_t)lda * es, (size_t)m * es, (size_t)n, hipMemcpyHostToDevice,
                           ctx->stream);
    } else {
      // the host matrix is column-major (Julia's): slabs of columns through a staging buffer, transposed into the row storage
      const int64_t cols = std::max<int64_t>(32, std::min<int64_t>(n, (int64_t)(kStageBytes / ((size_t)A->ld * es)) / 32 * 32));
      DBuf stage;
      if (int32_t rc = stage.alloc_zero((size_t)A->ld * (size_t)cols * es, ctx->stream); rc != APX_OK) {
        apx_matrix_free(A);
        *out = nullptr;
        return rc;
      }
      for (int64_t j0 = 0; j0 < n && e == hipSuccess; j0 += cols) {
        const int64_t nj = std::min<int64_t>(cols, n - j0);
        e = hipMemcpy2DAsync(stage.p, (size_t)A->ld * es, (const char*)host + (size_t)j0 * lda * es, (size_t)lda * es, (size_t)m * es,
                             (size_t)nj, hipMemcpyHostToDevice, ctx->stream);
        if (e != hipSuccess) break;
        if (dtype == APX_F32) transpose_block<float>(ctx->stream, stage.p, A->ld, (float*)A->dT + j0, A->ldT, m, nj);
        else transpose_block<double>(ctx->stream, stage.p, A->ld, (double*)A->dT + j0, A->ldT, m, nj);
        e = hipGetLastError();
      }
      if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);      // (the staging buffer goes out of scope)
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
      apx_matrix_free(A);
      *out = nullptr;
      return fail(APX_ERR_HIP, std::string("apx_matrix_upload: ") + hipGetErrorString(e));
    }
  }
  return APX_OK;
}

int32_t apx_matrix_upload(apx_ctx* ctx, const void* host, int64_t m, int64_t n, int64_t lda, int32_t dtype,
                          apx_matrix** out) {
  return apx_matrix_upload_layout(ctx, host, m, n, lda, dtype, APX_LAYOUT_COLS, out);
}

int32_t apx_matrix_download(const apx_matrix* A, void* host, int64_t lda) {
  APX_REQUIRE(A != nullptr && host != nullptr, "apx_matrix_download: NULL argument");
  APX_REQUIRE(!A->sparse, "apx_matrix_download: dense matrices only");
  APX_REQUIRE(lda >= A->m, "apx_matrix_download: lda < m");
  APX_HIP(hipSetDevice(A->ctx->device));
  const size_t es = dtype_size(A->dtype);
  if (A->m > 0 && A->n > 0 && A->d != nullptr) {
    APX_HIP(hipMemcpy2DAsync(host, (size_t)lda * es, A->d, (size_t)A->ld * es, (size_t)A->m * es, (size_t)A->n,
                             hipMemcpyDeviceToHost, A->ctx->stream));
  } else if (A->m > 0 && A->n > 0) {
    // rows-only: slabs of ROWS of A transposed back into a staging buffer (column-major, rows x n), copied out column by column
    const int64_t rows = std::max<int64_t>(32, std::min<int64_t>(A->ld, (int64_t)(kStageBytes / ((size_t)A->n * es)) / 32 * 32));
    DBuf stage;
    APX_TRY(stage.alloc_zero((size_t)rows * (size_t)A->n * es, A->ctx->stream));
    for (int64_t i0 = 0; i0 < A->m; i0 += rows) {
      const int64_t ni = std::min<int64_t>(rows, A->m - i0);
      // (the block of the row storage: an n x ni column-major operand at dT + i0 * ldT)
      if (A->dtype == APX_F32) transpose_block<float>(A->ctx->stream, (const float*)A->dT + (size_t)i0 * A->ldT, A->ldT, stage.p, rows, A->n, ni);
      else transpose_block<double>(A->ctx->stream, (const double*)A->dT + (size_t)i0 * A->ldT, A->ldT, stage.p, rows, A->n, ni);
      APX_HIP(hipGetLastError());
      APX_HIP(hipMemcpy2DAsync((char*)host + (size_t)i0 * es, (size_t)lda * es, stage.p, (size_t)rows * es, (size_t)ni * es, (size_t)A->n,
                               hipMemcpyDeviceToHost, A->ctx->stream));
    }
    APX_HIP(hipStreamSynchronize(A->ctx->stream));      // (the staging buffer goes out of scope)
  }
  APX_HIP(hipStreamSynchronize(A->ctx->stream));
  return APX_OK;
}

int32_t apx_matrix_free(apx_matrix* A) {
  if (A == nullptr || process_exiting()) return APX_OK;
  try {
    (void)hipSetDevice(A->ctx->device);
    (void)hipStreamSynchronize(A->ctx->stream);
    if (A->dense_view) apx_matrix_free(A->dense_view);
    if (A->d) (void)hipFree(A->d);
    if (A->dT) (void)hipFree(A->dT);
    for (void* q : {(void*)A->rptr, (void*)A->rind, A->rval, (void*)A->tptr, (void*)A->tind, A->tval, (void*)A->rwin, (void*)A->twin, (void*)A->rblk, (void*)A->tblk,
                    (void*)A->rtile.items, (void*)A->rtile.off, (void*)A->rtile.ind, A->rtile.val, (void*)A->ttile.items, (void*)A->ttile.off,
                    (void*)A->ttile.ind, A->ttile.val})
      if (q) (void)hipFree(q);
    apx_ctx* ctx = A->ctx;
    delete A;
    ctx_release(ctx);
  } catch (...) {
  }
  return APX_OK;
}

static void drop_transpose(apx_matrix* A) {
  if (A->dT == nullptr || A->d == nullptr) return;      // (a rows-only matrix: that IS the matrix — its edits work on it in place)
  (void)hipStreamSynchronize(A->ctx->stream);
  (void)hipFree(A->dT);
  A->dT = nullptr;
  A->ldT = 0;
}

}  // extern "C"

namespace apx {
int32_t transpose_into(apx_ctx* ctx, const apx_matrix* A, void* dT, int64_t ldT) {
  if (A->m > 0 && A->n > 0) {
    // gridDim.y is limited to 65535 tiles: walk the columns in slabs
    const int64_t slab = 65535LL * 32;
    for (int64_t j0 = 0; j0 < A->n; j0 += slab) {
      const int64_t nj = std::min<int64_t>(slab, A->n - j0);
      dim3 grid((unsigned)((A->m + 31) / 32), (unsigned)((nj + 31) / 32));
      if (A->dtype == APX_F32)
        hipLaunchKernelGGL(k_transpose<float>, grid, dim3(kBlock), 0, ctx->stream,
                           (const float*)A->d + (size_t)j0 * A->ld, A->ld, (float*)dT + j0, ldT, A->m, nj);
      else
        hipLaunchKernelGGL(k_transpose<double>, grid, dim3(kBlock), 0, ctx->stream,
                           (const double*)A->d + (size_t)j0 * A->ld, A->ld, (double*)dT + j0, ldT, A->m, nj);
    }
  }
  APX_HIP(hipGetLastError());
  return APX_OK;
}
}  // namespace apx

extern "C" {

int32_t apx_matrix_build_transpose(apx_matrix* A) {
  APX_REQUIRE(A != nullptr, "apx_matrix_build_transpose: A is NULL");
  APX_REQUIRE(!A->sparse, "apx_matrix_build_transpose: dense matrices only (sparse storage keeps both orientations)");
  apx_ctx* ctx = A->ctx;
  APX_HIP(hipSetDevice(ctx->device));
  if (A->d == nullptr) return APX_OK;      // a rows-only matrix already is what this builds
  if (A->dT != nullptr) {
    APX_HIP(hipStreamSynchronize(ctx->stream));
    (void)hipFree(A->dT);
    A->dT = nullptr;
  }
  const size_t es = dtype_size(A->dtype);
  const int64_t ldT = round_up(std::max<int64_t>(A->n, 1), kPad);
  const size_t bytes = (size_t)ldT * (size_t)A->ld * es;      // n_pad(rows of A') x ld(columns = rows of A, padded)
  void* dT = nullptr;
  hipError_t e = hipMalloc(&dT, bytes);
  if (e != hipSuccess)
    return fail(APX_ERR_HIP, std::string("apx_matrix_build_transpose: hipMalloc of ") + std::to_string(bytes) +
                                 " B: " + hipGetErrorString(e));
  APX_HIP(hipMemsetAsync(dT, 0, bytes, ctx->stream));
  if (int32_t rc = transpose_into(ctx, A, dT, ldT); rc != APX_OK) {
    (void)hipFree(dT);
    return rc;
  }
  APX_HIP(hipGetLastError());
  APX_HIP(hipStreamSynchronize(ctx->stream));
  A->dT = dT;
  A->ldT = ldT;
  return APX_OK;
}

int32_t apx_matrix_drop_transpose(apx_matrix* A) {
  APX_REQUIRE(A != nullptr, "apx_matrix_drop_transpose: A is NULL");
  APX_REQUIRE(A->sparse || A->d != nullptr, "apx_matrix_drop_transpose: a rows-only matrix has no other copy (apx_matrix_free)");
  APX_HIP(hipSetDevice(A->ctx->device));
  drop_transpose(A);
  return APX_OK;
}

int32_t apx_matrix_drop_primary(apx_matrix* A) {
  APX_REQUIRE(A != nullptr, "apx_matrix_drop_primary: A is NULL");
  APX_REQUIRE(!A->sparse, "apx_matrix_drop_primary: dense matrices only");
  if (A->d == nullptr) return APX_OK;                              // rows-only already
  APX_REQUIRE(A->users == 0, "apx_matrix_drop_primary: problems created on this matrix still exist — their solvers may be reading its "
                             "column-major storage (the small-size kernels): drop the primary copy before creating problems, or destroy them first");
  if (A->dT == nullptr) APX_TRY(apx_matrix_build_transpose(A));   // (both copies are resident for the moment of the transpose)
  APX_HIP(hipSetDevice(A->ctx->device));
  APX_HIP(hipStreamSynchronize(A->ctx->stream));
  (void)hipFree(A->d);
  A->d = nullptr;
  return APX_OK;
}

int32_t apx_matrix_layout(const apx_matrix* A, int32_t* mask) {
  APX_REQUIRE(A != nullptr && mask != nullptr, "apx_matrix_layout: NULL argument");
  *mask = A->sparse ? 0 : ((A->d != nullptr ? 1 : 0) | (A->dT != nullptr ? 2 : 0));
  return APX_OK;
}

int32_t apx_matrix_shape(const apx_matrix* A, int64_t* m, int64_t* n, int32_t* dtype) {
  APX_REQUIRE(A != nullptr, "apx_matrix_shape: A is NULL");
  if (m) *m = A->m;
  if (n) *n = A->n;
  if (dtype) *dtype = A->dtype;
  return APX_OK;
}

static inline dim3 grid_mn(int64_t m, int64_t n) {
  return dim3((unsigned)std::max<int64_t>(1, std::min<int64_t>(64, (m + kBlock - 1) / kBlock)), (unsigned)n);
}

int32_t apx_matrix_fill_uniform(apx_matrix* A, uint64_t seed, double lo, double hi, int64_t row_offset,
                                int64_t m_total) {
  APX_REQUIRE(A != nullptr, "apx_matrix_fill_uniform: A is NULL");
  APX_REQUIRE(!A->sparse, "apx_matrix_fill_uniform: dense matrices only");
  drop_transpose(A);   // the transposed copy (if any) no longer matches
  if (m_total <= 0) m_total = A->m;
  APX_REQUIRE(row_offset >= 0 && row_offset + A->m <= m_total, "apx_matrix_fill_uniform: shard outside the global matrix");
  if (A->m == 0 || A->n == 0) return APX_OK;
  APX_HIP(hipSetDevice(A->ctx->device));
  if (A->d == nullptr) {      // rows-only: the same values, written along the rows (gridDim.y: slabs of 65535 rows)
    const unsigned gx = (unsigned)std::max<int64_t>(1, std::min<int64_t>(64, (A->n + kBlock - 1) / kBlock));
    for (int64_t i0 = 0; i0 < A->m; i0 += 65535) {
      const int64_t ni = std::min<int64_t>(65535, A->m - i0);
      if (A->dtype == APX_F32)
        hipLaunchKernelGGL(k_fill_uniform_rows<float>, dim3(gx, (unsigned)ni), dim3(kBlock), 0, A->ctx->stream,
                           (float*)A->dT + (size_t)i0 * A->ldT, A->ldT, A->n, seed, lo, hi, row_offset + i0, m_total);
      else
        hipLaunchKernelGGL(k_fill_uniform_rows<double>, dim3(gx, (unsigned)ni), dim3(kBlock), 0, A->ctx->stream,
                           (double*)A->dT + (size_t)i0 * A->ldT, A->ldT, A->n, seed, lo, hi, row_offset + i0, m_total);
    }
    APX_HIP(hipGetLastError());
    return APX_OK;
  }
  // gridDim.y is limited to 65535: walk the columns in slabs
  for (int64_t j0 = 0; j0 < A->n; j0 += 65535) {
    const int64_t nj = std::min<int64_t>(65535, A->n - j0);
    const size_t es = dtype_size(A->dtype);
    char* base = (char*)A->d + (size_t)j0 * A->ld * es;
    const uint64_t seed_eff = seed;
    // the column index enters the hash through idx = j*m_total + i, so slab launches pass j0 via row offset trick:
    // idx = (j0 + jj)*m_total + row_offset + i  ==  jj*m_total + (j0*m_total + row_offset + i)
    const int64_t off = j0 * m_total + row_offset;
    if (A->dtype == APX_F32)
      hipLaunchKernelGGL(k_fill_uniform<float>, grid_mn(A->m, nj), dim3(kBlock), 0, A->ctx->stream, (float*)base,
                         A->ld, A->m, nj, seed_eff, lo, hi, off, m_total);
    else
      hipLaunchKernelGGL(k_fill_uniform<double>, grid_mn(A->m, nj), dim3(kBlock), 0, A->ctx->stream, (double*)base,
                         A->ld, A->m, nj, seed_eff, lo, hi, off, m_total);
  }
  APX_HIP(hipGetLastError());
  return APX_OK;
}

int32_t apx_matrix_scale_columns(apx_matrix* A, const double* alpha) {
  APX_REQUIRE(A != nullptr && alpha != nullptr, "apx_matrix_scale_columns: NULL argument");
  APX_REQUIRE(!A->sparse, "apx_matrix_scale_columns: dense matrices only");
  drop_transpose(A);   // the transposed copy (if any) no longer matches
  if (A->m == 0 || A->n == 0) return APX_OK;
  APX_HIP(hipSetDevice(A->ctx->device));
  DBuf d_alpha;
  APX_TRY(d_alpha.alloc_zero((size_t)A->n * sizeof(double), A->ctx->stream));
  APX_HIP(hipMemcpyAsync(d_alpha.p, alpha, (size_t)A->n * sizeof(double), hipMemcpyHostToDevice, A->ctx->stream));
  if (A->d == nullptr) {      // rows-only
    const unsigned gx = (unsigned)std::max<int64_t>(1, std::min<int64_t>(64, (A->n + kBlock - 1) / kBlock));
    for (int64_t i0 = 0; i0 < A->m; i0 += 65535) {
      const int64_t ni = std::min<int64_t>(65535, A->m - i0);
      if (A->dtype == APX_F32)
        hipLaunchKernelGGL(k_scale_columns_rows<float>, dim3(gx, (unsigned)ni), dim3(kBlock), 0, A->ctx->stream,
                           (float*)A->dT + (size_t)i0 * A->ldT, A->ldT, A->n, (const double*)d_alpha.p);
      else
        hipLaunchKernelGGL(k_scale_columns_rows<double>, dim3(gx, (unsigned)ni), dim3(kBlock), 0, A->ctx->stream,
                           (double*)A->dT + (size_t)i0 * A->ldT, A->ldT, A->n, (const double*)d_alpha.p);
    }
    APX_HIP(hipGetLastError());
    APX_HIP(hipStreamSynchronize(A->ctx->stream));
    return APX_OK;
  }
  for (int64_t j0 = 0; j0 < A->n; j0 += 65535) {
    const int64_t nj = std::min<int64_t>(65535, A->n - j0);
    const size_t es = dtype_size(A->dtype);
    char* base = (char*)A->d + (size_t)j0 * A->ld * es;
    if (A->dtype == APX_F32)
      hipLaunchKernelGGL(k_scale_columns<float>, grid_mn(A->m, nj), dim3(kBlock), 0, A->ctx->stream, (float*)base,
                         A->ld, A->m, nj, (const double*)d_alpha.p + j0);
    else
      hipLaunchKernelGGL(k_scale_columns<double>, grid_mn(A->m, nj), dim3(kBlock), 0, A->ctx->stream, (double*)base,
                         A->ld, A->m, nj, (const double*)d_alpha.p + j0);
  }
  APX_HIP(hipGetLastError());
  APX_HIP(hipStreamSynchronize(A->ctx->stream));
  return APX_OK;
}

int32_t apx_matrix_fill_column(apx_matrix* A, int64_t j, double value) {
  APX_REQUIRE(A != nullptr, "apx_matrix_fill_column: A is NULL");
  APX_REQUIRE(!A->sparse, "apx_matrix_fill_column: dense matrices only");
  drop_transpose(A);   // the transposed copy (if any) no longer matches
  APX_REQUIRE(j >= 0 && j < A->n, "apx_matrix_fill_column: column out of range");
  if (A->m == 0) return APX_OK;
  APX_HIP(hipSetDevice(A->ctx->device));
  const size_t es = dtype_size(A->dtype);
  const unsigned g = (unsigned)std::min<int64_t>(1024, (A->m + kBlock - 1) / kBlock);
  if (A->d == nullptr) {      // rows-only: a strided write
    if (A->dtype == APX_F32)
      hipLaunchKernelGGL(k_fill_column_rows<float>, dim3(g), dim3(kBlock), 0, A->ctx->stream, (float*)A->dT, A->ldT, A->m, j, value);
    else
      hipLaunchKernelGGL(k_fill_column_rows<double>, dim3(g), dim3(kBlock), 0, A->ctx->stream, (double*)A->dT, A->ldT, A->m, j, value);
    APX_HIP(hipGetLastError());
    return APX_OK;
  }
  char* col = (char*)A->d + (size_t)j * A->ld * es;
  if (A->dtype == APX_F32)
    hipLaunchKernelGGL(k_fill_column<float>, dim3(g), dim3(kBlock), 0, A->ctx->stream, (float*)col, A->m, value);
  else
    hipLaunchKernelGGL(k_fill_column<double>, dim3(g), dim3(kBlock), 0, A->ctx->stream, (double*)col, A->m, value);
  APX_HIP(hipGetLastError());
  return APX_OK;
}

int32_t apx_matrix_sumsq(const apx_matrix* A, double* out) {
  APX_REQUIRE(A != nullptr && out != nullptr, "apx_matrix_sumsq: NULL argument");
  apx_ctx* ctx = A->ctx;
  APX_HIP(hipSetDevice(ctx->device));
  const int nblk = kMaxRedBlocks;
  if (A->sparse) {
    if (A->dtype == APX_F32)
      hipLaunchKernelGGL(k_sumsq_elems<float>, dim3(nblk), dim3(kBlock), 0, ctx->stream, (const float*)A->rval, A->nnz, ctx->red);
    else
      hipLaunchKernelGGL(k_sumsq_elems<double>, dim3(nblk), dim3(kBlock), 0, ctx->stream, (const double*)A->rval, A->nnz, ctx->red);
  } else {
    const int64_t total = A->ld * A->n_pad;                       // (either storage: the same number of elements, pads zero)
    const void* base = A->d != nullptr ? A->d : A->dT;
    if (A->dtype == APX_F32)
      hipLaunchKernelGGL(k_sumsq<float>, dim3(nblk), dim3(kBlock), 0, ctx->stream, (const float*)base, total, ctx->red);
    else
      hipLaunchKernelGGL(k_sumsq<double>, dim3(nblk), dim3(kBlock), 0, ctx->stream, (const double*)base, total, ctx->red);
  }
  APX_HIP(hipGetLastError());
  std::vector<double> h(nblk);
  APX_HIP(hipMemcpyAsync(h.data(), ctx->red, sizeof(double) * nblk, hipMemcpyDeviceToHost, ctx->stream));
  APX_HIP(hipStreamSynchronize(ctx->stream));
  double s = 0.0;
  for (int i = 0; i < nblk; ++i) s += h[i];
  *out = s;
  return APX_OK;
}

// `A*x` (src/counting.jl:68-74): x host (n), y host (m)
int32_t apx_gemv_n(const apx_matrix* A, const void* x, void* y) {
  APX_REQUIRE(A != nullptr && x != nullptr && y != nullptr, "apx_gemv_n: NULL argument");
  apx_ctx* ctx = A->ctx;
  APX_HIP(hipSetDevice(ctx->device));
  const size_t es = dtype_size(A->dtype);
  DBuf dx, dy;
  APX_TRY(dx.alloc_zero((size_t)A->n_pad * es, ctx->stream));
  APX_TRY(dy.alloc_zero((size_t)A->ld * es, ctx->stream));
  APX_HIP(hipMemcpyAsync(dx.p, x, (size_t)A->n * es, hipMemcpyHostToDevice, ctx->stream));
  int splits = 0;
  APX_TRY(launch_gemv_n(ctx, A, dx.p, &splits, nullptr));
  APX_TRY(launch_sum_partials_rows(ctx, A->dtype, splits, A->m, A->ld, dy.p, nullptr));
  APX_HIP(hipMemcpyAsync(y, dy.p, (size_t)A->m * es, hipMemcpyDeviceToHost, ctx->stream));
  APX_HIP(hipStreamSynchronize(ctx->stream));
  return APX_OK;
}

// `A'*y` (src/counting.jl:76-82): r host (m), g host (n)
int32_t apx_gemv_t(const apx_matrix* A, const void* r, void* g) {
  APX_REQUIRE(A != nullptr && r != nullptr && g != nullptr, "apx_gemv_t: NULL argument");
  apx_ctx* ctx = A->ctx;
  APX_HIP(hipSetDevice(ctx->device));
  const size_t es = dtype_size(A->dtype);
  DBuf dr, dg;
  APX_TRY(dr.alloc_zero((size_t)A->ld * es, ctx->stream));
  APX_TRY(dg.alloc_zero((size_t)A->n_pad * es, ctx->stream));
  APX_HIP(hipMemcpyAsync(dr.p, r, (size_t)A->m * es, hipMemcpyHostToDevice, ctx->stream));
  APX_TRY(launch_gemv_t(ctx, A, dr.p, dg.p, nullptr));
  APX_HIP(hipMemcpyAsync(g, dg.p, (size_t)A->n * es, hipMemcpyDeviceToHost, ctx->stream));
  APX_HIP(hipStreamSynchronize(ctx->stream));
  return APX_OK;
}

}  // extern "C"
