// comm.hip — RCCL communicators inside the library: the collective of the row-sharded path.
//
// The reference is single-process Julia and has no collective; what is summed over the row shards is the gradient
// `A' * res` its pullback returns (experiments/lasso/runme.jl:23, consumed at src/AdaProx.jl:336) together with the
// partial f scalars, and — with a row-sharded linear operator — `A' * y` and the dual residual (src/AdaProx.jl:344-358).
//
// A communicator is bound to one context (device + stream).  The native loops enqueue `ncclAllReduce` on that stream
// between their kernels — stream-ordered like everything else, no host round trip per iteration, no callback.
//
//   one process per GPU (torch.distributed.run, MPI, ...):  rank 0 calls apx_comm_unique_id, the launcher hands the
//       128 bytes to every rank, each rank calls apx_comm_init_rank on its own context      (ncclCommInitRank)
//   one process, several GPUs (a Julia session):  apx_comm_init_all over one context per device (ncclCommInitAll);
//       every device's loop then needs its own host thread — apx_adaptive_primal_dual_group runs them
//
// xGMI is a point-to-point mesh (7 links per GPU) and the message is n + 2 elements (0.8 MB at n = 100 000) against
// 5.8 ms of local streaming per iteration at C4's shard (one pass over 40 GB ; 0.77 ms at an eighth of it — DESIGN.md section 6):
// latency-bound, one collective per gradient, nothing to bucket or overlap.
// (librccl is 0.3 - 0.6 GB of device code: on a box whose image is not in the page cache its first use demand-pages it at ~2 MB/s
//  — minutes.  The host mirror reads the file sequentially first, sharded.prefetch_library_image ; a Julia host would do the same.)
//
// librccl is bound at the FIRST apx_comm_* call, not at load time (rccl_api below): a single-GPU user never maps its
// 570 MB, and — what decides it — the copy must be the one that belongs to the HIP runtime this library is bound to.
// PyTorch-ROCm ships its own libamdhip64 + librccl pair next to the system's (/opt/rocm), two HIP runtimes in one process
// do not share the device (adaptive-proximal-algorithms_amd/_lib.py), and a librccl mapped before torch maps its own
// aborts the process at exit (measured: `double free or corruption` from the exit handlers).  So: a librccl already in the
// process is taken (torch imported first: its pair); else the librccl in the directory of the libamdhip64 in use;
// else the loader's default.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "solver_internal.h"

struct apx_comm {
  apx_ctx* ctx = nullptr;
  ncclComm_t comm = nullptr;
  int nranks = 0, rank = 0;
  apx::DBuf scratch;          // device staging of apx_comm_allreduce_host
  // problems created on this communicator keep it alive, like matrices keep their context: apx_comm_destroy on a communicator
  // that still has problems only marks it, and the last problem to go destroys it (finalizers run in any order)
  int64_t users = 0;
  bool zombie = false;
};

namespace apx {

// the RCCL entry points this file uses, resolved once
struct RcclApi {
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommInitAll) CommInitAll = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclCommCount) CommCount = nullptr;
  decltype(&ncclCommUserRank) CommUserRank = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  decltype(&ncclGetLastError) GetLastError = nullptr;
  std::string path, error;
  bool ok = false;
};

static RcclApi* rccl_api() {
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);          // already in the process (torch imported first)
    if (h == nullptr) {
      Dl_info info;                                                    // next to the HIP runtime this library runs on
      if (dladdr((void*)&hipGetDeviceCount, &info) != 0 && info.dli_fname != nullptr) {
        std::string dir(info.dli_fname);
        const size_t cut = dir.rfind('/');
        if (cut != std::string::npos) {
          dir.resize(cut + 1);
          for (const char* name : {"librccl.so.1", "librccl.so"}) {
            h = dlopen((dir + name).c_str(), RTLD_NOW | RTLD_GLOBAL);
            if (h != nullptr) break;
          }
        }
      }
    }
    if (h == nullptr) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (h == nullptr) {
      const char* e = dlerror();
      api.error = std::string("librccl.so.1 could not be loaded: ") + (e != nullptr ? e : "?");
      return;
    }
    bool all = true;
    auto sym = [&](auto& fp, const char* name) {
      fp = reinterpret_cast<std::remove_reference_t<decltype(fp)>>(dlsym(h, name));
      if (fp == nullptr) {
        all = false;
        api.error = std::string("librccl: missing symbol ") + name;
      }
    };
    sym(api.GetUniqueId, "ncclGetUniqueId");
    sym(api.CommInitRank, "ncclCommInitRank");
    sym(api.CommInitAll, "ncclCommInitAll");
    sym(api.CommDestroy, "ncclCommDestroy");
    sym(api.CommCount, "ncclCommCount");
    sym(api.CommUserRank, "ncclCommUserRank");
    sym(api.AllReduce, "ncclAllReduce");
    sym(api.GetErrorString, "ncclGetErrorString");
    sym(api.GetLastError, "ncclGetLastError");
    Dl_info where;
    if (all && dladdr((void*)api.AllReduce, &where) != 0 && where.dli_fname != nullptr) api.path = where.dli_fname;
    api.ok = all;
  });
  return &api;
}

#define APX_RCCL_API(R)                                                   \
  ::apx::RcclApi* R = ::apx::rccl_api();                                  \
  if (!R->ok) return ::apx::fail(APX_ERR_COMM, R->error)

static int32_t nccl_fail(const char* what, ncclResult_t r, ncclComm_t comm) {
  RcclApi* R = rccl_api();
  std::string msg = std::string(what) + ": " + R->GetErrorString(r);
  if (comm != nullptr) {
    const char* last = R->GetLastError(comm);
    if (last != nullptr && last[0] != 0) msg += std::string(" (") + last + ")";
  }
  return fail(APX_ERR_COMM, msg);
}

#define APX_NCCL(expr, comm)                                           \
  do {                                                                 \
    ncclResult_t _r = (expr);                                          \
    if (_r != ncclSuccess) return ::apx::nccl_fail(#expr, _r, (comm)); \
  } while (0)

// in-place sum of `count` elements at the DEVICE pointer buf over the ranks, enqueued on the context's stream
int32_t comm_allreduce(apx_comm* c, void* buf, int64_t count, int32_t dtype) {
  RcclApi* R = rccl_api();                        // (a communicator exists: the API is bound)
  APX_NCCL(R->AllReduce(buf, buf, (size_t)count, dtype == APX_F32 ? ncclFloat32 : ncclFloat64, ncclSum, c->comm,
                         c->ctx->stream),
           c->comm);
  return APX_OK;
}

apx_ctx* comm_ctx(const apx_comm* c) { return c->ctx; }

static void comm_free(apx_comm* c) {
  (void)hipSetDevice(c->ctx->device);
  (void)hipStreamSynchronize(c->ctx->stream);
  if (c->comm != nullptr) (void)rccl_api()->CommDestroy(c->comm);
  apx_ctx* ctx = c->ctx;
  delete c;
  ctx_release(ctx);
}
void comm_retain(apx_comm* c) {
  if (c != nullptr) c->users += 1;
}
void comm_release(apx_comm* c) {
  if (c == nullptr) return;
  c->users -= 1;
  if (c->users <= 0 && c->zombie) comm_free(c);
}

}  // namespace apx

using namespace apx;

extern "C" {

int32_t apx_comm_unique_id(void* id_out) {
  APX_REQUIRE(id_out != nullptr, "apx_comm_unique_id: id_out is NULL");
  static_assert(sizeof(ncclUniqueId) == APX_COMM_ID_BYTES, "APX_COMM_ID_BYTES must match ncclUniqueId");
  APX_RCCL_API(R);
  ncclUniqueId id;
  APX_NCCL(R->GetUniqueId(&id), nullptr);
  memcpy(id_out, &id, sizeof(id));
  return APX_OK;
}

int32_t apx_comm_init_rank(apx_ctx* ctx, int32_t nranks, int32_t rank, const void* id, apx_comm** out) {
  APX_REQUIRE(ctx != nullptr && id != nullptr && out != nullptr, "apx_comm_init_rank: NULL argument");
  *out = nullptr;
  APX_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, "apx_comm_init_rank: need 0 <= rank < nranks");
  APX_RCCL_API(R);
  APX_HIP(hipSetDevice(ctx->device));
  ncclUniqueId uid;
  memcpy(&uid, id, sizeof(uid));
  std::unique_ptr<apx_comm> c(new apx_comm());
  c->ctx = ctx;
  APX_NCCL(R->CommInitRank(&c->comm, nranks, uid, rank), nullptr);
  ncclResult_t r1 = R->CommCount(c->comm, &c->nranks), r2 = r1 == ncclSuccess ? R->CommUserRank(c->comm, &c->rank) : r1;
  if (r2 != ncclSuccess) {                         // (do not leak the communicator that did initialise)
    const int32_t rc = nccl_fail("ncclCommCount / ncclCommUserRank", r2, c->comm);
    (void)R->CommDestroy(c->comm);
    return rc;
  }
  ctx_retain(ctx);
  *out = c.release();
  return APX_OK;
}

int32_t apx_comm_init_all(apx_ctx* const* ctxs, int32_t ndev, apx_comm** comms_out) {
  APX_REQUIRE(ctxs != nullptr && comms_out != nullptr && ndev >= 1, "apx_comm_init_all: bad argument");
  std::vector<int> devs((size_t)ndev);
  for (int i = 0; i < ndev; ++i) {
    comms_out[i] = nullptr;
    APX_REQUIRE(ctxs[i] != nullptr, "apx_comm_init_all: a context is NULL");
    devs[(size_t)i] = ctxs[i]->device;
    for (int k = 0; k < i; ++k)
      APX_REQUIRE(devs[(size_t)k] != devs[(size_t)i], "apx_comm_init_all: one context per DEVICE (RCCL refuses two ranks on a GPU)");
  }
  APX_RCCL_API(R);
  std::vector<ncclComm_t> cs((size_t)ndev, nullptr);
  APX_NCCL(R->CommInitAll(cs.data(), ndev, devs.data()), nullptr);
  for (int i = 0; i < ndev; ++i) {
    apx_comm* c = new apx_comm();
    c->ctx = ctxs[i];
    c->comm = cs[(size_t)i];
    (void)R->CommCount(c->comm, &c->nranks);
    (void)R->CommUserRank(c->comm, &c->rank);
    ctx_retain(ctxs[i]);
    comms_out[i] = c;
  }
  return APX_OK;
}

int32_t apx_comm_info(const apx_comm* c, int32_t* nranks, int32_t* rank) {
  APX_REQUIRE(c != nullptr, "apx_comm_info: comm is NULL");
  APX_RCCL_API(R);
  int cnt = 0, rk = 0;
  APX_NCCL(R->CommCount(c->comm, &cnt), c->comm);       // (asked of RCCL, not remembered: bench.py prints it)
  APX_NCCL(R->CommUserRank(c->comm, &rk), c->comm);
  if (nranks != nullptr) *nranks = cnt;
  if (rank != nullptr) *rank = rk;
  return APX_OK;
}

const char* apx_comm_library(void) {
  RcclApi* R = rccl_api();
  return R->ok ? R->path.c_str() : "";
}

int32_t apx_comm_allreduce_host(apx_comm* c, double* buf, int64_t count) {
  APX_REQUIRE(c != nullptr && buf != nullptr && count > 0, "apx_comm_allreduce_host: bad argument");
  apx_ctx* ctx = c->ctx;
  APX_HIP(hipSetDevice(ctx->device));
  const size_t bytes = (size_t)count * sizeof(double);
  if (c->scratch.bytes < bytes) APX_TRY(c->scratch.alloc_zero(bytes, ctx->stream));
  APX_TRY(h2d(ctx, c->scratch.p, buf, bytes));
  APX_TRY(comm_allreduce(c, c->scratch.p, count, APX_F64));
  APX_TRY(d2h(ctx, buf, c->scratch.p, bytes));
  return APX_OK;
}

int32_t apx_comm_destroy(apx_comm* c) {
  if (c == nullptr || process_exiting()) return APX_OK;
  if (c->users > 0) {        // problems still enqueue collectives on it: deferred to the last of them
    c->zombie = true;
    return APX_OK;
  }
  try {
    comm_free(c);
  } catch (...) {
  }
  return APX_OK;
}

// One process, one problem per device: every device's loop on its own host thread (RCCL's collectives block the
// enqueueing thread until all ranks of the communicator have arrived, so one thread cannot drive several ranks outside
// a group call).  All ranks run the same scalar recurrences on identical all-reduced inputs and take identical steps;
// x_out / res / records come from rank 0, y_outs[i] is shard i of y.
int32_t apx_adaptive_primal_dual_group(int32_t ndev, apx_problem* const* ps, const apx_rule* rule, const void* x0,
                                       const void* const* y0s, double tol, int64_t maxit, const apx_solve_opts* opts,
                                       void* x_out, void* const* y_outs, apx_result* res) {
  APX_REQUIRE(ndev >= 1 && ps != nullptr && rule != nullptr && x0 != nullptr, "apx_adaptive_primal_dual_group: bad argument");
  for (int i = 0; i < ndev; ++i) {
    APX_REQUIRE(ps[i] != nullptr, "apx_adaptive_primal_dual_group: a problem is NULL");
    APX_REQUIRE(ndev == 1 || ps[i]->comm != nullptr, "apx_adaptive_primal_dual_group: every problem needs its communicator");
    APX_REQUIRE(ps[i]->n == ps[0]->n && ps[i]->dtype == ps[0]->dtype, "apx_adaptive_primal_dual_group: x is replicated: same n and dtype on every device");
  }
  std::vector<int32_t> rc((size_t)ndev, APX_OK);
  std::vector<std::string> err((size_t)ndev);
  std::vector<std::thread> th;
  // the convergence poll copies the flag at fixed iteration numbers on every rank: identical decisions
  apx_solve_opts quiet{};
  if (opts != nullptr) {
    quiet.use_graph = opts->use_graph;
    quiet.poll_every = opts->poll_every;
  }
  auto run = [&](int i) {
    rc[(size_t)i] = apx_adaptive_primal_dual(ps[i], rule, x0, y0s != nullptr ? y0s[i] : nullptr, tol, maxit,
                                             i == 0 ? opts : &quiet, i == 0 ? x_out : nullptr,
                                             y_outs != nullptr ? y_outs[i] : nullptr, i == 0 ? res : nullptr);
    if (rc[(size_t)i] != APX_OK) err[(size_t)i] = apx_last_error();     // (thread-local: carry it to the caller's thread)
  };
  for (int i = 1; i < ndev; ++i) th.emplace_back(run, i);
  run(0);
  for (auto& t : th) t.join();
  for (int i = 0; i < ndev; ++i)
    if (rc[(size_t)i] != APX_OK) return fail(rc[(size_t)i], "device " + std::to_string(i) + ": " + err[(size_t)i]);
  return APX_OK;
}

}  // extern "C"
