/* exit_probe.c — a host program whose "finalizers" call the destroy entry points of libadaprox_hip.so while the process
 * exits, like a Julia session does for every object registered with `finalizer` (julia/AdaProxHIP.jl).  Built and run
 * by tests/test_gpu_exit.py.
 *
 *   exit_probe <lib> atexit      destroys registered with atexit() after the GPU was used: they run BEFORE the library's
 *                                own exit handler and before the HIP runtime's teardown — real frees, context first
 *                                (the order a garbage collector may pick)
 *   exit_probe <lib> destructor  destroys called from a destructor of the main program, i.e. from _dl_fini, AFTER the
 *                                exit handlers of the library and of the HIP runtime: must be harmless no-ops
 * Exit code 0 = clean exit. */
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "adaprox_hip.h"

static int32_t (*p_ctx_destroy)(apx_ctx*);
static int32_t (*p_matrix_free)(apx_matrix*);
static int32_t (*p_problem_destroy)(apx_problem*);
static int32_t (*p_comm_destroy)(apx_comm*);
static apx_ctx* g_ctx;
static apx_matrix* g_A;
static apx_problem* g_p;
static apx_comm* g_comm;
static int g_late;

static void finalize_all(void) {
  /* context first, its children afterwards: the library must cope (refcounted context) */
  int rc = p_ctx_destroy(g_ctx) | p_comm_destroy(g_comm) | p_matrix_free(g_A) | p_problem_destroy(g_p);
  if (rc != 0) _Exit(3);
}
__attribute__((destructor)) static void late_finalizers(void) {
  if (g_late) finalize_all();
}

#define SYM(var, name)                                   \
  do {                                                   \
    *(void**)(&var) = dlsym(h, name);                    \
    if (!var) { fprintf(stderr, "no %s\n", name); return 2; } \
  } while (0)
#define OK(expr)                                                                  \
  do {                                                                            \
    int32_t rc_ = (expr);                                                         \
    if (rc_ != 0) { fprintf(stderr, "%s -> %d: %s\n", #expr, rc_, p_err()); return 2; } \
  } while (0)

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  void* h = dlopen(argv[1], RTLD_NOW | RTLD_GLOBAL);
  if (!h) { fprintf(stderr, "%s\n", dlerror()); return 2; }
  int32_t (*p_ctx_create)(int32_t, void*, apx_ctx**);
  int32_t (*p_matrix_create)(apx_ctx*, int64_t, int64_t, int32_t, apx_matrix**);
  int32_t (*p_fill)(apx_matrix*, uint64_t, double, double, int64_t, int64_t);
  int32_t (*p_problem_create)(apx_ctx*, const apx_problem_desc*, apx_problem**);
  int32_t (*p_eval)(apx_problem*, const void*, double*);
  int32_t (*p_uid)(void*);
  int32_t (*p_comm_init)(apx_ctx*, int32_t, int32_t, const void*, apx_comm**);
  const char* (*p_err)(void);
  SYM(p_ctx_create, "apx_ctx_create"); SYM(p_matrix_create, "apx_matrix_create"); SYM(p_fill, "apx_matrix_fill_uniform");
  SYM(p_problem_create, "apx_problem_create"); SYM(p_eval, "apx_eval"); SYM(p_err, "apx_last_error");
  SYM(p_ctx_destroy, "apx_ctx_destroy"); SYM(p_matrix_free, "apx_matrix_free"); SYM(p_problem_destroy, "apx_problem_destroy");
  SYM(p_uid, "apx_comm_unique_id"); SYM(p_comm_init, "apx_comm_init_rank"); SYM(p_comm_destroy, "apx_comm_destroy");
  enum { M = 300, N = 200 };
  static double b[M], x[N];
  OK(p_ctx_create(0, NULL, &g_ctx));
  OK(p_matrix_create(g_ctx, M, N, APX_F64, &g_A));
  OK(p_fill(g_A, 1, -1.0, 1.0, 0, M));
  if (argc > 3 && strcmp(argv[3], "comm") == 0) {        /* a live RCCL communicator among the finalized objects */
    char id[APX_COMM_ID_BYTES];
    OK(p_uid(id));
    OK(p_comm_init(g_ctx, 1, 0, id, &g_comm));
  }
  apx_problem_desc d;
  memset(&d, 0, sizeof d);
  d.f_kind = APX_F_LEAST_SQUARES;
  d.dtype = APX_F64;
  d.n = N;
  d.A_f = g_A;
  d.b = b;
  d.comm = g_comm;
  for (int i = 0; i < N; ++i) x[i] = 0.01 * i;
  OK(p_problem_create(g_ctx, &d, &g_p));
  double fx = 0.0;
  OK(p_eval(g_p, x, &fx));
  if (!(fx > 0.0)) return 2;
  if (strcmp(argv[2], "atexit") == 0) atexit(finalize_all);
  else g_late = 1;
  printf("f(x) = %.6f, exiting\n", fx);
  return 0;
}
