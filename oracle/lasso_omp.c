/*
 * lasso_omp.c — TEST INFRASTRUCTURE, not product code: a plain C + OpenMP restatement of ONE configuration of the
 * reference's hot path, used only as the multi-threaded CPU baseline of bench.py (`cpu_baseline`, kind "port") and
 * checked against the numpy oracle (oracle/adaprox_oracle.py) in tests/test_oracle_c.py.  Nothing under
 * adaptive-proximal-algorithms_amd/ links or loads it.  Parity status: like the numpy oracle, UNPINNED at the iterate
 * level (no Julia runtime here; the reference's tests hold no iterate vectors — DESIGN.md §2).
 *
 * What it restates (file:line of /root/reference):
 *   adaptive_proxgrad(x; f, g, rule)            src/AdaProx.jl:418-421  ->  adaptive_primal_dual with y = 0, h = Zero, A = 0
 *   the loop                                    src/AdaProx.jl:324-363
 *   OurRule with norm_A = 0, t = 1, delta = 0   src/AdaProx.jl:258-272
 *   f = LinearLeastSquares(A, b)                experiments/lasso/runme.jl:16-27   (res = A*w - b ; 0.5*norm(res)^2 ; A'*res)
 *   g = NormL1(lam)  (ProximalOperators, third-party, unpinned: soft-threshold)
 *
 * A is column-major m x n with leading dimension lda (Julia layout).  The two products are the whole cost
 * (experiments/lasso/runme.jl:179-180): both are parallel over blocks of columns (`A*x` into a private m-vector per
 * thread, summed afterwards; `A'*r` one dot product per column), the blocks the generator touched first.
 * Summation order differs from OpenBLAS dgemv; agreement with the numpy oracle is to rounding (tests: 1e-10 over 30
 * iterations at 300 x 700).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* (the host may give this job a share of its CPUs: bench.py sizes the team to it — an environment variable would come too
 * late when another library has initialised the OpenMP runtime already) */
void apx_oracle_omp_set_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}

int apx_oracle_omp_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* res = A*x - b ; returns norm(res)^2 (experiments/lasso/runme.jl:22,24).  The threads split the COLUMNS in the same static
 * blocks as the generator and the gradient below — on a two-socket host every thread then streams pages it touched first
 * (NUMA-local) — each into a private m-vector, summed over the threads in thread order afterwards. */
static double residual(const double* A, int64_t m, int64_t n, int64_t lda, const double* x, const double* b, double* res) {
  int T = 1;
#ifdef _OPENMP
  T = omp_get_max_threads();
#endif
  double* acc = (double*)malloc(sizeof(double) * (size_t)T * (size_t)m);
  if (!acc) return NAN;
#pragma omp parallel num_threads(T)
  {
    int t = 0, nt = 1;
#ifdef _OPENMP
    t = omp_get_thread_num();
    nt = omp_get_num_threads();
#endif
    double* a = acc + (size_t)t * (size_t)m;
    for (int64_t i = 0; i < m; ++i) a[i] = 0.0;
    /* the static schedule of `#pragma omp parallel for schedule(static)` over n columns */
    const int64_t q = n / nt, r = n % nt;
    const int64_t j0 = t * q + (t < r ? t : r), j1 = j0 + q + (t < r ? 1 : 0);
    for (int64_t j = j0; j < j1; ++j) {
      const double xj = x[j];
      const double* col = A + j * lda;
      for (int64_t i = 0; i < m; ++i) a[i] += col[i] * xj;
    }
  }
  double ss = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : ss)
  for (int64_t i = 0; i < m; ++i) {
    double s = 0.0;
    for (int t = 0; t < T; ++t) s += acc[(size_t)t * (size_t)m + i];
    const double r = s - b[i];
    res[i] = r;
    ss += r * r;
  }
  free(acc);
  return ss;
}

/* grad = A' * res (experiments/lasso/runme.jl:23) */
static void gradient(const double* A, int64_t m, int64_t n, int64_t lda, const double* res, double* grad) {
#pragma omp parallel for schedule(static)
  for (int64_t j = 0; j < n; ++j) {
    const double* col = A + j * lda;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int64_t i = 0;
    for (; i + 4 <= m; i += 4) {
      s0 += col[i] * res[i];
      s1 += col[i + 1] * res[i + 1];
      s2 += col[i + 2] * res[i + 2];
      s3 += col[i + 3] * res[i + 3];
    }
    for (; i < m; ++i) s0 += col[i] * res[i];
    grad[j] = (s0 + s1) + (s2 + s3);
  }
}

static double nan_to_zero(double v) { return isnan(v) ? 0.0 : v; }     /* src/AdaProx.jl:24 */
static double jl_min(double a, double b) { return (isnan(a) || isnan(b)) ? NAN : (a < b ? a : b); }   /* Julia min */

/* prox of gamma * lam * ||.||_1 (soft-threshold, the recalled upstream form — see the numpy oracle's NormL1) */
static void soft_threshold(const double* v, double gl, int64_t n, double* x) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) {
    const double t = v[i];
    x[i] = t + (t <= -gl ? gl : (t >= gl ? -gl : -t));
  }
}

/*
 * Runs at most `maxit` iterations from x (in: x0; out: the tested iterate when `norm_res <= tol` stopped the loop,
 * src/AdaProx.jl:354-356, else the next one, :361-363); *numit = the `it` the reference returns.  gammas / norm_res (length maxit, may be NULL) receive the per-iteration
 * record fields of src/AdaProx.jl:351.  Returns 0, or 1 if memory ran out.
 */
int apx_oracle_lasso_adaptive_proxgrad(const double* A, int64_t m, int64_t n, int64_t lda, const double* b, double lam,
                                       double gamma0, double tol, int64_t maxit, double* x, double* gammas,
                                       double* norm_res_out, double* f_out, int64_t* numit) {
  double* res = (double*)malloc(sizeof(double) * (size_t)m);
  double* grad = (double*)malloc(sizeof(double) * (size_t)n);
  double* grad_prev = (double*)malloc(sizeof(double) * (size_t)n);
  double* x_prev = (double*)malloc(sizeof(double) * (size_t)n);
  double* v = (double*)malloc(sizeof(double) * (size_t)n);
  if (!res || !grad || !grad_prev || !x_prev || !v) {
    free(res); free(grad); free(grad_prev); free(x_prev); free(v);
    return 1;
  }
  double gamma = gamma0, g1 = gamma0, g0 = gamma0;                 /* :252-256  state = (gamma, gamma) */
  (void)residual(A, m, n, lda, x, b, res);                        /* :328 */
  gradient(A, m, n, lda, res, grad);
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) {                               /* :330-331 */
    v[i] = x[i] - gamma * grad[i];
    x_prev[i] = x[i];
    grad_prev[i] = grad[i];
  }
  soft_threshold(v, gamma * lam, n, x);                           /* :332 */
  for (int64_t it = 1; it <= maxit; ++it) {                       /* :334 */
    const double ss = residual(A, m, n, lda, x, b, res);          /* :336 */
    const double nr = sqrt(ss);
    const double f_x = 0.5 * (nr * nr);                           /* runme.jl:24: nrm2 then square */
    gradient(A, m, n, lda, res, grad);
    double npr2 = 0.0, ndg2 = 0.0, dgdx = 0.0, ndx2 = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : npr2, ndg2, dgdx, ndx2)
    for (int64_t i = 0; i < n; ++i) {
      const double pr = (v[i] - x[i]) / gamma + grad[i];          /* :338 */
      const double dg = grad[i] - grad_prev[i], dx = x[i] - x_prev[i];
      npr2 += pr * pr;
      ndg2 += dg * dg;
      dgdx += dg * dx;
      ndx2 += dx * dx;
    }
    /* :258-272 with norm_A = 0 (xi = 0, second term 1/0 = Inf), t = 1, delta = 0 */
    const double ndg = sqrt(ndg2), ndx = sqrt(ndx2);
    const double C = nan_to_zero((ndg * ndg) / dgdx);             /* :260 */
    const double L = nan_to_zero(dgdx / (ndx * ndx));             /* :261 */
    const double D = g1 * L * (g1 * C - 1.0);                     /* :262 */
    const double t1 = g1 * sqrt(1.0 + g1 / g0);
    const double inner = D + sqrt(D * D);                         /* D + sqrt(D^2 + xi*(...)), xi = 0 */
    const double t3 = (g1 * 1.0) / sqrt(2.0 * inner);             /* gamma1*sqrt(1) / sqrt(2*(1+delta)*inner) */
    const double gnew = jl_min(jl_min(t1, INFINITY), t3);         /* :263-270 */
    g0 = g1;                                                      /* :272  state = (gamma, gamma1) */
    g1 = gnew;
    gamma = gnew;
    const double norm_res = sqrt(sqrt(npr2) * sqrt(npr2) + 0.0);  /* :348 (dual_res = 0) */
    if (gammas) gammas[it - 1] = gamma;
    if (norm_res_out) norm_res_out[it - 1] = norm_res;
    if (f_out) f_out[it - 1] = f_x;
    if (norm_res <= tol) {                                        /* :354-356 */
      if (numit) *numit = it;
      free(res); free(grad); free(grad_prev); free(x_prev); free(v);
      return 0;
    }
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {                             /* :359-360 */
      v[i] = x[i] - gamma * grad[i];
      x_prev[i] = x[i];
      grad_prev[i] = grad[i];
    }
    soft_threshold(v, gamma * lam, n, x);                         /* :361 */
  }
  if (numit) *numit = maxit;
  free(res); free(grad); free(grad_prev); free(x_prev); free(v);
  return 0;
}

/* ---- the synthetic lasso of experiments/lasso/runme.jl:40-79, at sizes where numpy takes minutes ------------------
 * Same counter-based generator as oracle/problems.py::uniform_matrix (element (i, j) of the global m_total x n matrix
 * = lo + (hi - lo) * u(seed, j * m_total + row_offset + i), splitmix64) — bit-identical, tests/test_oracle_c.py. */
static double hash_uniform(uint64_t seed, uint64_t idx) {
  uint64_t z = seed * 0x9E3779B97F4A7C15ULL + idx;
  z += 0x9E3779B97F4A7C15ULL;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  z = z ^ (z >> 31);
  return (double)(z >> 11) * (1.0 / 9007199254740992.0);
}

void apx_oracle_fill_uniform(double* A, int64_t m, int64_t n, int64_t lda, uint64_t seed, double lo, double hi,
                             int64_t row_offset, int64_t m_total) {
#pragma omp parallel for schedule(static)
  for (int64_t j = 0; j < n; ++j)
    for (int64_t i = 0; i < m; ++i)
      A[j * lda + i] = lo + (hi - lo) * hash_uniform(seed, (uint64_t)j * (uint64_t)m_total + (uint64_t)(row_offset + i));
}

/* A[:, j] *= alpha[j]  (`C * diagm(0 => alpha)`, experiments/lasso/runme.jl:68) */
void apx_oracle_scale_columns(double* A, int64_t m, int64_t n, int64_t lda, const double* alpha) {
#pragma omp parallel for schedule(static)
  for (int64_t j = 0; j < n; ++j)
    for (int64_t i = 0; i < m; ++i) A[j * lda + i] *= alpha[j];
}

/* y = A*x and g = A'*r with the kernels of the iteration above */
void apx_oracle_gemv_n(const double* A, int64_t m, int64_t n, int64_t lda, const double* x, double* y) {
  double* zero = (double*)calloc((size_t)m, sizeof(double));
  if (!zero) return;
  (void)residual(A, m, n, lda, x, zero, y);
  free(zero);
}
void apx_oracle_gemv_t(const double* A, int64_t m, int64_t n, int64_t lda, const double* r, double* g) {
  gradient(A, m, n, lda, r, g);
}
