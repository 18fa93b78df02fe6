"""ctypes wrapper of oracle/lasso_omp.c — TEST INFRASTRUCTURE (the multi-threaded CPU baseline of bench.py and its
check against the numpy oracle).  Never imported by the product package."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(_HERE, "lasso_omp.c")
OUT_DIR = os.path.join(_HERE, "_build")
LIB = os.path.join(OUT_DIR, "liblasso_omp.so")


def build(native: bool = False, out: str = LIB) -> str:
    """gcc -O3 -fopenmp.  native=True adds -march=native: only for a library that is used on the machine that built
    it (bench.py's cpu_baseline leg rebuilds on the GPU box's host); the default build is portable."""
    os.makedirs(os.path.dirname(out), exist_ok=True)
    cmd = ["gcc", "-O3", "-fopenmp", "-shared", "-fPIC", "-o", out, SRC, "-lm"]
    if native:
        cmd.insert(2, "-march=native")
    subprocess.run(cmd, check=True)
    return out


def load(path: str = LIB) -> C.CDLL:
    if not os.path.exists(path):
        build(out=path)
    lib = C.CDLL(path)
    lib.apx_oracle_omp_threads.restype = C.c_int
    lib.apx_oracle_omp_set_threads.restype = None
    lib.apx_oracle_omp_set_threads.argtypes = [C.c_int]
    lib.apx_oracle_lasso_adaptive_proxgrad.restype = C.c_int
    lib.apx_oracle_lasso_adaptive_proxgrad.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_double,
                                                       C.c_double, C.c_double, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                       C.POINTER(C.c_int64)]
    lib.apx_oracle_fill_uniform.restype = None
    lib.apx_oracle_fill_uniform.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_uint64, C.c_double, C.c_double,
                                            C.c_int64, C.c_int64]
    lib.apx_oracle_scale_columns.restype = None
    lib.apx_oracle_scale_columns.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p]
    for fn in (lib.apx_oracle_gemv_n, lib.apx_oracle_gemv_t):
        fn.restype = None
        fn.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]
    return lib


def gemv_n(A: np.ndarray, x: np.ndarray, lib=None) -> np.ndarray:
    lib = lib or load()
    m, n = A.shape
    y = np.empty(m)
    lib.apx_oracle_gemv_n(A.ctypes.data, m, n, max(m, 1), np.ascontiguousarray(x, dtype=np.float64).ctypes.data, y.ctypes.data)
    return y


def gemv_t(A: np.ndarray, r: np.ndarray, lib=None) -> np.ndarray:
    lib = lib or load()
    m, n = A.shape
    g = np.empty(n)
    lib.apx_oracle_gemv_t(A.ctypes.data, m, n, max(m, 1), np.ascontiguousarray(r, dtype=np.float64).ctypes.data, g.ctypes.data)
    return g


def random_lasso(m: int, n: int, pfactor: float = 10, seed: int = 0, lam: float = 1.0, rho: float = 1.0, lib=None):
    """oracle/problems.py::random_lasso (experiments/lasso/runme.jl:40-79) with the O(mn) parts — the matrix fill, the
    column scaling and the two products — in C + OpenMP: the same matrix bit for bit, b to rounding, in seconds
    instead of minutes at the 10 000 x 100 000 of bench.py's CPU sample."""
    from . import problems as P
    lib = lib or load()
    y_star = P.uniform_vector(seed + 1, m)                            # :47
    y_star = y_star / np.sqrt(np.dot(y_star, y_star))                 # :48
    A = np.empty((m, n), order="F")
    lib.apx_oracle_fill_uniform(A.ctypes.data, m, n, max(m, 1), seed, -1.0, 1.0, 0, m)      # :49
    cty_signed = gemv_t(A, y_star, lib)
    cty = np.abs(cty_signed)                                          # :51
    p, perm, alpha, r_x = P.lasso_column_scaling(cty, n, pfactor, lam, seed)
    lib.apx_oracle_scale_columns(A.ctypes.data, m, n, max(m, 1), alpha.ctypes.data)         # :68
    x_star = np.zeros(n)
    head = perm[:int(np.floor(p))] if p >= 1 else perm[:0]            # i <= p  (:72)
    k = np.arange(len(head))
    x_star[head] = r_x[k] * rho / np.sqrt(p) * np.sign(alpha[head] * cty_signed[head])      # :70-75
    b = gemv_n(A, x_star, lib) + y_star                               # :76
    optimum = np.sqrt(np.dot(y_star, y_star)) / 2 + lam * np.sum(np.abs(x_star))            # :77
    return P.LassoProblem(A, b, x_star, y_star, lam, float(optimum), alpha, perm)


def adaptive_proxgrad_lasso(A: np.ndarray, b: np.ndarray, lam: float, gamma0: float, maxit: int, x0=None, lib=None,
                            tol: float = 0.0):
    """adaptive_proxgrad(x0; f = LinearLeastSquares(A, b), g = NormL1(lam), rule = OurRule(gamma = gamma0), tol, maxit)
    (src/AdaProx.jl:418-421).  Returns (x, gammas, norm_res, f_x) — the last three per iteration done."""
    lib = lib or load()
    A = np.asfortranarray(A, dtype=np.float64)
    m, n = A.shape
    b = np.ascontiguousarray(b, dtype=np.float64)
    x = np.zeros(n) if x0 is None else np.array(x0, dtype=np.float64)
    gam, nres, fx = np.empty(maxit), np.empty(maxit), np.empty(maxit)
    it = C.c_int64(0)
    rc = lib.apx_oracle_lasso_adaptive_proxgrad(A.ctypes.data, m, n, max(m, 1), b.ctypes.data, float(lam), float(gamma0),
                                                float(tol), int(maxit), x.ctypes.data, gam.ctypes.data, nres.ctypes.data,
                                                fx.ctypes.data, C.byref(it))
    if rc != 0:
        raise MemoryError("lasso_omp: allocation failed")
    k = it.value
    return x, gam[:k], nres[:k], fx[:k]
