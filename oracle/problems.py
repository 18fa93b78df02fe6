"""Synthetic problem generators for the oracle side.  TEST INFRASTRUCTURE ONLY.

Restates the experiment set-ups of the reference with a counter-based RNG
(``u(seed, e)`` = splitmix64 finaliser of a linear element index) so that the
HIP library can generate the *same* matrices on device, element for element
(``apx_matrix_fill_uniform`` in include/adaprox_hip.h), without ever moving a
40 GB matrix over PCIe.  Julia's own RNG stream cannot be reproduced, so the
random draws differ from the reference's; the *construction* is the same.

* ``random_lasso``      — experiments/lasso/runme.jl:40-84
* ``random_logreg``     — experiments/sparse_logreg/runme.jl:52-63 (dense synthetic X, BASELINE C3)
* ``random_lad``        — experiments/least_absolute_deviation/runme.jl:39-46 (dense synthetic X, BASELINE C5)
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

_GOLD = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def hash_uniform(seed: int, idx: np.ndarray) -> np.ndarray:
    """u in [0,1): splitmix64(seed*GOLD + idx) >> 11, times 2^-53.  idx: uint64 array."""
    with np.errstate(over="ignore"):
        z = np.uint64(seed) * _GOLD + idx.astype(np.uint64)
        z = z + _GOLD
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def uniform_matrix(seed: int, m: int, n: int, lo: float, hi: float, dtype=np.float64,
                   row_offset: int = 0, m_total: int | None = None) -> np.ndarray:
    """Element (i, j) of the GLOBAL m_total x n matrix has linear index j*m_total + i
    (column-major, like Julia).  Returns rows [row_offset, row_offset+m) as an
    (m, n) Fortran-ordered array."""
    m_total = m if m_total is None else m_total
    i = np.arange(row_offset, row_offset + m, dtype=np.uint64)[:, None]
    j = np.arange(n, dtype=np.uint64)[None, :]
    idx = j * np.uint64(m_total) + i
    u = hash_uniform(seed, idx)
    return np.asfortranarray((lo + (hi - lo) * u).astype(dtype))


def uniform_vector(seed: int, n: int, lo: float = 0.0, hi: float = 1.0) -> np.ndarray:
    return lo + (hi - lo) * hash_uniform(seed, np.arange(n, dtype=np.uint64))


@dataclass
class LassoProblem:
    A: np.ndarray
    b: np.ndarray
    x_star: np.ndarray
    y_star: np.ndarray
    lam: float
    optimum: float
    alpha: np.ndarray
    perm: np.ndarray


def lasso_column_scaling(CTy: np.ndarray, n: int, pfactor: float, lam: float, seed: int):
    """experiments/lasso/runme.jl:51-67,70-75 — the O(n) host part of the generator,
    shared with the device-side generator (which obtains CTy from a HIP GEMV-T)."""
    p = n / pfactor                                                  # :45 (Float64, Appendix A.12)
    perm = np.argsort(-CTy, kind="stable")                           # :52 sortperm(rev=true)
    r_alpha = uniform_vector(seed + 2, n)                            # stands in for rand() at :63
    r_x = uniform_vector(seed + 3, n)                                # stands in for rand() at :73
    alpha = np.zeros(n)
    for i in range(1, n + 1):                                        # :55-67
        pi = perm[i - 1]
        if i <= p:
            alpha[pi] = lam / CTy[pi]
        else:
            temp = CTy[pi]
            if temp < 0.1 * lam:
                alpha[pi] = lam
            else:
                alpha[pi] = lam * r_alpha[i - 1] / temp
    return p, perm, alpha, r_x


def random_lasso(m: int, n: int, pfactor: float = 10, seed: int = 0, lam: float = 1.0,
                 rho: float = 1.0) -> LassoProblem:
    """experiments/lasso/runme.jl:40-79 with the counter-based RNG."""
    y_star = uniform_vector(seed + 1, m)                             # :47
    y_star = y_star / np.sqrt(np.dot(y_star, y_star))                # :48
    C = uniform_matrix(seed, m, n, -1.0, 1.0)                        # :49
    CTy_signed = C.T @ y_star
    CTy = np.abs(CTy_signed)                                         # :51
    p, perm, alpha, r_x = lasso_column_scaling(CTy, n, pfactor, lam, seed)
    A = np.asfortranarray(C * alpha[None, :])                        # :68
    x_star = np.zeros(n)
    for i in range(1, n + 1):                                        # :70-75
        if i <= p:
            pi = perm[i - 1]
            # sign(dot(A[:,pi], y_star)) = sign(alpha*C'y) ; alpha > 0
            x_star[pi] = r_x[i - 1] * rho / np.sqrt(p) * np.sign(alpha[pi] * CTy_signed[pi])
    b = A @ x_star + y_star                                          # :76
    optimum = np.sqrt(np.dot(y_star, y_star)) / 2 + lam * np.sum(np.abs(x_star))   # :77
    return LassoProblem(A, b, x_star, y_star, lam, float(optimum), alpha, perm)


def opnorm_sq_power(A: np.ndarray, iters: int = 50, seed: int = 7) -> float:
    """lambda_max(A'A) by power iteration — stands in for ``opnorm(A)^2``
    (experiments/lasso/runme.jl:81), same recurrence as apx_matrix_opnorm_sq."""
    n = A.shape[1]
    v = uniform_vector(seed, n, -1.0, 1.0).astype(A.dtype)
    v = v / np.sqrt(np.dot(v.astype(np.float64), v.astype(np.float64))).astype(A.dtype)
    lam = 0.0
    for _ in range(iters):
        w = A.T @ (A @ v)
        nw = float(np.sqrt(np.dot(w.astype(np.float64), w.astype(np.float64))))
        lam = nw
        v = (w / nw).astype(A.dtype)
    return lam


@dataclass
class LogRegProblem:
    X: np.ndarray
    y: np.ndarray
    lam: float
    gamma0: float


def random_logreg(m: int, n: int, seed: int = 0, lam: float = 0.01, dtype=np.float32) -> LogRegProblem:
    """BASELINE C3 synthetic stand-in for experiments/sparse_logreg/runme.jl:52-63.
    X_ij ~ U(-1,1); w_true 5 % dense; y_i = 1[sigma(x_i.w_true) > u_i];
    gamma0 = 4m / ||[X 1]||_F^2 (upper-bounds 1/Lf of :58-59)."""
    X = uniform_matrix(seed, m, n, -1.0, 1.0, dtype=dtype)
    u_sel = uniform_vector(seed + 1, n)
    u_val = uniform_vector(seed + 2, n, -1.0, 1.0)
    w_true = np.where(u_sel < 0.05, 4.0 * u_val, 0.0)
    logits = X.astype(np.float64) @ w_true
    probs = 1.0 / (1.0 + np.exp(-logits))
    y = (probs > uniform_vector(seed + 3, m)).astype(dtype)
    fro2 = float(np.sum(X.astype(np.float64) ** 2)) + m
    return LogRegProblem(X, y, lam, 4.0 * m / fro2)


@dataclass
class LadProblem:
    A: np.ndarray
    y: np.ndarray
    lam: float
    norm_A: float


def random_lad(m: int, n: int, seed: int = 0, lam: float = 10.0) -> LadProblem:
    """BASELINE C5 synthetic stand-in for experiments/least_absolute_deviation/runme.jl:39-46:
    A = [X 1] dense (last column ones), y = A x_true + Laplace noise, norm_A Frobenius."""
    A = uniform_matrix(seed, m, n, -1.0, 1.0)
    A[:, -1] = 1.0                                                   # hcat(X, ones(m,1)) :42
    u_sel = uniform_vector(seed + 1, n)
    u_val = uniform_vector(seed + 2, n, -1.0, 1.0)
    x_true = np.where(u_sel < 0.1, u_val, 0.0)
    u = uniform_vector(seed + 3, m) - 0.5
    noise = -np.sign(u) * np.log1p(-2.0 * np.abs(u) + 1e-300)       # Laplace(0,1)
    y = A @ x_true + noise
    return LadProblem(np.asfortranarray(A), y, lam, float(np.linalg.norm(A)))
