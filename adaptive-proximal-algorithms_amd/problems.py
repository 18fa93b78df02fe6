"""Experiment set-ups of the reference, built ON THE DEVICE through the C-ABI, so that a
40 GB design matrix never crosses PCIe.

* ``device_lasso``  — experiments/lasso/runme.jl:40-84 (Nesterov-style random lasso with a
  known optimum).  The O(mn) parts (`rand(m,n)`, `C' * y_star`, column scaling,
  `A * x_star`, `opnorm(A)^2`) run as HIP kernels; the O(n) bookkeeping (sortperm, alpha,
  x_star) stays on the host exactly as in the reference.
* ``device_lad``    — experiments/least_absolute_deviation/runme.jl:39-46 on synthetic dense X.
* ``device_logreg`` — experiments/sparse_logreg/runme.jl:52-63 on synthetic dense X.

Row sharding: each rank builds rows [rank*m_local, (rank+1)*m_local) of the global matrix
(the RNG is counter-based) and the few global reductions of the set-up go through
``allreduce_host`` (a callable summing a numpy vector over ranks; identity when None).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Callable, Optional

import numpy as np

from ._lib import ApxError
from .api import Context, DeviceMatrix, _hash_uniform


@dataclass
class DeviceLasso:
    A: DeviceMatrix
    b: np.ndarray
    x_star: np.ndarray
    lam: float
    optimum: float
    Lf: float
    dual_layout: bool = False      # a transposed copy of A is resident too
    layout: str = "cols"           # DeviceMatrix.layout of A: "cols", "dual" or "rows" (one resident copy, row-contiguous)


def _ident(v):
    return v


def device_lasso(m_local: int, n: int, *, pfactor: float = 10, seed: int = 0, lam: float = 1.0, rho: float = 1.0,
                 rank: int = 0, world: int = 1, power_iters: int = 50, ctx: Optional[Context] = None,
                 allreduce_host: Optional[Callable[[np.ndarray], np.ndarray]] = None,
                 dual_layout: bool = False, m_total: Optional[int] = None,
                 row_offset: Optional[int] = None, layout: str = "cols") -> DeviceLasso:
    """``dual_layout``: keep the transposed copy of A too (DeviceMatrix.build_transpose).
    ``layout="rows"``: A lives in the row-contiguous storage ALONE from the start (one resident copy ; ``dual_layout`` is moot).
    ``m_total`` / ``row_offset``: an uneven row split (default: ``world`` equal shards of ``m_local`` rows)."""
    ar = allreduce_host or _ident
    m_total = m_local * world if m_total is None else m_total
    r0 = rank * m_local if row_offset is None else row_offset
    # :47-48  y_star = rand(m); y_star ./= norm(y_star)
    y_loc = _hash_uniform(seed + 1, m_total, 0.0, 1.0)[r0:r0 + m_local]
    y_norm = math.sqrt(float(ar(np.array([np.dot(y_loc, y_loc)]))[0]))
    y_loc = y_loc / y_norm
    # :49  C = rand(m, n) .* 2 .- 1   (on device)
    A = DeviceMatrix.uniform(m_local, n, seed=seed, lo=-1.0, hi=1.0, dtype=np.float64, row_offset=r0,
                             m_total=m_total, ctx=ctx, layout=layout)
    # :51  CTy = abs.(C' * y_star)    (HIP GEMV-T, partial sums all-reduced over the row shards)
    cty_signed = ar(A.rmatvec(y_loc))
    cty = np.abs(cty_signed)
    # :52-67  perm, alpha   (O(n), host — vectorised form of the reference loop)
    p = n / pfactor
    perm = np.argsort(-cty, kind="stable")
    r_alpha = _hash_uniform(seed + 2, n, 0.0, 1.0)
    r_x = _hash_uniform(seed + 3, n, 0.0, 1.0)
    pos = np.arange(1, n + 1)
    cp = cty[perm]
    head = pos <= p
    a_sorted = np.where(head, lam / cp, np.where(cp < 0.1 * lam, lam, lam * r_alpha / cp))
    alpha = np.empty(n)
    alpha[perm] = a_sorted
    A.scale_columns(alpha)                                      # :68  A = C * diagm(alpha)
    have_dual = False
    if dual_layout and layout != "rows":                        # A is final from here on
        try:
            A.build_transpose()
            have_dual = True
        except ApxError as e:                                   # e.g. not enough free HBM for the second copy
            import warnings
            warnings.warn(f"dual layout not available, continuing with the single layout: {e}")
    # :70-75  x_star
    xs_sorted = np.where(head, r_x * rho / math.sqrt(p) * np.sign(alpha[perm] * cty_signed[perm]), 0.0)
    x_star = np.zeros(n)
    x_star[perm] = xs_sorted
    b = A.matvec(x_star) + y_loc                                # :76  (local rows)
    optimum = 1.0 / 2 + lam * float(np.sum(np.abs(x_star)))     # :77  norm(y_star) = 1
    # :81  Lf = opnorm(A)^2 by power iteration on A'A (global over shards)
    v = _hash_uniform(7, n, -1.0, 1.0)
    v = v / math.sqrt(float(np.dot(v, v)))
    Lf = 0.0
    for _ in range(power_iters):
        w = ar(A.rmatvec(A.matvec(v)))
        Lf = math.sqrt(float(np.dot(w, w)))
        v = w / Lf
    return DeviceLasso(A, b, x_star, lam, optimum, Lf, have_dual, A.layout)


@dataclass
class DeviceLad:
    A: DeviceMatrix
    y: np.ndarray
    lam: float
    norm_A: float


def device_lad(m: int, n: int, *, seed: int = 0, lam: float = 10.0, ctx: Optional[Context] = None) -> DeviceLad:
    A = DeviceMatrix.uniform(m, n, seed=seed, lo=-1.0, hi=1.0, dtype=np.float64, ctx=ctx)
    A.fill_column(n - 1, 1.0)                                   # hcat(X, ones(m, 1))  :42
    u_sel = _hash_uniform(seed + 1, n, 0.0, 1.0)
    u_val = _hash_uniform(seed + 2, n, -1.0, 1.0)
    x_true = np.where(u_sel < 0.1, u_val, 0.0)
    u = _hash_uniform(seed + 3, m, 0.0, 1.0) - 0.5
    noise = -np.sign(u) * np.log1p(-2.0 * np.abs(u) + 1e-300)
    y = A.matvec(x_true) + noise
    return DeviceLad(A, y, lam, A.norm())                       # norm(A) Frobenius  :46


@dataclass
class DeviceLogReg:
    X: DeviceMatrix
    y: np.ndarray
    lam: float
    gamma0: float


def device_logreg(m: int, n: int, *, seed: int = 0, lam: float = 0.01, dtype=np.float32,
                  ctx: Optional[Context] = None) -> DeviceLogReg:
    X = DeviceMatrix.uniform(m, n, seed=seed, lo=-1.0, hi=1.0, dtype=dtype, ctx=ctx)
    u_sel = _hash_uniform(seed + 1, n, 0.0, 1.0)
    u_val = _hash_uniform(seed + 2, n, -1.0, 1.0)
    w_true = np.where(u_sel < 0.05, 4.0 * u_val, 0.0).astype(dtype)
    logits = X.matvec(w_true).astype(np.float64)
    probs = 1.0 / (1.0 + np.exp(-logits))
    y = (probs > _hash_uniform(seed + 3, m, 0.0, 1.0)).astype(dtype)
    fro2 = X.sumsq() + m
    return DeviceLogReg(X, y, lam, 4.0 * m / fro2)
