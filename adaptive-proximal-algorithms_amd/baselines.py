"""Host mirror of the comparison methods the experiments run next to AdaPGM / AdaPDM
(SURVEY §8f rank 3), same names and keyword arguments as the reference:

    backtracking_nesterov(x0; f, g, gamma0, shrink, tol, maxit, name)      src/AdaProx.jl:66-84
    fixed_nesterov(x0; f, g, Lf | gamma, muf, mug, theta, tol, maxit)      src/AdaProx.jl:91-142
    agraal(x1; f, g, x0, gamma0, gamma_max, phi, tol, maxit)               src/AdaProx.jl:150-192
    malitsky_pock(x, y; f, g, h, A, sigma, t, tol, maxit)                  src/AdaProx.jl:555-629
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Optional

import numpy as np

from . import _lib as L
from .api import (Context, Record, Zero, _Problem, _advance_counts, _as_vec, _at, _count_bases, _opts,
                  _problem_dtype, _ptr)


def backtracking_nesterov(x0, *, f, g, gamma0, shrink=0.5, tol=1e-5, maxit=100_000, name="Backtracking Nesterov",
                          record: Record = None, ctx: Optional[Context] = None):
    """src/AdaProx.jl:66-84.  Returns (z, it)."""
    maxit = int(math.floor(maxit))
    dtype = _problem_dtype(x0, f, None)
    x0 = _as_vec(x0, dtype)
    prob = _Problem(f, g, Zero(), None, dtype, ctx).finalize(x0.size)
    try:
        opts, recs = _opts(maxit, record is not None)
        res = L.Result()
        z = np.empty_like(x0)
        base = _count_bases(f, g, None, None)
        prob.check(prob.lib.apx_backtracking_nesterov(prob.h, _ptr(x0), float(gamma0), float(shrink), float(tol),
                                                      maxit, C.byref(opts), _ptr(z), C.byref(res)))
        _advance_counts(f, g, None, None, res, prob.f_is_callback)
        if record is not None:
            # prox calls so far = trials = f_evals - 1 - (#extrapolation evals = it - 1)
            for i in range(int(res.n_records)):
                rr = recs[i]
                k = int(rr.it)
                record(dict(method=name, it=k, gamma=rr.gamma, norm_res=rr.norm_res, objective=rr.objective,
                            grad_f_evals=_at(base, "grad", int(rr.grad_evals)),
                            prox_g_evals=_at(base, "prox_g", int(rr.f_evals) - k),
                            f_evals=_at(base, "eval", int(rr.f_evals))))
        return z, int(res.numit)
    finally:
        prob.close()


def fixed_nesterov(x0, *, f, g, Lf=None, muf=0, mug=0, gamma=None, theta=None, tol=1e-5, maxit=100_000,
                   name="Fixed Nesterov", record: Record = None, ctx: Optional[Context] = None):
    """src/AdaProx.jl:91-142.  Returns (x, it)."""
    maxit = int(math.floor(maxit))
    assert (gamma is None) != (Lf is None)                       # :104
    if gamma is None:                                            # :105-107
        gamma = 1 / Lf
    mu = muf + mug
    q = gamma * mu / (1 + gamma * mug)
    assert q < 1                                                 # :110
    if theta is not None:
        assert 0 <= theta <= (1 / math.sqrt(q) if q > 0 else math.inf)      # :118
    dtype = _problem_dtype(x0, f, None)
    x0 = _as_vec(x0, dtype)
    prob = _Problem(f, g, Zero(), None, dtype, ctx).finalize(x0.size)
    try:
        opts, recs = _opts(maxit, record is not None)
        res = L.Result()
        x = np.empty_like(x0)
        base = _count_bases(f, g, None, None)
        prob.check(prob.lib.apx_fixed_nesterov(prob.h, _ptr(x0), float(gamma), float(muf), float(mug),
                                               -1.0 if theta is None else float(theta), float(tol), maxit,
                                               C.byref(opts), _ptr(x), C.byref(res)))
        _advance_counts(f, g, None, None, res, prob.f_is_callback)
        if record is not None:
            for i in range(int(res.n_records)):
                rr = recs[i]
                k = int(rr.it)
                record(dict(method=name, it=k, gamma=rr.gamma, norm_res=rr.norm_res, objective=rr.objective,
                            grad_f_evals=_at(base, "grad", k), prox_g_evals=_at(base, "prox_g", k),
                            f_evals=_at(base, "eval", k)))
        return x, int(res.numit)
    finally:
        prob.close()


def agraal(x1, *, f, g, x0=None, gamma0=None, gamma_max=1e6, phi=1.5, tol=1e-5, maxit=100_000, name="aGRAAL",
           record: Record = None, ctx: Optional[Context] = None, rng: Optional[np.random.Generator] = None):
    """src/AdaProx.jl:150-192.  Returns (x, it).  ``x0 = x1 + randn`` when not given (:162-164)."""
    maxit = int(math.floor(maxit))
    dtype = _problem_dtype(x1, f, None)
    x1 = _as_vec(x1, dtype)
    if x0 is None:
        x0 = x1 + (rng or np.random.default_rng()).standard_normal(x1.shape).astype(dtype)
    x0 = _as_vec(x0, dtype)
    prob = _Problem(f, g, Zero(), None, dtype, ctx).finalize(x1.size)
    try:
        opts, recs = _opts(maxit, record is not None)
        res = L.Result()
        x = np.empty_like(x1)
        base = _count_bases(f, g, None, None)
        prob.check(prob.lib.apx_agraal(prob.h, _ptr(x1), _ptr(x0), 0.0 if gamma0 is None else float(gamma0),
                                       float(gamma_max), float(phi), float(tol), maxit, C.byref(opts), _ptr(x),
                                       C.byref(res)))
        _advance_counts(f, g, None, None, res, prob.f_is_callback)
        if record is not None:
            for i in range(int(res.n_records)):
                rr = recs[i]
                k = int(rr.it)
                record(dict(method=name, it=k, gamma=rr.gamma, norm_res=rr.norm_res, objective=rr.objective,
                            grad_f_evals=_at(base, "grad", k + 1), prox_g_evals=_at(base, "prox_g", k),
                            f_evals=_at(base, "eval", k + 1)))
        return x, int(res.numit)
    finally:
        prob.close()


def malitsky_pock(x, y, *, f, g, h, A, sigma, t=1.0, tol=1e-5, maxit=10_000, name="MP-ls", record: Record = None,
                  ctx: Optional[Context] = None, A_allreduce=None):
    """src/AdaProx.jl:581-629 (+ backtrack_stepsize_MP :555-579).  Returns (x, y, it).  (``A_allreduce``: the rows of A, y and
    h's shift are sharded over ranks — A' y, a trial's ||A_x - A_x_prev||^2 and the residual sums are all-reduced, every rank
    takes the same linesearch decisions ; a separable h only.)"""
    maxit = int(math.floor(maxit))
    dtype = _problem_dtype(x, f, A)
    x = _as_vec(x, dtype)
    prob = _Problem(f, g, h, A, dtype, ctx, lin_allreduce=A_allreduce).finalize(x.size)
    try:
        yv = _as_vec(y, dtype)
        if yv.size != prob._d.m:
            raise ValueError(f"y has length {yv.size} but A has {prob._d.m} rows")
        opts, recs = _opts(maxit, record is not None)
        res = L.Result()
        x_out, y_out = np.empty_like(x), np.empty_like(yv)
        base = _count_bases(f, g, h, A)
        prob.check(prob.lib.apx_malitsky_pock(prob.h, _ptr(x), _ptr(yv), float(sigma), float(t), float(tol), maxit,
                                              C.byref(opts), _ptr(x_out), _ptr(y_out), C.byref(res)))
        _advance_counts(f, g, h, A, res, prob.f_is_callback)
        if record is not None:
            for i in range(int(res.n_records)):
                rr = recs[i]
                k = int(rr.it)
                trials = int(rr.f_evals) - k                    # f evals = 1 per iteration + 1 per trial
                record(dict(method=name, it=k, gamma=rr.gamma, sigma=rr.sigma, norm_res=rr.norm_res,
                            objective=rr.objective,
                            grad_f_evals=_at(base, "grad", 2 * k), prox_g_evals=_at(base, "prox_g", trials),
                            prox_h_evals=_at(base, "prox_h", k), A_evals=_at(base, "mul", 1 + trials),
                            At_evals=_at(base, "amul", 1 + k), f_evals=_at(base, "eval", int(rr.f_evals))))
        return x_out, y_out, int(res.numit)
    finally:
        prob.close()
