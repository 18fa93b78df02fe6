"""CPU oracle for the AdaProx hot path.  TEST INFRASTRUCTURE ONLY.

This file is a numpy restatement of the reference's algorithm (AdaProx.jl,
pylat/adaptive-proximal-algorithms).  It is the *checker* for the HIP path:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it.  Nothing in the product package
(``adaptive-proximal-algorithms_amd/``) imports or calls it.

Every function cites the reference file:line it follows (paths relative to the
reference checkout).  The operation ORDER of the Julia source is kept (norm
then square, separate C / L quotients, ``nan_to_zero`` on C and L only, Julia's
NaN-propagating ``min``), because the adaptive step-size recurrences are
sensitive to it.

PARITY PIN STATUS ("parity unpinned" at the iterate level)
----------------------------------------------------------
* The reference is 100 % Julia and no Julia runtime exists in this image, so the
  reference cannot be run here and no golden vectors can be produced from it.
* The reference's own tests hold NO iterate-level vectors; they pin only
  end-state known answers and counter semantics (``test/runtests.jl:25-90``).
  This oracle is checked against all of them (tests/test_oracle_kats.py), plus
  the analytic optima at ``experiments/lasso/runme.jl:77`` and
  ``experiments/nesterov_worst_case/runme.jl:53``.
* The prox operators (NormL1, IndBox, NormL2, Translate, Zero, IndZero,
  ConvexConjugate) live in ProximalOperators.jl / ProximalCore.jl, which are
  third-party, un-vendored and *version-unpinned* in the reference
  (``Project.toml:6-9`` has no ``[compat]``; ``Manifest.toml`` is git-ignored).
  They are restated here from their published definitions / recalled upstream
  forms; the reference holds no test at that boundary.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Any, Callable, Optional

import numpy as np

__all__ = [
    "Counting", "without_counting", "eval_with_pullback", "eval_with_gradient",
    "nan_to_zero", "upper_bound", "backtrack_stepsize", "backtracking_proxgrad",
    "FixedStepsize", "MalitskyMishchenkoRule", "OurRule", "OurRulePlus", "stepsize",
    "adaptive_primal_dual", "condat_vu", "adaptive_proxgrad", "fixed_proxgrad",
    "NormL1", "NormL2", "IndBox", "Zero", "IndZero", "Translate", "ConvexConjugate",
    "convex_conjugate", "prox",
    "LinearLeastSquares", "LogisticLoss", "Quadratic", "AdjointCounting",
    "DomainError", "jl_min", "jl_norm",
]


class DomainError(ValueError):
    """Julia's ``sqrt`` of a negative real throws DomainError; mirrored here."""


# ----------------------------------------------------------------------------
# Julia scalar semantics
# ----------------------------------------------------------------------------

def jl_min(*vals: float) -> float:
    """Julia ``min`` propagates NaN (C ``fmin`` does not).  SURVEY Appendix A.6."""
    out = vals[0]
    for v in vals:
        if math.isnan(v):
            return float("nan")
        if v < out:
            out = v
    return out


def jl_sqrt(v: float) -> float:
    if v < 0:
        raise DomainError(f"sqrt of negative value {v}")
    return math.sqrt(v)


def jl_div(a: float, b: float) -> float:
    """IEEE float division (Julia ``/`` on Float64: 1/0 = Inf, 0/0 = NaN)."""
    with np.errstate(divide="ignore", invalid="ignore"):
        return float(np.float64(a) / np.float64(b))


def jl_norm(v: np.ndarray) -> float:
    """``LinearAlgebra.norm`` of a vector = 2-norm (Appendix A.1).  Accumulated
    in float64 also for float32 storage (matches the device reductions)."""
    v = np.asarray(v)
    return math.sqrt(float(np.dot(v.astype(np.float64, copy=False).ravel(),
                                  v.astype(np.float64, copy=False).ravel())))


def jl_dot(a: np.ndarray, b: np.ndarray) -> float:
    return float(np.dot(np.asarray(a, dtype=np.float64).ravel(), np.asarray(b, dtype=np.float64).ravel()))


# ----------------------------------------------------------------------------
# src/counting.jl
# ----------------------------------------------------------------------------

_counting_enabled = True  # src/counting.jl:4


def is_counting_enabled() -> bool:  # src/counting.jl:6
    return _counting_enabled


def without_counting(fn: Callable[[], Any]) -> Any:
    """src/counting.jl:8-14 (not exception safe in the reference either)."""
    global _counting_enabled
    _counting_enabled = False
    ret = fn()
    _counting_enabled = True
    return ret


class Counting:
    """src/counting.jl:16-34.  Wraps f, g or a matrix A and counts calls."""

    def __init__(self, f: Any):
        self.f = f
        self.eval_count = 0
        self.grad_count = 0
        self.prox_count = 0
        self.mul_count = 0
        self.amul_count = 0

    def __call__(self, *args):  # src/counting.jl:34
        return self.f(*args)

    # src/counting.jl:36-51
    def eval_with_pullback(self, x):
        if is_counting_enabled():
            self.eval_count += 1
        f_x, pb = eval_with_pullback(self.f, x)

        def counting_pullback():
            if is_counting_enabled():
                self.grad_count += 1
            return pb()

        return f_x, counting_pullback

    # src/counting.jl:53-59
    def prox(self, x, gamma=1.0):
        if is_counting_enabled():
            self.prox_count += 1
        return prox(self.f, x, gamma)

    # src/counting.jl:65-82
    def norm(self):
        return matrix_norm(self.f)

    @property
    def T(self):
        return AdjointCounting(self)

    def __matmul__(self, x):
        if is_counting_enabled():
            self.mul_count += 1
        return mul(self.f, x)


class AdjointCounting:  # src/counting.jl:61-63,76-82
    def __init__(self, op: Counting):
        self.op = op

    def __matmul__(self, x):
        if is_counting_enabled():
            self.op.amul_count += 1
        return amul(self.op.f, x)


def grad_count(c):  # src/counting.jl:84-85
    return c.grad_count if isinstance(c, Counting) else None


def prox_count(c):  # src/counting.jl:87-88
    return c.prox_count if isinstance(c, Counting) else None


def mul_count(c):  # src/counting.jl:90-91
    return c.mul_count if isinstance(c, Counting) else None


def amul_count(c):  # src/counting.jl:93-94
    return c.amul_count if isinstance(c, Counting) else None


def eval_count(c):  # src/counting.jl:96-97
    return c.eval_count if isinstance(c, Counting) else None


# ----------------------------------------------------------------------------
# Linear operator convention: ``A*x``, ``A'*y``, ``norm(A)``.
# ``A`` may be a numpy matrix, a Counting wrapper, or the integer 0
# (adaptive_proxgrad passes ``A = 0``: src/AdaProx.jl:419, Appendix A.7).
# ----------------------------------------------------------------------------

def mul(A, x):
    if isinstance(A, Counting):
        return A @ x
    if np.isscalar(A):
        return A * x  # 0 * x -> zeros(n)
    return A @ x


def amul(A, y):
    if isinstance(A, Counting):
        return A.T @ y
    if np.isscalar(A):
        return A * y  # 0' * y -> zeros
    return A.T @ y


def matrix_norm(A) -> float:
    """``norm(A)`` of a matrix is FROBENIUS (Appendix A.1)."""
    if isinstance(A, Counting):
        return A.norm()
    return float(np.linalg.norm(np.asarray(A, dtype=np.float64)))


# ----------------------------------------------------------------------------
# Gradient evaluation interface  (src/AdaProx.jl:9-16)
# ----------------------------------------------------------------------------

def eval_with_pullback(f, x):
    """src/AdaProx.jl:11 — duck-typed dispatch on ``f.eval_with_pullback``."""
    if not hasattr(f, "eval_with_pullback"):
        raise TypeError(f"eval_with_pullback not defined for type {type(f)}")
    return f.eval_with_pullback(x)


def eval_with_gradient(f, x):  # src/AdaProx.jl:13-16
    f_x, pb = eval_with_pullback(f, x)
    return f_x, pb()


def call_f(f, x):
    """``f(x)`` — experiments define ``(f::T)(w) = eval_with_pullback(f, w)[1]``."""
    if isinstance(f, Counting):
        return call_f(f.f, x)
    if callable(f):
        return f(x)
    return eval_with_pullback(f, x)[0]


# ----------------------------------------------------------------------------
# Utilities (src/AdaProx.jl:22-26)
# ----------------------------------------------------------------------------

def nan_to_zero(v: float) -> float:  # src/AdaProx.jl:24
    return 0.0 if math.isnan(v) else v


def upper_bound(x, f_x, grad_x, z, gamma):  # src/AdaProx.jl:26
    d = z - x
    return f_x + jl_dot(grad_x, d) + jl_div(1.0, 2 * gamma) * jl_norm(d) ** 2


# ----------------------------------------------------------------------------
# Prox operators (ProximalCore.jl / ProximalOperators.jl — third-party,
# unpinned, restated from definitions).
# ``prox(g, x, gamma) -> (y, g(y))``.
# ----------------------------------------------------------------------------

@dataclass
class Zero:
    """ProximalCore.Zero: f = 0, prox = identity (copy)."""

    def __call__(self, x):
        return 0.0

    def prox(self, x, gamma=1.0):
        return np.array(x, copy=True), 0.0

    # experiments/least_absolute_deviation/runme.jl:18-21
    def eval_with_pullback(self, x):
        return self(x), (lambda: np.zeros_like(x))


@dataclass
class IndZero:
    """ProximalCore.IndZero: indicator of {0}; prox = 0."""

    def __call__(self, x):
        return 0.0 if not np.any(x) else float("inf")

    def prox(self, x, gamma=1.0):
        return np.zeros_like(x), 0.0


@dataclass
class NormL1:
    """ProximalOperators.NormL1(lambda): f = lambda*||x||_1.
    Upstream prox form as SURVEY §8a P1 recalls it (the package is un-vendored):
        gl = gamma*lambda
        y[i] = x[i] + (x[i] <= -gl ? gl : (x[i] >= gl ? -gl : -x[i]))
    Verified against the DEFINITION argmin_y lam|y| + (y - x)^2 / (2 gamma) by brute force, and through the end states
    of three third-party solvers, in tests/test_independent_crosscheck.py
    (test_prox_operators_against_their_definition, test_lad_reaches_the_linear_programs_solution,
    test_l1_logistic_with_penalised_bias_reaches_liblinears_solution, test_oracle_vs_sklearn_lasso)."""
    lam: float = 1.0

    def __call__(self, x):
        return self.lam * float(np.sum(np.abs(np.asarray(x, dtype=np.float64))))

    def prox(self, x, gamma=1.0):
        gl = x.dtype.type(gamma * self.lam) if x.dtype == np.float32 else gamma * self.lam
        adj = np.where(x <= -gl, gl, np.where(x >= gl, -gl, -x))
        y = x + adj
        return y, self.lam * float(np.sum(np.abs(y.astype(np.float64, copy=False))))


@dataclass
class IndBox:
    """ProximalOperators.IndBox(lb, ub): prox = clamp.  lb/ub scalars or arrays."""
    lb: Any = -math.inf
    ub: Any = math.inf

    def __call__(self, x):
        return 0.0 if (np.all(x >= self.lb) and np.all(x <= self.ub)) else float("inf")

    def prox(self, x, gamma=1.0):
        y = np.where(x < self.lb, self.lb, np.where(x > self.ub, self.ub, x)).astype(x.dtype)
        return y, 0.0


@dataclass
class NormL2:
    """ProximalOperators.NormL2(lambda): f = lambda*||x||_2.
    prox: scale = max(0, 1 - lambda*gamma/||x||); y = scale*x."""
    lam: float = 1.0

    def __call__(self, x):
        return self.lam * jl_norm(x)

    def prox(self, x, gamma=1.0):
        normx = jl_norm(x)
        scale = max(0.0, 1 - jl_div(self.lam * gamma, normx))
        y = (scale * x).astype(x.dtype)
        return y, self.lam * scale * normx


@dataclass
class Translate:
    """ProximalOperators.Translate(f, b): x -> f(x + b).
    prox: z = x + b; y = prox_f(z); y -= b."""
    f: Any
    b: np.ndarray

    def __call__(self, x):
        return self.f(x + self.b)

    def prox(self, x, gamma=1.0):
        z = x + self.b
        y, v = prox(self.f, z, gamma)
        y = y - self.b
        return y, v


@dataclass
class ConvexConjugate:
    """ProximalCore.ConvexConjugate — Moreau identity (recalled ``prox_conjugate!``):
        u = x / gamma ; p = prox_{f, 1/gamma}(u) ; y = x - gamma * p
    """
    f: Any

    def prox(self, x, gamma=1.0):
        u = x / gamma
        p, v = prox(self.f, u, jl_div(1.0, gamma))
        v = jl_dot(x, p) - gamma * jl_dot(p, p) - v
        y = x - gamma * p
        return y.astype(x.dtype), v


def convex_conjugate(h):
    """ProximalCore.convex_conjugate (used at src/AdaProx.jl:325)."""
    # a Counting-wrapped h is not `::Zero` for Julia's dispatch: it takes the generic
    # ConvexConjugate(Counting(h)) route, whose prox! calls the wrapped prox! and therefore
    # advances h.prox_count (that is what counting.jl:25-27 `is_convex(::Counting)` is for)
    if isinstance(h, Zero):
        return IndZero()
    if isinstance(h, IndZero):
        return Zero()
    if isinstance(h, ConvexConjugate):
        return h.f
    return ConvexConjugate(h)


def prox(g, x, gamma=1.0):
    """ProximalCore.prox(g, x, gamma) -> (y, g(y))."""
    return g.prox(x, gamma)


def call_g(g, x):
    if isinstance(g, Counting):
        return g.f(x)
    return g(x)


# ----------------------------------------------------------------------------
# Smooth terms defined by the experiments
# ----------------------------------------------------------------------------

def storage_rounding(storage):
    """The product keeps its vectors in the problem's element type T and computes element arithmetic in Float64
    (csrc/vecops_dev.h): every STORED vector is rounded to T once.  ``storage=np.float32`` makes the oracle round at the same
    places — `a -> Float64(Float32(a))` — so that a Float32 run of the product is compared with the arithmetic it implements
    instead of with a Float64 run of Float32-representable data.  The reference itself never runs Float32
    (SURVEY 8a F2): this mode has no counterpart there and is identity for the default ``storage=None``."""
    if storage is None:
        return lambda a: a
    return lambda a: np.asarray(a).astype(storage).astype(np.float64)


@dataclass
class LinearLeastSquares:
    """experiments/lasso/runme.jl:16-27.  (``emulate=np.float32``: `A x` and the residual are stored Float32 vectors, sums in
    Float64 — csrc/team_rows.h team_row_finish.)"""
    A: np.ndarray
    b: np.ndarray
    emulate: Any = None

    def eval_with_pullback(self, w):
        if self.emulate is not None:
            st = storage_rounding(self.emulate)
            res = st(st(self.A @ w) - self.b)
            return 0.5 * jl_norm(res) ** 2, (lambda: self.A.T @ res)
        res = self.A @ w - self.b                       # :22
        pb = lambda: self.A.T @ res                     # :23
        return 0.5 * jl_norm(res) ** 2, pb              # :24 (nrm2 then square)

    def __call__(self, w):                              # :27
        return self.eval_with_pullback(w)[0]


@dataclass
class LogisticLoss:
    """experiments/sparse_logreg/runme.jl:18-39.  Bias is the LAST coordinate
    (and is l1-penalised by g); ``exp(-logits)`` is unguarded (Appendix A.11)."""
    X: np.ndarray
    y: np.ndarray
    emulate: Any = None

    def _eval_emulated(self, w):
        """Float32 storage as the product implements it (csrc/team_rows.h team_row_finish, csrc/slab.hip slab_group): `X w` summed
        in Float64 and stored Float32, the logit and the link in Float32 (`expf` / `logf`, as Julia does on Float32 arrays), the
        loss and the gradient summed in Float64."""
        T = np.dtype(self.emulate).type
        X, w = np.asarray(self.X, dtype=np.float64), np.asarray(w, dtype=np.float64)
        ax = (X @ w[:-1]).astype(T)
        with np.errstate(over="ignore"):
            lf = (ax.astype(np.float64) + w[-1]).astype(T)
            ef = T(1) + np.exp(-lf)
            yT = np.asarray(self.y).astype(T)
            loss = ((yT - T(1)) * lf - np.log(ef)).astype(np.float64)
            d = (T(1) / ef - yT).astype(np.float64)
        N = self.y.shape[0]

        def pb():
            return np.concatenate([(X.T @ d) / N, [np.sum(d) / N]])

        return float(-(np.sum(loss) / N)), pb

    def eval_with_pullback(self, w):
        if self.emulate is not None:
            return self._eval_emulated(w)
        dt = np.result_type(self.X.dtype, w.dtype)
        with np.errstate(over="ignore"):
            logits = (self.X @ w[:-1] + w[-1]).astype(np.float64)   # :24
            u = 1 + np.exp(-logits)                                  # :25

        def pb():
            probs = 1 / u                                            # :28
            N = self.y.shape[0]                                      # :29
            grad = np.zeros_like(w)                                  # :30
            d = probs - self.y
            grad[:-1] = (self.X.T @ d.astype(dt)) / N                # :31
            grad[-1] = np.mean(d)                                    # :32
            return grad

        return float(-np.mean((self.y - 1) * logits - np.log(u))), pb   # :36

    def __call__(self, w):
        return self.eval_with_pullback(w)[0]


@dataclass
class Quadratic:
    """experiments/dual_svm/runme.jl:19-28."""
    Q: np.ndarray
    q: np.ndarray

    def eval_with_pullback(self, x):
        temp = self.Q @ x
        return 0.5 * jl_dot(x, temp) + jl_dot(x, self.q), (lambda: temp + self.q)

    def __call__(self, x):
        return self.eval_with_pullback(x)[0]


# ----------------------------------------------------------------------------
# Record sink (mirror of ``@logmsg Record`` — src/AdaProx.jl:7,56,351).
# ``record`` is an optional callable receiving one dict per iteration; like the
# lazy ``@logmsg`` the objective terms are only evaluated when a sink is given
# (Appendix A.9).
# ----------------------------------------------------------------------------

Record = Optional[Callable[[dict], None]]


# ----------------------------------------------------------------------------
# Backtracking proximal gradient  (src/AdaProx.jl:34-64)
# ----------------------------------------------------------------------------

def backtrack_stepsize(gamma, f, g, x, f_x, grad_x, shrink=0.5, warn=None):
    """src/AdaProx.jl:34-48."""
    z, g_z = prox(g, x - gamma * grad_x, gamma)                  # :35
    ub_z = upper_bound(x, f_x, grad_x, z, gamma)                 # :36
    f_z, pb = eval_with_pullback(f, z)                           # :37
    while f_z > ub_z:                                            # :38
        gamma *= shrink                                          # :39
        if gamma < 1e-12:                                        # :40-42 (logs, continues)
            if warn is not None:
                warn(gamma)
        z, g_z = prox(g, x - gamma * grad_x, gamma)              # :43
        ub_z = upper_bound(x, f_x, grad_x, z, gamma)             # :44
        f_z, pb = eval_with_pullback(f, z)                       # :45
    return gamma, z, f_z, g_z, pb                                # :47


def backtracking_proxgrad(x0, *, f, g, gamma0, xi=1.0, shrink=0.5, tol=1e-5,
                          maxit=100_000, name="Backtracking PG", record: Record = None,
                          trace: Optional[list] = None):
    """src/AdaProx.jl:50-64.  Returns (z, it)."""
    maxit = int(math.floor(maxit))
    x, z, gamma = x0, x0, gamma0                                 # :51
    f_x, grad_x = eval_with_gradient(f, x)                       # :52
    for it in range(1, maxit + 1):                               # :53
        gamma, z, f_z, g_z, pb = backtrack_stepsize(xi * gamma, f, g, x, f_x, grad_x, shrink)  # :54
        norm_res = jl_norm(z - x) / gamma                        # :55
        if record is not None:                                   # :56
            record(dict(method=name, it=it, gamma=gamma, norm_res=norm_res,
                        objective=(f_z + g_z), grad_f_evals=grad_count(f),
                        prox_g_evals=prox_count(g), f_evals=eval_count(f)))
        if trace is not None:
            trace.append(dict(it=it, gamma=gamma, norm_res=norm_res, x=np.array(z, copy=True)))
        if norm_res <= tol:                                      # :57-59
            return z, it
        x, f_x = z, f_z                                          # :60
        grad_x = pb()                                            # :61
    return z, maxit                                              # :63


# ----------------------------------------------------------------------------
# Step-size rules  (src/AdaProx.jl:208-308)
# ----------------------------------------------------------------------------

@dataclass
class FixedStepsize:                                             # :208-211
    gamma: float
    t: float = 1.0


@dataclass
class MalitskyMishchenkoRule:                                    # :217-220
    gamma: float
    t: float = 1.0


class OurRule:                                                   # :232-250
    def __init__(self, *, gamma=0, t=1, norm_A=0, delta=0, Theta=1.2):
        if gamma > 0:
            _gamma = gamma
        elif norm_A > 0:
            _gamma = 1 / (2 * Theta * t * norm_A)
        else:
            raise ValueError("you must provide gamma > 0 if norm_A = 0")   # :246
        self.gamma = float(_gamma)
        self.t = float(t)
        self.norm_A = float(norm_A)
        self.delta = float(delta)
        self.Theta = float(Theta)


class OurRulePlus:                                               # :277-292
    def __init__(self, *, gamma=0, nu=1, xi=1, r=1 / 2):
        if gamma > 0:
            _gamma = gamma
        else:
            raise ValueError("you must provide gamma > 0")       # :288
        self.gamma = float(_gamma)
        self.xi = float(xi)
        self.nu = float(nu)
        self.r = float(r)


def stepsize(rule, state=None, x1=None, grad_x1=None, x0=None, grad_x0=None):
    """``stepsize(rule)`` (init) and ``stepsize(rule, state, x1, g1, x0, g0)`` (update).
    Returns ((gamma, sigma), state)."""
    init = x1 is None
    if isinstance(rule, FixedStepsize):                          # :213-215
        return (rule.gamma, rule.gamma * rule.t ** 2), None

    if isinstance(rule, MalitskyMishchenkoRule):
        if init:                                                 # :222-224
            return (rule.gamma, rule.gamma * rule.t ** 2), (rule.gamma, math.inf)
        gamma_prev, rho = state                                  # :226-230
        L = jl_div(jl_norm(grad_x1 - grad_x0), jl_norm(x1 - x0))
        gamma = jl_min(jl_sqrt_nan(1 + rho) * gamma_prev, jl_div(1.0, 2 * L))
        return (gamma, gamma * rule.t ** 2), (gamma, jl_div(gamma, gamma_prev))

    if isinstance(rule, OurRule):
        if init:                                                 # :252-256
            gamma = rule.gamma
            sigma = rule.gamma * rule.t ** 2
            return (gamma, sigma), (gamma, gamma)
        gamma1, gamma0 = state                                   # :258
        dg = grad_x1 - grad_x0
        dx = x1 - x0
        xi = rule.t ** 2 * gamma1 ** 2 * rule.norm_A ** 2        # :259
        C = nan_to_zero(jl_div(jl_norm(dg) ** 2, jl_dot(dg, dx)))  # :260
        L = nan_to_zero(jl_div(jl_dot(dg, dx), jl_norm(dx) ** 2))  # :261
        D = gamma1 * L * (gamma1 * C - 1)                        # :262
        d1 = 1 + rule.delta
        inner = 1 - 4 * xi * d1 ** 2
        gamma = jl_min(                                          # :263-270
            gamma1 * jl_sqrt_nan(1 + jl_div(gamma1, gamma0)),
            jl_div(1.0, 2 * rule.Theta * rule.t * rule.norm_A),
            jl_div(gamma1 * jl_sqrt_nan(inner),
                   jl_sqrt_nan(2 * d1 * (D + jl_sqrt_nan(D ** 2 + xi * inner)))),
        )
        sigma = gamma * rule.t ** 2                              # :271
        return (gamma, sigma), (gamma, gamma1)                   # :272

    if isinstance(rule, OurRulePlus):
        if init:                                                 # :294-297
            gamma = rule.gamma
            return (gamma, gamma), (gamma, gamma)
        gamma1, gamma0 = state                                   # :299
        dg = grad_x1 - grad_x0
        dx = x1 - x0
        C = nan_to_zero(jl_div(jl_norm(dg) ** 2, jl_dot(dg, dx)))  # :300
        L = nan_to_zero(jl_div(jl_dot(dg, dx), jl_norm(dx) ** 2))  # :301
        D = nan_to_zero(1 - 2 * rule.r + gamma1 * L * (gamma1 * C + 2 * (rule.r - 1)))  # :302
        gamma = gamma1 * jl_min(                                 # :303-306
            jl_sqrt_nan(jl_div(1.0, rule.r * (rule.nu + rule.xi)) + jl_div(gamma1, gamma0)),
            jl_div(jl_sqrt_nan(jl_div(rule.nu * (1 + rule.xi) - 1, rule.nu * (rule.nu + rule.xi))),
                   jl_sqrt_nan(max(D, 0.0))),
        )
        return (gamma, gamma), (gamma, gamma1)                   # :307

    raise TypeError(f"unknown rule {type(rule)}")


def jl_sqrt_nan(v: float) -> float:
    """sqrt that throws DomainError for negative reals (Julia) but passes NaN through."""
    if math.isnan(v):
        return float("nan")
    return jl_sqrt(v)


# ----------------------------------------------------------------------------
# Generic primal-dual loop  (src/AdaProx.jl:312-364)
# ----------------------------------------------------------------------------

def adaptive_primal_dual(x, y, *, f, g, h, A, rule, tol=1e-5, maxit=10_000, name="AdaPDM",
                         record: Record = None, trace: Optional[list] = None, storage=None):
    """src/AdaProx.jl:312-364.  Returns (x, y, it).  (``storage``: see storage_rounding — identity by default.)"""
    maxit = int(math.floor(maxit))
    st = storage_rounding(storage)
    (gamma, sigma), state = stepsize(rule)                       # :324
    h_conj = convex_conjugate(h)                                 # :325

    if storage is not None:
        x, y = st(x), st(y)
    A_x = st(mul(A, x))                                          # :327
    _, grad_x = eval_with_gradient(f, x)                         # :328
    grad_x = st(grad_x)
    At_y = st(amul(A, y))                                        # :329
    v = st(x - gamma * (grad_x + At_y))                          # :330
    x_prev, A_x_prev, grad_x_prev = x, A_x, grad_x               # :331
    x, _ = prox(g, v, gamma)                                     # :332
    x = st(x)

    for it in range(1, maxit + 1):                               # :334
        A_x = st(mul(A, x))                                      # :335
        f_x, grad_x = eval_with_gradient(f, x)                   # :336
        grad_x = st(grad_x)

        primal_res = (v - x) / gamma + grad_x + At_y             # :338

        gamma_prev = gamma                                       # :340
        (gamma, sigma), state = stepsize(rule, state, x, grad_x, x_prev, grad_x_prev)  # :341
        rho = jl_div(gamma, gamma_prev)                          # :342

        w = y + sigma * ((1 + rho) * A_x - rho * A_x_prev)       # :344
        y, _ = prox(h_conj, w, sigma)                            # :345
        y = st(y)

        dual_res = (w - y) / sigma - A_x                         # :347
        norm_res = math.sqrt(jl_norm(primal_res) ** 2 + jl_norm(dual_res) ** 2)  # :348

        if record is not None:                                   # :350-352 (lazy)
            def _emit():
                record(dict(method=name, it=it, gamma=gamma, sigma=sigma, norm_res=norm_res,
                            objective=(f_x + call_g(g, x) + call_g(h, A_x)),
                            grad_f_evals=grad_count(f), prox_g_evals=prox_count(g),
                            prox_h_evals=prox_count(h), A_evals=mul_count(A),
                            At_evals=amul_count(A), f_evals=eval_count(f)))
            without_counting(_emit)
        if trace is not None:
            trace.append(dict(it=it, gamma=gamma, sigma=sigma, norm_res=norm_res, f_x=f_x,
                              x=np.array(x, copy=True), y=np.array(y, copy=True)))

        if norm_res <= tol:                                      # :354-356
            return x, y, it

        At_y = st(amul(A, y))                                    # :358
        v = st(x - gamma * (grad_x + At_y))                      # :359
        x_prev, A_x_prev, grad_x_prev = x, A_x, grad_x           # :360
        x, _ = prox(g, v, gamma)                                 # :361
        x = st(x)
    return x, y, maxit                                           # :363


def condat_vu(x, y, *, f, g, h, A, Lf, gamma=None, sigma=None, norm_A=None, tol=1e-5,
              maxit=10_000, name="Condat-Vu", record: Record = None, trace=None):
    """src/AdaProx.jl:367-416 (default parameter branch :398-412)."""
    if gamma is None and sigma is None:                          # :398
        par = 5.0                                                # :400
        par2 = 100.0                                             # :401
        if norm_A is None:                                       # :402-404
            norm_A = matrix_norm(A)
        if norm_A > par * Lf:                                    # :405-409
            alpha = 1.0
        else:
            alpha = par2 * norm_A / Lf
        gamma = 1.0 / (Lf / 2 + norm_A / alpha)                  # :410
        sigma = 0.99 / (norm_A * alpha)                          # :411
    assert gamma is not None and sigma is not None               # :413
    rule = FixedStepsize(gamma, math.sqrt(sigma / gamma))        # :414
    return adaptive_primal_dual(x, y, f=f, g=g, h=h, A=A, rule=rule, tol=tol, maxit=maxit,
                                name=name, record=record, trace=trace)   # :415


def adaptive_proxgrad(x, *, f, g, rule, tol=1e-5, maxit=100_000, name="AdaPGM",
                      record: Record = None, trace=None, storage=None):
    """src/AdaProx.jl:418-421."""
    x, _, numit = adaptive_primal_dual(x, np.zeros_like(x), f=f, g=g, h=Zero(), A=0, rule=rule,
                                       tol=tol, maxit=maxit, name=name, record=record, trace=trace, storage=storage)
    return x, numit


def fixed_proxgrad(x, *, f, g, gamma, tol=1e-5, maxit=100_000, name="Fixed stepsize PGM",
                   record: Record = None, trace=None):
    """src/AdaProx.jl:457-459."""
    return adaptive_proxgrad(x, f=f, g=g, rule=FixedStepsize(gamma, 1.0), tol=tol, maxit=maxit,
                             name=name, record=record, trace=trace)


# ----------------------------------------------------------------------------
# Linesearch version of adaptive_primal_dual, "AdaPDM+"  (src/AdaProx.jl:463-550)
# what experiments/least_absolute_deviation actually showcases.
# ----------------------------------------------------------------------------

def adaptive_linesearch_primal_dual(x, y, *, f, g, h, A, gamma=None, eta=1.0, t=1.0, delta=1e-8,
                                    Theta=1.2, r=2, R=0.95, tol=1e-5, maxit=10_000, name="AdaPDM+",
                                    record: Record = None, trace: Optional[list] = None,
                                    max_linesearch: int = 10_000):
    """src/AdaProx.jl:463-550.  Returns (x, y, it).  ``max_linesearch`` bounds the inner
    ``while true`` (the reference has no bound and spins forever when ``y_next == y`` makes the
    ratio NaN); hitting it raises."""
    maxit = int(math.floor(maxit))
    assert eta > 0, "eta must be positive"                            # :481
    assert Theta > (delta + 1), "must be Theta > (delta + 1)"         # :482
    if gamma is None:                                                 # :484-486
        gamma = 1 / (2 * Theta * t * eta)
    assert gamma <= 1 / (2 * Theta * t * eta), "gamma is too large"   # :488

    delta1 = 1 + delta                                                # :490
    gamma_prev = gamma                                                # :491
    h_conj = convex_conjugate(h)                                      # :492

    A_x = mul(A, x)                                                   # :494
    _, grad_x = eval_with_gradient(f, x)                              # :495
    At_y = amul(A, y)                                                 # :496
    v = x - gamma * (grad_x + At_y)                                   # :497
    x_prev, A_x_prev, grad_x_prev = x, A_x, grad_x                    # :498
    x, _ = prox(g, v, gamma)                                          # :499

    for it in range(1, maxit + 1):                                    # :501
        A_x = mul(A, x)                                               # :502
        f_x, grad_x = eval_with_gradient(f, x)                        # :503

        primal_res = (v - x) / gamma + grad_x + At_y                  # :505

        dg = grad_x - grad_x_prev
        dx = x - x_prev
        C = nan_to_zero(jl_div(jl_norm(dg) ** 2, jl_dot(dg, dx)))     # :507
        L = nan_to_zero(jl_div(jl_dot(dg, dx), jl_norm(dx) ** 2))     # :508
        Delta = gamma * L * (gamma * C - 1)                           # :509
        xi_bar = t ** 2 * gamma ** 2 * eta ** 2 * delta1 ** 2         # :510
        m4xim1 = 1 - 4 * xi_bar                                       # :511

        eta = R * eta                                                 # :513
        w = y                                                         # :514
        sigma = t ** 2 * gamma                                        # :515
        trials = 0
        while True:                                                   # :516
            gamma_next = jl_min(                                      # :517-521
                gamma * jl_sqrt_nan(1 + jl_div(gamma, gamma_prev)),
                jl_div(1.0, 2 * Theta * t * eta),
                gamma * jl_sqrt_nan(jl_div(m4xim1, 2 * delta1 * (Delta + jl_sqrt_nan(
                    Delta ** 2 + m4xim1 * (t * eta * gamma) ** 2)))),
            )
            rho = jl_div(gamma_next, gamma)                           # :522
            sigma = t ** 2 * gamma_next                               # :523
            w = y + sigma * ((1 + rho) * A_x - rho * A_x_prev)        # :524
            y_next, _ = prox(h_conj, w, sigma)                        # :525
            At_y_next = amul(A, y_next)                               # :526
            if eta >= jl_div(jl_norm(At_y_next - At_y), jl_norm(y_next - y)):   # :527
                gamma, gamma_prev = gamma_next, gamma                 # :528
                y, At_y = y_next, At_y_next                           # :529
                break
            eta *= r                                                  # :532
            trials += 1
            if trials >= max_linesearch:
                raise RuntimeError("AdaPDM+ linesearch did not terminate (the reference would spin forever)")

        dual_res = (w - y) / sigma - A_x                              # :535
        norm_res = math.sqrt(jl_norm(primal_res) ** 2 + jl_norm(dual_res) ** 2)   # :536

        if record is not None:                                        # :538-540
            def _emit():
                record(dict(method=name, it=it, gamma=gamma, sigma=sigma, norm_res=norm_res,
                            objective=(f_x + call_g(g, x) + call_g(h, A_x)),
                            grad_f_evals=grad_count(f), prox_g_evals=prox_count(g),
                            prox_h_evals=prox_count(h), A_evals=mul_count(A),
                            At_evals=amul_count(A), f_evals=eval_count(f)))
            without_counting(_emit)
        if trace is not None:
            trace.append(dict(it=it, gamma=gamma, sigma=sigma, eta=eta, norm_res=norm_res,
                              x=np.array(x, copy=True), y=np.array(y, copy=True)))
        if norm_res <= tol:                                           # :541-543
            return x, y, it

        v = x - gamma * (grad_x + At_y)                               # :545
        x_prev, A_x_prev, grad_x_prev = x, A_x, grad_x                # :546
        x, _ = prox(g, v, gamma)                                      # :547
    return x, y, maxit                                                # :549


# ----------------------------------------------------------------------------
# Comparison baselines: Nesterov variants, aGRAAL, Malitsky-Pock
# ----------------------------------------------------------------------------

def backtracking_nesterov(x0, *, f, g, gamma0, shrink=0.5, tol=1e-5, maxit=100_000,
                          name="Backtracking Nesterov", record: Record = None, trace=None):
    """src/AdaProx.jl:66-84.  Returns (z, it)."""
    maxit = int(math.floor(maxit))
    x, z, gamma = x0, x0, gamma0                                     # :67
    theta = 1.0                                                      # :68
    f_x, grad_x = eval_with_gradient(f, x)                           # :69
    for it in range(1, maxit + 1):                                   # :70
        z_prev = z                                                   # :71
        gamma, z, f_z, g_z, _ = backtrack_stepsize(gamma, f, g, x, f_x, grad_x, shrink)   # :72
        norm_res = jl_norm(z - x) / gamma                            # :73
        if record is not None:                                       # :74
            record(dict(method=name, it=it, gamma=gamma, norm_res=norm_res, objective=(f_z + g_z),
                        grad_f_evals=grad_count(f), prox_g_evals=prox_count(g), f_evals=eval_count(f)))
        if trace is not None:
            trace.append(dict(it=it, gamma=gamma, norm_res=norm_res, x=np.array(z, copy=True)))
        if norm_res <= tol:                                          # :75-77
            return z, it
        theta_prev = theta                                           # :78
        theta = (1 + math.sqrt(1 + 4 * theta_prev ** 2)) / 2         # :79
        x = z + (theta_prev - 1) / theta * (z - z_prev)              # :80
        f_x, grad_x = eval_with_gradient(f, x)                       # :81
    return z, maxit                                                  # :83


def fixed_nesterov(x0, *, f, g, Lf=None, muf=0, mug=0, gamma=None, theta=None, tol=1e-5, maxit=100_000,
                   name="Fixed Nesterov", record: Record = None, trace=None):
    """src/AdaProx.jl:91-142.  Returns (x, it)."""
    maxit = int(math.floor(maxit))
    assert (gamma is None) != (Lf is None)                           # :104
    if gamma is None:                                                # :105-107
        gamma = 1 / Lf
    mu = muf + mug                                                   # :108
    q = gamma * mu / (1 + gamma * mug)                               # :109
    assert q < 1                                                     # :110
    if theta is None:                                                # :111-117
        theta = 1 / math.sqrt(q) if q > 0 else 0
    assert 0 <= theta <= jl_div(1.0, math.sqrt(q))                   # :118
    x, x_prev = x0, x0                                               # :119
    for it in range(1, maxit + 1):                                   # :120
        theta_prev = theta                                           # :121
        if mu == 0:                                                  # :122-128
            theta = (1 + math.sqrt(1 + 4 * theta_prev ** 2)) / 2
            beta = (theta_prev - 1) / theta
        else:
            theta = (1 - q * theta_prev ** 2 + math.sqrt((1 - q * theta_prev ** 2) ** 2 + 4 * theta_prev ** 2)) / 2
            beta = (theta_prev - 1) * (1 + gamma * mug - theta * gamma * mu) / theta / (1 - gamma * muf)
        z = x + beta * (x - x_prev)                                  # :129
        _, grad_z = eval_with_gradient(f, z)                         # :130
        x_prev = x                                                   # :131
        x, g_x = prox(g, z - gamma * grad_z, gamma)                  # :132
        norm_res = jl_norm(x - z) / gamma                            # :133
        if record is not None:                                       # :134-136 (lazy, inside without_counting)
            def _emit():
                record(dict(method=name, it=it, gamma=gamma, norm_res=norm_res, objective=(call_f(f, x) + g_x),
                            grad_f_evals=grad_count(f), prox_g_evals=prox_count(g), f_evals=eval_count(f)))
            without_counting(_emit)
        if trace is not None:
            trace.append(dict(it=it, gamma=gamma, norm_res=norm_res, x=np.array(x, copy=True)))
        if norm_res <= tol:                                          # :137-139
            return x, it
    return x, maxit                                                  # :141


def agraal(x1, *, f, g, x0=None, gamma0=None, gamma_max=1e6, phi=1.5, tol=1e-5, maxit=100_000, name="aGRAAL",
           record: Record = None, trace=None):
    """src/AdaProx.jl:150-192.  ``x0`` must be given to be reproducible (the reference draws
    ``x1 + randn(size(x1))`` when it is ``nothing``, :162-164)."""
    maxit = int(math.floor(maxit))
    if x0 is None:                                                   # :162-164
        x0 = x1 + np.random.default_rng(0).standard_normal(x1.shape)
    x, x_prev, x_bar = x1, x0, x1                                    # :165
    _, grad_x = eval_with_gradient(f, x)                             # :166
    _, grad_x_prev = eval_with_gradient(f, x_prev)                   # :167
    if gamma0 is None:                                               # :168-170
        gamma0 = jl_div(jl_norm(x - x_prev), jl_norm(grad_x - grad_x_prev))
    gamma = gamma0                                                   # :171
    rho = 1 / phi + 1 / phi ** 2                                     # :172
    theta = 1.0                                                      # :173
    for it in range(1, maxit + 1):                                   # :174
        C = jl_div(jl_norm(x - x_prev) ** 2, jl_norm(grad_x - grad_x_prev) ** 2)   # :175
        gamma_prev = gamma                                           # :176
        gamma = jl_min(rho * gamma_prev, jl_div(phi * theta * C, 4 * gamma_prev), gamma_max)   # :177
        theta = phi * gamma / gamma_prev                             # :178
        x_bar = ((phi - 1) * x + x_bar) / phi                        # :179
        x_prev, grad_x_prev = x, grad_x                              # :180
        x, g_x = prox(g, x_bar - gamma * grad_x_prev, gamma)         # :181
        norm_res = jl_norm(x - x_prev) / gamma                       # :182
        if record is not None:                                       # :183-185
            def _emit():
                record(dict(method=name, it=it, gamma=gamma, norm_res=norm_res, objective=(call_f(f, x) + g_x),
                            grad_f_evals=grad_count(f), prox_g_evals=prox_count(g), f_evals=eval_count(f)))
            without_counting(_emit)
        if trace is not None:
            trace.append(dict(it=it, gamma=gamma, norm_res=norm_res, x=np.array(x, copy=True)))
        if norm_res <= tol:                                          # :186-188
            return x, it
        _, grad_x = eval_with_gradient(f, x)                         # :189
    return x, maxit                                                  # :191


def backtrack_stepsize_MP(sigma, sigma_prev, t, x_prev, y, y_prev, grad_x_prev, A_x_prev, At_y, At_y_prev, f, g, A,
                          f_x_prev, warn=None):
    """src/AdaProx.jl:555-579."""
    def attempt(sigma):
        theta = sigma / sigma_prev                                   # :556 / :569
        gamma = t ** 2 * sigma                                       # :557 / :570
        At_ybar = (1 + theta) * At_y - theta * At_y_prev             # :558 / :571
        v = x_prev - gamma * (At_ybar + grad_x_prev)                 # :559 / :572
        x, _ = prox(g, v, gamma)                                     # :560 / :573
        A_x = mul(A, x)                                              # :561 / :574
        f_x, pb = eval_with_pullback(f, x)                           # :562 / :575
        lhs = gamma * sigma * jl_norm(A_x - A_x_prev) ** 2 + 2 * gamma * (
            f_x - f_x_prev - jl_dot(grad_x_prev, x - x_prev))        # :563 / :576
        return gamma, x, v, A_x, f_x, pb, lhs
    gamma, x, v, A_x, f_x, pb, lhs = attempt(sigma)
    while lhs > 0.95 * jl_norm(x - x_prev) ** 2:                     # :564
        sigma /= 2                                                   # :565
        if sigma < 1e-12 and warn is not None:                       # :566-568
            warn(gamma)
        gamma, x, v, A_x, f_x, pb, lhs = attempt(sigma)
    return sigma, gamma, x, v, A_x, f_x, pb                          # :578


def malitsky_pock(x, y, *, f, g, h, A, sigma, t=1.0, tol=1e-5, maxit=10_000, name="MP-ls", record: Record = None,
                  trace=None):
    """src/AdaProx.jl:581-629.  Returns (x, y, it)."""
    maxit = int(math.floor(maxit))
    h_conj = convex_conjugate(h)                                     # :594
    theta = 1.0                                                      # :595
    y_prev = y                                                       # :596
    A_x = mul(A, x)                                                  # :597
    At_y = amul(A, y)                                                # :598
    for it in range(1, maxit + 1):                                   # :599
        At_y_prev = At_y                                             # :600
        w = y + sigma * A_x                                          # :601
        y, _ = prox(h_conj, w, sigma)                                # :602
        At_y = amul(A, y)                                            # :603
        sigma_prev = sigma                                           # :605
        sigma = sigma * math.sqrt(1 + theta)                         # :606
        f_x_prev, grad_x_prev = eval_with_gradient(f, x)             # :608
        x_prev, A_x_prev = x, A_x                                    # :609
        sigma, gamma, x, v, A_x, f_x, pb = backtrack_stepsize_MP(    # :610-611
            sigma, sigma_prev, t, x_prev, y, y_prev, grad_x_prev, A_x_prev, At_y, At_y_prev, f, g, A, f_x_prev)
        grad_x = pb()                                                # :612
        y_prev = y                                                   # :614
        primal_res = (v - x) / gamma + grad_x + At_y                 # :616
        dual_res = (w - y) / sigma_prev - A_x                        # :617
        norm_res = math.sqrt(jl_norm(primal_res) ** 2 + jl_norm(dual_res) ** 2)   # :618
        if record is not None:                                       # :620-622
            def _emit():
                record(dict(method=name, it=it, gamma=gamma, sigma=sigma, norm_res=norm_res,
                            objective=(f_x + call_g(g, x) + call_g(h, A_x)),
                            grad_f_evals=grad_count(f), prox_g_evals=prox_count(g), prox_h_evals=prox_count(h),
                            A_evals=mul_count(A), At_evals=amul_count(A), f_evals=eval_count(f)))
            without_counting(_emit)
        if trace is not None:
            trace.append(dict(it=it, gamma=gamma, sigma=sigma, norm_res=norm_res, x=np.array(x, copy=True),
                              y=np.array(y, copy=True)))
        if norm_res <= tol:                                          # :624-626
            return x, y, it
    return x, y, maxit                                               # :628
