"""Host-side mirror of the AdaProx.jl interface for the hot path (Python over the C-ABI).

Same names, argument meaning and error behaviour as the reference:

    adaptive_proxgrad(x; f, g, rule, tol, maxit, name)            src/AdaProx.jl:418-421
    fixed_proxgrad(x; f, g, gamma, tol, maxit, name)              src/AdaProx.jl:457-459
    backtracking_proxgrad(x0; f, g, gamma0, xi, shrink, ...)      src/AdaProx.jl:50-64
    adaptive_primal_dual(x, y; f, g, h, A, rule, tol, maxit)      src/AdaProx.jl:312-364
    condat_vu(x, y; f, g, h, A, Lf, norm_A, tol, maxit)           src/AdaProx.jl:367-416
    eval_with_pullback / eval_with_gradient                       src/AdaProx.jl:11-16
    prox(g, v, gamma), convex_conjugate(h)                        ProximalCore
    Counting / without_counting                                   src/counting.jl
    FixedStepsize / MalitskyMishchenkoRule / OurRule / OurRulePlus   src/AdaProx.jl:208-308

``f`` is one of the built-in smooth terms (LinearLeastSquares, LogisticLoss, Zero) —
whose products with A run as the HIP GEMV kernels — or ANY object with an
``eval_with_pullback(x) -> (f_x, pb)`` method (the reference's callable convention);
the latter is bridged through the C-ABI callback slots and the rest of the iteration
still runs on the GPU.  All arithmetic happens in libadaprox_hip.so; nothing here
computes on the CPU.
"""
from __future__ import annotations

import ctypes as C
import math
import os
import sys
from typing import Any, Callable, Optional

import numpy as np

from . import _lib as L

Record = Optional[Callable[[dict], None]]


def _np_dtype(dtype_code: int):
    return np.float32 if dtype_code == L.APX_F32 else np.float64


def _code(dtype) -> int:
    dt = np.dtype(dtype)
    if dt == np.float64:
        return L.APX_F64
    if dt == np.float32:
        return L.APX_F32
    raise TypeError(f"unsupported dtype {dt}: the HIP path stores fp64 or fp32")


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


# --------------------------------------------------------------------------------------
# context
# --------------------------------------------------------------------------------------
class Context:
    """One device + one HIP stream.  ``stream`` may be an existing hipStream_t handle
    (e.g. ``torch.cuda.current_stream().cuda_stream``)."""

    def __init__(self, device: Optional[int] = None, stream: Optional[int] = None):
        self._lib = L.load()
        if device is None:
            device = int(os.environ.get("LOCAL_RANK", "0"))
        h = C.c_void_p()
        L.check(self._lib.apx_ctx_create(int(device), C.c_void_p(stream) if stream else None, C.byref(h)))
        self._h = h
        self.device = int(device)

    def sync(self):
        L.check(self._lib.apx_ctx_sync(self._h))

    @property
    def stream(self) -> int:
        s = C.c_void_p()
        L.check(self._lib.apx_ctx_stream(self._h, C.byref(s)))
        return s.value or 0

    def profile(self, enable: bool):
        L.check(self._lib.apx_ctx_profile(self._h, 1 if enable else 0))

    def kernel_stats(self, reset: bool = False) -> dict:
        arr = (L.KernelStat * L.APX_KSTAT_COUNT)()
        L.check(self._lib.apx_ctx_kernel_stats(self._h, arr, 1 if reset else 0))
        out = {}
        for name, k in (("gemv_n", L.APX_KSTAT_GEMV_N), ("gemv_t", L.APX_KSTAT_GEMV_T), ("one_pass", L.APX_KSTAT_ONE_PASS)):
            out[name] = dict(launches=int(arr[k].launches), total_ms=float(arr[k].total_ms), bytes=float(arr[k].bytes))
        return out

    def device_clock(self) -> tuple:
        """apx_ctx_device_clock: (device wall clock ns, host CLOCK_REALTIME ns, uncertainty ns) for one instant."""
        d, h, u = C.c_int64(), C.c_int64(), C.c_int64()
        L.check(self._lib.apx_ctx_device_clock(self._h, C.byref(d), C.byref(h), C.byref(u)))
        return int(d.value), int(h.value), int(u.value)

    def kernel_name(self, slot: str) -> str:
        """The kernels last booked under a timing slot ("gemv_n", "gemv_t", "one_pass"), as the library launched them."""
        k = {"gemv_n": L.APX_KSTAT_GEMV_N, "gemv_t": L.APX_KSTAT_GEMV_T, "one_pass": L.APX_KSTAT_ONE_PASS}[slot]
        return self._lib.apx_ctx_kernel_name(self._h, k).decode()

    def one_pass(self, mode: int = 1):
        """apx_ctx_one_pass: 0 = the two GEMV passes, 1 = one pass over A per iteration for large dense matrices (default),
        2 = one pass, but never the team kernels (a GPU shared with other work)."""
        L.check(self._lib.apx_ctx_one_pass(self._h, int(mode)))
        if int(mode) != 0:
            self._one_pass_on = int(mode)       # what "on" means for this context (fused_small switches it back on as it was)
        return self

    def set_option(self, key: int, value: int):
        """apx_ctx_set_option (keys: _lib.APX_OPT_*): A/B switches and test hooks — the library reads no environment."""
        L.check(self._lib.apx_ctx_set_option(self._h, int(key), int(value)))
        return self

    def get_option(self, key: int) -> int:
        v = C.c_int64()
        L.check(self._lib.apx_ctx_get_option(self._h, int(key), C.byref(v)))
        return int(v.value)

    def set_gemv_config(self, n_variant: int = -1, t_variant: int = -1, n_blocks: int = 0, t_blocks: int = 0):
        L.check(self._lib.apx_ctx_set_gemv_config(self._h, n_variant, t_variant, n_blocks, t_blocks))

    def one_pass_threshold(self, min_bytes: int):
        """Matrix size from which f and grad f (and the linear operator's half of the primal-dual loop) are formed in one
        pass over A (csrc/slab.hip, csrc/slab_pd.hip); default 192 MB."""
        L.check(self._lib.apx_ctx_one_pass_threshold(self._h, int(min_bytes)))
        return self

    def auto_transpose(self, enable: bool = True):
        """Default on: a one-pass-eligible problem gives its matrix (up to 16 GB) the transposed copy by itself;
        off: only where the caller built it (DeviceMatrix.build_transpose)."""
        L.check(self._lib.apx_ctx_auto_transpose(self._h, 1 if enable else 0))
        return self

    def fused_small(self, enable: bool = True, one_pass: Optional[bool] = None):
        """Both fused execution policies at once — the tests' way to the generic kernels: apx_ctx_fused_small (small dense
        problems as two kernels per iteration, csrc/small.hip and its family) and, unless ``one_pass`` says otherwise,
        apx_ctx_one_pass (large matrices: one pass over A per iteration).  The C-ABI has one switch for each."""
        L.check(self._lib.apx_ctx_fused_small(self._h, 1 if enable else 0))
        on = enable if one_pass is None else one_pass
        # (switching it back ON keeps an earlier one_pass(2) — "never the team kernels" — instead of overriding it: ADVICE r3)
        self.one_pass(getattr(self, "_one_pass_on", 1) if on else 0)

    def exploit_sparsity(self, enable: bool = True):
        """Opt-in: `A*x` streams only the columns with x_j != 0 (dense column-major A)."""
        L.check(self._lib.apx_ctx_exploit_sparsity(self._h, 1 if enable else 0))

    def gemv_variants(self, transposed: bool):
        k = self._lib.apx_gemv_variant_count(1 if transposed else 0)
        return [self._lib.apx_gemv_variant_name(1 if transposed else 0, i).decode() for i in range(k)]

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.apx_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        if sys is None or sys.is_finalizing():      # interpreter teardown: the HIP runtime may already be gone
            return
        try:
            self.close()
        except Exception:
            pass


_default_ctx: Optional[Context] = None


def default_context() -> Context:
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context()
    return _default_ctx


def set_default_context(ctx: Optional[Context]):
    global _default_ctx
    _default_ctx = ctx


# --------------------------------------------------------------------------------------
# dense matrix resident in HBM — plays the role of the Julia `Matrix` A
# --------------------------------------------------------------------------------------
class DeviceMatrix:
    def __init__(self, ctx: Context, handle, m: int, n: int, dtype_code: int):
        self.ctx, self._h, self.shape, self._dtype_code = ctx, handle, (int(m), int(n)), dtype_code
        self._lib = ctx._lib

    # -- constructors
    @staticmethod
    def _layout(layout) -> int:
        """"cols" (column-major in HBM, Julia's layout — the default) or "rows" (ONLY the row-contiguous storage the one-pass
        kernels read: one resident copy instead of two ; include/adaprox_hip.h APX_LAYOUT_*)."""
        try:
            return {"cols": L.APX_LAYOUT_COLS, "rows": L.APX_LAYOUT_ROWS, 0: 0, 1: 1}[layout]
        except KeyError:
            raise ValueError(f"layout must be 'cols' or 'rows', not {layout!r}") from None

    @classmethod
    def from_numpy(cls, A: np.ndarray, ctx: Optional[Context] = None, layout="cols") -> "DeviceMatrix":
        ctx = ctx or default_context()
        A = np.asarray(A)
        if A.ndim != 2:
            raise ValueError("DeviceMatrix.from_numpy: need a 2-D array")
        Af = np.asfortranarray(A)          # column-major, like Julia
        m, n = Af.shape
        h = C.c_void_p()
        L.check(ctx._lib.apx_matrix_upload_layout(ctx._h, _ptr(Af), m, n, max(m, 1), _code(Af.dtype), cls._layout(layout),
                                                  C.byref(h)))
        return cls(ctx, h, m, n, _code(Af.dtype))

    @classmethod
    def from_sparse(cls, A, ctx: Optional[Context] = None) -> "DeviceMatrix":
        """A scipy.sparse matrix (any format) -> compressed storage on the device; plays the role of
        the `SparseMatrixCSC` of experiments/libsvm.jl.  CSC input is passed through as such
        (that is Julia's own layout), anything else as CSR."""
        ctx = ctx or default_context()
        fmt = getattr(A, "format", None)
        if fmt not in ("csr", "csc"):
            A = A.tocsr()
            fmt = "csr"
        A = A.copy()
        A.sum_duplicates()
        A.sort_indices()
        if A.dtype not in (np.float32, np.float64):
            A = A.astype(np.float64)
        m, n = A.shape
        ptr = np.ascontiguousarray(A.indptr, dtype=np.int64)
        idx = np.ascontiguousarray(A.indices, dtype=np.int32)
        val = np.ascontiguousarray(A.data)
        h = C.c_void_p()
        L.check(ctx._lib.apx_matrix_upload_sparse(ctx._h, m, n, int(A.nnz), 0 if fmt == "csr" else 1, _ptr(ptr),
                                                  _ptr(idx), _ptr(val), _code(val.dtype), C.byref(h)))
        M = cls(ctx, h, m, n, _code(val.dtype))
        M.nnz = int(A.nnz)
        return M

    @classmethod
    def zeros(cls, m: int, n: int, dtype=np.float64, ctx: Optional[Context] = None, layout="cols") -> "DeviceMatrix":
        ctx = ctx or default_context()
        h = C.c_void_p()
        L.check(ctx._lib.apx_matrix_create_layout(ctx._h, m, n, _code(dtype), cls._layout(layout), C.byref(h)))
        return cls(ctx, h, m, n, _code(dtype))

    @classmethod
    def uniform(cls, m: int, n: int, seed: int, lo: float = -1.0, hi: float = 1.0, dtype=np.float64,
                row_offset: int = 0, m_total: Optional[int] = None, ctx: Optional[Context] = None, layout="cols"):
        """`rand(m, n) .* 2 .- 1` generated on the device (counter-based; a row shard of a
        global m_total x n matrix can be generated independently on every GPU ; the same values in either layout)."""
        A = cls.zeros(m, n, dtype, ctx, layout)
        L.check(A._lib.apx_matrix_fill_uniform(A._h, seed, lo, hi, row_offset, m_total or m))
        return A

    def drop_primary(self):
        """Keeps ONLY the row-contiguous storage (building it first if it is not there): one resident copy of A."""
        L.check(self._lib.apx_matrix_drop_primary(self._h))
        return self

    @property
    def layout(self) -> str:
        """"cols", "rows" (rows-only), "dual" (both copies resident) or "sparse"."""
        mask = C.c_int32()
        L.check(self._lib.apx_matrix_layout(self._h, C.byref(mask)))
        return {0: "sparse", 1: "cols", 2: "rows", 3: "dual"}[int(mask.value)]

    # -- element access helpers used by the generators
    def scale_columns(self, alpha: np.ndarray):
        alpha = np.ascontiguousarray(alpha, dtype=np.float64)
        if alpha.shape != (self.shape[1],):
            raise ValueError("scale_columns: alpha must have one entry per column")
        L.check(self._lib.apx_matrix_scale_columns(self._h, _ptr(alpha)))

    def fill_column(self, j: int, value: float):
        L.check(self._lib.apx_matrix_fill_column(self._h, j, value))

    def build_transpose(self):
        """Optional dual layout: a transposed copy in HBM so that `A*x` streams contiguously too."""
        L.check(self._lib.apx_matrix_build_transpose(self._h))
        return self

    def drop_transpose(self):
        """Releases the transposed copy (problems on this matrix go back to the two GEMV passes)."""
        L.check(self._lib.apx_matrix_drop_transpose(self._h))
        return self

    def to_numpy(self) -> np.ndarray:
        m, n = self.shape
        out = np.empty((m, n), dtype=self.dtype, order="F")
        L.check(self._lib.apx_matrix_download(self._h, _ptr(out), max(m, 1)))
        return out

    @property
    def dtype(self):
        return np.dtype(_np_dtype(self._dtype_code))

    @property
    def nbytes(self) -> int:
        return self.shape[0] * self.shape[1] * self.dtype.itemsize

    # -- `A*x`, `A'*y`, `norm(A)` (src/counting.jl:65-82)
    def matvec(self, x: np.ndarray) -> np.ndarray:
        x = np.ascontiguousarray(x, dtype=self.dtype)
        if x.shape != (self.shape[1],):
            raise ValueError(f"A*x: x has shape {x.shape}, need ({self.shape[1]},)")
        y = np.empty(self.shape[0], dtype=self.dtype)
        L.check(self._lib.apx_gemv_n(self._h, _ptr(x), _ptr(y)))
        return y

    def rmatvec(self, r: np.ndarray) -> np.ndarray:
        r = np.ascontiguousarray(r, dtype=self.dtype)
        if r.shape != (self.shape[0],):
            raise ValueError(f"A'*y: y has shape {r.shape}, need ({self.shape[0]},)")
        g = np.empty(self.shape[1], dtype=self.dtype)
        L.check(self._lib.apx_gemv_t(self._h, _ptr(r), _ptr(g)))
        return g

    def __matmul__(self, x):
        return self.matvec(x)

    @property
    def T(self):
        return _Adjoint(self)

    def sumsq(self) -> float:
        out = C.c_double()
        L.check(self._lib.apx_matrix_sumsq(self._h, C.byref(out)))
        return out.value

    def norm(self) -> float:
        """`norm(A)` of a matrix = FROBENIUS."""
        return math.sqrt(self.sumsq())

    def opnorm_sq(self, iters: int = 50, seed: int = 7) -> float:
        """lambda_max(A'A) by power iteration with the HIP GEMVs — stands in for
        `opnorm(A)^2` (experiments/lasso/runme.jl:81)."""
        n = self.shape[1]
        v = _hash_uniform(seed, n, -1.0, 1.0).astype(self.dtype)
        v = (v / math.sqrt(float(np.dot(v.astype(np.float64), v.astype(np.float64))))).astype(self.dtype)
        lam = 0.0
        for _ in range(iters):
            w = self.rmatvec(self.matvec(v))
            lam = math.sqrt(float(np.dot(w.astype(np.float64), w.astype(np.float64))))
            v = (w / lam).astype(self.dtype)
        return lam

    def free(self):
        if getattr(self, "_h", None):
            self._lib.apx_matrix_free(self._h)
            self._h = None

    def __del__(self):
        if sys is None or sys.is_finalizing():      # interpreter teardown: the HIP runtime may already be gone
            return
        try:
            self.free()
        except Exception:
            pass


class _Adjoint:
    def __init__(self, A):
        self.A = A

    def __matmul__(self, y):
        return self.A.rmatvec(y)


def _hash_uniform(seed: int, n: int, lo: float, hi: float) -> np.ndarray:
    """Host copy of the library's counter-based RNG (splitmix64 finaliser); O(n) set-up only."""
    gold, m1, m2 = np.uint64(0x9E3779B97F4A7C15), np.uint64(0xBF58476D1CE4E5B9), np.uint64(0x94D049BB133111EB)
    with np.errstate(over="ignore"):
        z = np.uint64(seed) * gold + np.arange(n, dtype=np.uint64)
        z = z + gold
        z = (z ^ (z >> np.uint64(30))) * m1
        z = (z ^ (z >> np.uint64(27))) * m2
        z = z ^ (z >> np.uint64(31))
    return lo + (hi - lo) * ((z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0))


def as_device_matrix(A, ctx: Optional[Context] = None) -> DeviceMatrix:
    if isinstance(A, DeviceMatrix):
        return A
    if isinstance(A, np.ndarray):
        return DeviceMatrix.from_numpy(A, ctx)
    if hasattr(A, "tocsr") and hasattr(A, "nnz"):          # scipy.sparse
        return DeviceMatrix.from_sparse(A, ctx)
    raise TypeError(f"cannot use {type(A)} as a matrix")


# --------------------------------------------------------------------------------------
# counting (src/counting.jl)
# --------------------------------------------------------------------------------------
_counting_enabled = True


def is_counting_enabled() -> bool:
    return _counting_enabled


def without_counting(fn: Callable[[], Any]) -> Any:
    """src/counting.jl:8-14."""
    global _counting_enabled
    _counting_enabled = False
    try:
        return fn()
    finally:
        _counting_enabled = True


class Counting:
    """src/counting.jl:16-34 — wraps f, g or A; the counters advance by what the native
    loop reports (identities of SURVEY §3.4)."""

    def __init__(self, f):
        self.f = f
        self.eval_count = self.grad_count = self.prox_count = self.mul_count = self.amul_count = 0

    def __call__(self, *args):
        return self.f(*args)

    def eval_with_pullback(self, x):
        if is_counting_enabled():
            self.eval_count += 1
        f_x, pb = eval_with_pullback(self.f, x)

        def counting_pullback():
            if is_counting_enabled():
                self.grad_count += 1
            return pb()

        return f_x, counting_pullback

    def prox(self, x, gamma=1.0):
        if is_counting_enabled():
            self.prox_count += 1
        return prox(self.f, x, gamma)

    def norm(self):
        return _unwrap(self).norm()

    @property
    def T(self):
        return _AdjointCounting(self)

    def __matmul__(self, x):
        if is_counting_enabled():
            self.mul_count += 1
        return self.f @ x


class _AdjointCounting:
    def __init__(self, op):
        self.op = op

    def __matmul__(self, y):
        if is_counting_enabled():
            self.op.amul_count += 1
        return self.op.f.T @ y


def _unwrap(o):
    return o.f if isinstance(o, Counting) else o


def _cnt(o, attr):
    return getattr(o, attr) if isinstance(o, Counting) else None


# --------------------------------------------------------------------------------------
# proximable terms
# --------------------------------------------------------------------------------------
class _ProxTerm:
    _kind = L.APX_G_ZERO

    def _desc(self, dtype, keep: list, shift=None) -> L.ProxDesc:
        d = L.ProxDesc()
        d.kind = self._kind
        d.lam = float(getattr(self, "lam", 0.0))
        lb, ub = getattr(self, "lb", -math.inf), getattr(self, "ub", math.inf)
        if self._kind == L.APX_G_BOXV:
            lbv = np.ascontiguousarray(np.broadcast_to(np.asarray(lb, dtype=dtype), self._len), dtype=dtype)
            ubv = np.ascontiguousarray(np.broadcast_to(np.asarray(ub, dtype=dtype), self._len), dtype=dtype)
            keep += [lbv, ubv]
            d.lb_vec, d.ub_vec = lbv.ctypes.data, ubv.ctypes.data
        else:
            d.lb, d.ub = float(lb), float(ub)
        if shift is not None:
            sh = np.ascontiguousarray(shift, dtype=dtype)
            keep.append(sh)
            d.shift = sh.ctypes.data
        return d

    def prox(self, x, gamma=1.0):
        return prox(self, x, gamma)


class Zero(_ProxTerm):
    """ProximalCore.Zero; also a smooth term (experiments/least_absolute_deviation/runme.jl:18-21)."""
    _kind = L.APX_G_ZERO

    def __call__(self, x):
        return 0.0

    def eval_with_pullback(self, x):
        return 0.0, (lambda: np.zeros_like(x))


class IndZero(_ProxTerm):
    _kind = L.APX_G_INDZERO

    def __call__(self, x):
        return prox_value(self, x)


class NormL1(_ProxTerm):
    """ProximalOperators.NormL1(lambda)."""
    _kind = L.APX_G_L1

    def __init__(self, lam: float = 1.0):
        self.lam = float(lam)

    def __call__(self, x):
        return prox_value(self, x)


class NormL2(_ProxTerm):
    """ProximalOperators.NormL2(lambda)."""
    _kind = L.APX_G_L2

    def __init__(self, lam: float = 1.0):
        self.lam = float(lam)

    def __call__(self, x):
        return prox_value(self, x)


class IndBox(_ProxTerm):
    """ProximalOperators.IndBox(lb, ub); bounds may be scalars or arrays."""

    def __init__(self, lb=-math.inf, ub=math.inf):
        self.lb, self.ub = lb, ub
        self._vec = np.ndim(lb) > 0 or np.ndim(ub) > 0
        self._kind = L.APX_G_BOXV if self._vec else L.APX_G_BOX
        self._len = max(np.size(lb), np.size(ub)) if self._vec else 0

    def __call__(self, x):
        return prox_value(self, x)


class Translate(_ProxTerm):
    """ProximalOperators.Translate(f, b): x -> f(x + b)."""

    def __init__(self, f: _ProxTerm, b):
        if isinstance(f, Translate):
            raise TypeError("nested Translate is not supported")
        self.f, self.b = f, np.asarray(b)
        self._kind = f._kind

    def _desc(self, dtype, keep, shift=None):
        return self.f._desc(dtype, keep, shift=self.b)

    def __call__(self, x):
        return prox_value(self, x)


class ConvexConjugate:
    def __init__(self, f):
        self.f = f

    def prox(self, x, gamma=1.0):
        return prox(self, x, gamma)


def convex_conjugate(h):
    """ProximalCore.convex_conjugate (src/AdaProx.jl:325)."""
    h = _unwrap(h)
    if isinstance(h, Zero):
        return IndZero()
    if isinstance(h, IndZero):
        return Zero()
    if isinstance(h, ConvexConjugate):
        return h.f
    return ConvexConjugate(h)


def prox(g, x, gamma=1.0, ctx: Optional[Context] = None):
    """`y, g_y = prox(g, x, gamma)` (ProximalCore.prox) evaluated by the HIP prox kernel."""
    if isinstance(g, Counting):
        return g.prox(x, gamma)
    if not isinstance(g, (_ProxTerm, ConvexConjugate)):
        if hasattr(g, "prox"):
            return g.prox(x, gamma)      # user-defined term with its own prox
        raise TypeError(f"prox not defined for {type(g)}")
    ctx = ctx or default_context()
    conj = isinstance(g, ConvexConjugate)
    term = g.f if conj else g
    x = np.ascontiguousarray(x)
    if x.dtype not in (np.float32, np.float64):
        x = x.astype(np.float64)
    keep: list = []
    desc = term._desc(x.dtype, keep)
    y = np.empty_like(x)
    val = C.c_double()
    L.check(ctx._lib.apx_prox(ctx._h, C.byref(desc), 1 if conj else 0, _code(x.dtype), _ptr(x), x.size,
                              float(gamma), _ptr(y), C.byref(val)))
    return y, val.value


def prox_value(g, x, ctx: Optional[Context] = None):
    """`g(x)` — value of a proximable term (the logged objective, src/AdaProx.jl:351)."""
    ctx = ctx or default_context()
    x = np.ascontiguousarray(x)
    if x.dtype not in (np.float32, np.float64):
        x = x.astype(np.float64)
    if isinstance(g, IndBox) and g._vec:
        g._len = x.size
    keep: list = []
    desc = g._desc(x.dtype, keep)
    val = C.c_double()
    L.check(ctx._lib.apx_prox_value(ctx._h, C.byref(desc), _code(x.dtype), _ptr(x), x.size, C.byref(val)))
    return val.value


# --------------------------------------------------------------------------------------
# smooth terms
# --------------------------------------------------------------------------------------
class LinearLeastSquares:
    """experiments/lasso/runme.jl:16-27.  A: DeviceMatrix (or numpy, uploaded once)."""

    def __init__(self, A, b, ctx: Optional[Context] = None, allreduce=None):
        self.A = as_device_matrix(A, ctx)
        self.b = np.ascontiguousarray(b, dtype=self.A.dtype)
        if self.b.shape != (self.A.shape[0],):
            raise ValueError("LinearLeastSquares: b must have one entry per row of A")
        self.allreduce = allreduce
        self._op: Optional[_Problem] = None

    @property
    def n(self):
        return self.A.shape[1]

    def eval_with_pullback(self, w):
        return _builtin_eval(self, w)

    def __call__(self, w):
        return self.eval_with_pullback(w)[0]


class LogisticLoss:
    """experiments/sparse_logreg/runme.jl:18-39 — bias is the LAST coordinate of w."""

    def __init__(self, X, y, ctx: Optional[Context] = None, allreduce=None, n_total_rows: int = 0):
        self.A = as_device_matrix(X, ctx)
        self.b = np.ascontiguousarray(y, dtype=self.A.dtype)
        if self.b.shape != (self.A.shape[0],):
            raise ValueError("LogisticLoss: y must have one entry per row of X")
        self.allreduce = allreduce
        self.n_total_rows = n_total_rows
        self._op: Optional[_Problem] = None

    @property
    def X(self):
        return self.A

    @property
    def n(self):
        return self.A.shape[1] + 1

    def eval_with_pullback(self, w):
        return _builtin_eval(self, w)

    def __call__(self, w):
        return self.eval_with_pullback(w)[0]


class Quadratic:
    """experiments/dual_svm/runme.jl:19-28: f(x) = 0.5*x'Qx + q'x, pullback `temp + q` with
    `temp = Q*x` kept from the evaluation (one pass over Q per eval+grad)."""

    def __init__(self, Q, q, ctx: Optional[Context] = None):
        self.A = as_device_matrix(Q, ctx)
        if self.A.shape[0] != self.A.shape[1]:
            raise ValueError("Quadratic: Q must be square")
        self.b = np.ascontiguousarray(q, dtype=self.A.dtype)
        if self.b.shape != (self.A.shape[0],):
            raise ValueError("Quadratic: q must have one entry per row of Q")
        self.allreduce = None
        self._op: Optional[_Problem] = None

    @property
    def n(self):
        return self.A.shape[1]

    def eval_with_pullback(self, x):
        return _builtin_eval(self, x)

    def __call__(self, x):
        return self.eval_with_pullback(x)[0]


def _builtin_eval(f, w):
    """`eval_with_pullback` of a built-in f at op level: GEMV-N now, GEMV-T when pb() is called."""
    w = np.ascontiguousarray(w, dtype=f.A.dtype)
    if f._op is None:
        f._op = _Problem(f, Zero(), Zero(), None, f.A.dtype, f.A.ctx).finalize(f.n)
    op = f._op
    fx = C.c_double()
    L.check(op.lib.apx_eval(op.h, _ptr(w), C.byref(fx)))

    def pb():
        g = np.empty(f.n, dtype=f.A.dtype)
        L.check(op.lib.apx_pullback(op.h, _ptr(g)))
        return g

    return fx.value, pb


def eval_with_pullback(f, x):
    """src/AdaProx.jl:11."""
    if not hasattr(f, "eval_with_pullback"):
        raise TypeError(f"eval_with_pullback not defined for type {type(f)}")
    return f.eval_with_pullback(x)


def eval_with_gradient(f, x):
    """src/AdaProx.jl:13-16."""
    f_x, pb = eval_with_pullback(f, x)
    return f_x, pb()


# --------------------------------------------------------------------------------------
# step-size rules (src/AdaProx.jl:208-308)
# --------------------------------------------------------------------------------------
class FixedStepsize:
    def __init__(self, gamma, t=1.0):
        self.gamma, self.t = float(gamma), float(t)

    def _rule(self):
        r = L.Rule()
        r.kind, r.gamma, r.t = L.APX_RULE_FIXED, self.gamma, self.t
        return r


class MalitskyMishchenkoRule:
    def __init__(self, gamma, t=1.0):
        self.gamma, self.t = float(gamma), float(t)

    def _rule(self):
        r = L.Rule()
        r.kind, r.gamma, r.t = L.APX_RULE_MM, self.gamma, self.t
        return r


class OurRule:
    def __init__(self, *, gamma=0, t=1, norm_A=0, delta=0, Theta=1.2):
        if gamma > 0:
            _gamma = gamma
        elif norm_A > 0:
            _gamma = 1 / (2 * Theta * t * norm_A)
        else:
            raise ValueError("you must provide gamma > 0 if norm_A = 0")     # src/AdaProx.jl:246
        self.gamma, self.t, self.norm_A, self.delta, self.Theta = map(float, (_gamma, t, norm_A, delta, Theta))

    def _rule(self):
        r = L.Rule()
        r.kind = L.APX_RULE_OURS
        r.gamma, r.t, r.norm_A, r.delta, r.Theta = self.gamma, self.t, self.norm_A, self.delta, self.Theta
        return r


class OurRulePlus:
    def __init__(self, *, gamma=0, nu=1, xi=1, r=1 / 2):
        if not gamma > 0:
            raise ValueError("you must provide gamma > 0")                  # src/AdaProx.jl:288
        self.gamma, self.xi, self.nu, self.r = map(float, (gamma, xi, nu, r))

    def _rule(self):
        r = L.Rule()
        r.kind = L.APX_RULE_OURS_PLUS
        r.gamma, r.xi, r.nu, r.r = self.gamma, self.xi, self.nu, self.r
        return r


# --------------------------------------------------------------------------------------
# problem handle
# --------------------------------------------------------------------------------------
class _Problem:
    """Owns an apx_problem and everything its descriptor points to."""

    def __init__(self, f, g, h, A, dtype, ctx: Optional[Context], lin_allreduce=None):
        fu, gu, hu, Au = _unwrap(f), _unwrap(g), _unwrap(h), _unwrap(A)
        self.keep: list = []
        dtype = np.dtype(dtype)
        d = L.ProblemDesc()
        d.dtype = _code(dtype)
        if lin_allreduce is not None:        # rows of A (hence y, A_x) are sharded over ranks
            self._set_collective(d, lin_allreduce)
            d.shard_mask = 2
        self._cb_exc: Optional[BaseException] = None
        if isinstance(fu, (LinearLeastSquares, LogisticLoss, Quadratic)):
            ctx = ctx or fu.A.ctx
            if fu.A.dtype != dtype:
                raise TypeError(f"x is {dtype} but the matrix of f is {fu.A.dtype}")
            d.f_kind = (L.APX_F_LEAST_SQUARES if isinstance(fu, LinearLeastSquares) else
                        L.APX_F_LOGISTIC if isinstance(fu, LogisticLoss) else L.APX_F_QUADRATIC)
            d.A_f = fu.A._h
            d.b = fu.b.ctypes.data
            d.n = fu.n
            d.n_total_rows = int(getattr(fu, "n_total_rows", 0) or 0)
            self.keep += [fu.A, fu.b]
            if fu.allreduce is not None:
                if lin_allreduce is None:
                    self._set_collective(d, fu.allreduce)
                d.shard_mask |= 1            # rows of the matrix of f are sharded (one hook serves both)
        elif isinstance(fu, Zero):
            d.f_kind = L.APX_F_ZERO
            d.n = 0    # filled in below from x / A
        elif hasattr(fu, "eval_with_pullback"):
            d.f_kind = L.APX_F_CALLBACK
            self._user_f = f          # keep the Counting wrapper in the loop: its counters advance
            self._pb = None
            self._eval_c = L.EVAL_FN(self._eval_cb)
            self._pb_c = L.PULLBACK_FN(self._pullback_cb)
            d.eval_cb, d.pullback_cb = self._eval_c, self._pb_c
            d.n = 0
        else:
            raise TypeError(f"eval_with_pullback not defined for type {type(fu)}")
        ctx = ctx or (Au.ctx if isinstance(Au, DeviceMatrix) else default_context())
        self.ctx, self.lib, self.dtype = ctx, ctx._lib, dtype
        self.f_is_callback = d.f_kind == L.APX_F_CALLBACK
        self._d = d
        self._g, self._h_term, self._A = gu, hu, Au
        self.h = None

    def finalize(self, n: int):
        d = self._d
        if d.n == 0:
            d.n = n
        if d.n != n:
            raise ValueError(f"x has length {n} but f expects {d.n}")
        Au = self._A
        if Au is None or (np.isscalar(Au) and Au == 0):
            pass
        else:
            Au = as_device_matrix(Au, self.ctx)
            if Au.dtype != self.dtype:
                raise TypeError("A and x must have the same dtype")
            d.A_lin = Au._h
            d.m = Au.shape[0]
            self.keep.append(Au)
            self._A = Au
        for name, term in (("g", self._g), ("h", self._h_term)):
            if not isinstance(term, _ProxTerm):
                raise TypeError(f"{name}: prox not built into the HIP loop for {type(term)}; "
                                "use NormL1 / IndBox / Zero / IndZero / NormL2 / Translate")
            if isinstance(term, IndBox) and term._vec:
                term._len = n if name == "g" else int(d.m)
            setattr(d, name, term._desc(self.dtype, self.keep))
        h = C.c_void_p()
        L.check(self.lib.apx_problem_create(self.ctx._h, C.byref(d), C.byref(h)))
        self.h = h
        return self

    # callback bridges for a user-defined f (the reference's callable convention)
    def _eval_cb(self, user, xptr, n, fout):
        try:
            x = np.ctypeslib.as_array(C.cast(xptr, C.POINTER(C.c_double if self.dtype == np.float64 else C.c_float)),
                                      shape=(n,)).copy()
            f_x, self._pb = eval_with_pullback(self._user_f, x)
            fout[0] = float(f_x)
            return 0
        except BaseException as e:      # never unwind through the C frames
            self._cb_exc = e
            return 1

    def _pullback_cb(self, user, gptr, n):
        try:
            g = np.asarray(self._pb(), dtype=self.dtype)
            out = np.ctypeslib.as_array(C.cast(gptr, C.POINTER(C.c_double if self.dtype == np.float64 else C.c_float)),
                                        shape=(n,))
            out[:] = g
            return 0
        except BaseException as e:
            self._cb_exc = e
            return 1

    def _set_collective(self, d, how):
        """``how``: a sharded.Comm (RCCL inside the library: the loop enqueues ncclAllReduce itself) or a
        callable ``(buf, count, dtype, stream)`` — the callback transport of the CPU / shared-GPU tests."""
        from .sharded import Comm
        if isinstance(how, Comm):
            d.comm = how._h
            self.keep.append(how)
        else:
            self._ar = L.ALLREDUCE_FN(self._wrap_allreduce(how))
            d.allreduce = self._ar

    def _wrap_allreduce(self, fn):
        def cb(user, buf, count, dtype, stream):
            try:
                fn(buf, count, dtype, stream)
                return 0
            except BaseException as e:
                self._cb_exc = e
                return 1
        return cb

    def check(self, code):
        if code == L.APX_ERR_CALLBACK and self._cb_exc is not None:
            e, self._cb_exc = self._cb_exc, None
            raise e
        L.check(code)

    def close(self):
        if self.h:
            self.lib.apx_problem_destroy(self.h)
            self.h = None

    def __del__(self):
        if sys is None or sys.is_finalizing():      # interpreter teardown: the HIP runtime may already be gone
            return
        try:
            self.close()
        except Exception:
            pass


def _as_vec(x, dtype=None) -> np.ndarray:
    x = np.ascontiguousarray(x)
    if dtype is not None:
        x = x.astype(dtype, copy=False)
    elif x.dtype not in (np.float32, np.float64):
        x = x.astype(np.float64)
    if x.ndim != 1:
        raise ValueError("x must be a vector")
    return x


def _problem_dtype(x, f, A):
    fu, Au = _unwrap(f), _unwrap(A)
    if isinstance(fu, (LinearLeastSquares, LogisticLoss, Quadratic)):
        return fu.A.dtype
    if isinstance(Au, DeviceMatrix):
        return Au.dtype
    x = np.asarray(x)
    return x.dtype if x.dtype in (np.float32, np.float64) else np.dtype(np.float64)


def _graph_flag(use_graph) -> int:
    """None = the library's choice (eager launches today: replay measured no faster, tools/small_bench.py)."""
    return 2 if use_graph is None else (1 if use_graph else 0)


def _opts(maxit: int, want_records: bool, poll_every: int = 0, use_graph=None):
    o = L.SolveOpts()
    o.use_graph = _graph_flag(use_graph)
    recs = None
    if want_records and maxit > 0:
        recs = (L.Record * int(maxit))()
        o.records = C.cast(recs, C.POINTER(L.Record))
        o.record_cap = int(maxit)
    o.poll_every = poll_every
    return o, recs


def _nan_none(v):
    return None if (isinstance(v, float) and math.isnan(v)) else v


# --------------------------------------------------------------------------------------
# solvers
# --------------------------------------------------------------------------------------
def adaptive_primal_dual(x, y, *, f, g, h, A, rule, tol=1e-5, maxit=10_000, name="AdaPDM",
                         record: Record = None, ctx: Optional[Context] = None, poll_every: int = 0,
                         use_graph: Optional[bool] = None, A_allreduce=None):
    """src/AdaProx.jl:312-364.  Returns (x, y, it).  ``use_graph`` replays the iteration as a
    captured hipGraph (same kernels, same results; pays off for launch-bound small problems)."""
    maxit = int(math.floor(maxit))                     # maxit may be a Float64 in the experiments
    dtype = _problem_dtype(x, f, A)
    x = _as_vec(x, dtype)
    no_lin = A is None or (np.isscalar(_unwrap(A)) and _unwrap(A) == 0)
    prob = _Problem(f, g, h, None if no_lin else A, dtype, ctx, lin_allreduce=A_allreduce).finalize(x.size)
    try:
        yv = None if no_lin else _as_vec(y, dtype)
        if not no_lin and yv.size != prob._d.m:
            raise ValueError(f"y has length {yv.size} but A has {prob._d.m} rows")
        opts, recs = _opts(maxit, record is not None, poll_every, use_graph)
        res = L.Result()
        x_out = np.empty_like(x)
        y_out = None if no_lin else np.empty_like(yv)
        r = rule._rule()
        base = _count_bases(f, g, h, A)
        prob.check(prob.lib.apx_adaptive_primal_dual(prob.h, C.byref(r), _ptr(x), _ptr(yv), float(tol), maxit,
                                                     C.byref(opts), _ptr(x_out), _ptr(y_out), C.byref(res)))
        _advance_counts(f, g, h, A, res, prob.f_is_callback)
        if record is not None:
            for i in range(int(res.n_records)):
                rr = recs[i]
                k = int(rr.it)
                record(dict(method=name, it=k, gamma=rr.gamma, sigma=rr.sigma, norm_res=rr.norm_res,
                            objective=rr.objective,
                            grad_f_evals=_at(base, "grad", k + 1), prox_g_evals=_at(base, "prox_g", k),
                            prox_h_evals=_at(base, "prox_h", k), A_evals=_at(base, "mul", k + 1),
                            At_evals=_at(base, "amul", k), f_evals=_at(base, "eval", k + 1)))
        if no_lin:
            return x_out, np.zeros_like(x_out), int(res.numit)
        return x_out, y_out, int(res.numit)
    finally:
        prob.close()


def _count_bases(f, g, h, A):
    return dict(eval=_cnt(f, "eval_count"), grad=_cnt(f, "grad_count"), prox_g=_cnt(g, "prox_count"),
                prox_h=_cnt(h, "prox_count"), mul=_cnt(A, "mul_count"), amul=_cnt(A, "amul_count"))


def _at(base, key, inc):
    return None if base[key] is None else base[key] + inc


def last_incident() -> str:
    """apx_last_incident(): what the last team-kernel failure on this thread looked like ("" if none)."""
    return L.load().apx_last_incident().decode("utf-8", "replace")


def _warn_team_fallback(res):
    if int(res.flags) & L.APX_FLAG_TEAM_FALLBACK:
        import warnings
        warnings.warn("adaprox_hip: a hand-off of the one-pass team kernels failed; the call was run again on the two GEMV "
                      "passes and its result is valid (APX_FLAG_TEAM_FALLBACK). " + last_incident(), RuntimeWarning, stacklevel=3)


def _advance_counts(f, g, h, A, res, f_is_callback):
    _warn_team_fallback(res)          # (every one-shot wrapper comes through here with its apx_result)
    # a callback f went through Counting.eval_with_pullback itself
    if isinstance(f, Counting) and not f_is_callback:
        f.eval_count += int(res.f_evals)
        f.grad_count += int(res.grad_f_evals)
    if isinstance(g, Counting):
        g.prox_count += int(res.prox_g_evals)
    if isinstance(h, Counting) and res.prox_h_evals >= 0:
        h.prox_count += int(res.prox_h_evals)
    if isinstance(A, Counting) and res.A_evals >= 0:
        A.mul_count += int(res.A_evals)
        A.amul_count += int(res.At_evals)


def adaptive_proxgrad(x, *, f, g, rule, tol=1e-5, maxit=100_000, name="AdaPGM", record: Record = None,
                      ctx: Optional[Context] = None, poll_every: int = 0, use_graph: Optional[bool] = None):
    """src/AdaProx.jl:418-421.  Returns (x, numit)."""
    x, _, numit = adaptive_primal_dual(x, None, f=f, g=g, h=Zero(), A=0, rule=rule, tol=tol, maxit=maxit,
                                       name=name, record=record, ctx=ctx, poll_every=poll_every,
                                       use_graph=use_graph)
    return x, numit


def fixed_proxgrad(x, *, f, g, gamma, tol=1e-5, maxit=100_000, name="Fixed stepsize PGM", record: Record = None,
                   ctx: Optional[Context] = None):
    """src/AdaProx.jl:457-459."""
    return adaptive_proxgrad(x, f=f, g=g, rule=FixedStepsize(gamma, 1.0), tol=tol, maxit=maxit, name=name,
                             record=record, ctx=ctx)


def condat_vu(x, y, *, f, g, h, A, Lf, gamma=None, sigma=None, norm_A=None, tol=1e-5, maxit=10_000,
              name="Condat-Vu", record: Record = None, ctx: Optional[Context] = None, A_allreduce=None,
              poll_every: int = 0, use_graph: Optional[bool] = None):
    """src/AdaProx.jl:367-416.  ``A_allreduce``: the rows of A (and y) are sharded over ranks; then
    ``norm_A`` must be given (the Frobenius norm of the WHOLE matrix)."""
    if A_allreduce is not None and gamma is None and sigma is None and norm_A is None:
        raise ValueError("condat_vu with a row-sharded A needs norm_A of the whole matrix")
    if gamma is None and sigma is None:                             # :398
        par, par2 = 5.0, 100.0
        if norm_A is None:                                          # :402-404
            norm_A = as_device_matrix(_unwrap(A), ctx).norm()
        alpha = 1.0 if norm_A > par * Lf else par2 * norm_A / Lf    # :405-409
        gamma = 1.0 / (Lf / 2 + norm_A / alpha)                     # :410
        sigma = 0.99 / (norm_A * alpha)                             # :411
    assert gamma is not None and sigma is not None                  # :413
    rule = FixedStepsize(gamma, math.sqrt(sigma / gamma))           # :414
    return adaptive_primal_dual(x, y, f=f, g=g, h=h, A=A, rule=rule, tol=tol, maxit=maxit, name=name,
                                record=record, ctx=ctx, A_allreduce=A_allreduce, poll_every=poll_every,
                                use_graph=use_graph)


def adaptive_linesearch_primal_dual(x, y, *, f, g, h, A, gamma=None, eta=1.0, t=1.0, delta=1e-8, Theta=1.2, r=2,
                                    R=0.95, tol=1e-5, maxit=10_000, name="AdaPDM+", record: Record = None,
                                    ctx: Optional[Context] = None, max_linesearch: int = 0, A_allreduce=None):
    """src/AdaProx.jl:463-550 ("AdaPDM+").  Returns (x, y, it).  (``A_allreduce``: the rows of A, y and h's shift are
    sharded over ranks — A' y_next and the trial's three sums are all-reduced per linesearch trial, so every rank takes the
    same decisions ; a separable h only: the l2 ball on a shard is refused.)"""
    maxit = int(math.floor(maxit))
    # the @assert's of :481-488 (AssertionError in Julia)
    assert eta > 0, "eta must be positive"
    assert Theta > (delta + 1), "must be Theta > (delta + 1)"
    if gamma is not None:
        assert gamma <= 1 / (2 * Theta * t * eta), "gamma is too large"
    dtype = _problem_dtype(x, f, A)
    x = _as_vec(x, dtype)
    prob = _Problem(f, g, h, A, dtype, ctx, lin_allreduce=A_allreduce).finalize(x.size)
    try:
        yv = _as_vec(y, dtype)
        if yv.size != prob._d.m:
            raise ValueError(f"y has length {yv.size} but A has {prob._d.m} rows")
        opts, recs = _opts(maxit, record is not None)
        res = L.Result()
        prm = L.LinesearchParams(float(gamma) if gamma is not None else 0.0, float(eta), float(t), float(delta),
                                 float(Theta), float(r), float(R), int(max_linesearch))
        x_out, y_out = np.empty_like(x), np.empty_like(yv)
        base = _count_bases(f, g, h, A)
        prob.check(prob.lib.apx_adaptive_linesearch_primal_dual(prob.h, C.byref(prm), _ptr(x), _ptr(yv), float(tol),
                                                                maxit, C.byref(opts), _ptr(x_out), _ptr(y_out),
                                                                C.byref(res)))
        _advance_counts(f, g, h, A, res, prob.f_is_callback)
        if record is not None:
            for i in range(int(res.n_records)):
                rr = recs[i]
                k, at = int(rr.it), int(rr.grad_evals)        # at = A'*y products so far (1 + linesearch trials)
                record(dict(method=name, it=k, gamma=rr.gamma, sigma=rr.sigma, norm_res=rr.norm_res,
                            objective=rr.objective,
                            grad_f_evals=_at(base, "grad", k + 1), prox_g_evals=_at(base, "prox_g", k),
                            prox_h_evals=_at(base, "prox_h", at - 1), A_evals=_at(base, "mul", k + 1),
                            At_evals=_at(base, "amul", at), f_evals=_at(base, "eval", k + 1)))
        return x_out, y_out, int(res.numit)
    finally:
        prob.close()


def backtracking_proxgrad(x0, *, f, g, gamma0, xi=1.0, shrink=0.5, tol=1e-5, maxit=100_000,
                          name="Backtracking PG", record: Record = None, ctx: Optional[Context] = None):
    """src/AdaProx.jl:50-64.  Returns (z, it)."""
    maxit = int(math.floor(maxit))
    dtype = _problem_dtype(x0, f, None)
    x0 = _as_vec(x0, dtype)
    prob = _Problem(f, g, Zero(), None, dtype, ctx).finalize(x0.size)
    try:
        opts, recs = _opts(maxit, record is not None)
        res = L.Result()
        z = np.empty_like(x0)
        base = _count_bases(f, g, None, None)
        prob.check(prob.lib.apx_backtracking_proxgrad(prob.h, _ptr(x0), float(gamma0), float(xi), float(shrink),
                                                      float(tol), maxit, C.byref(opts), _ptr(z), C.byref(res)))
        _advance_counts(f, g, None, None, res, prob.f_is_callback)
        if res.flags & L.APX_FLAG_TINY_GAMMA:
            import warnings
            warnings.warn(f"step size became too small ({res.gamma})")       # src/AdaProx.jl:40-42
        if record is not None:
            nprox = 0
            for i in range(int(res.n_records)):
                rr = recs[i]
                # prox calls are 1 + #backtracks per iteration = f_evals - 1 in total
                nprox = int(rr.f_evals) - 1
                record(dict(method=name, it=int(rr.it), gamma=rr.gamma, norm_res=rr.norm_res,
                            objective=rr.objective, grad_f_evals=_at(base, "grad", int(rr.grad_evals)),
                            prox_g_evals=_at(base, "prox_g", nprox), f_evals=_at(base, "eval", int(rr.f_evals))))
        return z, int(res.numit)
    finally:
        prob.close()


# --------------------------------------------------------------------------------------
# stepping interface (bench.py): K iterations with everything resident in HBM
# --------------------------------------------------------------------------------------
class Solver:
    def __init__(self, x, y=None, *, f, g, h=None, A=None, rule, tol=0.0, ctx: Optional[Context] = None,
                 poll_every: int = 0, use_graph: Optional[bool] = None):
        dtype = _problem_dtype(x, f, A)
        x = _as_vec(x, dtype)
        no_lin = A is None or (np.isscalar(_unwrap(A)) and _unwrap(A) == 0)
        self.prob = _Problem(f, g, h if h is not None else Zero(), None if no_lin else A, dtype, ctx).finalize(x.size)
        self.n, self.m, self.dtype = x.size, int(self.prob._d.m), dtype
        self._no_lin, self._rule, self._tol = no_lin, rule, float(tol)
        self._opts = L.SolveOpts()
        self._opts.poll_every = poll_every
        self._opts.use_graph = _graph_flag(use_graph)
        self._h = None
        self._create(x, y)

    def _create(self, x, y):
        yv = None if self._no_lin else _as_vec(y, self.dtype)
        h_ = C.c_void_p()
        r = self._rule._rule()
        self.prob.check(self.prob.lib.apx_solver_create(self.prob.h, C.byref(r), _ptr(x), _ptr(yv), self._tol,
                                                        C.byref(self._opts), C.byref(h_)))
        self._h = h_

    def restart(self, x, y=None):
        """A new native solver on the SAME problem, from (x, y): what a caller does after the stepping interface reported a
        failed team hand-off (APX_ERR_HIP) — the problem has been taken off the team kernels, the new solver runs on the two
        GEMV passes."""
        if self._h:
            self.prob.lib.apx_solver_destroy(self._h)
            self._h = None
        self._create(_as_vec(x, self.dtype), y)

    def step(self, iters: int, sync: bool = False):
        self.prob.check(self.prob.lib.apx_solver_step(self._h, int(iters), 1 if sync else 0))

    def result(self):
        res = L.Result()
        x = np.empty(self.n, dtype=self.dtype)
        y = np.empty(self.m, dtype=self.dtype) if self.m else None
        self.prob.check(self.prob.lib.apx_solver_result(self._h, _ptr(x), _ptr(y), C.byref(res)))
        return x, y, res

    def team_waits(self) -> dict:
        """apx_problem_team_waits: hand-off statistics of this problem's team kernels since it was set up."""
        out = (C.c_int64 * 4)()
        self.prob.check(self.prob.lib.apx_problem_team_waits(self.prob.h, out))
        return dict(waits=int(out[0]), max_polls=int(out[1]), max_wait_ns=int(out[2]), team_launches=int(out[3]))

    def team_wait_slots(self) -> list:
        """apx_problem_team_wait_slots: per workgroup slot of the team launches, where and when it saw its longest hand-off wait
        (launch, row, team / member / XCD, the device's wall clock) — tools/soak.py tells a queue that was not running (many
        slots at one moment) from one late workgroup (one team's partners) by it."""
        buf = (L.TeamWaitSlot * 512)()
        cnt = C.c_int32()
        self.prob.check(self.prob.lib.apx_problem_team_wait_slots(self.prob.h, buf, 512, C.byref(cnt)))
        return [{k: int(getattr(buf[i], k)) for k, _ in L.TeamWaitSlot._fields_ if k != "_pad"} for i in range(min(cnt.value, 512))]

    def close(self):
        if getattr(self, "_h", None):
            self.prob.lib.apx_solver_destroy(self._h)
            self._h = None
        self.prob.close()

    def __del__(self):
        if sys is None or sys.is_finalizing():      # interpreter teardown: the HIP runtime may already be gone
            return
        try:
            self.close()
        except Exception:
            pass
