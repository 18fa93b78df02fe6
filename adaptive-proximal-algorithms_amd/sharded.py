"""Row-sharded multi-GPU path: A = [A_1; ...; A_P], one process per GPU.

    r_k = A_k x - b_k            local
    grad = sum_k A_k' r_k        ONE all-reduce per gradient (n elements + 2 scalars)
    f    = 0.5 * || [r_1; ...; r_P] ||^2 = 0.5 * sqrt(sum_k ||r_k||^2)^2

x, the prox and the step-size state are replicated: every rank runs the same scalar
recurrences on identical all-reduced inputs, so all ranks take identical decisions (the
convergence poll happens at fixed iteration numbers, never on a timer).

There is no reference counterpart (the reference is single-process Julia).  The collective
of the product path is RCCL INSIDE the library (`Comm`: apx_comm_* in include/adaprox_hip.h,
csrc/comm.hip — the native loop enqueues ncclAllReduce on its own stream; a launcher such as
torch.distributed.run only starts the ranks and carries the 128-byte id).  The callback
transport below (`device_allreduce` over torch.distributed) is what the CPU tests ("gloo")
and the two-ranks-on-one-GPU rehearsal use, where RCCL cannot run.
"""
from __future__ import annotations

from typing import Callable, Tuple

import numpy as np

import ctypes as C
import sys

from . import _lib as L


def shard_rows(m_total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous row block of rank `rank`: (first row, number of rows).  The first
    m_total % world ranks get one extra row."""
    base, extra = divmod(m_total, world)
    r0 = rank * base + min(rank, extra)
    return r0, base + (1 if rank < extra else 0)


# ---- RCCL inside the library -----------------------------------------------------------
_PREFETCHED = set()


def prefetch_library_image(path: str) -> float:
    """Reads a shared library's file once, sequentially, so that the loader's page faults hit the page cache.  librccl is
    0.3 - 0.6 GB of device code ; on a box whose image is not cached yet the first communicator demand-pages it at ~2 MB/s —
    141 s for PyTorch's copy and more than 300 s for /opt/rocm's on one box of round 5 (profiles/incidents/README.md), against
    seconds for a sequential read.  Returns the seconds it took (0.0 when there is nothing to read)."""
    import os
    import time
    real = os.path.realpath(path)
    if real in _PREFETCHED or not os.path.isfile(real):
        return 0.0
    _PREFETCHED.add(real)
    t0 = time.perf_counter()
    try:
        with open(real, "rb", buffering=0) as fh:
            while fh.read(64 << 20):
                pass
    except OSError:
        pass
    return time.perf_counter() - t0


def rccl_candidates() -> list:
    """The librccl files a process may end up binding (csrc/comm.hip rccl_api: the copy next to PyTorch's HIP runtime where
    torch is installed, else the system's)."""
    import os
    out = []
    if "torch" in sys.modules:
        out.append(os.path.join(os.path.dirname(sys.modules["torch"].__file__), "lib", "librccl.so"))
    for d in (os.environ.get("ROCM_PATH", "/opt/rocm"),):
        out.append(os.path.join(d, "lib", "librccl.so"))
    return [p for p in out if os.path.exists(p)]


class Comm:
    """An RCCL communicator bound to one Context (apx_comm_*).  Pass it where a row-sharded
    term takes its collective: ``LinearLeastSquares(A_k, b_k, allreduce=comm)``,
    ``adaptive_primal_dual(..., A_allreduce=comm)``; ``comm(v)`` sums a host fp64 vector over
    the ranks (set-up code)."""

    def __init__(self, handle, ctx):
        self._h, self.ctx, self._lib = handle, ctx, ctx._lib

    @staticmethod
    def _plumbing():
        """The library binds librccl at its first apx_comm_* call and takes the copy already in the process.  Where
        PyTorch-ROCm is installed its HIP runtime is the one in use (_lib._share_hip_runtime_with_torch), so its RCCL
        must be too: importing torch first lets torch map its own pair in its own order (a librccl mapped before torch
        maps its copy aborts the process at exit)."""
        if "torch" not in sys.modules:
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        for path in rccl_candidates()[:1]:          # the copy this process will bind (see rccl_candidates)
            prefetch_library_image(path)

    @staticmethod
    def library() -> str:
        """Path of the librccl in use ("" before the first communicator call failed to find one)."""
        Comm._plumbing()
        return L.load().apx_comm_library().decode()

    @staticmethod
    def unique_id() -> bytes:
        """ncclGetUniqueId (rank 0); hand the bytes to every rank (store, MPI, file)."""
        Comm._plumbing()
        buf = C.create_string_buffer(L.APX_COMM_ID_BYTES)
        L.check(L.load().apx_comm_unique_id(buf))
        return buf.raw

    @classmethod
    def init_rank(cls, ctx, nranks: int, rank: int, uid: bytes) -> "Comm":
        """One process per GPU: ncclCommInitRank on the context's device."""
        if len(uid) != L.APX_COMM_ID_BYTES:
            raise ValueError("Comm.init_rank: the id must be APX_COMM_ID_BYTES long")
        cls._plumbing()
        h = C.c_void_p()
        L.check(ctx._lib.apx_comm_init_rank(ctx._h, int(nranks), int(rank), C.c_char_p(uid), C.byref(h)))
        return cls(h, ctx)

    @classmethod
    def init_all(cls, ctxs) -> list:
        """One process, one Context per device: ncclCommInitAll."""
        cls._plumbing()
        n = len(ctxs)
        hs = (C.c_void_p * n)(*[c._h for c in ctxs])
        out = (C.c_void_p * n)()
        L.check(ctxs[0]._lib.apx_comm_init_all(hs, n, out))
        return [cls(C.c_void_p(out[i]), ctxs[i]) for i in range(n)]

    def info(self):
        """(nranks, rank) as RCCL reports them (ncclCommCount, ncclCommUserRank)."""
        n, r = C.c_int32(), C.c_int32()
        L.check(self._lib.apx_comm_info(self._h, C.byref(n), C.byref(r)))
        return n.value, r.value

    @property
    def nranks(self) -> int:
        return self.info()[0]

    @property
    def rank(self) -> int:
        return self.info()[1]

    def __call__(self, v: np.ndarray) -> np.ndarray:
        out = np.ascontiguousarray(v, dtype=np.float64).copy()
        if out.size:
            L.check(self._lib.apx_comm_allreduce_host(self._h, out.ctypes.data_as(C.POINTER(C.c_double)), out.size))
        return out.reshape(np.shape(v))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.apx_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        if sys is None or sys.is_finalizing():
            return
        try:
            self.close()
        except Exception:
            pass


# ---- collectives through torch.distributed (test transport) ----------------------------
def host_allreduce(dist, local_rank=None) -> Callable[[np.ndarray], np.ndarray]:
    """Sum of a (small) numpy vector over the ranks — used by the O(n) set-up code."""
    import torch

    def ar(v: np.ndarray) -> np.ndarray:
        dev = "cpu" if dist.get_backend() == "gloo" or local_rank is None else f"cuda:{local_rank}"
        t = torch.as_tensor(np.ascontiguousarray(v, dtype=np.float64)).to(dev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return t.cpu().numpy()

    return ar


class _DevArray:
    """Zero-copy view of library-owned device memory for torch (`__cuda_array_interface__`)."""

    def __init__(self, ptr: int, count: int, typestr: str):
        self.__cuda_array_interface__ = {"shape": (count,), "typestr": typestr, "data": (ptr, False), "version": 2}


def device_allreduce(dist, local_rank: int):
    """The `apx_allreduce_fn` hook: sums `count` elements at a DEVICE pointer over the ranks
    with RCCL, ordered after / before the library's work on `stream`."""
    import torch

    cache = {}

    def ar(buf, count, dtype, stream):
        key = (buf, count, dtype)
        t = cache.get(key)
        if t is None:
            t = torch.as_tensor(_DevArray(int(buf), int(count), "<f4" if dtype == L.APX_F32 else "<f8"),
                                device=f"cuda:{local_rank}")
            cache[key] = t
        ext = torch.cuda.ExternalStream(int(stream), device=f"cuda:{local_rank}") if stream else None
        if ext is not None:
            with torch.cuda.stream(ext):
                dist.all_reduce(t, op=dist.ReduceOp.SUM)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)

    return ar
