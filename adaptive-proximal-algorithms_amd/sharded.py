"""Row-sharded multi-GPU path: A = [A_1; ...; A_P], one process per GPU.

    r_k = A_k x - b_k            local
    grad = sum_k A_k' r_k        ONE all-reduce per gradient (n elements + 2 scalars)
    f    = 0.5 * || [r_1; ...; r_P] ||^2 = 0.5 * sqrt(sum_k ||r_k||^2)^2

x, the prox and the step-size state are replicated: every rank runs the same scalar
recurrences on identical all-reduced inputs, so all ranks take identical decisions (the
convergence poll happens at fixed iteration numbers, never on a timer).

There is no reference counterpart (the reference is single-process Julia).  The collective
is torch.distributed — backend "nccl" is RCCL over xGMI on ROCm; CPU tests use "gloo".
"""
from __future__ import annotations

from typing import Callable, Tuple

import numpy as np

from . import _lib as L


def shard_rows(m_total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous row block of rank `rank`: (first row, number of rows).  The first
    m_total % world ranks get one extra row."""
    base, extra = divmod(m_total, world)
    r0 = rank * base + min(rank, extra)
    return r0, base + (1 if rank < extra else 0)


# ---- collectives -----------------------------------------------------------------------
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
