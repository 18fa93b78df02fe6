"""Row-sharded smooth terms in host form — TEST INFRASTRUCTURE for the world_size-2 gloo
tests (tests/test_sharded_gloo.py).  Same sharding algebra as the native HIP path
(csrc/solver.hip::pullback_f with an all-reduce hook):

    r_k = A_k x - b_k ;  grad = allreduce(A_k' r_k) ;  f = 0.5*sqrt(allreduce(||r_k||^2))^2
"""
import math
from typing import Callable

import numpy as np


class RowShardedLeastSquares:
    """`eval_with_pullback` of LinearLeastSquares (experiments/lasso/runme.jl:16-27) over row
    shards, written against two local operators so that the same host logic runs on HIP
    (`DeviceMatrix.matvec / rmatvec`) and in the world_size-2 gloo tests on CPU."""

    def __init__(self, matvec: Callable, rmatvec: Callable, b_local: np.ndarray,
                 allreduce: Callable[[np.ndarray], np.ndarray]):
        self.matvec, self.rmatvec, self.b, self.allreduce = matvec, rmatvec, np.asarray(b_local), allreduce

    def eval_with_pullback(self, w):
        res = self.matvec(w) - self.b
        ss = float(self.allreduce(np.array([float(np.dot(res, res))]))[0])
        nr = math.sqrt(ss)

        def pb():
            return self.allreduce(self.rmatvec(res))

        return 0.5 * (nr * nr), pb

    def __call__(self, w):
        return self.eval_with_pullback(w)[0]


class RowShardedLogisticLoss:
    """LogisticLoss (experiments/sparse_logreg/runme.jl:18-39) over row shards; N is the
    GLOBAL number of rows."""

    def __init__(self, matvec: Callable, rmatvec: Callable, y_local: np.ndarray, n_total: int,
                 allreduce: Callable[[np.ndarray], np.ndarray]):
        self.matvec, self.rmatvec, self.y, self.N, self.allreduce = matvec, rmatvec, np.asarray(y_local), n_total, allreduce

    def eval_with_pullback(self, w):
        logits = self.matvec(w[:-1]) + w[-1]
        u = 1 + np.exp(-logits)
        loss = float(self.allreduce(np.array([float(np.sum((self.y - 1) * logits - np.log(u)))]))[0])

        def pb():
            d = 1 / u - self.y
            packed = np.concatenate([self.rmatvec(d), [float(np.sum(d))]])
            return self.allreduce(packed) / self.N

        return -loss / self.N, pb

    def __call__(self, w):
        return self.eval_with_pullback(w)[0]
