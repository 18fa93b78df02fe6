#!/usr/bin/env python3
"""SpMV rates of the compressed-rows storage (csrc/sparse.hip) on LIBSVM-like shapes: rows x cols with a
given average number of entries per row.  Algorithmic bytes per product = nnz * (sizeof(T) + 4) (value + index;
the gathered operand and the row pointers are excluded).  `python tools/sparse_bench.py` on an MI355X."""
import json
import os
import sys
import time

import numpy as np
import scipy.sparse as sp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import adaprox_amd as apx  # noqa: E402


def rand_csr(m, n, per_row, seed, dtype):
    rng = np.random.default_rng(seed)
    nnz_row = np.clip(rng.poisson(per_row, m), 1, n)
    indptr = np.concatenate([[0], np.cumsum(nnz_row)]).astype(np.int64)
    idx = rng.integers(0, n, indptr[-1]).astype(np.int32)
    val = rng.uniform(-1, 1, indptr[-1]).astype(dtype)
    A = sp.csr_matrix((val, idx, indptr), shape=(m, n))
    A.sum_duplicates()
    return A


def planted_labels(A, dtype):
    """y_i = 1[sigmoid(<a_i, w>) > u_i] for a dense N(0, 1) w: coin-flip labels make x = 0 optimal at lam = 0.01 and
    the loop converges (bit for bit) at its first iteration — then every timed iteration is a no-op."""
    rng = np.random.default_rng(3)
    w = rng.standard_normal(A.shape[1])
    return (1.0 / (1.0 + np.exp(-(A @ w))) > rng.uniform(size=A.shape[0])).astype(dtype)


def dataset_shapes():
    """The reference's three LIBSVM datasets (experiments/sparse_logreg/runme.jl:177-204) by shape and fill: with the
    small-size policy a problem on such a matrix runs on its dense copy (csrc/skinny_pg.hip), without it on the SpMV
    kernels."""
    spmv = apx.Context(0)
    spmv.fused_small(False)
    for name, m, n, per_row in (("mushrooms", 8124, 112, 21), ("a5a", 6414, 123, 14), ("phishing", 11055, 68, 30)):
        A = rand_csr(m, n, per_row, 0, np.float64)
        y = planted_labels(A, np.float64)
        gam = 4.0 * m / (A.multiply(A).sum() + m)
        out = {"dataset_shape": name, "shape": [m, n], "nnz": int(A.nnz)}
        for label, ctx in (("dense_copy_us_per_iteration", apx.default_context()), ("spmv_us_per_iteration", spmv)):
            f = apx.LogisticLoss(apx.DeviceMatrix.from_sparse(A, ctx), y, ctx=ctx)
            s = apx.Solver(np.zeros(n + 1), f=f, g=apx.NormL1(1e-4), rule=apx.OurRule(gamma=float(gam)), tol=0.0, ctx=ctx,
                           poll_every=1 << 30)
            s.step(50, sync=True)
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter()
                s.step(300, sync=True)
                best = min(best, (time.perf_counter() - t0) / 300)
            _, _, res = s.result()
            s.close()
            assert res.numit == 950 and not res.converged, (res.numit, res.converged)   # (every timed iteration was a real one)
            out[label] = round(best * 1e6, 1)
        print(json.dumps(out), flush=True)


def main():
    dataset_shapes()
    ctx = apx.default_context()
    for m, n, per_row, dtype in ((49749, 300, 12, np.float64),        # w8a-like
                                 (581012, 54, 12, np.float64),        # covtype-like
                                 (400000, 2000, 40, np.float64),
                                 (1000000, 100000, 60, np.float64),
                                 (1000000, 100000, 60, np.float32)):
        A = rand_csr(m, n, per_row, 0, dtype)
        dA = apx.DeviceMatrix.from_sparse(A, ctx)
        es = np.dtype(dtype).itemsize
        x = np.random.default_rng(1).standard_normal(n).astype(dtype)
        r = np.random.default_rng(2).standard_normal(m).astype(dtype)
        per = {}
        for name, fn in (("A@x", lambda: dA @ x), ("A.T@r", lambda: dA.T @ r)):
            fn()
            t0 = time.perf_counter()
            for _ in range(5):
                fn()
            per[name] = (time.perf_counter() - t0) / 5
        lr_y = planted_labels(A, dtype)
        f = apx.LogisticLoss(dA, lr_y, ctx=ctx)
        gam = 4.0 * m / (A.multiply(A).sum() + m)
        s = apx.Solver(np.zeros(n + 1, dtype=dtype), f=f, g=apx.NormL1(1e-4), rule=apx.OurRule(gamma=float(gam)), tol=0.0,
                       ctx=ctx, poll_every=1 << 30)
        s.step(20, sync=True)
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            s.step(100, sync=True)
            best = min(best, (time.perf_counter() - t0) / 100)
        _, _, res = s.result()
        s.close()
        assert res.numit == 320 and not res.converged, (res.numit, res.converged)       # (every timed iteration was a real one)
        bytes_iter = 2.0 * A.nnz * (es + 4)
        print(json.dumps({"shape": [m, n], "nnz": int(A.nnz), "dtype": np.dtype(dtype).name,
                          "us_per_iteration": round(best * 1e6, 1),
                          "op_call_us_incl_host_copies": {k: round(v * 1e6) for k, v in per.items()},
                          "GBps_algorithmic": round(bytes_iter / best / 1e9, 1)}), flush=True)
        del f
        dA.free()


if __name__ == "__main__":
    main()
