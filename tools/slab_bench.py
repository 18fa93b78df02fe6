#!/usr/bin/env python3
"""f and grad f in ONE pass over A (csrc/slab.hip) against the two GEMV passes: BASELINE config C3 (logistic
100 000 x 10 000 fp32) and a few least-squares shapes.  `python tools/slab_bench.py` on an MI355X."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import adaprox_amd as apx  # noqa: E402


def run(label, m, n, dtype, logistic, one_pass, iters=60, threshold=None):
    ctx = apx.Context(0)
    if threshold is not None:
        ctx.one_pass_threshold(threshold)
    ctx.fused_small(one_pass)                 # the one-pass kernel belongs to the same execution policy
    A = apx.DeviceMatrix.uniform(m, n, seed=1, dtype=dtype, ctx=ctx)
    A.build_transpose()
    rng = np.random.default_rng(0)
    if logistic:
        y = (rng.uniform(size=m) < 0.5).astype(dtype)
        f = apx.LogisticLoss(A, y, ctx=ctx)
        nx, gam = n + 1, 4.0 / (n / 3.0 + 1.0)
    else:
        f = apx.LinearLeastSquares(A, rng.standard_normal(m).astype(dtype), ctx=ctx)
        nx, gam = n, 1.0 / (m * n / 3.0)
    s = apx.Solver(np.zeros(nx, dtype=dtype), f=f, g=apx.NormL1(1e-4), rule=apx.OurRule(gamma=gam), tol=0.0, ctx=ctx,
                   poll_every=1 << 30)
    s.step(5, sync=True)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        s.step(iters, sync=True)
        best = min(best, (time.perf_counter() - t0) / iters)
    _, _, res = s.result()
    s.close()
    assert res.numit == 5 + 3 * iters and not res.converged
    es = np.dtype(dtype).itemsize
    print(json.dumps({"config": label, "one_pass": one_pass, "ms_per_iteration": round(best * 1e3, 4),
                      "iters_per_sec": round(1 / best, 1), "two_pass_algorithmic_GBps": round(2 * m * n * es / best / 1e9),
                      "objective": res.objective}), flush=True)
    A.free()


def threshold_sweep():
    """`python tools/slab_bench.py threshold`: where one pass starts to pay (apx_ctx_one_pass_threshold's default)."""
    for m, n in ((8300, 1024), (16384, 1024), (24576, 1024), (32768, 1024), (8300, 4096), (16384, 4096), (16384, 8192)):
        for dt in (np.float64, np.float32):
            label = "lasso %dx%d %s (%d MB)" % (m, n, np.dtype(dt).name, m * n * np.dtype(dt).itemsize >> 20)
            for one_pass in (True, False):
                run(label, m, n, dt, False, one_pass, iters=200, threshold=1 << 20)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "threshold":
        return threshold_sweep()
    for label, m, n, dt, lg in (("C3 logistic 100000x10000 fp32", 100_000, 10_000, np.float32, True),
                                ("lasso 200000x5000 fp64", 200_000, 5000, np.float64, False),
                                ("lasso 50000x10000 fp64", 50_000, 10_000, np.float64, False),
                                ("logistic 60000x8000 fp64", 60_000, 8000, np.float64, True),
                                ("lasso 400000x2000 fp32", 400_000, 2000, np.float32, False),
                                ("logistic 50000x4000 fp64", 50_000, 4000, np.float64, True)):
        for one_pass in (True, False):
            run(label, m, n, dt, lg, one_pass)


if __name__ == "__main__":
    main()
