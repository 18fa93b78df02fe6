#!/usr/bin/env python3
"""The linear operator's half of the primal-dual iteration in ONE pass over A (csrc/slab_pd.hip) against the two GEMV
passes, on least-absolute-deviation problems (Condat-Vu step sizes).  `python tools/slab_pd_bench.py` on an MI355X."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import adaprox_amd as apx  # noqa: E402


def run(label, m, n, dtype, one_pass, iters=60):
    ctx = apx.Context(0)
    ctx.fused_small(one_pass)
    A = apx.DeviceMatrix.uniform(m, n, seed=1, dtype=dtype, ctx=ctx)
    A.fill_column(n - 1, 1.0)
    if not one_pass:
        A.build_transpose()                     # (dual layout for the two passes, as in bench_configs.py)
    rng = np.random.default_rng(0)
    yv = rng.laplace(size=m).astype(dtype)
    nA = float(np.sqrt(m * (n - 1) / 3.0 + m))
    gamma, sigma = 1.0 / nA, 0.99 / nA
    s = apx.Solver(np.zeros(n, dtype=dtype), np.zeros(m, dtype=dtype), f=apx.Zero(), g=apx.NormL1(10.0),
                   h=apx.Translate(apx.NormL1(1.0), -yv), A=A, rule=apx.FixedStepsize(gamma, float(np.sqrt(sigma / gamma))),
                   tol=0.0, ctx=ctx, poll_every=1 << 30)
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


def main():
    for label, m, n, dt in (("LAD 100000x10000 fp64", 100_000, 10_000, np.float64), ("LAD 200000x5000 fp64", 200_000, 5000, np.float64),
                            ("LAD 400000x2000 fp32", 400_000, 2000, np.float32), ("LAD 100000x10000 fp32", 100_000, 10_000, np.float32)):
        for one_pass in (True, False):
            run(label, m, n, dt, one_pass)


if __name__ == "__main__":
    main()
