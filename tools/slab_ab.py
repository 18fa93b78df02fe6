#!/usr/bin/env python3
"""A/B of the two generations of the long-row one-pass kernel (csrc/slab.hip: k_slab_rows vs k_slab2_rows) in ONE process,
interleaved — runs of the same binary minutes apart differ by +-3 % on this part (clocks), more than the kernels do.
`python tools/slab_ab.py` on an MI355X."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import adaprox_amd as apx  # noqa: E402


def make(ctx, A, y, logistic, dtype, n, m, v1):
    from importlib import import_module
    L = import_module("adaptive-proximal-algorithms_amd._lib")
    ctx.set_option(L.APX_OPT_SLAB_GENERATION, 1 if v1 else 2)      # (read when the problem is set up: its first evaluation)
    if logistic:
        f, nx, gam = apx.LogisticLoss(A, y, ctx=ctx), n + 1, 4.0 / (n / 3.0 + 1.0)
    else:
        f, nx, gam = apx.LinearLeastSquares(A, y, ctx=ctx), n, 1.0 / (m * n / 3.0)
    s = apx.Solver(np.zeros(nx, dtype=dtype), f=f, g=apx.NormL1(1e-4), rule=apx.OurRule(gamma=gam), tol=0.0, ctx=ctx,
                   poll_every=1 << 30)
    s.step(3, sync=True)          # (the variant is chosen at the first evaluation)
    return s


for label, m, n, dtype, logistic in (("C3 logistic 100000x10000 fp32", 100_000, 10_000, np.float32, True),
                                     ("lasso 100000x10000 fp32", 100_000, 10_000, np.float32, False),
                                     ("lasso 50000x10000 fp64", 50_000, 10_000, np.float64, False),
                                     ("lasso 60000x8000 fp64", 60_000, 8_000, np.float64, False),
                                     ("lasso 120000x6000 fp32", 120_000, 6_000, np.float32, False)):
    ctx = apx.Context(0)
    A = apx.DeviceMatrix.uniform(m, n, seed=1, dtype=dtype, ctx=ctx).build_transpose()
    rng = np.random.default_rng(0)
    y = (rng.uniform(size=m) < 0.5).astype(dtype) if logistic else rng.standard_normal(m).astype(dtype)
    solvers = {"gen2": make(ctx, A, y, logistic, dtype, n, m, False), "gen1": make(ctx, A, y, logistic, dtype, n, m, True)}
    t = {k: [] for k in solvers}
    for _ in range(8):
        for k, s in solvers.items():
            t0 = time.perf_counter()
            s.step(40, sync=True)
            t[k].append((time.perf_counter() - t0) / 40 * 1e3)
    print(json.dumps({"config": label, **{k + "_ms_per_iteration": [round(min(v), 4), round(float(np.median(v)), 4)] for k, v in t.items()}}),
          flush=True)
    for s in solvers.values():
        s.close()
    A.free()
