"""One-pass kernels on a given dense shape: ms per iteration of adaptive_proxgrad (least squares and logistic), HIP-event time
of the pass over A.   python tools/shape_bench.py m n [f64|f32]"""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import adaprox_amd as apx  # noqa: E402

apx.load()
m, n = int(sys.argv[1]), int(sys.argv[2])
dtype = np.float32 if (len(sys.argv) > 3 and sys.argv[3] == "f32") else np.float64
ctx = apx.Context(0)
A = apx.DeviceMatrix.uniform(m, n, seed=3, dtype=dtype, ctx=ctx)
A.build_transpose()
rng = np.random.default_rng(0)
out = {}
for kind in ("least_squares", "logistic"):
    if kind == "logistic":
        f = apx.LogisticLoss(A, (rng.uniform(size=m) < 0.5).astype(dtype), ctx=ctx)
        x0, gam, lam = np.zeros(n + 1, dtype=dtype), 4.0 / (n / 3.0 + 1.0), 1e-3
    else:
        f = apx.LinearLeastSquares(A, rng.standard_normal(m).astype(dtype), ctx=ctx)
        x0, gam, lam = np.zeros(n, dtype=dtype), 1.0 / (m * n / 3.0), 1.0
    s = apx.Solver(x0, f=f, g=apx.NormL1(lam), rule=apx.OurRule(gamma=gam), tol=0.0, ctx=ctx, poll_every=1 << 30)
    s.step(5, sync=True)
    ctx.kernel_stats(reset=True)
    ctx.profile(True)
    t0 = time.perf_counter()
    s.step(30, sync=True)
    dt = time.perf_counter() - t0
    ctx.profile(False)
    st = ctx.kernel_stats(reset=True)
    k = {q: dict(launches=v["launches"], avg_ms=v["total_ms"] / v["launches"], GBps=v["bytes"] / v["total_ms"] / 1e6) for q, v in st.items() if v["launches"]}
    out[kind] = dict(ms_per_iter=dt / 30 * 1e3, kernels=k)
    s.close()
print(json.dumps({"shape": [m, n], "dtype": dtype.__name__, **out}))
