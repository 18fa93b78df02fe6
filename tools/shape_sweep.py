"""Sweep of the row length at ~constant matrix bytes: GB/s of the pass(es) over A per iteration of adaptive_proxgrad, to find
cliffs between the kernel shapes.   python tools/shape_sweep.py [f64|f32] [ls|log] [GB]"""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import adaprox_amd as apx  # noqa: E402

apx.load()
dtype = np.float32 if (len(sys.argv) > 1 and sys.argv[1] == "f32") else np.float64
kind = sys.argv[2] if len(sys.argv) > 2 else "ls"
gb = float(sys.argv[3]) if len(sys.argv) > 3 else 6.0
two_pass = len(sys.argv) > 4 and sys.argv[4] == "two"
only = [int(v) for v in sys.argv[5].split(",")] if len(sys.argv) > 5 and sys.argv[5] != "-" else None
mask = int(sys.argv[6]) if len(sys.argv) > 6 else None       # APX_OPT_MULTI_ROWS_MASK (A/B: 7 = Float32 on the first-generation shapes)
es = np.dtype(dtype).itemsize
NS = [130, 200, 300, 450, 600, 1000, 1500, 2000, 2048, 2100, 2600, 3072, 3100, 3600, 4096, 4200, 5000, 5120, 5200, 6000, 6144, 6200, 7000, 7680, 7700,
      8500, 9216, 9300, 10000, 10752, 10800, 11264, 11300, 12000, 14000, 16000, 20000, 24000, 28160, 28200, 33000, 40000, 50000,
      56320, 56400, 70000, 85000, 100000, 112640, 113000, 130000]
ctx = apx.Context(0)
if two_pass:
    ctx.fused_small(False)
if mask is not None:
    from importlib import import_module
    ctx.set_option(import_module("adaptive-proximal-algorithms_amd._lib").APX_OPT_MULTI_ROWS_MASK, mask)
if only:
    NS = only
rng = np.random.default_rng(0)
for n in NS:
    m = int(gb * 1e9 / es / n)
    if m < 4096:
        continue
    A = apx.DeviceMatrix.uniform(m, n, seed=3, dtype=dtype, ctx=ctx)
    A.build_transpose()
    if kind == "pd":      # least absolute deviation by Condat-Vu (FixedStepsize): A x, the dual step and A' y
        y = rng.standard_normal(m).astype(dtype)
        nA = float(np.sqrt(m * n / 3.0))
        gam, sig = 1.0 / nA, 0.99 / nA
        s = apx.Solver(np.zeros(n, dtype=dtype), np.zeros(m, dtype=dtype), f=apx.Zero(), g=apx.NormL1(10.0), h=apx.Translate(apx.NormL1(1.0), -y), A=A,
                       rule=apx.FixedStepsize(gam, np.sqrt(sig / gam)), tol=0.0, ctx=ctx, poll_every=1 << 30)
    elif kind == "log":
        f = apx.LogisticLoss(A, (rng.uniform(size=m) < 0.5).astype(dtype), ctx=ctx)
        x0, gam, lam = np.zeros(n + 1, dtype=dtype), 4.0 / (n / 3.0 + 1.0), 1e-3
    else:
        f = apx.LinearLeastSquares(A, rng.standard_normal(m).astype(dtype), ctx=ctx)
        x0, gam, lam = np.zeros(n, dtype=dtype), 1.0 / (m * n / 3.0), 1.0
    if kind != "pd":
        s = apx.Solver(x0, f=f, g=apx.NormL1(lam), rule=apx.OurRule(gamma=gam), tol=0.0, ctx=ctx, poll_every=1 << 30)
    s.step(4, sync=True)
    ctx.kernel_stats(reset=True)
    ctx.profile(True)
    t0 = time.perf_counter()
    s.step(20, sync=True)
    dt = time.perf_counter() - t0
    ctx.profile(False)
    st = ctx.kernel_stats(reset=True)
    passes = sum(v["launches"] for v in st.values()) / 20
    ms = sum(v["total_ms"] for v in st.values()) / 20
    if passes == 0:      # (a path that books no pass over A: the skinny / small families)
        passes, ms = 1.0, dt / 20 * 1e3
    print(json.dumps({"n": n, "m": m, "passes_per_iter": passes, "ms_per_iter": round(dt / 20 * 1e3, 4), "kernel_ms_per_iter": round(ms, 4),
                      "GBps_of_bytes_read": round(passes * m * n * es / (ms * 1e-3) / 1e9), "iter_GBps_two_pass_model": round(2 * m * n * es / (dt / 20) / 1e9)}),
          flush=True)
    s.close()
    A.free()
