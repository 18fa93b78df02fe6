"""How close does a Float32 run of the product come to the oracle's Float32-storage mode (oracle storage_rounding,
`emulate=np.float32`) and to a Float64 run of the same Float32 data ?  One JSON line per case: the largest relative error of the
iterates and of gamma over the first K iterations, per kernel family.  (The numbers behind the tolerances of
tests/test_gpu_slab.py's Float32 cases — VERDICT r4 #6.)"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))      # (lives in tests/: only tests may import oracle/)
import adaprox_amd as apx  # noqa: E402
from oracle import adaprox_oracle as O  # noqa: E402

CASES = [(8700, 600, False), (9000, 900, True), (8600, 1536, True), (4200, 3072, True), (4099, 3500, False), (9000, 1900, True),
         (4099, 10239, True), (4100, 10240, False), (140000, 200, False), (20000, 2500, True), (9800, 10000, True), (6000, 5500, True),
         (4300, 9216, False), (60000, 130, False), (50000, 601, True), (4101, 11264, True), (4100, 12000, False), (4099, 16001, True),
         (4100, 22529, False), (4099, 28160, True), (4100, 50001, False), (4098, 90112, True)]
K = int(os.environ.get("K", "12"))
for m, n, logistic in CASES:
    rng = np.random.default_rng(m + n)
    A = np.asfortranarray(rng.uniform(-1, 1, (m, n)).astype(np.float32))
    A64 = A.astype(np.float64)
    nx = n + 1 if logistic else n
    if logistic:
        w = rng.standard_normal(n) * (rng.uniform(size=n) < 0.05)
        b = (1 / (1 + np.exp(-(A64 @ w))) > rng.uniform(size=m)).astype(np.float32)
        gam, lam = 4.0 * m / (np.linalg.norm(A64) ** 2 + m), 0.01
        mk = lambda em: O.LogisticLoss(A64, b.astype(np.float64), emulate=em)      # noqa: E731
    else:
        b = rng.standard_normal(m).astype(np.float32)
        gam, lam = 1.0 / (m * n / 3.0), 1.0
        mk = lambda em: O.LinearLeastSquares(A64, b.astype(np.float64), emulate=em)      # noqa: E731
    x0 = (0.01 * rng.standard_normal(nx)).astype(np.float32)
    tr = {}
    for mode in ("emulated", "float64"):
        t = []
        O.adaptive_proxgrad(x0.astype(np.float64), f=mk(np.float32 if mode == "emulated" else None), g=O.NormL1(lam), rule=O.OurRule(gamma=gam),
                            tol=0.0, maxit=K, trace=t, storage=np.float32 if mode == "emulated" else None)
        tr[mode] = t
    ctx = apx.Context(0)
    ctx.one_pass_threshold(16 << 20)
    dA = apx.DeviceMatrix.from_numpy(A, ctx, layout="rows")
    f = apx.LogisticLoss(dA, b, ctx=ctx) if logistic else apx.LinearLeastSquares(dA, b, ctx=ctx)
    out = dict(m=m, n=n, logistic=logistic)
    xs, gs = [], []
    for k in range(1, K + 1):          # the device's iterate after k iterations = the oracle's trace entry k + 1's x ... compare like with like:
        recs = []
        x, _ = apx.adaptive_proxgrad(x0, f=f, g=apx.NormL1(lam), rule=apx.OurRule(gamma=gam), tol=0.0, maxit=k, ctx=ctx, record=recs.append)
        xs.append(x)
        gs.append(recs[-1]["gamma"])
    out["kernel"] = ctx.kernel_name("one_pass")
    for mode in ("emulated", "float64"):
        t = tr[mode]
        # a k-iteration run returns the iterate iteration k + 1 would test (src/AdaProx.jl:361-363): trace[k].x for k < K
        ex = [float(np.max(np.abs(xs[k - 1] - t[k]["x"])) / max(np.max(np.abs(t[k]["x"])), 1e-300)) for k in range(1, K)]
        eg = [abs(gs[k - 1] - t[k - 1]["gamma"]) / t[k - 1]["gamma"] for k in range(1, K + 1)]
        out[mode] = dict(x_first10=max(ex[:10]), gamma_first10=max(eg[:10]), x_all=max(ex), gamma_all=max(eg))
    print(json.dumps(out), flush=True)
    dA.free()
