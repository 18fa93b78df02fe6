"""Times the non-headline BASELINE configs at full size on one MI355X (C3 logreg fp32,
C5 LAD via Condat-Vu, C2 backtracking, C1 small lasso).  Same protocol as bench.py: data
resident, warm-up, K timed iterations, HIP-event kernel times from the library.
(C3 runs f and grad f in one pass over A, csrc/slab.hip, booked under the one_pass slot; APX_SINGLE_LAYOUT=1 no longer gives a
single-layout C3: the one-pass kernel builds the transposed copy it reads.)"""
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import adaprox_amd as apx
from importlib import import_module
problems = import_module("adaptive-proximal-algorithms_amd.problems")

ctx = apx.Context(0)
apx.set_default_context(ctx)
out = {}

def timed(solver, steps, warm, bytes_per_iter):
    solver.step(warm, sync=True)
    ctx.kernel_stats(reset=True); ctx.profile(True); ctx.sync()
    t0 = time.perf_counter()
    solver.step(steps, sync=True)
    dt = time.perf_counter() - t0
    ctx.profile(False)
    st = ctx.kernel_stats(reset=True)
    k = {n: dict(avg_ms=v["total_ms"] / v["launches"], GBps=v["bytes"] / v["total_ms"] / 1e6) for n, v in st.items() if v["launches"]}
    x, y, res = solver.result()
    assert np.all(np.isfinite(x))
    r = dict(iters_per_s=steps / dt, ms_per_iter=dt / steps * 1e3, frac_of_8TBs=bytes_per_iter * steps / dt / 8e12, kernels=k,
             objective=res.objective, norm_res=res.norm_res)
    if "one_pass" in k:      # csrc/slab.hip / slab_pd.hip: ONE pass over A per iteration
        r["note"] = ("one pass over A per iteration: frac_of_8TBs is of the TWO-pass algorithmic bytes (SURVEY 8d) and can "
                     "exceed 1; the bytes actually read per iteration are half, at kernels.one_pass.GBps")
        r["frac_of_8TBs_bytes_read"] = k["one_pass"]["GBps"] / 8e3
        r["one_pass_kernel"] = ctx.kernel_name("one_pass")
    return r

which = sys.argv[1:] or ["c3", "c5", "c2bt", "c1"]
if "c3" in which:
    m, n = 100_000, 10_000
    p = problems.device_logreg(m, n, seed=0, lam=0.01, dtype=np.float32, ctx=ctx)
    if os.environ.get("APX_SINGLE_LAYOUT") != "1":
        p.X.build_transpose()
    f = apx.LogisticLoss(p.X, p.y)
    s = apx.Solver(np.zeros(n + 1, dtype=np.float32), f=f, g=apx.NormL1(p.lam), rule=apx.OurRule(gamma=p.gamma0), tol=0.0, ctx=ctx, poll_every=1 << 30)
    out["C3 sparse_logreg dense 100000x10000 fp32, adaptive_proxgrad"] = timed(s, 200, 20, 2 * m * n * 4)
    s.close(); p.X.free()
if "c3f64" in which:      # C3's shape in the reference's own precision (experiments/sparse_logreg/runme.jl runs Float64)
    m, n = 100_000, 10_000
    p = problems.device_logreg(m, n, seed=0, lam=0.01, dtype=np.float64, ctx=ctx)
    p.X.build_transpose()
    f = apx.LogisticLoss(p.X, p.y)
    s = apx.Solver(np.zeros(n + 1), f=f, g=apx.NormL1(p.lam), rule=apx.OurRule(gamma=p.gamma0), tol=0.0, ctx=ctx, poll_every=1 << 30)
    # (solved by iteration 40 and NaN by 80 with tol = 0, DESIGN.md 9: 10 + 40 iterations)
    out["C3 shape in Float64: logistic 100000x10000 fp64, adaptive_proxgrad"] = timed(s, 40, 10, 2 * m * n * 8)
    s.close(); p.X.free()
    # and the least-squares term on the same matrix (the second-generation row kernel)
    A = apx.DeviceMatrix.uniform(m, n, seed=3, ctx=ctx)
    A.build_transpose()
    b = np.random.default_rng(0).standard_normal(m)
    s = apx.Solver(np.zeros(n), f=apx.LinearLeastSquares(A, b), g=apx.NormL1(1.0), rule=apx.OurRule(gamma=1.0 / (m * n / 3.0)), tol=0.0, ctx=ctx, poll_every=1 << 30)
    out["least squares 100000x10000 fp64, adaptive_proxgrad"] = timed(s, 200, 20, 2 * m * n * 8)
    s.close(); A.free()
if "c5" in which:
    m, n = 50_000, 20_000
    p = problems.device_lad(m, n, seed=0, lam=10.0, ctx=ctx)
    norm_A = p.norm_A
    if os.environ.get("APX_SINGLE_LAYOUT") != "1":
        p.A.build_transpose()
    gamma = 1.0 / (0.0 / 2 + norm_A / 1.0); sigma = 0.99 / (norm_A * 1.0)       # condat_vu, Lf = 0 (src/AdaProx.jl:405-411)
    rule = apx.FixedStepsize(gamma, np.sqrt(sigma / gamma))
    s = apx.Solver(np.zeros(n), np.zeros(m), f=apx.Zero(), g=apx.NormL1(p.lam), h=apx.Translate(apx.NormL1(1.0), -p.y), A=p.A, rule=rule, tol=0.0, ctx=ctx, poll_every=1 << 30)
    out["C5 LAD 50000x20000 fp64, condat_vu via adaptive_primal_dual"] = timed(s, 100, 10, 2 * m * n * 8)
    s.close(); p.A.free()
if "sqrtlasso" in which:      # the square-root lasso's l2 ball on C5's matrix (experiments/square_root_lasso/runme.jl:39-41): one pass (round 4) / two passes
    m, n = 50_000, 20_000
    for one_pass in (True, False):
        cx = apx.Context(0)
        cx.one_pass(1 if one_pass else 0)
        p = problems.device_lad(m, n, seed=0, lam=10.0, ctx=cx)
        if one_pass:
            p.A.build_transpose()
        ctx_save, ctx = ctx, cx
        # (lambda = 0.5: with the l2 ball |A' y| stays below 2.5 or so on this matrix — LAD's lambda = 10 would keep x at 0)
        s = apx.Solver(np.zeros(n), np.zeros(m), f=apx.Zero(), g=apx.NormL1(0.5), h=apx.Translate(apx.NormL2(1.0), -p.y), A=p.A,
                       rule=apx.OurRule(t=1.0, norm_A=p.norm_A), tol=0.0, ctx=cx, poll_every=1 << 30)
        out["square-root lasso 50000x20000 fp64, adaptive_primal_dual (OurRule), %s" % ("one pass" if one_pass else "two passes")] = timed(s, 100, 10, 2 * m * n * 8)
        ctx = ctx_save
        s.close(); p.A.free()
if "adapdm" in which:      # AdaPDM+ (adaptive_linesearch_primal_dual) on C5's LAD: the first trial in one pass (round 4) / every product a GEMV
    m, n = 50_000, 20_000
    for one_pass in (True, False):
        cx = apx.Context(0)
        cx.one_pass(1 if one_pass else 0)
        p = problems.device_lad(m, n, seed=0, lam=10.0, ctx=cx)
        if one_pass:
            p.A.build_transpose()
        kw = dict(f=apx.Zero(), g=apx.NormL1(p.lam), h=apx.Translate(apx.NormL1(1.0), -p.y), A=apx.Counting(p.A), eta=p.norm_A, t=1.0, tol=0.0, ctx=cx)
        apx.adaptive_linesearch_primal_dual(np.zeros(n), np.zeros(m), maxit=5, **kw)
        cx.kernel_stats(reset=True); cx.profile(True)
        recs = []
        t0 = time.perf_counter()
        x, y, it = apx.adaptive_linesearch_primal_dual(np.zeros(n), np.zeros(m), maxit=100, record=recs.append, **kw)
        dt = time.perf_counter() - t0
        cx.profile(False)
        st = cx.kernel_stats(reset=True)
        out["AdaPDM+ LAD 50000x20000 fp64, %s" % ("first trial in one pass" if one_pass else "GEMV per product")] = dict(
            iters_per_s=100 / dt, ms_per_iter=dt * 10, trials=recs[-1]["At_evals"] - 1, objective=recs[-1]["objective"],
            kernels={k: dict(launches=v["launches"], avg_ms=v["total_ms"] / v["launches"]) for k, v in st.items() if v["launches"]},
            note="the call includes the upload of x0 / y0, two products in front of the loop and the read-back of the result")
        p.A.free()
if "mp" in which:      # Malitsky-Pock on C5's LAD: every trial in one pass (round 4) / every product a GEMV
    m, n = 50_000, 20_000
    for one_pass in (True, False):
        cx = apx.Context(0)
        cx.one_pass(1 if one_pass else 0)
        p = problems.device_lad(m, n, seed=0, lam=10.0, ctx=cx)
        if one_pass:
            p.A.build_transpose()
        kw = dict(f=apx.Zero(), g=apx.NormL1(p.lam), h=apx.Translate(apx.NormL1(1.0), -p.y), A=apx.Counting(p.A), sigma=1.0 / p.norm_A, t=1.0, tol=0.0, ctx=cx)
        apx.malitsky_pock(np.zeros(n), np.zeros(m), maxit=5, **kw)
        cx.kernel_stats(reset=True); cx.profile(True)
        recs = []
        t0 = time.perf_counter()
        x, y, it = apx.malitsky_pock(np.zeros(n), np.zeros(m), maxit=100, record=recs.append, **kw)
        dt = time.perf_counter() - t0
        cx.profile(False)
        st = cx.kernel_stats(reset=True)
        out["Malitsky-Pock LAD 50000x20000 fp64, %s" % ("trials in one pass" if one_pass else "GEMV per product")] = dict(
            iters_per_s=100 / dt, ms_per_iter=dt * 10, trials=recs[-1]["A_evals"] - 1, objective=recs[-1]["objective"],
            kernels={k: dict(launches=v["launches"], avg_ms=v["total_ms"] / v["launches"]) for k, v in st.items() if v["launches"]})
        p.A.free()
if "c2bt" in which:
    m, n = 50_000, 100_000
    p = problems.device_lasso(m, n, seed=0, power_iters=20, ctx=ctx, dual_layout=os.environ.get("APX_SINGLE_LAYOUT") != "1")
    f = apx.Counting(apx.LinearLeastSquares(p.A, p.b)); g = apx.NormL1(p.lam)
    apx.backtracking_proxgrad(np.zeros(n), f=f, g=g, gamma0=1 / p.Lf, xi=1.5, tol=0.0, maxit=2)
    e0, g0 = f.eval_count, f.grad_count
    t0 = time.perf_counter()
    z, it = apx.backtracking_proxgrad(np.zeros(n), f=f, g=g, gamma0=1 / p.Lf, xi=1.5, tol=0.0, maxit=20)
    dt = time.perf_counter() - t0
    passes = (f.eval_count - e0) + (f.grad_count - g0)
    out["C2 lasso 50000x100000 fp64, backtracking_proxgrad xi=1.5"] = dict(iters_per_s=20 / dt, ms_per_iter=dt / 20 * 1e3, passes_over_A=passes,
        frac_of_8TBs=passes * m * n * 8 / dt / 8e12)
    p.A.free()
if "c1" in which:
    m, n = 500, 1000
    p = problems.device_lasso(m, n, seed=0, power_iters=50, ctx=ctx)
    f = apx.LinearLeastSquares(p.A, p.b)
    s = apx.Solver(np.zeros(n), f=f, g=apx.NormL1(p.lam), rule=apx.OurRule(gamma=1 / p.Lf), tol=0.0, ctx=ctx, poll_every=1 << 30)
    out["C1 lasso 500x1000 fp64, adaptive_proxgrad (launch-bound)"] = timed(s, 500, 20, 2 * m * n * 8)
    s.close()
    s = apx.Solver(np.zeros(n), f=f, g=apx.NormL1(p.lam), rule=apx.OurRule(gamma=1 / p.Lf), tol=0.0, ctx=ctx, poll_every=1 << 30, use_graph=True)
    s.step(20, sync=True)
    t0 = time.perf_counter(); s.step(500, sync=True); dt = time.perf_counter() - t0
    out["C1 lasso 500x1000 fp64, adaptive_proxgrad, hipGraph replay"] = dict(iters_per_s=500 / dt, ms_per_iter=dt / 500 * 1e3)
    s.close()
print(json.dumps(out, indent=1))
json.dump(out, open("gpurun_out/bench_configs.json", "w"), indent=1)
