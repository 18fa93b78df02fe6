"""Randomised shapes / grid configurations for the two GEMV kernels and the fused vector
kernels (hypothesis, derandomised so that failures replay): ragged sizes around the 16-B
vector width, the 32-element padding, the row-tile and column-split edges."""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

pytestmark = pytest.mark.gpu

from oracle import adaprox_oracle as O

_ctx = {}


def _apx():
    import adaprox_amd as apx
    if "ctx" not in _ctx:
        _ctx["ctx"] = apx.Context(0)
    return apx, _ctx["ctx"]


dims = st.one_of(st.integers(1, 70), st.integers(500, 1100), st.sampled_from([31, 32, 33, 63, 64, 65, 511, 512, 513,
                                                                             1023, 1024, 1025, 2047, 2049, 4097]))


@settings(max_examples=60, deadline=None, derandomize=True, suppress_health_check=list(HealthCheck))
@given(m=dims, n=dims, f32=st.booleans(), nv=st.integers(0, 18), tv=st.integers(0, 10),
       nb=st.sampled_from([1, 7, 64, 1000, 4096, 30000]), tb=st.sampled_from([1, 9, 300, 8192, 65536]),
       seed=st.integers(0, 2**31 - 1))
def test_gemv_fuzz(m, n, f32, nv, tv, nb, tb, seed):
    apx, ctx = _apx()
    dtype = np.float32 if f32 else np.float64
    rng = np.random.default_rng(seed)
    A = np.asfortranarray(rng.uniform(-1, 1, (m, n)).astype(dtype))
    x = rng.standard_normal(n).astype(dtype)
    x[rng.uniform(size=n) < 0.3] = 0
    r = rng.standard_normal(m).astype(dtype)
    ctx.set_gemv_config(n_variant=nv, t_variant=tv, n_blocks=nb, t_blocks=tb)
    dA = apx.DeviceMatrix.from_numpy(A, ctx)
    tol = 2e-6 if f32 else 1e-12
    A64 = A.astype(np.float64)
    y_ref, g_ref = A64 @ x.astype(np.float64), A64.T @ r.astype(np.float64)
    y, g = dA @ x, dA.T @ r
    assert np.max(np.abs(y - y_ref)) <= tol * max(1.0, np.max(np.abs(y_ref)))
    assert np.max(np.abs(g - g_ref)) <= tol * max(1.0, np.max(np.abs(g_ref)))
    dA.free()


@settings(max_examples=25, deadline=None, derandomize=True, suppress_health_check=list(HealthCheck))
@given(m=st.integers(1, 400), n=st.integers(1, 400), lam=st.floats(0.01, 3.0), k=st.integers(1, 12),
       rule=st.sampled_from(["ours", "fixed", "plus"]), seed=st.integers(0, 2**31 - 1))
def test_lasso_iterations_fuzz(m, n, lam, k, rule, seed):
    apx, ctx = _apx()
    rng = np.random.default_rng(seed)
    A = np.asfortranarray(rng.uniform(-1, 1, (m, n)))
    b = rng.standard_normal(m)
    gam = 0.9 / max(np.linalg.norm(A, 2) ** 2, 1e-12)
    orule = {"ours": O.OurRule(gamma=gam), "fixed": O.FixedStepsize(gam), "plus": O.OurRulePlus(gamma=gam)}[rule]
    drule = {"ours": apx.OurRule(gamma=gam), "fixed": apx.FixedStepsize(gam), "plus": apx.OurRulePlus(gamma=gam)}[rule]
    x0 = rng.standard_normal(n)
    xo, _ = O.adaptive_proxgrad(x0, f=O.LinearLeastSquares(A, b), g=O.NormL1(lam), rule=orule, tol=0.0, maxit=k)
    ctx.set_gemv_config(n_variant=4, t_variant=2, n_blocks=4096, t_blocks=2048)
    x, it = apx.adaptive_proxgrad(x0, f=apx.LinearLeastSquares(A, b, ctx=ctx), g=apx.NormL1(lam), rule=drule, tol=0.0,
                                  maxit=k, ctx=ctx)
    assert it == k
    assert np.max(np.abs(x - xo)) <= 1e-10 * max(1.0, np.max(np.abs(xo)))


# ---------------------------------------------------------------------------------------
# the two-kernel small-problem paths (csrc/small.hip, small_pd.hip) against the generic kernels
# ---------------------------------------------------------------------------------------
def _generic_ctx():
    apx, _ = _apx()
    if "generic" not in _ctx:
        _ctx["generic"] = apx.Context(0)
        _ctx["generic"].fused_small(False)
    return _ctx["generic"]


small_dims = st.one_of(st.integers(1, 40), st.integers(60, 700), st.sampled_from([255, 256, 257, 1023, 1024, 1025, 2048,
                                                                                  2049, 3000, 4095, 4096]))


def _g_term(apx, kind, n, rng):
    if kind == "l1":
        return apx.NormL1(float(rng.uniform(0.01, 1.0)))
    if kind == "box":
        return apx.IndBox(-0.3, 0.5)
    if kind == "boxv":
        lb = -rng.uniform(0.1, 1.0, n)
        return apx.IndBox(lb, lb + rng.uniform(0.1, 2.0, n))
    if kind == "tl1":
        return apx.Translate(apx.NormL1(0.2), rng.uniform(-0.5, 0.5, n))
    return apx.Zero()


@settings(max_examples=40, deadline=None, derandomize=True, suppress_health_check=list(HealthCheck))
@given(m=small_dims, n=small_dims, logistic=st.booleans(), f32=st.booleans(),
       gk=st.sampled_from(["l1", "box", "boxv", "tl1", "zero"]), rule=st.sampled_from(["ours", "fixed", "plus", "mm"]),
       k=st.integers(0, 14), seed=st.integers(0, 2**31 - 1))
def test_small_proxgrad_vs_generic_fuzz(m, n, logistic, f32, gk, rule, k, seed):
    apx, _ = _apx()
    g_ctx = _generic_ctx()
    if m * n > 17_000_000:
        m = max(1, 17_000_000 // n)
    dtype = np.float32 if f32 else np.float64
    rng = np.random.default_rng(seed)
    A = np.asfortranarray(rng.uniform(-1, 1, (m, n)).astype(dtype))
    nx = n + 1 if logistic else n
    if logistic:
        b = (rng.uniform(size=m) < 0.5).astype(dtype)
        gam = 4.0 * m / (np.linalg.norm(A.astype(np.float64)) ** 2 + m)
    else:
        b = rng.standard_normal(m).astype(dtype)
        gam = 0.9 / max(np.linalg.norm(A.astype(np.float64), 2) ** 2, 1e-12)
    mk = {"ours": lambda: apx.OurRule(gamma=gam), "fixed": lambda: apx.FixedStepsize(gam),
          "plus": lambda: apx.OurRulePlus(gamma=gam), "mm": lambda: apx.MalitskyMishchenkoRule(gam)}[rule]
    x0 = (0.1 * rng.standard_normal(nx)).astype(dtype)
    out = []
    for ctx in (apx.default_context(), g_ctx):
        dA = apx.DeviceMatrix.from_numpy(A, ctx)
        f = apx.LogisticLoss(dA, b, ctx=ctx) if logistic else apx.LinearLeastSquares(dA, b, ctx=ctx)
        recs = []
        x, it = apx.adaptive_proxgrad(x0, f=f, g=_g_term(apx, gk, nx, np.random.default_rng(seed + 1)), rule=mk(), tol=0.0,
                                      maxit=k, ctx=ctx, record=recs.append)
        out.append((x, it, recs))
        dA.free()
    (xs, its, rs), (xg, itg, rg) = out
    tol = 5e-4 if f32 else 1e-9
    assert its == itg or min(m, n) == 1
    assert np.all(np.isfinite(xs)) == np.all(np.isfinite(xg))
    if np.all(np.isfinite(xg)):
        assert np.max(np.abs(xs - xg)) <= tol * max(1.0, np.max(np.abs(xg)))
        for a, b_ in zip(rs, rg):
            if np.isfinite(b_["objective"]):
                assert a["objective"] == pytest.approx(b_["objective"], rel=10 * tol, abs=10 * tol)


@settings(max_examples=30, deadline=None, derandomize=True, suppress_health_check=list(HealthCheck))
@given(m=small_dims, n=small_dims, quad=st.booleans(), hk=st.sampled_from(["l1", "l2", "tl1", "tl2", "indzero", "box"]),
       gk=st.sampled_from(["l1", "box", "zero"]), rule=st.sampled_from(["ours", "fixed"]), k=st.integers(0, 14),
       seed=st.integers(0, 2**31 - 1))
def test_small_primal_dual_vs_generic_fuzz(m, n, quad, hk, gk, rule, k, seed):
    apx, _ = _apx()
    g_ctx = _generic_ctx()
    if quad:
        n = min(n, 1500)
    if m * n > 17_000_000:
        m = max(1, 17_000_000 // n)
    rng = np.random.default_rng(seed)
    A = np.asfortranarray(rng.uniform(-1, 1, (m, n)))
    nrm = float(np.linalg.norm(A))
    bvec = rng.standard_normal(m)
    hmk = {"l1": lambda: apx.NormL1(0.7), "l2": lambda: apx.NormL2(1.3), "tl1": lambda: apx.Translate(apx.NormL1(1.0), -bvec),
           "tl2": lambda: apx.Translate(apx.NormL2(1.0), -bvec), "indzero": lambda: apx.IndZero(),
           "box": lambda: apx.IndBox(-0.5, 0.5)}[hk]
    if quad:
        B = rng.uniform(-1, 1, (n, max(2, n // 3)))
        Q = np.asfortranarray(B @ B.T / n)
        qv = rng.standard_normal(n)
        Lf = float(np.linalg.norm(Q, 2))
    else:
        Lf = 0.0
    t = 1.0
    gam = 1.0 / (2.0 * 1.2 * t * max(nrm, 1e-12) + Lf)
    mk = {"ours": lambda: apx.OurRule(gamma=gam, t=t, norm_A=nrm), "fixed": lambda: apx.FixedStepsize(gam, t)}[rule]
    x0, y0 = 0.1 * rng.standard_normal(n), 0.1 * rng.standard_normal(m)
    out = []
    for ctx in (apx.default_context(), g_ctx):
        dA = apx.DeviceMatrix.from_numpy(A, ctx)
        f = apx.Quadratic(apx.DeviceMatrix.from_numpy(Q, ctx), qv, ctx=ctx) if quad else apx.Zero()
        recs = []
        x, y, it = apx.adaptive_primal_dual(x0, y0, f=f, g=_g_term(apx, gk, n, np.random.default_rng(seed + 1)), h=hmk(),
                                            A=dA, rule=mk(), tol=0.0, maxit=k, ctx=ctx, record=recs.append)
        out.append((x, y, it, recs))
    (xs, ys, its, rs), (xg, yg, itg, rg) = out
    assert its == itg
    assert np.max(np.abs(xs - xg)) <= 1e-9 * max(1.0, np.max(np.abs(xg)))
    assert np.max(np.abs(ys - yg)) <= 1e-9 * max(1.0, np.max(np.abs(yg)))
    for a, b_ in zip(rs, rg):
        assert a["norm_res"] == pytest.approx(b_["norm_res"], rel=1e-6, abs=1e-9)
        if np.isfinite(b_["objective"]):
            assert a["objective"] == pytest.approx(b_["objective"], rel=1e-8, abs=1e-8)
        else:
            assert a["objective"] == b_["objective"]


def test_small_path_soak_long_run():
    """60 000 iterations through the two-kernel path: the buffer rotation and the control blocks stay consistent
    (same converged point as the generic kernels, objective at the optimum), and a converged run stays frozen."""
    apx, _ = _apx()
    g_ctx = _generic_ctx()
    rng = np.random.default_rng(5)
    A = np.asfortranarray(rng.uniform(-1, 1, (150, 260)))
    b = rng.standard_normal(150)
    gam = 1.0 / np.linalg.norm(A, 2) ** 2
    # (fixed step: with tol = 0 an adaptive rule keeps growing gamma once the iterate is exactly stationary —
    #  dx = dg = 0 makes C = L = D = 0 — until it overflows, in the reference as well)
    s = apx.Solver(np.zeros(260), f=apx.LinearLeastSquares(A, b), g=apx.NormL1(0.5), rule=apx.FixedStepsize(gam), tol=0.0)
    s.step(60_000, sync=True)
    xs, _, res = s.result()
    s.close()
    xg, _ = apx.adaptive_proxgrad(np.zeros(260), f=apx.LinearLeastSquares(apx.DeviceMatrix.from_numpy(A, g_ctx), b, ctx=g_ctx),
                                  g=apx.NormL1(0.5), rule=apx.OurRule(gamma=gam), tol=1e-12, maxit=60_000, ctx=g_ctx)
    obj = lambda x: 0.5 * np.linalg.norm(A @ x - b) ** 2 + 0.5 * np.sum(np.abs(x))   # noqa: E731
    assert res.numit == 60_000 and np.all(np.isfinite(xs))
    assert obj(xs) == pytest.approx(obj(xg), rel=1e-10) and np.max(np.abs(xs - xg)) < 1e-7
    # tol > 0: converges early, later iterations are no-ops, the tested iterate is returned
    x1, it1 = apx.adaptive_proxgrad(np.zeros(260), f=apx.LinearLeastSquares(A, b), g=apx.NormL1(0.5),
                                    rule=apx.OurRule(gamma=gam), tol=1e-9, maxit=60_000, poll_every=5000)
    assert it1 < 60_000 and np.max(np.abs(x1 - xg)) < 1e-6
    lad_A = np.asfortranarray(np.hstack([rng.uniform(-1, 1, (300, 9)), np.ones((300, 1))]))
    yv = lad_A @ rng.standard_normal(10) + rng.laplace(size=300)
    nrm = float(np.linalg.norm(lad_A))
    kw = dict(f=apx.Zero(), g=apx.NormL1(1.0), h=apx.Translate(apx.NormL1(1.0), -yv), Lf=0.0, norm_A=nrm, tol=1e-6,
              maxit=300_000)
    xa, ya, ita = apx.condat_vu(np.zeros(10), np.zeros(300), A=lad_A, poll_every=4096, **kw)
    xb, yb, itb = apx.condat_vu(np.zeros(10), np.zeros(300), A=apx.DeviceMatrix.from_numpy(lad_A, g_ctx), ctx=g_ctx, **kw)
    assert ita < 300_000 and abs(ita - itb) <= max(5, itb // 20)
    assert np.max(np.abs(xa - xb)) < 1e-5 * max(1.0, np.max(np.abs(xb)))


tall_rows = st.one_of(st.integers(2048, 9000), st.sampled_from([2048, 2049, 4095, 4096, 4097, 8191, 8192, 8193, 30000, 70000, 300007]))
skinny_cols = st.one_of(st.integers(1, 40), st.sampled_from([3, 4, 5, 15, 16, 17, 31, 32, 33, 63, 64, 65, 111, 122, 123]))


@settings(max_examples=60, deadline=None, derandomize=True, suppress_health_check=list(HealthCheck))
@given(m=tall_rows, n=skinny_cols, logistic=st.booleans(), f32=st.booleans(), gk=st.sampled_from(["l1", "box", "tl1", "zero"]),
       method=st.sampled_from(["adaptive", "fixed", "bt", "bt_nesterov", "nesterov", "agraal"]), k=st.integers(0, 9),
       records=st.booleans(), seed=st.integers(0, 2**31 - 1))
def test_tall_skinny_methods_vs_generic_fuzz(m, n, logistic, f32, gk, method, k, records, seed):
    """csrc/skinny_pg.hip, skinny_bt.hip, skinny_mom.hip against the host-driven loops on the generic kernels: ragged row
    counts around the 64-row groups and the block limits, column counts around the 4 / 16 columns-per-wave slices."""
    apx, _ = _apx()
    g_ctx = _generic_ctx()
    if logistic and n > 123:
        n = 123
    if 4 * (n + 1) > m:
        m = 4 * (n + 1)
    dtype = np.float32 if f32 else np.float64
    rng = np.random.default_rng(seed)
    A = np.asfortranarray((rng.uniform(-1, 1, (m, n)) * np.logspace(0, -1, n)).astype(dtype))
    nx = n + 1 if logistic else n
    if logistic:
        w = rng.standard_normal(n)
        b = (1 / (1 + np.exp(-(A.astype(np.float64) @ w))) > rng.uniform(size=m)).astype(dtype)
        gam = 4.0 * m / (np.linalg.norm(A.astype(np.float64)) ** 2 + m)
    else:
        b = rng.standard_normal(m).astype(dtype)
        gam = 0.9 / max(np.linalg.norm(A.astype(np.float64), 2) ** 2, 1e-12)
    x0 = (0.1 * rng.standard_normal(nx)).astype(dtype)
    out = []
    for ctx in (apx.default_context(), g_ctx):
        dA = apx.DeviceMatrix.from_numpy(A, ctx)
        f = apx.LogisticLoss(dA, b, ctx=ctx) if logistic else apx.LinearLeastSquares(dA, b, ctx=ctx)
        g = _g_term(apx, gk, nx, np.random.default_rng(seed + 1))
        recs = []
        kw = dict(f=f, g=g, tol=0.0, maxit=k, ctx=ctx, record=recs.append if records else None)
        if method == "adaptive":
            x, it = apx.adaptive_proxgrad(x0, rule=apx.OurRule(gamma=gam), **kw)
        elif method == "fixed":
            x, it = apx.fixed_proxgrad(x0, gamma=gam, **kw)
        elif method == "bt":
            x, it = apx.backtracking_proxgrad(x0, gamma0=5 * gam, xi=1.3, **kw)
        elif method == "bt_nesterov":
            x, it = apx.backtracking_nesterov(x0, gamma0=5 * gam, **kw)
        elif method == "nesterov":
            x, it = apx.fixed_nesterov(x0, gamma=gam, **kw)
        else:
            x, it = apx.agraal(x0, x0=x0 + 0.01, gamma0=gam, **kw)
        out.append((x, it, recs))
        dA.free()
    (xs, its, rs), (xg, itg, rg) = out
    # (a rounding-level linesearch decision near convergence may differ: the window is short and the test loose there)
    tol = 2e-3 if f32 else 1e-7
    if its != itg:          # an iterate that stops moving gives norm_res == 0.0 exactly or not, by rounding: legitimate
        assert max(its, itg) == k and np.all(np.isfinite(xs)) and np.all(np.isfinite(xg))
        return
    assert its <= k and len(rs) == len(rg)          # (its < k: both stopped at an exact fixed point)
    assert np.all(np.isfinite(xs)) == np.all(np.isfinite(xg))
    if np.all(np.isfinite(xg)):
        assert np.max(np.abs(xs - xg)) <= tol * max(1.0, np.max(np.abs(xg)))
        for a, b_ in zip(rs, rg):
            assert a["it"] == b_["it"]
            if np.isfinite(b_["objective"]):
                assert a["objective"] == pytest.approx(b_["objective"], rel=10 * tol, abs=10 * tol)


@settings(max_examples=60, deadline=None, derandomize=True, suppress_health_check=list(HealthCheck))
@given(m=st.one_of(st.integers(1024, 6000), st.sampled_from([1024, 1025, 3071, 3072, 3073, 4096, 4097, 20000])),
       n=st.one_of(st.integers(1, 30), st.sampled_from([7, 8, 9, 15, 16, 17, 64, 120, 124])),
       hk=st.sampled_from(["tl1", "tl2", "l1"]), method=st.sampled_from(["adapdm", "adapdm+", "mp", "cv"]),
       f32=st.booleans(), k=st.integers(0, 9), seed=st.integers(0, 2**31 - 1))
def test_tall_skinny_primal_dual_methods_vs_generic_fuzz(m, n, hk, method, f32, k, seed):
    """csrc/skinny_pd.hip, skinny_ls.hip, skinny_mp.hip (f = Zero, LAD / square-root-lasso terms) against the generic
    kernels and the host-driven linesearch loops."""
    apx, _ = _apx()
    g_ctx = _generic_ctx()
    if 4 * n > m:
        m = 4 * n
    if method in ("adapdm+", "mp") and hk == "l1":
        hk = "tl1"          # (h without data: y stays 0 and AdaPDM+'s test is 0 / 0 for ever — the reference would spin)
    dtype = np.float32 if f32 else np.float64
    rng = np.random.default_rng(seed)
    A = np.asfortranarray(np.hstack([rng.uniform(-1, 1, (m, max(n - 1, 0))), np.ones((m, 1))]).astype(dtype))
    yv = (A.astype(np.float64) @ rng.standard_normal(n) + rng.laplace(size=m)).astype(dtype)
    nA = float(np.linalg.norm(A.astype(np.float64)))
    x0, y0 = np.zeros(n, dtype=dtype), np.zeros(m, dtype=dtype)
    out = []
    for ctx in (apx.default_context(), g_ctx):
        dA = apx.DeviceMatrix.from_numpy(A, ctx)
        h = {"tl1": lambda: apx.Translate(apx.NormL1(1.0), -yv), "tl2": lambda: apx.Translate(apx.NormL2(1.0), -yv),
             "l1": lambda: apx.NormL1(0.7)}[hk]()
        kw = dict(f=apx.Zero(), g=apx.NormL1(0.05 if hk == "tl2" else 5.0), h=h, A=dA, tol=0.0, maxit=k, ctx=ctx)
        if method == "adapdm":
            x, y, it = apx.adaptive_primal_dual(x0, y0, rule=apx.OurRule(t=1.0, norm_A=nA), **kw)
        elif method == "adapdm+":
            x, y, it = apx.adaptive_linesearch_primal_dual(x0, y0, eta=nA, **kw)
        elif method == "mp":
            x, y, it = apx.malitsky_pock(x0, y0, sigma=1.0 / nA, **kw)
        else:
            x, y, it = apx.condat_vu(x0, y0, Lf=0.0, norm_A=nA, **kw)
        out.append((x, y, it))
        dA.free()
    (xs, ys, its), (xg, yg, itg) = out
    tol = 2e-3 if f32 else 1e-7
    if its != itg:          # (an exact fixed point reached by one summation order and not the other)
        assert max(its, itg) == k
        return
    assert np.all(np.isfinite(xs)) == np.all(np.isfinite(xg))
    if np.all(np.isfinite(xg)) and np.all(np.isfinite(yg)):
        assert np.max(np.abs(xs - xg)) <= tol * max(1.0, np.max(np.abs(xg)))
        assert np.max(np.abs(ys - yg)) <= tol * max(1.0, np.max(np.abs(yg)))


@settings(max_examples=30, deadline=None, derandomize=True, suppress_health_check=list(HealthCheck))
@given(N=st.one_of(st.integers(2, 600), st.sampled_from([255, 256, 257, 1023, 1024, 1025, 4095, 4096, 4097])),
       rows=st.sampled_from([1, 2, 5, 64]), t=st.sampled_from([0.05, 1.0, 8.0]), f32=st.booleans(), k=st.integers(0, 9),
       seed=st.integers(0, 2**31 - 1))
def test_malitsky_pock_quadratic_vs_host_driven_fuzz(N, rows, t, f32, k, seed):
    """csrc/small_mpq.hip (f = Quadratic, a few rows of A) against the host-driven loop."""
    apx, _ = _apx()
    g_ctx = _generic_ctx()
    dtype = np.float32 if f32 else np.float64
    rng = np.random.default_rng(seed)
    X = rng.uniform(-1, 1, (N, 9))
    yl = np.sign(X @ rng.standard_normal(9) + 0.1 * rng.standard_normal(N))
    Q = np.asfortranarray(((yl[:, None] * X) @ (yl[:, None] * X).T).astype(dtype))
    q = -np.ones(N, dtype=dtype)
    A = np.asfortranarray(np.vstack([yl[None, :]] + [rng.uniform(-1, 1, (1, N)) for _ in range(rows - 1)]).astype(dtype))
    nA = float(np.linalg.norm(A.astype(np.float64)))
    out = []
    for ctx in (apx.default_context(), g_ctx):
        f = apx.Quadratic(apx.DeviceMatrix.from_numpy(Q, ctx), q, ctx=ctx)
        x, y, it = apx.malitsky_pock(np.zeros(N, dtype=dtype), np.zeros(rows, dtype=dtype), f=f, g=apx.IndBox(0.0, 0.1),
                                     h=apx.IndZero(), A=apx.DeviceMatrix.from_numpy(A, ctx), sigma=1.0 / nA, t=t, tol=0.0,
                                     maxit=k, ctx=ctx)
        out.append((x, y, it))
    (xs, ys, its), (xg, yg, itg) = out
    tol = 2e-3 if f32 else 1e-8
    assert its == itg == k
    assert np.max(np.abs(xs - xg)) <= tol * max(1.0, np.max(np.abs(xg)))
    assert np.max(np.abs(ys - yg)) <= tol * max(1.0, np.max(np.abs(yg)))


@settings(max_examples=25, deadline=None, derandomize=True, suppress_health_check=list(HealthCheck))
@given(n=st.sampled_from([125, 1023, 1024, 1025, 2047, 2048, 2049, 3000, 5119, 5120, 5121, 7679, 10751, 10752]),
       extra_rows=st.integers(0, 9), logistic=st.booleans(), f32=st.booleans(), rule=st.sampled_from(["ours", "fixed", "mm"]),
       k=st.integers(0, 6), seed=st.integers(0, 2**31 - 1), layout=st.sampled_from(["cols", "rows"]))
def test_one_pass_vs_two_passes_fuzz(n, extra_rows, logistic, f32, rule, k, seed, layout):
    """csrc/slab.hip (f and grad f in one pass over A) against the two GEMV passes: column counts around the 2 / 5 / 14
    columns-per-thread shapes, row counts that are not multiples of the step or of the slab."""
    apx, _ = _apx()
    dtype = np.float32 if f32 else np.float64
    es = 4 if f32 else 8
    m = max(4096, -(-(64 * 1024 * 1024) // (n * es))) + extra_rows            # just past the 64 MB where the path starts
    if m * n * es > 400 * 1024 * 1024:
        return
    rng = np.random.default_rng(seed)
    A = np.asfortranarray(rng.uniform(-1, 1, (m, n)).astype(dtype))
    nx = n + 1 if logistic else n
    if logistic:
        w = rng.standard_normal(n) * (rng.uniform(size=n) < 0.05)
        b = (1 / (1 + np.exp(-(A.astype(np.float64) @ w))) > rng.uniform(size=m)).astype(dtype)
        gam = 4.0 * m / (float(np.sum(A.astype(np.float64) ** 2)) + m)
    else:
        b = rng.standard_normal(m).astype(dtype)
        gam = 1.0 / (m * n / 3.0)
    mk = {"ours": lambda: apx.OurRule(gamma=gam), "fixed": lambda: apx.FixedStepsize(gam),
          "mm": lambda: apx.MalitskyMishchenkoRule(gam)}[rule]
    x0 = (0.01 * rng.standard_normal(nx)).astype(dtype)
    out = []
    for one_pass in (True, False):
        ctx = apx.Context(0)
        ctx.fused_small(one_pass)
        ctx.one_pass_threshold(16 << 20)
        dA = apx.DeviceMatrix.from_numpy(A, ctx, layout=layout)      # (where the library keeps A's bytes must not show: round 5)
        f = apx.LogisticLoss(dA, b, ctx=ctx) if logistic else apx.LinearLeastSquares(dA, b, ctx=ctx)
        recs = []
        x, it = apx.adaptive_proxgrad(x0, f=f, g=apx.NormL1(0.01 if logistic else 1.0), rule=mk(), tol=0.0, maxit=k, ctx=ctx,
                                      record=recs.append)
        out.append((x, it, recs))
        dA.free()
    (x1, it1, r1), (x2, it2, r2) = out
    tol = 2e-3 if f32 else 1e-9
    assert it1 == it2 == k
    assert np.max(np.abs(x1 - x2)) <= tol * max(1.0, np.max(np.abs(x2)))
    for a, b_ in zip(r1, r2):
        assert a["objective"] == pytest.approx(b_["objective"], rel=1e-4 if f32 else 1e-10)


@settings(max_examples=16, deadline=None, derandomize=True, suppress_health_check=list(HealthCheck))
@given(n=st.sampled_from([125, 1024, 2048, 2049, 5120, 5121, 10752]), extra_rows=st.integers(0, 9),
       hk=st.sampled_from(["tl1", "tbox"]), f32=st.booleans(), fixed=st.booleans(), k=st.integers(0, 5),
       seed=st.integers(0, 2**31 - 1), layout=st.sampled_from(["cols", "rows"]))
def test_one_pass_primal_dual_vs_two_passes_fuzz(n, extra_rows, hk, f32, fixed, k, seed, layout):
    """csrc/slab_pd.hip (A x, the dual step, A' y in one pass over A) against the generic kernels."""
    apx, _ = _apx()
    dtype = np.float32 if f32 else np.float64
    es = 4 if f32 else 8
    m = max(4096, -(-(64 * 1024 * 1024) // (n * es))) + extra_rows
    if m * n * es > 400 * 1024 * 1024:
        return
    rng = np.random.default_rng(seed)
    A = np.asfortranarray(np.hstack([rng.uniform(-1, 1, (m, n - 1)), np.ones((m, 1))]).astype(dtype))
    yv = (rng.laplace(size=m) + 0.3).astype(dtype)
    nA = float(np.sqrt(m * (n - 1) / 3.0 + m))
    x0, y0 = (0.01 * rng.standard_normal(n)).astype(dtype), (0.1 * rng.standard_normal(m)).astype(dtype)
    out = []
    for one_pass in (True, False):
        ctx = apx.Context(0)
        ctx.fused_small(one_pass)
        ctx.one_pass_threshold(16 << 20)
        dA = apx.DeviceMatrix.from_numpy(A, ctx, layout=layout)
        h = apx.Translate(apx.NormL1(1.0) if hk == "tl1" else apx.IndBox(-0.5, 0.8), -yv)
        rule = apx.FixedStepsize(1.0 / nA, 0.99) if fixed else apx.OurRule(t=1.0, norm_A=nA)
        x, y, it = apx.adaptive_primal_dual(x0, y0, f=apx.Zero(), g=apx.NormL1(3.0), h=h, A=dA, rule=rule, tol=0.0, maxit=k,
                                            ctx=ctx)
        out.append((x, y, it))
        dA.free()
    (x1, y1, it1), (x2, y2, it2) = out
    tol = 2e-3 if f32 else 1e-9
    assert it1 == it2 == k
    assert np.max(np.abs(x1 - x2)) <= tol * max(1.0, np.max(np.abs(x2)))
    assert np.max(np.abs(y1 - y2)) <= tol * max(1.0, np.max(np.abs(y2)))
