"""GPU parity for the comparison methods of the experiments against the
oracle: backtracking_nesterov, fixed_nesterov, agraal (src/AdaProx.jl:66-192) and malitsky_pock
(:555-629), plus the reference's own third "Simple 2D" assertion through the HIP path."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import adaprox_oracle as O
from oracle import problems as P
from test_gpu_solvers import Simple2DObjective, _simple_2d_box


@pytest.fixture(scope="module")
def apx():
    import adaprox_amd as apx
    apx.load()
    return apx


def _rel_inf(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b)))) / max(float(np.max(np.abs(b))), 1e-300)


@pytest.fixture(scope="module")
def lasso():
    prob = P.random_lasso(500, 1000, pfactor=10, seed=0)
    return prob, 1.0 / P.opnorm_sq_power(prob.A, iters=50)


def _compare(orecs, drecs, keys, rel):
    assert len(orecs) == len(drecs)
    for ro, rd in zip(orecs, drecs):
        assert list(rd) == list(ro)                       # same record schema, same key order
        for k in keys:
            assert rd[k] == pytest.approx(ro[k], rel=rel), k
        for k in ro:
            if k.endswith("_evals"):
                assert rd[k] == ro[k], k


def test_reference_simple_2d_backtracking_nesterov(apx):
    f, g = Simple2DObjective(), _simple_2d_box(apx)
    sol, numit = apx.backtracking_nesterov(np.ones(2), f=f, g=g, gamma0=1.0)      # test/runtests.jl:45-50
    assert f(sol) < 1e-7 and g(sol) == 0


def test_backtracking_nesterov(apx, lasso):
    prob, gam = lasso
    K = 150
    of, og = O.Counting(O.LinearLeastSquares(prob.A, prob.b)), O.Counting(O.NormL1(1.0))
    df, dg = apx.Counting(apx.LinearLeastSquares(prob.A, prob.b)), apx.Counting(apx.NormL1(1.0))
    orecs, drecs = [], []
    zo, ito = O.backtracking_nesterov(np.zeros(1000), f=of, g=og, gamma0=gam * 8, tol=0.0, maxit=K, record=orecs.append)
    z, it = apx.backtracking_nesterov(np.zeros(1000), f=df, g=dg, gamma0=gam * 8, tol=0.0, maxit=K, record=drecs.append)
    assert it == ito == K and _rel_inf(z, zo) <= 1e-10
    _compare(orecs, drecs, ("gamma", "norm_res", "objective"), 1e-8)
    assert (df.eval_count, df.grad_count, dg.prox_count) == (of.eval_count, of.grad_count, og.prox_count)
    assert any(r["gamma"] < gam * 8 for r in drecs)      # the linesearch actually backtracked


def test_fixed_nesterov(apx, lasso):
    prob, gam = lasso
    K = 200
    for kw in (dict(gamma=gam), dict(Lf=1 / gam), dict(gamma=gam, muf=0.05 / gam, mug=0.0)):
        of, df = O.Counting(O.LinearLeastSquares(prob.A, prob.b)), apx.Counting(apx.LinearLeastSquares(prob.A, prob.b))
        orecs, drecs = [], []
        xo, _ = O.fixed_nesterov(np.zeros(1000), f=of, g=O.NormL1(1.0), tol=0.0, maxit=K, record=orecs.append, **kw)
        x, _ = apx.fixed_nesterov(np.zeros(1000), f=df, g=apx.NormL1(1.0), tol=0.0, maxit=K, record=drecs.append, **kw)
        assert _rel_inf(x, xo) <= 1e-10
        _compare(orecs, drecs, ("gamma", "norm_res", "objective"), 1e-8)
        assert (df.eval_count, df.grad_count) == (of.eval_count, of.grad_count) == (K, K)
    # without a record sink the lazy objective costs nothing: same iterate
    x2, _ = apx.fixed_nesterov(np.zeros(1000), f=apx.LinearLeastSquares(prob.A, prob.b), g=apx.NormL1(1.0), tol=0.0,
                               maxit=K, gamma=gam, muf=0.05 / gam)
    assert np.array_equal(x2, x)
    with pytest.raises(AssertionError):
        apx.fixed_nesterov(np.zeros(3), f=apx.Zero(), g=apx.Zero(), gamma=1.0, Lf=1.0)      # :104


def test_agraal(apx, lasso):
    prob, gam = lasso
    # aGRAAL's step (a ratio of differences, :175-177) amplifies rounding noise like the MM rule:
    # two CPU fp64 runs differing only in summation order are 1e-13 apart at k=50, 2e-10 at
    # k=100 and 1e-8 at k=120 on this problem — pinned over 50 iterations
    K = 50
    x0 = np.cos(np.arange(1000)) * 0.01
    for g0 in (gam, None):
        of, df = O.Counting(O.LinearLeastSquares(prob.A, prob.b)), apx.Counting(apx.LinearLeastSquares(prob.A, prob.b))
        orecs, drecs = [], []
        xo, _ = O.agraal(np.zeros(1000), f=of, g=O.NormL1(1.0), x0=x0, gamma0=g0, tol=0.0, maxit=K, record=orecs.append)
        x, _ = apx.agraal(np.zeros(1000), f=df, g=apx.NormL1(1.0), x0=x0, gamma0=g0, tol=0.0, maxit=K,
                          record=drecs.append)
        assert _rel_inf(x, xo) <= 1e-9
        _compare(orecs, drecs, ("gamma", "norm_res", "objective"), 1e-7)
        assert (df.eval_count, df.grad_count) == (of.eval_count, of.grad_count)


@pytest.mark.parametrize("t", [0.1, 1.0, 10.0])
def test_malitsky_pock_lad(apx, t):
    """experiments/least_absolute_deviation/runme.jl:64-78: sigma = 1, t in t_values."""
    m, n = 600, 240
    prob = P.random_lad(m, n, seed=1, lam=10.0)
    K = 120
    oA, dA = O.Counting(prob.A), apx.Counting(apx.DeviceMatrix.from_numpy(prob.A))
    orecs, drecs = [], []
    xo, yo, _ = O.malitsky_pock(np.zeros(n), np.zeros(m), f=O.Zero(), g=O.NormL1(prob.lam),
                                h=O.Translate(O.NormL1(1.0), -prob.y), A=oA, sigma=1.0, t=t, tol=0.0, maxit=K,
                                record=orecs.append)
    x, y, _ = apx.malitsky_pock(np.zeros(n), np.zeros(m), f=apx.Zero(), g=apx.NormL1(prob.lam),
                                h=apx.Translate(apx.NormL1(1.0), -prob.y), A=dA, sigma=1.0, t=t, tol=0.0, maxit=K,
                                record=drecs.append)
    assert _rel_inf(x, xo) <= 1e-10 and _rel_inf(y, yo) <= 1e-10
    _compare(orecs, drecs, ("gamma", "sigma", "norm_res", "objective"), 1e-8)
    assert (dA.mul_count, dA.amul_count) == (oA.mul_count, oA.amul_count)


def test_malitsky_pock_with_smooth_f(apx):
    rng = np.random.default_rng(9)
    m, n, k = 120, 70, 50
    A = np.asfortranarray(rng.uniform(-1, 1, (m, n)))
    Q = np.asfortranarray(rng.uniform(-1, 1, (k, n)) / np.sqrt(n))
    q, yv = rng.standard_normal(k), rng.standard_normal(m)
    kw = dict(sigma=0.5, t=1.0, tol=0.0, maxit=80)
    of, df = O.Counting(O.LinearLeastSquares(Q, q)), apx.Counting(apx.LinearLeastSquares(Q, q))
    xo, yo, _ = O.malitsky_pock(np.zeros(n), np.zeros(m), f=of, g=O.NormL1(0.3), h=O.Translate(O.NormL1(1.0), -yv),
                                A=A, **kw)
    x, y, _ = apx.malitsky_pock(np.zeros(n), np.zeros(m), f=df, g=apx.NormL1(0.3),
                                h=apx.Translate(apx.NormL1(1.0), -yv), A=A, **kw)
    assert _rel_inf(x, xo) <= 1e-9 and _rel_inf(y, yo) <= 1e-9
    assert (df.eval_count, df.grad_count) == (of.eval_count, of.grad_count)


# ---------------------------------------------------------------------------------------
# tall, skinny A: Malitsky-Pock with the device-resident linesearch of csrc/skinny_mp.hip
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("m,n,hkind,t", [(506, 14, "l1", 0.1), (506, 14, "l1", 10.0), (506, 14, "l2", 1.0), (4177, 9, "l1", 1.0),
                                         (8192, 13, "l2", 0.5), (300, 1, "l1", 1.0), (1000, 120, "l2", 2.0)])
def test_malitsky_pock_skinny_vs_oracle(apx, m, n, hkind, t):
    prob = P.random_lad(m, n, seed=7, lam=10.0 if hkind == "l1" else 0.05)
    mkh = (lambda M: M.Translate(M.NormL1(1.0), -prob.y)) if hkind == "l1" else (lambda M: M.Translate(M.NormL2(1.0), -prob.y))
    K = 60      # (a linesearch decision is a comparison of two nearly equal numbers once the iterates settle)
    oA, dA = O.Counting(prob.A), apx.Counting(apx.DeviceMatrix.from_numpy(prob.A))
    og, dg = O.Counting(O.NormL1(prob.lam)), apx.Counting(apx.NormL1(prob.lam))
    orecs, drecs = [], []
    xo, yo, ito = O.malitsky_pock(np.zeros(n), np.zeros(m), f=O.Zero(), g=og, h=mkh(O), A=oA, sigma=1.0, t=t, tol=0.0,
                                  maxit=K, record=orecs.append)
    x, y, it = apx.malitsky_pock(np.zeros(n), np.zeros(m), f=apx.Zero(), g=dg, h=mkh(apx), A=dA, sigma=1.0, t=t, tol=0.0,
                                 maxit=K, record=drecs.append)
    assert it == ito == K and len(drecs) == K
    assert _rel_inf(x, xo) <= 1e-9 and _rel_inf(y, yo) <= 1e-9
    _compare(orecs, drecs, ("gamma", "sigma", "norm_res", "objective"), 1e-7)
    for ro, rd in zip(orecs, drecs):
        assert (rd["A_evals"], rd["At_evals"], rd["prox_g_evals"]) == (ro["A_evals"], ro["At_evals"], ro["prox_g_evals"])
    assert (dA.mul_count, dA.amul_count, dg.prox_count) == (oA.mul_count, oA.amul_count, og.prox_count)


def test_malitsky_pock_skinny_termination_and_maxit(apx):
    prob = P.random_lad(2000, 10, seed=9, lam=5.0)
    terms = lambda M: dict(f=M.Zero(), g=M.NormL1(prob.lam), h=M.Translate(M.NormL1(1.0), -prob.y))   # noqa: E731
    orecs = []
    O.malitsky_pock(np.zeros(10), np.zeros(2000), A=prob.A, sigma=1.0, t=1.0, tol=0.0, maxit=150, record=orecs.append,
                    **terms(O))
    tol = orecs[99]["norm_res"] * (1 + 1e-7)
    xo, yo, ito = O.malitsky_pock(np.zeros(10), np.zeros(2000), A=prob.A, sigma=1.0, t=1.0, tol=tol, maxit=5000, **terms(O))
    x, y, it = apx.malitsky_pock(np.zeros(10), np.zeros(2000), A=prob.A, sigma=1.0, t=1.0, tol=tol, maxit=5000, **terms(apx))
    assert it == ito <= 100 and _rel_inf(x, xo) <= 1e-9 and _rel_inf(y, yo) <= 1e-9
    for K in (0, 1, 49, 97):
        xo, yo, _ = O.malitsky_pock(np.zeros(10), np.zeros(2000), A=prob.A, sigma=1.0, t=0.3, tol=0.0, maxit=K, **terms(O))
        x, y, it = apx.malitsky_pock(np.zeros(10), np.zeros(2000), A=prob.A, sigma=1.0, t=0.3, tol=0.0, maxit=K, **terms(apx))
        assert it == K and _rel_inf(x, xo) <= 1e-8 and (K == 0 or _rel_inf(y, yo) <= 1e-8)


# ---------------------------------------------------------------------------------------
# fixed Nesterov and aGRAAL as device-resident loops (csrc/small_mom.hip): more shapes, logistic, edge cases
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("m,n,logistic", [(100, 300, False), (4000, 1000, False), (900, 70, True), (64, 17, False)])
def test_small_mom_shapes(apx, m, n, logistic):
    if logistic:
        lr = P.random_logreg(m, n, seed=4, dtype=np.float64)
        A, b, lam, gam, nx = lr.X, lr.y, lr.lam, lr.gamma0, n + 1
        mkf = lambda M: M.LogisticLoss(A, b)      # noqa: E731
    else:
        rng = np.random.default_rng(m + n)
        A, b, lam, nx = rng.uniform(-1, 1, (m, n)), rng.standard_normal(m), 0.3, n
        gam = 1.0 / P.opnorm_sq_power(A, iters=30)
        mkf = lambda M: M.LinearLeastSquares(A, b)      # noqa: E731
    K = 40
    ro, rd = [], []
    xo, ito = O.fixed_nesterov(np.zeros(nx), f=mkf(O), g=O.NormL1(lam), gamma=gam, tol=0.0, maxit=K, record=ro.append)
    x, it = apx.fixed_nesterov(np.zeros(nx), f=mkf(apx), g=apx.NormL1(lam), gamma=gam, tol=0.0, maxit=K, record=rd.append)
    assert it == ito == K and _rel_inf(x, xo) <= 1e-10
    _compare(ro, rd, ("gamma", "norm_res", "objective"), 1e-8)
    x0 = 0.01 * np.cos(np.arange(nx))
    ro, rd = [], []
    xo, ito = O.agraal(np.zeros(nx), f=mkf(O), g=O.NormL1(lam), x0=x0, tol=0.0, maxit=K, record=ro.append)
    x, it = apx.agraal(np.zeros(nx), f=mkf(apx), g=apx.NormL1(lam), x0=x0, tol=0.0, maxit=K, record=rd.append)
    assert it == ito == K and _rel_inf(x, xo) <= 1e-9
    _compare(ro, rd, ("gamma", "norm_res", "objective"), 1e-7)


def test_small_mom_termination_and_maxit(apx, lasso):
    prob, gam = lasso
    f_o, f_d = O.LinearLeastSquares(prob.A, prob.b), apx.LinearLeastSquares(prob.A, prob.b)
    x0 = 0.01 * np.cos(np.arange(1000))
    for name, kw in (("fixed_nesterov", dict(gamma=gam)), ("agraal", dict(x0=x0, gamma0=gam))):
        ro = []
        getattr(O, name)(np.zeros(1000), f=f_o, g=O.NormL1(1.0), tol=0.0, maxit=60, record=ro.append, **kw)
        tol = min(r["norm_res"] for r in ro[:40]) * (1 + 1e-7)
        of, df = O.Counting(f_o), apx.Counting(f_d)
        xo, ito = getattr(O, name)(np.zeros(1000), f=of, g=O.NormL1(1.0), tol=tol, maxit=5000, **kw)
        x, it = getattr(apx, name)(np.zeros(1000), f=df, g=apx.NormL1(1.0), tol=tol, maxit=5000, **kw)
        assert it == ito <= 40 and _rel_inf(x, xo) <= 1e-10
        assert (df.eval_count, df.grad_count) == (of.eval_count, of.grad_count)
        for K in (0, 1, 2, 65):
            xo, _ = getattr(O, name)(np.zeros(1000), f=f_o, g=O.NormL1(1.0), tol=0.0, maxit=K, **kw)
            x, it = getattr(apx, name)(np.zeros(1000), f=f_d, g=apx.NormL1(1.0), tol=0.0, maxit=K, **kw)
            assert it == K and (np.max(np.abs(x - xo)) <= 1e-9 * max(1.0, np.max(np.abs(xo))))


@pytest.mark.parametrize("m,n,logistic", [(100, 300, False), (4000, 1000, False), (900, 70, True)])
def test_backtracking_nesterov_small_shapes(apx, m, n, logistic):
    if logistic:
        lr = P.random_logreg(m, n, seed=4, dtype=np.float64)
        A, b, lam, gam, nx = lr.X, lr.y, lr.lam, 4 * lr.gamma0, n + 1
        mkf = lambda M: M.LogisticLoss(A, b)      # noqa: E731
    else:
        rng = np.random.default_rng(m + n)
        A, b, lam, nx = rng.uniform(-1, 1, (m, n)), rng.standard_normal(m), 0.3, n
        gam = 6.0 / P.opnorm_sq_power(A, iters=30)
        mkf = lambda M: M.LinearLeastSquares(A, b)      # noqa: E731
    K = 30
    of, df = O.Counting(mkf(O)), apx.Counting(mkf(apx))
    og, dg = O.Counting(O.NormL1(lam)), apx.Counting(apx.NormL1(lam))
    ro, rd = [], []
    zo, ito = O.backtracking_nesterov(np.zeros(nx), f=of, g=og, gamma0=gam, tol=0.0, maxit=K, record=ro.append)
    z, it = apx.backtracking_nesterov(np.zeros(nx), f=df, g=dg, gamma0=gam, tol=0.0, maxit=K, record=rd.append)
    assert it == ito == K and _rel_inf(z, zo) <= 1e-10
    _compare(ro, rd, ("gamma", "norm_res", "objective"), 1e-8)
    assert (df.eval_count, df.grad_count, dg.prox_count) == (of.eval_count, of.grad_count, og.prox_count)
    assert logistic or any(r["gamma"] < gam for r in rd)          # (least squares: the linesearch did backtrack)
    # termination at a known iteration, and maxit edge cases
    tol = min(r["norm_res"] for r in ro[:20]) * (1 + 1e-7)
    of, df = O.Counting(mkf(O)), apx.Counting(mkf(apx))
    zo, ito = O.backtracking_nesterov(np.zeros(nx), f=of, g=O.NormL1(lam), gamma0=gam, tol=tol, maxit=500)
    z, it = apx.backtracking_nesterov(np.zeros(nx), f=df, g=apx.NormL1(lam), gamma0=gam, tol=tol, maxit=500)
    assert it == ito <= 20 and _rel_inf(z, zo) <= 1e-10
    assert (df.eval_count, df.grad_count) == (of.eval_count, of.grad_count)
    for K2 in (0, 1, 2):
        zo, _ = O.backtracking_nesterov(np.zeros(nx), f=mkf(O), g=O.NormL1(lam), gamma0=gam, tol=0.0, maxit=K2)
        z, it = apx.backtracking_nesterov(np.zeros(nx), f=mkf(apx), g=apx.NormL1(lam), gamma0=gam, tol=0.0, maxit=K2)
        assert it == K2 and np.max(np.abs(z - zo)) <= 1e-10 * max(1.0, np.max(np.abs(zo)))


# ---------------------------------------------------------------------------------------
# the backtracking methods on tall, skinny A (csrc/skinny_bt.hip: n <= 124, m >= 2048): the shapes of the
# reference's logistic datasets (experiments/sparse_logreg/runme.jl:177-204)
# ---------------------------------------------------------------------------------------
def _tall_problem(M, m, n, logistic):
    if logistic:
        lr = P.random_logreg(m, n, seed=6, dtype=np.float64)
        return (lambda: M.LogisticLoss(lr.X, lr.y)), lr.lam, lr.gamma0, n + 1
    rng = np.random.default_rng(m + n)
    A, b = rng.uniform(-1, 1, (m, n)) * np.logspace(0, -1, n), rng.standard_normal(m)
    return (lambda: M.LinearLeastSquares(A, b)), 0.3, 1.0 / P.opnorm_sq_power(A, iters=30), n


@pytest.mark.parametrize("m,n,logistic", [(8124, 112, True), (6414, 123, True), (11055, 68, True), (5000, 30, False),
                                          (2048, 3, False)])
def test_skinny_backtracking_methods(apx, m, n, logistic):
    mko, lam, gam, nx = _tall_problem(O, m, n, logistic)
    mkd = _tall_problem(apx, m, n, logistic)[0]
    # (these problems contract fast; past ~15 iterations — 5 with three unknowns — `f_z > ub_z` compares rounding errors)
    K = 12 if n >= 10 else 5
    for xi in (1.0, 2.0):
        of, og = O.Counting(mko()), O.Counting(O.NormL1(lam))
        df, dg = apx.Counting(mkd()), apx.Counting(apx.NormL1(lam))
        ro, rd = [], []
        zo, ito = O.backtracking_proxgrad(np.zeros(nx), f=of, g=og, gamma0=4 * gam, xi=xi, tol=0.0, maxit=K, record=ro.append)
        z, it = apx.backtracking_proxgrad(np.zeros(nx), f=df, g=dg, gamma0=4 * gam, xi=xi, tol=0.0, maxit=K, record=rd.append)
        assert it == ito == K and _rel_inf(z, zo) <= 1e-10
        _compare(ro, rd, ("gamma", "norm_res", "objective"), 1e-8)
        assert (df.eval_count, df.grad_count, dg.prox_count) == (of.eval_count, of.grad_count, og.prox_count)
    of, og = O.Counting(mko()), O.Counting(O.NormL1(lam))
    df, dg = apx.Counting(mkd()), apx.Counting(apx.NormL1(lam))
    ro, rd = [], []
    zo, ito = O.backtracking_nesterov(np.zeros(nx), f=of, g=og, gamma0=8 * gam, tol=0.0, maxit=K, record=ro.append)
    z, it = apx.backtracking_nesterov(np.zeros(nx), f=df, g=dg, gamma0=8 * gam, tol=0.0, maxit=K, record=rd.append)
    assert it == ito == K and _rel_inf(z, zo) <= 1e-10
    _compare(ro, rd, ("gamma", "norm_res", "objective"), 1e-8)
    assert (df.eval_count, df.grad_count, dg.prox_count) == (of.eval_count, of.grad_count, og.prox_count)


def test_skinny_backtracking_termination_maxit_fp32(apx):
    mko, lam, gam, nx = _tall_problem(O, 4000, 40, True)
    mkd = _tall_problem(apx, 4000, 40, True)[0]
    for name, kw in (("backtracking_proxgrad", dict(gamma0=4 * gam, xi=1.5)), ("backtracking_nesterov", dict(gamma0=4 * gam))):
        ro = []
        getattr(O, name)(np.zeros(nx), f=mko(), g=O.NormL1(lam), tol=0.0, maxit=12, record=ro.append, **kw)
        tol = ro[7]["norm_res"] * (1 + 1e-7)
        kstar = 1 + next(k for k, r in enumerate(ro) if r["norm_res"] <= tol)
        of, df = O.Counting(mko()), apx.Counting(mkd())
        zo, ito = getattr(O, name)(np.zeros(nx), f=of, g=O.NormL1(lam), tol=tol, maxit=5000, **kw)
        z, it = getattr(apx, name)(np.zeros(nx), f=df, g=apx.NormL1(lam), tol=tol, maxit=5000, **kw)
        assert it == ito == kstar and _rel_inf(z, zo) <= 1e-10
        assert (df.eval_count, df.grad_count) == (of.eval_count, of.grad_count)
        for K in (0, 1, 2, 9):
            zo, ito = getattr(O, name)(np.zeros(nx), f=mko(), g=O.NormL1(lam), tol=0.0, maxit=K, **kw)
            z, it = getattr(apx, name)(np.zeros(nx), f=mkd(), g=apx.NormL1(lam), tol=0.0, maxit=K, **kw)
            assert it == ito == K and _rel_inf(z, zo) <= 1e-10
    # fp32 storage: same loop, looser agreement
    lr = P.random_logreg(4000, 40, seed=6, dtype=np.float32)
    zo, _ = O.backtracking_proxgrad(np.zeros(41), f=O.LogisticLoss(lr.X.astype(np.float64), lr.y.astype(np.float64)),
                                    g=O.NormL1(lr.lam), gamma0=4 * lr.gamma0, xi=1.5, tol=0.0, maxit=10)
    z, _ = apx.backtracking_proxgrad(np.zeros(41, dtype=np.float32), f=apx.LogisticLoss(lr.X, lr.y), g=apx.NormL1(lr.lam),
                                     gamma0=4 * lr.gamma0, xi=1.5, tol=0.0, maxit=10)
    assert z.dtype == np.float32 and _rel_inf(z, zo) <= 5e-3


@pytest.mark.parametrize("m,n,logistic", [(8124, 112, True), (6414, 123, True), (11055, 68, True), (5000, 30, False),
                                          (2048, 3, False)])
def test_skinny_mom_shapes(apx, m, n, logistic):
    """fixed_nesterov and agraal on tall, skinny A (csrc/skinny_mom.hip)."""
    mko, lam, gam, nx = _tall_problem(O, m, n, logistic)
    mkd = _tall_problem(apx, m, n, logistic)[0]
    K = 14 if n >= 10 else 6
    for kw in (dict(gamma=gam), dict(gamma=gam, muf=0.05 / gam, mug=0.0)):
        of, df = O.Counting(mko()), apx.Counting(mkd())
        ro, rd = [], []
        xo, ito = O.fixed_nesterov(np.zeros(nx), f=of, g=O.NormL1(lam), tol=0.0, maxit=K, record=ro.append, **kw)
        x, it = apx.fixed_nesterov(np.zeros(nx), f=df, g=apx.NormL1(lam), tol=0.0, maxit=K, record=rd.append, **kw)
        assert it == ito == K and _rel_inf(x, xo) <= 1e-10
        _compare(ro, rd, ("gamma", "norm_res", "objective"), 1e-8)
        assert (df.eval_count, df.grad_count) == (of.eval_count, of.grad_count) == (K, K)
        # without a record sink there is no objective pass: one tick per iteration, same iterate
        x2, _ = apx.fixed_nesterov(np.zeros(nx), f=mkd(), g=apx.NormL1(lam), tol=0.0, maxit=K, **kw)
        assert np.array_equal(x2, x)
    x0 = 0.01 * np.cos(np.arange(nx))
    for g0 in (gam, None):
        of, df = O.Counting(mko()), apx.Counting(mkd())
        ro, rd = [], []
        xo, ito = O.agraal(np.zeros(nx), f=of, g=O.NormL1(lam), x0=x0, gamma0=g0, tol=0.0, maxit=K, record=ro.append)
        x, it = apx.agraal(np.zeros(nx), f=df, g=apx.NormL1(lam), x0=x0, gamma0=g0, tol=0.0, maxit=K, record=rd.append)
        assert it == ito == K and _rel_inf(x, xo) <= 1e-9
        _compare(ro, rd, ("gamma", "norm_res", "objective"), 1e-7)
        assert (df.eval_count, df.grad_count) == (of.eval_count, of.grad_count)


def test_skinny_mom_termination_and_maxit(apx):
    mko, lam, gam, nx = _tall_problem(O, 4000, 40, True)
    mkd = _tall_problem(apx, 4000, 40, True)[0]
    x0 = 0.01 * np.cos(np.arange(nx))
    for name, kw in (("fixed_nesterov", dict(gamma=gam)), ("agraal", dict(x0=x0, gamma0=gam))):
        ro = []
        getattr(O, name)(np.zeros(nx), f=mko(), g=O.NormL1(lam), tol=0.0, maxit=14, record=ro.append, **kw)
        tol = min(r["norm_res"] for r in ro[:10]) * (1 + 1e-7)
        for with_records in (False, True):
            of, df = O.Counting(mko()), apx.Counting(mkd())
            rd = []
            xo, ito = getattr(O, name)(np.zeros(nx), f=of, g=O.NormL1(lam), tol=tol, maxit=5000, **kw)
            x, it = getattr(apx, name)(np.zeros(nx), f=df, g=apx.NormL1(lam), tol=tol, maxit=5000,
                                       record=rd.append if with_records else None, **kw)
            assert it == ito <= 10 and _rel_inf(x, xo) <= 1e-10 and len(rd) == (it if with_records else 0)
            assert (df.eval_count, df.grad_count) == (of.eval_count, of.grad_count)
        for K in (0, 1, 2, 9):
            xo, _ = getattr(O, name)(np.zeros(nx), f=mko(), g=O.NormL1(lam), tol=0.0, maxit=K, **kw)
            rd = []
            x, it = getattr(apx, name)(np.zeros(nx), f=mkd(), g=apx.NormL1(lam), tol=0.0, maxit=K, record=rd.append, **kw)
            assert it == K and len(rd) == K and _rel_inf(x, xo) <= 1e-10
