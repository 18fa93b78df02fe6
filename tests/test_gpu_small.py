"""The two-kernel small-problem iteration (csrc/small.hip, default for dense proximal-gradient
problems with m, n <= 4096) against the oracle and against the generic kernels: the reference's
own lasso sizes (experiments/lasso/runme.jl:191-195), every rule, box / l1 / translated prox,
logistic fp64 and fp32, termination, records, counters, the stepping interface and the size
limits.  Tolerance: rtol 1e-10 (fp64) inside the drift window DESIGN.md §2 documents."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import adaprox_oracle as O
from oracle import problems as P

RTOL = 1e-10


@pytest.fixture(scope="module")
def apx():
    import adaprox_amd as apx
    apx.load()
    return apx


@pytest.fixture(scope="module")
def generic_ctx(apx):
    ctx = apx.Context(0)
    ctx.fused_small(False)
    return ctx


def _rel(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(1e-300, float(np.max(np.abs(b)))))


@pytest.mark.parametrize("m,n,K", [(100, 300, 60), (500, 1000, 100), (4000, 1000, 100), (64, 17, 15), (1, 5, 10),
                                   (7, 1, 10)])
def test_lasso_reference_sizes_vs_oracle_and_generic(apx, generic_ctx, m, n, K):
    prob = P.random_lasso(m, n, pfactor=10, seed=0) if n >= 10 else None
    if prob is None:
        rng = np.random.default_rng(5)
        A, b, lam = rng.uniform(-1, 1, (m, n)), rng.uniform(-1, 1, m), 0.1
    else:
        A, b, lam = prob.A, prob.b, prob.lam
    gam = 1.0 / max(1e-12, P.opnorm_sq_power(A, iters=30))
    ro, rf, rg = [], [], []
    xo, _ = O.adaptive_proxgrad(np.zeros(n), f=O.LinearLeastSquares(A, b), g=O.NormL1(lam), rule=O.OurRule(gamma=gam),
                                tol=0.0, maxit=K, record=ro.append)
    xf, itf = apx.adaptive_proxgrad(np.zeros(n), f=apx.LinearLeastSquares(A, b), g=apx.NormL1(lam),
                                    rule=apx.OurRule(gamma=gam), tol=0.0, maxit=K, record=rf.append)
    xg, itg = apx.adaptive_proxgrad(np.zeros(n), f=apx.LinearLeastSquares(apx.DeviceMatrix.from_numpy(A, generic_ctx), b),
                                    g=apx.NormL1(lam), rule=apx.OurRule(gamma=gam), tol=0.0, maxit=K, ctx=generic_ctx,
                                    record=rg.append)
    if min(m, n) == 1:
        # an exactly solved tiny problem reaches norm_res == 0 <= tol = 0 and stops early, in the oracle
        # too; WHEN the residual rounds to exactly zero depends on the last bits of the gradient
        assert itf <= len(ro) + 2 and itg <= len(ro) + 2 and len(rf) == itf
    else:
        assert itf == itg == len(ro) == K and len(rf) == len(rg) == K
    assert _rel(xf, xo) <= RTOL and _rel(xf, xg) <= RTOL
    assert _rel(xg, xo) <= RTOL                             # the generic kernels against the oracle directly, not via xf
    for a, b_ in zip(rg, [] if min(m, n) == 1 else ro):
        assert a["gamma"] == pytest.approx(b_["gamma"], rel=1e-8)
        assert a["objective"] == pytest.approx(b_["objective"], rel=1e-10)
    for a, b_ in zip(rf, [] if min(m, n) == 1 else ro):     # (n = 1: the rule's ratios turn into 0/0 noise)
        assert a["it"] == b_["it"]
        assert a["gamma"] == pytest.approx(b_["gamma"], rel=1e-8)
        assert a["objective"] == pytest.approx(b_["objective"], rel=1e-10)
        assert a["norm_res"] == pytest.approx(b_["norm_res"], rel=1e-7, abs=1e-13)


@pytest.mark.parametrize("rule_name", ["fixed", "mm", "ours", "ours_plus"])
def test_every_rule(apx, rule_name):
    m, n = 300, 500
    prob = P.random_lasso(m, n, pfactor=10, seed=2)
    Lf = P.opnorm_sq_power(prob.A, iters=50)
    mk = {"fixed": lambda M: M.FixedStepsize(1.0 / Lf, 1.0), "mm": lambda M: M.MalitskyMishchenkoRule(gamma=1.0 / Lf),
          "ours": lambda M: M.OurRule(gamma=1.0 / Lf), "ours_plus": lambda M: M.OurRulePlus(gamma=1.0 / Lf)}[rule_name]
    K = 25 if rule_name == "mm" else 50     # drift horizon: tests/drift_probe.py (generic kernels: 3.6e-10 at 80 too)
    xo, _ = O.adaptive_proxgrad(np.zeros(n), f=O.LinearLeastSquares(prob.A, prob.b), g=O.NormL1(1.0), rule=mk(O),
                                tol=0.0, maxit=K)
    xf, _ = apx.adaptive_proxgrad(np.zeros(n), f=apx.LinearLeastSquares(prob.A, prob.b), g=apx.NormL1(1.0), rule=mk(apx),
                                  tol=0.0, maxit=K)
    assert _rel(xf, xo) <= RTOL


def test_box_and_translated_prox(apx):
    rng = np.random.default_rng(11)
    m, n = 200, 120
    A, b = rng.uniform(-1, 1, (m, n)), rng.uniform(-1, 1, m)
    gam = 1.0 / P.opnorm_sq_power(A, iters=50)
    shift = rng.uniform(-0.2, 0.2, n)
    lb, ub = -0.05 * np.ones(n), rng.uniform(0.01, 0.1, n)
    for go, ga in ((O.IndBox(-0.05, 0.08), apx.IndBox(-0.05, 0.08)), (O.IndBox(lb, ub), apx.IndBox(lb, ub)),
                   (O.Translate(O.NormL1(0.3), shift), apx.Translate(apx.NormL1(0.3), shift)), (O.Zero(), apx.Zero())):
        xo, _ = O.adaptive_proxgrad(np.zeros(n), f=O.LinearLeastSquares(A, b), g=go, rule=O.OurRule(gamma=gam), tol=0.0,
                                    maxit=60)
        xf, _ = apx.adaptive_proxgrad(np.zeros(n), f=apx.LinearLeastSquares(A, b), g=ga, rule=apx.OurRule(gamma=gam),
                                      tol=0.0, maxit=60)
        assert _rel(xf, xo) <= RTOL


@pytest.mark.parametrize("dtype,tol_x,K", [(np.float64, 1e-10, 30), (np.float32, 2e-3, 30)])
def test_logistic(apx, generic_ctx, dtype, tol_x, K):
    lr = P.random_logreg(900, 70, seed=4, dtype=dtype)
    n = 71
    xo, _ = O.adaptive_proxgrad(np.zeros(n), f=O.LogisticLoss(lr.X.astype(np.float64), lr.y.astype(np.float64)),
                                g=O.NormL1(lr.lam), rule=O.OurRule(gamma=lr.gamma0), tol=0.0, maxit=K)
    rf = []
    xf, _ = apx.adaptive_proxgrad(np.zeros(n, dtype=dtype), f=apx.LogisticLoss(lr.X, lr.y), g=apx.NormL1(lr.lam),
                                  rule=apx.OurRule(gamma=lr.gamma0), tol=0.0, maxit=K, record=rf.append)
    xg, _ = apx.adaptive_proxgrad(np.zeros(n, dtype=dtype),
                                  f=apx.LogisticLoss(apx.DeviceMatrix.from_numpy(lr.X, generic_ctx), lr.y),
                                  g=apx.NormL1(lr.lam), rule=apx.OurRule(gamma=lr.gamma0), tol=0.0, maxit=K, ctx=generic_ctx)
    assert xf.dtype == dtype
    assert _rel(xf, xo) <= tol_x and _rel(xf, xg) <= tol_x
    assert np.isfinite(rf[-1]["objective"]) and rf[-1]["objective"] < rf[0]["objective"]


def test_termination_counters_and_tested_iterate(apx, generic_ctx):
    prob = P.random_lasso(500, 1000, pfactor=10, seed=0)
    gam = 1.0 / P.opnorm_sq_power(prob.A, iters=30)
    tol = 1e-6
    fo = O.Counting(O.LinearLeastSquares(prob.A, prob.b))
    xo, ito = O.adaptive_proxgrad(np.zeros(1000), f=fo, g=O.NormL1(1.0), rule=O.OurRule(gamma=gam), tol=tol, maxit=20000)
    ff = apx.Counting(apx.LinearLeastSquares(prob.A, prob.b))
    gf = apx.Counting(apx.NormL1(1.0))
    xf, itf = apx.adaptive_proxgrad(np.zeros(1000), f=ff, g=gf, rule=apx.OurRule(gamma=gam), tol=tol, maxit=20000,
                                    poll_every=7)
    fg = apx.Counting(apx.LinearLeastSquares(apx.DeviceMatrix.from_numpy(prob.A, generic_ctx), prob.b))
    xg, itg = apx.adaptive_proxgrad(np.zeros(1000), f=fg, g=apx.NormL1(1.0), rule=apx.OurRule(gamma=gam), tol=tol,
                                    maxit=20000, ctx=generic_ctx)
    assert ito < 20000 and abs(itf - ito) <= max(3, ito // 20) and abs(itf - itg) <= max(3, ito // 20)
    # SURVEY 3.4 identities on convergence at iteration k: f k+1, grad k+1, prox_g k
    assert ff.eval_count == itf + 1 and ff.grad_count == itf + 1 and gf.prox_count == itf
    assert _rel(xf, prob.x_star) < 1e-4 and _rel(xf, xo) < 1e-5
    obj = 0.5 * np.linalg.norm(prob.A @ xf - prob.b) ** 2 + np.sum(np.abs(xf))
    assert obj == pytest.approx(prob.optimum, rel=1e-9)


def test_stepping_interface_matches_one_shot(apx):
    prob = P.random_lasso(200, 400, pfactor=10, seed=1)
    gam = 1.0 / P.opnorm_sq_power(prob.A, iters=30)
    f, g = apx.LinearLeastSquares(prob.A, prob.b), apx.NormL1(1.0)
    x1, _ = apx.adaptive_proxgrad(np.zeros(400), f=f, g=g, rule=apx.OurRule(gamma=gam), tol=0.0, maxit=90)
    x30, _ = apx.adaptive_proxgrad(np.zeros(400), f=f, g=g, rule=apx.OurRule(gamma=gam), tol=0.0, maxit=30)
    for ug in (False, True):
        s = apx.Solver(np.zeros(400), f=f, g=g, rule=apx.OurRule(gamma=gam), tol=0.0, use_graph=ug)
        s.step(30)
        xa, _, ra = s.result()          # reading the staged iterate must not disturb the run
        s.step(45)
        s.step(15)
        xb, _, rb = s.result()
        s.close()
        assert np.array_equal(xa, x30) and np.array_equal(xb, x1)
        assert ra.numit == 30 and rb.numit == 90 and not rb.converged


def test_size_limits_fall_back_to_generic_kernels(apx):
    rng = np.random.default_rng(3)
    for m, n in ((4096, 64), (64, 4096), (4097, 40), (40, 4097)):
        A, b = rng.uniform(-1, 1, (m, n)), rng.uniform(-1, 1, m)
        gam = 1.0 / P.opnorm_sq_power(A, iters=30)
        K = 14 if m > n else 40            # (the tall ones are at rounding level after ~20 iterations)
        xo, _ = O.adaptive_proxgrad(np.zeros(n), f=O.LinearLeastSquares(A, b), g=O.NormL1(0.5), rule=O.OurRule(gamma=gam),
                                    tol=0.0, maxit=K)
        xf, _ = apx.adaptive_proxgrad(np.zeros(n), f=apx.LinearLeastSquares(A, b), g=apx.NormL1(0.5),
                                      rule=apx.OurRule(gamma=gam), tol=0.0, maxit=K)
        assert _rel(xf, xo) <= RTOL


def test_domain_error_and_nan_flags_survive(apx):
    """OurRule with norm_A > 0 and a huge gamma: sqrt of a negative number inside the rule (Julia: DomainError)."""
    rng = np.random.default_rng(8)
    A, b = rng.uniform(-1, 1, (50, 30)), rng.uniform(-1, 1, 50)
    with pytest.raises(apx.DomainError):
        apx.adaptive_proxgrad(np.zeros(30), f=apx.LinearLeastSquares(A, b), g=apx.NormL1(0.1),
                              rule=apx.OurRule(gamma=10.0, norm_A=5.0, t=1.0), tol=0.0, maxit=5)
    with pytest.raises(ValueError):
        O.adaptive_proxgrad(np.zeros(30), f=O.LinearLeastSquares(A, b), g=O.NormL1(0.1),
                            rule=O.OurRule(gamma=10.0, norm_A=5.0, t=1.0), tol=0.0, maxit=5)


# ---------------------------------------------------------------------------------------
# primal-dual loop (csrc/small_pd.hip): f = Zero, dense A, the reference's LAD / sqrt-lasso shapes
# ---------------------------------------------------------------------------------------
def _lad_terms(M, lad, hkind):
    h = M.Translate(M.NormL1(1.0) if hkind == "l1" else M.NormL2(1.0), -lad.y)
    return dict(f=M.Zero(), g=M.NormL1(lad.lam), h=h)


@pytest.mark.parametrize("m,n,lam,hkind,K", [(506, 14, 10.0, "l1", 150), (400, 150, 5.0, "l2", 100), (300, 1200, 2.0, "l1", 80),
                                             (1500, 40, 1.0, "l2", 80), (3, 2, 0.1, "l1", 20)])
def test_pd_condat_vu_and_adaptive_vs_oracle_and_generic(apx, generic_ctx, m, n, lam, hkind, K):
    lad = P.random_lad(m, n, seed=1, lam=lam)
    Ag = apx.DeviceMatrix.from_numpy(lad.A, generic_ctx)
    x0, y0 = np.zeros(n), np.zeros(m)
    ro, rf = [], []
    xo, yo, ito = O.condat_vu(x0, y0, A=lad.A, Lf=0.0, norm_A=lad.norm_A, tol=0.0, maxit=K, record=ro.append,
                              **_lad_terms(O, lad, hkind))
    xf, yf, itf = apx.condat_vu(x0, y0, A=lad.A, Lf=0.0, norm_A=lad.norm_A, tol=0.0, maxit=K, record=rf.append,
                                **_lad_terms(apx, lad, hkind))
    xg, yg, itg = apx.condat_vu(x0, y0, A=Ag, Lf=0.0, norm_A=lad.norm_A, tol=0.0, maxit=K, ctx=generic_ctx,
                                **_lad_terms(apx, lad, hkind))
    assert ito == itf == itg == K and len(rf) == K
    assert _rel(xf, xo) <= RTOL and _rel(yf, yo) <= RTOL and _rel(xf, xg) <= RTOL and _rel(yf, yg) <= RTOL
    for a, b_ in zip(rf, ro):
        assert a["gamma"] == pytest.approx(b_["gamma"], rel=1e-12) and a["sigma"] == pytest.approx(b_["sigma"], rel=1e-12)
        assert a["objective"] == pytest.approx(b_["objective"], rel=1e-9)
        assert a["norm_res"] == pytest.approx(b_["norm_res"], rel=1e-7, abs=1e-12)
    # the adaptive rule with a linear operator (norm_A > 0 enters xi and the second candidate of the min)
    Ka = min(K, 60)
    xo, yo, _ = O.adaptive_primal_dual(x0, y0, A=lad.A, rule=O.OurRule(t=0.8, norm_A=lad.norm_A), tol=0.0, maxit=Ka,
                                       **_lad_terms(O, lad, hkind))
    xf, yf, _ = apx.adaptive_primal_dual(x0, y0, A=lad.A, rule=apx.OurRule(t=0.8, norm_A=lad.norm_A), tol=0.0, maxit=Ka,
                                         **_lad_terms(apx, lad, hkind))
    assert _rel(xf, xo) <= 1e-9 and _rel(yf, yo) <= 1e-9


def test_pd_termination_returns_tested_x_and_updated_y(apx, generic_ctx):
    lad = P.random_lad(506, 14, seed=3, lam=10.0)
    x0, y0 = np.zeros(14), np.zeros(506)
    tol = 1e-3
    xo, yo, ito = O.condat_vu(x0, y0, A=lad.A, Lf=0.0, norm_A=lad.norm_A, tol=tol, maxit=200000, **_lad_terms(O, lad, "l1"))
    cA = apx.Counting(apx.DeviceMatrix.from_numpy(lad.A))
    terms = _lad_terms(apx, lad, "l1")
    terms["h"] = apx.Counting(terms["h"])
    for poll in (1, 16, 1000):
        xf, yf, itf = apx.condat_vu(x0, y0, A=cA, Lf=0.0, norm_A=lad.norm_A, tol=tol, maxit=200000, poll_every=poll,
                                    **terms)
        assert ito < 200000 and itf == ito
        assert _rel(xf, xo) <= 1e-8 and _rel(yf, yo) <= 1e-8
    # converged at k: A k+1, A' k, prox_h k  (three runs accumulated)
    assert cA.mul_count == 3 * (ito + 1) and cA.amul_count == 3 * ito and terms["h"].prox_count == 3 * ito


def test_pd_stepping_and_fp32(apx):
    lad = P.random_lad(400, 150, seed=5, lam=5.0)
    terms = _lad_terms(apx, lad, "l2")
    rule = apx.OurRule(t=1.0, norm_A=lad.norm_A)
    x1, y1, _ = apx.adaptive_primal_dual(np.zeros(150), np.zeros(400), A=lad.A, rule=rule, tol=0.0, maxit=77, **terms)
    s = apx.Solver(np.zeros(150), np.zeros(400), A=lad.A, rule=rule, tol=0.0, **terms)
    for k in (1, 2, 3, 30, 41):          # every residue of the 6-cycle of buffer roles
        s.step(k)
        s.result()
    xs, ys, res = s.result()
    s.close()
    assert res.numit == 77 and np.array_equal(xs, x1) and np.array_equal(ys, y1)
    # fp32 end to end
    A32, y32 = lad.A.astype(np.float32), lad.y.astype(np.float32)
    t32 = dict(f=apx.Zero(), g=apx.NormL1(lad.lam), h=apx.Translate(apx.NormL1(1.0), -y32))
    xo, yo, _ = O.condat_vu(np.zeros(150), np.zeros(400), A=lad.A, Lf=0.0, norm_A=lad.norm_A, tol=0.0, maxit=40,
                            **_lad_terms(O, lad, "l1"))
    xf, yf, _ = apx.condat_vu(np.zeros(150, dtype=np.float32), np.zeros(400, dtype=np.float32), A=A32, Lf=0.0,
                              norm_A=lad.norm_A, tol=0.0, maxit=40, **t32)
    assert xf.dtype == np.float32 and _rel(xf, xo) <= 5e-3 and _rel(yf, yo) <= 5e-3


def test_pd_dual_svm_quadratic(apx, generic_ctx):
    """experiments/dual_svm/runme.jl:47-76 on the two-kernel path: f = Quadratic(Q, q) (the third product Q x
    rides in kernel A), g = IndBox(0, C), h = IndZero, A = y'."""
    rng = np.random.default_rng(0)
    N, dfeat, C = 700, 25, 0.1
    X = rng.uniform(-1, 1, (N, dfeat))
    yl = np.sign(X @ rng.standard_normal(dfeat) + 0.1 * rng.standard_normal(N))
    Q = np.asfortranarray((yl[:, None] * X) @ (yl[:, None] * X).T)
    Q[3, 5] += 1e-3                                  # not symmetric: rows of Q, not its columns, must be used
    q = -np.ones(N)
    A = np.asfortranarray(yl[None, :])
    nrm = float(np.linalg.norm(A))
    for t in (0.05, 1.0, 5.0):
        K = 120
        ro, rf = [], []
        fo, ff = O.Counting(O.Quadratic(Q, q)), apx.Counting(apx.Quadratic(Q, q))
        xo, yo, _ = O.adaptive_primal_dual(np.zeros(N), np.zeros(1), f=fo, g=O.IndBox(0.0, C), h=O.IndZero(), A=A,
                                           rule=O.OurRule(t=t, norm_A=nrm), maxit=K, tol=0.0, record=ro.append)
        xf, yf, _ = apx.adaptive_primal_dual(np.zeros(N), np.zeros(1), f=ff, g=apx.IndBox(0.0, C), h=apx.IndZero(), A=A,
                                             rule=apx.OurRule(t=t, norm_A=nrm), maxit=K, tol=0.0, record=rf.append)
        fg = apx.Quadratic(apx.DeviceMatrix.from_numpy(Q, generic_ctx), q, ctx=generic_ctx)
        xg, yg, _ = apx.adaptive_primal_dual(np.zeros(N), np.zeros(1), f=fg, g=apx.IndBox(0.0, C), h=apx.IndZero(),
                                             A=apx.DeviceMatrix.from_numpy(A, generic_ctx), ctx=generic_ctx,
                                             rule=apx.OurRule(t=t, norm_A=nrm), maxit=K, tol=0.0)
        assert _rel(xf, xo) <= RTOL and _rel(yf, yo) <= 1e-9 and _rel(xf, xg) <= RTOL
        assert (ff.eval_count, ff.grad_count) == (fo.eval_count, fo.grad_count)
        for a, b_ in zip(rf, ro):
            assert a["gamma"] == pytest.approx(b_["gamma"], rel=1e-8)
            assert a["objective"] == pytest.approx(b_["objective"], rel=1e-10, abs=1e-12)


# ---------------------------------------------------------------------------------------
# tall matrices (4096 < m <= 8192; the larger shapes below fall back to the generic kernels): the m-vector in
# dynamic LDS, m-side operands re-read per pass —
# the shapes of the reference's own datasets: cpusmall_scale 8192 x 12(+1), abalone 4177 x 8(+1) (LAD,
# square-root lasso), mushrooms 8124 x 112, phishing 11055 x 68 (logistic regression, loaded dense)
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("m,n,hkind", [(8192, 13, "l1"), (8192, 13, "l2"), (4177, 9, "l2"), (16384, 40, "l1")])
def test_pd_tall_dataset_shapes(apx, generic_ctx, m, n, hkind):
    lad = P.random_lad(m, n, seed=2, lam=10.0)
    Ag = apx.DeviceMatrix.from_numpy(lad.A, generic_ctx)
    x0, y0, K = np.zeros(n), np.zeros(m), 80
    xo, yo, _ = O.condat_vu(x0, y0, A=lad.A, Lf=0.0, norm_A=lad.norm_A, tol=0.0, maxit=K, **_lad_terms(O, lad, hkind))
    xf, yf, _ = apx.condat_vu(x0, y0, A=lad.A, Lf=0.0, norm_A=lad.norm_A, tol=0.0, maxit=K, **_lad_terms(apx, lad, hkind))
    xg, yg, _ = apx.condat_vu(x0, y0, A=Ag, Lf=0.0, norm_A=lad.norm_A, tol=0.0, maxit=K, ctx=generic_ctx,
                              **_lad_terms(apx, lad, hkind))
    assert _rel(xf, xo) <= RTOL and _rel(yf, yo) <= RTOL and _rel(xf, xg) <= RTOL and _rel(yf, yg) <= RTOL
    xo, yo, _ = O.adaptive_primal_dual(x0, y0, A=lad.A, rule=O.OurRule(t=0.8, norm_A=lad.norm_A), tol=0.0, maxit=50,
                                       **_lad_terms(O, lad, hkind))
    xf, yf, _ = apx.adaptive_primal_dual(x0, y0, A=lad.A, rule=apx.OurRule(t=0.8, norm_A=lad.norm_A), tol=0.0, maxit=50,
                                         **_lad_terms(apx, lad, hkind))
    assert _rel(xf, xo) <= 1e-9 and _rel(yf, yo) <= 1e-9


@pytest.mark.parametrize("m,n,logistic", [(8124, 112, True), (11055, 68, True), (6414, 123, False), (16384, 200, False)])
def test_proxgrad_tall_dataset_shapes(apx, generic_ctx, m, n, logistic):
    rng = np.random.default_rng(m + n)
    if logistic:
        lr = P.random_logreg(m, n, seed=3, dtype=np.float64)
        A, b, lam, gam, nx = lr.X, lr.y, lr.lam, lr.gamma0, n + 1
        fo, mkf = O.LogisticLoss(A, b), apx.LogisticLoss
    else:
        A, b, lam, nx = rng.uniform(-1, 1, (m, n)), rng.standard_normal(m), 0.5, n
        gam = 1.0 / P.opnorm_sq_power(A, iters=30)
        fo, mkf = O.LinearLeastSquares(A, b), apx.LinearLeastSquares
    K = 30
    xo, _ = O.adaptive_proxgrad(np.zeros(nx), f=fo, g=O.NormL1(lam), rule=O.OurRule(gamma=gam), tol=0.0, maxit=K)
    rf = []
    xf, _ = apx.adaptive_proxgrad(np.zeros(nx), f=mkf(A, b), g=apx.NormL1(lam), rule=apx.OurRule(gamma=gam), tol=0.0,
                                  maxit=K, record=rf.append)
    xg, _ = apx.adaptive_proxgrad(np.zeros(nx), f=mkf(apx.DeviceMatrix.from_numpy(A, generic_ctx), b, ctx=generic_ctx),
                                  g=apx.NormL1(lam), rule=apx.OurRule(gamma=gam), tol=0.0, maxit=K, ctx=generic_ctx)
    assert _rel(xf, xo) <= RTOL and _rel(xf, xg) <= RTOL
    assert len(rf) == K and rf[-1]["objective"] < rf[0]["objective"]


# ---------------------------------------------------------------------------------------
# tall, skinny A (csrc/skinny_pd.hip: n <= 120, m >= 4 n): one thread per row, a closing block
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("m,n,hkind", [(1024, 14, "l1"), (3072, 14, "l2"), (1025, 1, "l1"), (3500, 120, "l2"), (40000, 30, "l1"),
                                       (262144, 6, "l2"), (9000, 120, "l1")])
def test_skinny_pd_shapes(apx, generic_ctx, m, n, hkind):
    # (lam = 5 at 262144 x 6 with the l2 ball has the solution x = 0: the residual is at rounding level after five
    #  iterations and reaches 0.0 exactly a few later, whenever the summation order lets it)
    lad = P.random_lad(m, n, seed=4, lam=0.05 if (m, hkind) == (262144, "l2") else 5.0)
    Ag = apx.DeviceMatrix.from_numpy(lad.A, generic_ctx)
    x0, y0, K = np.zeros(n), np.zeros(m), 60
    rf, rg = [], []
    xf, yf, itf = apx.adaptive_primal_dual(x0, y0, A=lad.A, rule=apx.OurRule(t=1.0, norm_A=lad.norm_A), tol=0.0, maxit=K,
                                           record=rf.append, **_lad_terms(apx, lad, hkind))
    xg, yg, itg = apx.adaptive_primal_dual(x0, y0, A=Ag, rule=apx.OurRule(t=1.0, norm_A=lad.norm_A), tol=0.0, maxit=K,
                                           ctx=generic_ctx, record=rg.append, **_lad_terms(apx, lad, hkind))
    assert itf == itg == K and len(rf) == K
    assert _rel(xf, xg) <= 1e-9 and _rel(yf, yg) <= 1e-9
    for a, b_ in zip(rf, rg):
        assert a["gamma"] == pytest.approx(b_["gamma"], rel=1e-8) and a["sigma"] == pytest.approx(b_["sigma"], rel=1e-8)
        assert a["norm_res"] == pytest.approx(b_["norm_res"], rel=1e-6, abs=1e-10)
        assert a["objective"] == pytest.approx(b_["objective"], rel=1e-9)
    if m <= 50000:
        xo, yo, _ = O.adaptive_primal_dual(x0, y0, A=lad.A, rule=O.OurRule(t=1.0, norm_A=lad.norm_A), tol=0.0, maxit=K,
                                           **_lad_terms(O, lad, hkind))
        assert _rel(xf, xo) <= 1e-9 and _rel(yf, yo) <= 1e-9


def test_skinny_pd_graph_stepping_fp32_and_termination(apx):
    lad = P.random_lad(3200, 11, seed=6, lam=5.0)
    terms = _lad_terms(apx, lad, "l2")
    rule = apx.OurRule(t=1.0, norm_A=lad.norm_A)
    x1, y1, _ = apx.adaptive_primal_dual(np.zeros(11), np.zeros(3200), A=lad.A, rule=rule, tol=0.0, maxit=70, **terms)
    x2, y2, _ = apx.adaptive_primal_dual(np.zeros(11), np.zeros(3200), A=lad.A, rule=rule, tol=0.0, maxit=70,
                                         use_graph=True, **terms)
    assert np.array_equal(x1, x2) and np.array_equal(y1, y2)            # fixed pointers: replayable, identical kernels
    s = apx.Solver(np.zeros(11), np.zeros(3200), A=lad.A, rule=rule, tol=0.0, **terms)
    for k in (1, 2, 30, 37):
        s.step(k)
        s.result()
    xs, ys, res = s.result()
    s.close()
    assert res.numit == 70 and np.array_equal(xs, x1) and np.array_equal(ys, y1)
    # termination: a tolerance that the residual first reaches at a known iteration k* of the run above;
    # the run must stop there whatever the polling period and return x_k* with the already updated y
    t1 = _lad_terms(apx, lad, "l1")
    recs = []
    apx.adaptive_primal_dual(np.zeros(11), np.zeros(3200), A=lad.A, rule=rule, tol=0.0, maxit=300, record=recs.append, **t1)
    res_k = [r["norm_res"] for r in recs]
    tol = res_k[249] * (1 + 1e-9)
    kstar = 1 + next(k for k, v in enumerate(res_k) if v <= tol)
    s = apx.Solver(np.zeros(11), np.zeros(3200), A=lad.A, rule=rule, tol=0.0, **t1)
    s.step(kstar - 1)
    x_prev_iter, _, _ = s.result()            # x_k* (the iterate that iteration k* tests)
    s.step(1)
    _, y_after, _ = s.result()                # y after the dual update of iteration k*
    s.close()
    for poll in (1, 64):
        xf, yf, itf = apx.adaptive_primal_dual(np.zeros(11), np.zeros(3200), A=lad.A, rule=rule, tol=tol, maxit=5000,
                                               poll_every=poll, **t1)
        assert itf == kstar <= 250 and np.array_equal(xf, x_prev_iter) and np.array_equal(yf, y_after)
    # fp32
    A32, y32 = lad.A.astype(np.float32), lad.y.astype(np.float32)
    xf, yf, _ = apx.condat_vu(np.zeros(11, dtype=np.float32), np.zeros(3200, dtype=np.float32), A=A32, Lf=0.0,
                              norm_A=lad.norm_A, tol=0.0, maxit=40, f=apx.Zero(), g=apx.NormL1(lad.lam),
                              h=apx.Translate(apx.NormL1(1.0), -y32))
    xo, yo, _ = O.condat_vu(np.zeros(11), np.zeros(3200), A=lad.A, Lf=0.0, norm_A=lad.norm_A, tol=0.0, maxit=40,
                            **_lad_terms(O, lad, "l1"))
    assert xf.dtype == np.float32 and _rel(xf, xo) <= 5e-3 and _rel(yf, yo) <= 5e-3


# ---------------------------------------------------------------------------------------
# backtracking_proxgrad with the device-resident linesearch (csrc/small_bt.hip)
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("m,n,logistic,xi", [(100, 300, False, 1.0), (500, 1000, False, 2.0), (4000, 1000, False, 1.5),
                                             (900, 70, True, 1.5), (64, 17, False, 1.1), (5, 1, False, 2.0)])
def test_backtracking_small_vs_oracle_and_generic(apx, generic_ctx, m, n, logistic, xi):
    if logistic:
        lr = P.random_logreg(m, n, seed=4, dtype=np.float64)
        A, b, lam, gam, nx = lr.X, lr.y, lr.lam, lr.gamma0, n + 1
        fo, mkf = O.LogisticLoss(A, b), apx.LogisticLoss
    else:
        rng = np.random.default_rng(m + n)
        A, b, lam, nx = rng.uniform(-1, 1, (m, n)), rng.standard_normal(m), 0.3, n
        gam = 3.0 / P.opnorm_sq_power(A, iters=30)            # too long on purpose: the linesearch has to shrink
        fo, mkf = O.LinearLeastSquares(A, b), apx.LinearLeastSquares
    # (near the solution `f_z > ub_z` compares two numbers that agree to rounding: any two fp64 evaluations
    #  then take different linesearch branches — 4000 x 1000 at K = 50: 8e-9 — so the window is short)
    K = 25
    of, og = O.Counting(fo), O.Counting(O.NormL1(lam))
    ro, rf = [], []
    zo, ito = O.backtracking_proxgrad(np.zeros(nx), f=of, g=og, gamma0=gam, xi=xi, tol=0.0, maxit=K, record=ro.append)
    ff, gf = apx.Counting(mkf(A, b)), apx.Counting(apx.NormL1(lam))
    zf, itf = apx.backtracking_proxgrad(np.zeros(nx), f=ff, g=gf, gamma0=gam, xi=xi, tol=0.0, maxit=K, record=rf.append)
    zg, itg = apx.backtracking_proxgrad(np.zeros(nx), f=mkf(apx.DeviceMatrix.from_numpy(A, generic_ctx), b, ctx=generic_ctx),
                                        g=apx.NormL1(lam), gamma0=gam, xi=xi, tol=0.0, maxit=K, ctx=generic_ctx)
    if min(m, n) == 1:          # solved exactly after a few iterations: WHEN z == x bit for bit is rounding
        assert _rel(zf, zo) <= 1e-6 and _rel(zf, zg) <= 1e-6
        return
    assert itf == ito == itg and len(rf) == len(ro)
    assert _rel(zf, zo) <= RTOL and _rel(zf, zg) <= RTOL
    for a, b_ in zip(rf, ro):
        assert a["gamma"] == pytest.approx(b_["gamma"], rel=1e-12)
        assert a["objective"] == pytest.approx(b_["objective"], rel=1e-10)
        assert a["norm_res"] == pytest.approx(b_["norm_res"], rel=1e-7, abs=1e-13)
        assert (a["f_evals"], a["grad_f_evals"], a["prox_g_evals"]) == (b_["f_evals"], b_["grad_f_evals"], b_["prox_g_evals"])
    assert (ff.eval_count, ff.grad_count, gf.prox_count) == (of.eval_count, of.grad_count, og.prox_count)
    assert of.eval_count > K + 1                              # trials were rejected along the way


def test_backtracking_small_termination_maxit_fp32(apx):
    prob = P.random_lasso(200, 400, pfactor=10, seed=3)
    gam = 2.0 / P.opnorm_sq_power(prob.A, iters=30)
    kw = dict(gamma0=gam, xi=1.5)
    of, ff = O.Counting(O.LinearLeastSquares(prob.A, prob.b)), apx.Counting(apx.LinearLeastSquares(prob.A, prob.b))
    zo, ito = O.backtracking_proxgrad(np.zeros(400), f=of, g=O.NormL1(1.0), tol=1e-6, maxit=20000, **kw)
    zf, itf = apx.backtracking_proxgrad(np.zeros(400), f=ff, g=apx.NormL1(1.0), tol=1e-6, maxit=20000, **kw)
    assert ito < 20000 and abs(itf - ito) <= max(2, ito // 50) and _rel(zf, zo) <= 1e-6
    if itf == ito:
        assert (ff.eval_count, ff.grad_count) == (of.eval_count, of.grad_count)     # converged: no pullback after the last trial
    obj = 0.5 * np.linalg.norm(prob.A @ zf - prob.b) ** 2 + np.sum(np.abs(zf))
    assert obj == pytest.approx(prob.optimum, rel=1e-8)
    for K in (0, 1, 33):
        zo, _ = O.backtracking_proxgrad(np.zeros(400), f=O.LinearLeastSquares(prob.A, prob.b), g=O.NormL1(1.0), tol=0.0,
                                        maxit=K, **kw)
        zf, itf = apx.backtracking_proxgrad(np.zeros(400), f=apx.LinearLeastSquares(prob.A, prob.b), g=apx.NormL1(1.0),
                                            tol=0.0, maxit=K, **kw)
        assert itf == K and (K == 0 and not np.any(zf) or _rel(zf, zo) <= RTOL)
    A32, b32 = prob.A.astype(np.float32), prob.b.astype(np.float32)
    zf, itf = apx.backtracking_proxgrad(np.zeros(400, dtype=np.float32), f=apx.LinearLeastSquares(A32, b32), g=apx.NormL1(1.0),
                                        tol=0.0, maxit=40, **kw)
    zo, _ = O.backtracking_proxgrad(np.zeros(400), f=O.LinearLeastSquares(prob.A, prob.b), g=O.NormL1(1.0), tol=0.0,
                                    maxit=40, **kw)
    assert zf.dtype == np.float32 and itf == 40 and _rel(zf, zo) <= 5e-3


# ---------------------------------------------------------------------------------------
# tall, skinny A in the proximal-gradient loop (csrc/skinny_pg.hip: n <= 120, m >= 2048)
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("m,n,logistic,dtype", [(8124, 112, True, np.float64), (11055, 68, True, np.float64),
                                                (2048, 119, True, np.float64), (6000, 40, False, np.float64),
                                                (100000, 7, False, np.float64), (2049, 2, False, np.float64),
                                                (262144, 4, False, np.float64), (3000, 1, False, np.float64), (700001, 20, False, np.float64),
                                                (2048, 123, True, np.float64),
                                                (8124, 112, True, np.float32)])
def test_skinny_pg_shapes(apx, generic_ctx, m, n, logistic, dtype):
    if logistic:
        lr = P.random_logreg(m, n, seed=5, dtype=dtype)
        A, b, lam, gam, nx = lr.X, lr.y, lr.lam, lr.gamma0, n + 1
        mko, mkf = (lambda: O.LogisticLoss(A.astype(np.float64), b.astype(np.float64))), apx.LogisticLoss
    else:
        rng = np.random.default_rng(m + n)
        A, b, lam, nx = rng.uniform(-1, 1, (m, n)).astype(dtype), rng.standard_normal(m).astype(dtype), 0.5, n
        gam = 1.0 / P.opnorm_sq_power(A.astype(np.float64), iters=30)
        mko, mkf = (lambda: O.LinearLeastSquares(A.astype(np.float64), b.astype(np.float64))), apx.LinearLeastSquares
    # (these tall problems are well conditioned: the residual reaches rounding level after ~20 iterations and every
    #  later step size is a ratio of rounding errors, so the compared window ends before that)
    K = 12
    tol_x = 1e-10 if dtype == np.float64 else 2e-3
    ro, rf = [], []
    xo, _ = O.adaptive_proxgrad(np.zeros(nx), f=mko(), g=O.NormL1(lam), rule=O.OurRule(gamma=gam), tol=0.0, maxit=K,
                                record=ro.append)
    ff = apx.Counting(mkf(A, b))
    xf, itf = apx.adaptive_proxgrad(np.zeros(nx, dtype=dtype), f=ff, g=apx.NormL1(lam), rule=apx.OurRule(gamma=gam), tol=0.0,
                                    maxit=K, record=rf.append)
    xg, _ = apx.adaptive_proxgrad(np.zeros(nx, dtype=dtype), f=mkf(apx.DeviceMatrix.from_numpy(A, generic_ctx), b, ctx=generic_ctx),
                                  g=apx.NormL1(lam), rule=apx.OurRule(gamma=gam), tol=0.0, maxit=K, ctx=generic_ctx)
    assert itf == K and len(rf) == K and (ff.eval_count, ff.grad_count) == (K + 1, K + 1)
    assert _rel(xf, xo) <= tol_x and _rel(xf, xg) <= tol_x
    if dtype == np.float64 and n > 1:          # (one unknown: solved at once, every later step size is 0 / 0)
        for a, b_ in zip(rf, ro):
            # (the step size is a ratio of differences of iterates: its relative accuracy is eps * |x| / |dx|, and
            #  these problems contract by 10x per iteration)
            if b_["norm_res"] <= 1e-4 * ro[0]["norm_res"]:
                break
            assert a["gamma"] == pytest.approx(b_["gamma"], rel=1e-8)
            assert a["objective"] == pytest.approx(b_["objective"], rel=1e-10)
            assert a["norm_res"] == pytest.approx(b_["norm_res"], rel=1e-7, abs=1e-13)


def test_skinny_pg_rules_graph_stepping_termination(apx):
    lr = P.random_logreg(5000, 60, seed=8, dtype=np.float64)
    f, g = apx.LogisticLoss(lr.X, lr.y), apx.NormL1(lr.lam)
    fo, go = O.LogisticLoss(lr.X, lr.y), O.NormL1(lr.lam)
    for mk in (lambda M: M.FixedStepsize(lr.gamma0, 1.0), lambda M: M.MalitskyMishchenkoRule(gamma=lr.gamma0),
               lambda M: M.OurRulePlus(gamma=lr.gamma0)):
        xo, _ = O.adaptive_proxgrad(np.zeros(61), f=fo, g=go, rule=mk(O), tol=0.0, maxit=14)
        xf, _ = apx.adaptive_proxgrad(np.zeros(61), f=f, g=g, rule=mk(apx), tol=0.0, maxit=14)
        assert _rel(xf, xo) <= RTOL
    rule = apx.OurRule(gamma=lr.gamma0)
    # (20 iterations: past ~24 the residual of this problem is at rounding level and may hit 0.0, which terminates)
    x1, _ = apx.adaptive_proxgrad(np.zeros(61), f=f, g=g, rule=rule, tol=0.0, maxit=20)
    x2, _ = apx.adaptive_proxgrad(np.zeros(61), f=f, g=g, rule=rule, tol=0.0, maxit=20, use_graph=True)
    assert np.array_equal(x1, x2)
    s = apx.Solver(np.zeros(61), f=f, g=g, rule=rule, tol=0.0)
    for k in (1, 9, 10):
        s.step(k)
        s.result()
    xs, _, res = s.result()
    s.close()
    assert res.numit == 20 and np.array_equal(xs, x1)
    # termination: tolerance reached at a known iteration
    recs = []
    apx.adaptive_proxgrad(np.zeros(61), f=f, g=g, rule=rule, tol=0.0, maxit=20, record=recs.append)
    res_k = [r["norm_res"] for r in recs]
    tol = res_k[11] * (1 + 1e-9)            # (rounding level is reached near iteration 24)
    kstar = 1 + next(k for k, v in enumerate(res_k) if v <= tol)
    xo, ito = O.adaptive_proxgrad(np.zeros(61), f=fo, g=go, rule=O.OurRule(gamma=lr.gamma0), tol=tol, maxit=1000)
    for poll in (1, 64):
        ff = apx.Counting(f)
        xf, itf = apx.adaptive_proxgrad(np.zeros(61), f=ff, g=g, rule=rule, tol=tol, maxit=1000, poll_every=poll)
        assert itf == kstar == ito and _rel(xf, xo) <= RTOL and (ff.eval_count, ff.grad_count) == (itf + 1, itf + 1)


# ---------------------------------------------------------------------------------------
# both dimensions large at once: the in-block dot products need several passes per block (more than 1024 outputs
# with four waves per dot).  small.hip skipped outputs there while its grid carried the rotation blocks behind the
# dot blocks — found by the one-pass fuzz test, which compares shapes of exactly 64 MB
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("m,n", [(3000, 3000), (2049, 2048), (4096, 4096), (4096, 1100), (2500, 4000)])
def test_small_paths_with_several_passes_per_block(apx, generic_ctx, m, n):
    rng = np.random.default_rng(m + n)
    A = np.asfortranarray(rng.uniform(-1, 1, (m, n)))
    b = rng.standard_normal(m)
    x0 = 0.01 * rng.standard_normal(n)
    gam = 1.0 / (m * n / 3.0)
    K = 6
    fo, go = O.LinearLeastSquares(A, b), O.NormL1(1.0)
    f, g = apx.LinearLeastSquares(A, b), apx.NormL1(1.0)
    for k in (0, K):
        xo, _ = O.adaptive_proxgrad(x0, f=fo, g=go, rule=O.OurRule(gamma=gam), tol=0.0, maxit=k)
        x, _ = apx.adaptive_proxgrad(x0, f=f, g=g, rule=apx.OurRule(gamma=gam), tol=0.0, maxit=k)
        assert _rel(x, xo) <= RTOL
    # the other device-resident proximal-gradient loops on the same scheme (m, n <= 4096)
    zo, _ = O.backtracking_proxgrad(x0, f=fo, g=go, gamma0=4 * gam, xi=1.5, tol=0.0, maxit=K)
    z, _ = apx.backtracking_proxgrad(x0, f=f, g=g, gamma0=4 * gam, xi=1.5, tol=0.0, maxit=K)
    assert _rel(z, zo) <= RTOL
    zo, _ = O.fixed_nesterov(x0, f=fo, g=go, gamma=gam, tol=0.0, maxit=K)
    z, _ = apx.fixed_nesterov(x0, f=f, g=g, gamma=gam, tol=0.0, maxit=K)
    assert _rel(z, zo) <= RTOL
    zo, _ = O.agraal(x0, f=fo, g=go, x0=x0 + 0.01, gamma0=gam, tol=0.0, maxit=K)
    z, _ = apx.agraal(x0, f=f, g=g, x0=x0 + 0.01, gamma0=gam, tol=0.0, maxit=K)
    assert _rel(z, zo) <= 1e-9
    # the primal-dual loop (csrc/small_pd.hip) with the same matrix as the linear operator
    yv = rng.standard_normal(m)
    nA = float(np.linalg.norm(A))
    kw = dict(Lf=0.0, norm_A=nA, tol=0.0, maxit=K)
    xo, yo, _ = O.condat_vu(np.zeros(n), np.zeros(m), f=O.Zero(), g=O.NormL1(2.0), h=O.Translate(O.NormL1(1.0), -yv), A=A, **kw)
    x, y, _ = apx.condat_vu(np.zeros(n), np.zeros(m), f=apx.Zero(), g=apx.NormL1(2.0), h=apx.Translate(apx.NormL1(1.0), -yv), A=A,
                            **kw)
    assert _rel(x, xo) <= RTOL and _rel(y, yo) <= 1e-9
