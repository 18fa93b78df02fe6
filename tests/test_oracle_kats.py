"""Pins the CPU oracle against every known answer the reference's own tests and
experiments hold for the hot path.  No GPU needed.

* test/runtests.jl:25-51  "Simple 2D problem"  (end-state KAT)
* test/runtests.jl:53-90  "Counting"           (counter semantics)
* experiments/lasso/runme.jl:77                (analytic optimum of the generator)
* experiments/nesterov_worst_case/runme.jl:53  (analytic optimum)
* SURVEY.md §3.4 evaluation-count identities derived from src/AdaProx.jl:324-363
"""
import math

import numpy as np
import pytest

from oracle import adaprox_oracle as O
from oracle import problems as P


class Simple2DObjective:
    """test/runtests.jl:6-13."""

    def eval_with_pullback(self, x):
        pb = lambda: np.array([2 * math.log(1 + x[0] ** 2) * 2 * x[0] / (1 + x[0] ** 2), 20 * x[1]])
        return math.log(1 + x[0] ** 2) ** 2 + 10 * x[1] ** 2, pb

    def __call__(self, x):
        return self.eval_with_pullback(x)[0]


class Simple2DBox:
    """test/runtests.jl:15-23."""

    def __call__(self, x):
        return 0.0 if abs(x[0]) <= 2.9 else math.inf

    def prox(self, x, gamma=1.0):
        y = np.array([min(max(x[0], -2.9), 2.9), x[1]])
        return y, 0.0


def test_simple_2d_adaptive_proxgrad():
    f, g = Simple2DObjective(), Simple2DBox()
    sol, numit = O.adaptive_proxgrad(np.ones(2), f=f, g=g, rule=O.OurRule(gamma=1.0))
    assert f(sol) < 1e-7
    assert g(sol) == 0


def test_simple_2d_backtracking_proxgrad():
    f, g = Simple2DObjective(), Simple2DBox()
    sol, numit = O.backtracking_proxgrad(np.ones(2), f=f, g=g, gamma0=1.0, xi=1.1)
    assert f(sol) < 1e-7
    assert g(sol) == 0


def test_counting():
    f = O.Counting(Simple2DObjective())
    g = O.Counting(Simple2DBox())
    A = O.Counting(np.eye(2))
    x = np.ones(2)

    _, pb = O.eval_with_pullback(f, x)
    O.prox(g, x)
    A @ x
    assert (f.eval_count, f.grad_count, g.prox_count, A.mul_count, A.amul_count) == (1, 0, 1, 1, 0)
    pb()
    assert f.grad_count == 1
    A.T @ x
    assert A.amul_count == 1

    def quiet():
        _, pb2 = O.eval_with_pullback(f, x)
        pb2()
        O.prox(g, x)
        A @ x

    O.without_counting(quiet)
    assert (f.eval_count, f.grad_count, g.prox_count, A.mul_count, A.amul_count) == (1, 1, 1, 1, 1)


@pytest.mark.parametrize("rule_name", ["ours", "mm", "fixed"])
def test_lasso_analytic_optimum(rule_name):
    prob = P.random_lasso(100, 300, pfactor=10, seed=0)
    Lf = float(np.linalg.norm(prob.A, 2) ** 2)
    gam = 1 / Lf
    f = O.LinearLeastSquares(prob.A, prob.b)
    g = O.NormL1(prob.lam)
    rule = {"ours": O.OurRule(gamma=gam), "mm": O.MalitskyMishchenkoRule(gamma=gam),
            "fixed": O.FixedStepsize(gam, 1.0)}[rule_name]
    recs = []
    sol, numit = O.adaptive_proxgrad(np.zeros(300), f=O.Counting(f), g=g, rule=rule, tol=1e-9,
                                     maxit=200_000, record=recs.append)
    obj = f(sol) + g(sol)
    assert obj == pytest.approx(prob.optimum, rel=1e-9)
    assert np.max(np.abs(sol - prob.x_star)) < 1e-6
    assert recs[-1]["it"] == numit


class WorstQuadratic:
    """experiments/nesterov_worst_case/runme.jl:14-40."""

    def __init__(self, k, L):
        self.k, self.L = k, L

    def eval_with_pullback(self, x):
        k, L = self.k, self.L
        s = x[0] ** 2 + x[k - 1] ** 2
        for i in range(k - 1):
            s += (x[i] - x[i + 1]) ** 2

        def pb():
            grad = np.zeros_like(x)
            grad[0] = (L / 4) * (2 * x[0] - x[1] - 1)
            for i in range(1, k - 1):
                grad[i] = (L / 4) * (2 * x[i] - x[i - 1] - x[i + 1])
            grad[k - 1] = (L / 4) * (2 * x[k - 1] - x[k - 2])
            return grad

        return (L / 4) * (s / 2 - x[0]), pb

    def __call__(self, x):
        return self.eval_with_pullback(x)[0]


def test_nesterov_worst_case_optimum():
    k = n = 100
    L = 100.0
    f = WorstQuadratic(k, L)
    optimum = (L / 8) * (1 / (k + 1) - 1)        # runme.jl:53
    sol, numit = O.adaptive_proxgrad(np.zeros(n), f=O.Counting(f), g=O.Zero(),
                                     rule=O.OurRule(gamma=1 / L), tol=1e-6, maxit=10_000)
    # the worst-case function exhausts maxit (as in the reference run); the objective must
    # approach the analytic optimum from above
    assert numit == 10_000
    assert f(sol) >= optimum - 1e-12
    assert f(sol) - optimum < 1e-6


def test_count_identities_pd():
    """converge at k -> evals = grads = A*x = k+1 ; A'y = prox_g = prox_h = k."""
    prob = P.random_lad(60, 20, seed=3, lam=1.0)
    f = O.Counting(O.Zero())
    g = O.Counting(O.NormL1(prob.lam))
    h = O.Counting(O.Translate(O.NormL1(1.0), -prob.y))
    A = O.Counting(prob.A)
    x, y, k = O.condat_vu(np.zeros(20), np.zeros(60), f=f, g=g, h=h, A=A, Lf=0.0,
                          norm_A=prob.norm_A, tol=1e-5, maxit=200_000)
    assert k < 200_000
    assert (f.eval_count, f.grad_count, A.mul_count) == (k + 1, k + 1, k + 1)
    # amul: one before the loop (:329) + (k-1) in-loop (:358) ; prox_g likewise (:332,:361)
    assert (A.amul_count, g.prox_count) == (k, k)
    # maxit exhaustion: one more A'y / prox_g and the returned x is the fresh iterate
    f2, g2, A2 = O.Counting(O.Zero()), O.Counting(O.NormL1(prob.lam)), O.Counting(prob.A)
    O.condat_vu(np.zeros(20), np.zeros(60), f=f2, g=g2, h=h, A=A2, Lf=0.0, norm_A=prob.norm_A,
                tol=0.0, maxit=7)
    assert (f2.eval_count, A2.mul_count, A2.amul_count, g2.prox_count) == (8, 8, 8, 8)


def test_rule_scalar_semantics():
    # Appendix A.5: MM first update is 1/(2L) because rho0 = Inf
    rule = O.MalitskyMishchenkoRule(gamma=0.3)
    (_, _), st = O.stepsize(rule)
    x1, x0 = np.array([1.0, 0.0]), np.array([0.0, 0.0])
    g1, g0 = np.array([2.0, 0.0]), np.array([0.0, 0.0])
    (gam, _), st = O.stepsize(rule, st, x1, g1, x0, g0)
    assert gam == pytest.approx(1 / (2 * 2.0))
    # OurRule, norm_A = 0: identical points -> C = L = 0 (nan_to_zero) -> D = 0 -> gamma1*sqrt(2)
    rule = O.OurRule(gamma=0.5)
    (_, _), st = O.stepsize(rule)
    (gam, sig), st = O.stepsize(rule, st, x1, g1, x1, g1)
    assert gam == pytest.approx(0.5 * math.sqrt(2))
    assert st == (gam, 0.5)
    # Julia min propagates NaN
    assert math.isnan(O.jl_min(1.0, float("nan"), 0.5))
    with pytest.raises(ValueError):
        O.OurRule(gamma=0, norm_A=0)


def test_prox_definitions():
    v = np.array([-3.0, -0.5, 0.0, 0.2, 2.0])
    y, gy = O.prox(O.NormL1(2.0), v, 0.5)          # threshold gl = 1
    assert np.array_equal(y, np.array([-2.0, 0.0, 0.0, 0.0, 1.0]))
    assert gy == pytest.approx(2.0 * 3.0)
    y, _ = O.prox(O.IndBox(0.0, 1.0), v, 0.7)
    assert np.array_equal(y, np.array([0.0, 0.0, 0.0, 0.2, 1.0]))
    # conj of translated l1 == clamp(w - sigma*b, -1, 1)  
    b = np.array([0.3, -0.2, 1.0, 0.0, -4.0])
    hc = O.convex_conjugate(O.Translate(O.NormL1(1.0), -b))
    sigma = 0.37
    y, _ = O.prox(hc, v, sigma)
    assert np.allclose(y, np.clip(v - sigma * b, -1, 1), atol=1e-15)
    # conj of translated l2 == projection of (w - sigma*b) on the unit l2 ball
    hc2 = O.convex_conjugate(O.Translate(O.NormL2(1.0), -b))
    y, _ = O.prox(hc2, v, sigma)
    u = v - sigma * b
    assert np.allclose(y, u / max(1.0, np.linalg.norm(u)), atol=1e-14)
    assert isinstance(O.convex_conjugate(O.Zero()), O.IndZero)
    assert isinstance(O.convex_conjugate(O.IndZero()), O.Zero)


def test_tol_zero_runs_past_the_fixed_point_into_nan():
    """Reference semantics worth knowing before timing a loop with tol = 0: once the iterates have stopped
    moving, OurRule's C = |dg|^2 / <dg, dx> is x / 0 = Inf (nan_to_zero maps only NaN, src/AdaProx.jl:24), L = 0,
    D = gamma L (gamma C - 1) = 0 * Inf = NaN, and Julia's min hands the NaN on (:264-271).  The oracle restates that: a small
    lasso reaches its analytic optimum, and some iterations later gamma is NaN for good."""
    import warnings

    from oracle import adaprox_oracle as O
    from oracle import problems as P

    p = P.random_lasso(60, 120, seed=0)
    Lf = np.linalg.norm(p.A, 2) ** 2
    recs = []
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        O.adaptive_proxgrad(np.zeros(120), f=O.LinearLeastSquares(p.A, p.b), g=O.NormL1(p.lam), rule=O.OurRule(gamma=1.0 / Lf),
                            tol=0.0, maxit=3000, record=recs.append)
    gam = np.array([r["gamma"] for r in recs])
    obj = np.array([r["objective"] for r in recs])
    bad = np.flatnonzero(~np.isfinite(gam))
    assert bad.size and bad[0] > 100                      # it does happen, and only after the problem is solved:
    assert obj[bad[0] - 1] == pytest.approx(p.optimum, rel=1e-12)
    assert not np.isfinite(gam[bad[0]:]).any()            # NaN for good
