"""Oracle pins for the comparison methods: the reference's third
assertion block of "Simple 2D problem" (test/runtests.jl:45-50, backtracking_nesterov) and the
analytic optima of the lasso generator / Nesterov's worst-case function."""
import numpy as np
import pytest

from oracle import adaprox_oracle as O
from oracle import problems as P
from test_oracle_kats import Simple2DBox, Simple2DObjective, WorstQuadratic


def test_simple_2d_backtracking_nesterov():
    f, g = Simple2DObjective(), Simple2DBox()
    sol, numit = O.backtracking_nesterov(np.ones(2), f=f, g=g, gamma0=1.0)       # test/runtests.jl:45-47
    assert f(sol) < 1e-7                                                          # :49
    assert g(sol) == 0                                                            # :50


@pytest.mark.parametrize("method", ["fixed_nesterov", "backtracking_nesterov", "agraal"])
def test_baselines_reach_the_lasso_optimum(method):
    prob = P.random_lasso(100, 300, pfactor=10, seed=0)
    Lf = float(np.linalg.norm(prob.A, 2) ** 2)
    f, g = O.LinearLeastSquares(prob.A, prob.b), O.NormL1(prob.lam)
    kw = dict(fixed_nesterov=dict(gamma=1 / Lf), backtracking_nesterov=dict(gamma0=1 / Lf),
              agraal=dict(gamma0=1 / Lf, x0=np.full(300, 0.01)))[method]
    x, it = getattr(O, method)(np.zeros(300), f=O.Counting(f), g=g, tol=1e-7, maxit=50_000, **kw)
    assert it < 50_000
    assert f(x) + g(x) == pytest.approx(prob.optimum, rel=1e-9)      # experiments/lasso/runme.jl:77


def test_nesterov_worst_case_with_accelerated_methods():
    k = n = 100
    L = 100.0
    f = WorstQuadratic(k, L)
    optimum = (L / 8) * (1 / (k + 1) - 1)                                         # nesterov_worst_case/runme.jl:53
    x, it = O.fixed_nesterov(np.zeros(n), f=f, g=O.Zero(), gamma=1 / L, tol=1e-6, maxit=10_000)
    assert optimum - 1e-12 <= f(x) < optimum + 1e-9
    x, it = O.backtracking_nesterov(np.zeros(n), f=f, g=O.Zero(), gamma0=1.0, tol=1e-6, maxit=10_000)
    assert optimum - 1e-12 <= f(x) < optimum + 1e-6          # exhausts maxit like the reference run (gap 1e-7)


def test_malitsky_pock_and_adapdm_plus_agree_on_lad():
    lad = P.random_lad(200, 60, seed=5, lam=3.0)
    h = O.Translate(O.NormL1(1.0), -lad.y)
    obj = lambda x: lad.lam * np.sum(np.abs(x)) + np.sum(np.abs(lad.A @ x - lad.y))
    A = O.Counting(lad.A)
    x1, _, it1 = O.malitsky_pock(np.zeros(60), np.zeros(200), f=O.Zero(), g=O.NormL1(lad.lam), h=h, A=A, sigma=1.0,
                                 t=1.0, tol=1e-3, maxit=50_000)
    x2, _, it2 = O.adaptive_linesearch_primal_dual(np.zeros(60), np.zeros(200), f=O.Zero(), g=O.NormL1(lad.lam), h=h,
                                                   A=lad.A, eta=lad.norm_A, t=1.0, tol=1e-3, maxit=50_000)
    assert it1 < 50_000 and it2 < 50_000
    assert obj(x1) == pytest.approx(obj(x2), rel=1e-5)
    assert A.amul_count == it1 + 1                      # one A'y before the loop (:598) + one per iteration (:603)
