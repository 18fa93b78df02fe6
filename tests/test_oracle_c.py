"""The C + OpenMP restatement behind bench.py's cpu_baseline (oracle/lasso_omp.c) against the numpy oracle: same
iterates, step sizes and residuals (src/AdaProx.jl:324-363 with OurRule :258-272, f of experiments/lasso/runme.jl:16-27)."""
import numpy as np
import pytest

from oracle import adaprox_oracle as O
from oracle import lasso_omp
from oracle import problems as P


@pytest.fixture(scope="module")
def clib(tmp_path_factory):
    return lasso_omp.load(lasso_omp.build(out=str(tmp_path_factory.mktemp("omp") / "liblasso_omp.so")))


@pytest.mark.parametrize("m,n,K", [(300, 700, 30), (2500, 90, 25), (64, 64, 10)])
def test_c_port_matches_numpy_oracle(clib, m, n, K):
    prob = P.random_lasso(m, n, pfactor=10, seed=3)
    gam = 1.0 / P.opnorm_sq_power(prob.A, iters=30)
    recs = []
    xo, _ = O.adaptive_proxgrad(np.zeros(n), f=O.LinearLeastSquares(prob.A, prob.b), g=O.NormL1(prob.lam),
                                rule=O.OurRule(gamma=gam), tol=0.0, maxit=K, record=recs.append)
    x, gammas, nres, fx = lasso_omp.adaptive_proxgrad_lasso(prob.A, prob.b, prob.lam, gam, K, lib=clib)
    assert np.max(np.abs(x - xo)) <= 1e-10 * max(np.max(np.abs(xo)), 1e-300)
    og = np.array([r["gamma"] for r in recs])
    on = np.array([r["norm_res"] for r in recs])
    assert np.max(np.abs(gammas - og) / og) <= 1e-8
    assert np.max(np.abs(nres - on) / on) <= 1e-8
    # objective of the record = f_x + g(x) at the iterate the record belongs to: f_x alone is checked through the last x
    assert np.all(np.isfinite(fx)) and fx[-1] <= fx[0]


def test_c_port_reaches_the_generator_optimum(clib):
    """experiments/lasso/runme.jl:77: the generator knows its optimum (0.5 + lam * ||x*||_1)."""
    prob = P.random_lasso(200, 400, pfactor=10, seed=0)
    gam = 1.0 / P.opnorm_sq_power(prob.A, iters=50)
    x, _, nres, _ = lasso_omp.adaptive_proxgrad_lasso(prob.A, prob.b, prob.lam, gam, 200_000, lib=clib, tol=1e-9)
    obj = 0.5 * np.linalg.norm(prob.A @ x - prob.b) ** 2 + prob.lam * np.abs(x).sum()
    assert abs(obj - prob.optimum) <= 1e-9 * prob.optimum
    assert np.max(np.abs(x - prob.x_star)) < 1e-6
    assert nres[-1] <= 1e-9 and len(nres) < 200_000
    # the numpy oracle stops nearby (not AT the same iteration: beyond ~700 iterations two fp64 evaluations of this
    # recurrence that differ only in summation order have drifted apart, DESIGN.md §2 / tests/test_oracle_sensitivity.py)
    _, numit = O.adaptive_proxgrad(np.zeros(400), f=O.LinearLeastSquares(prob.A, prob.b), g=O.NormL1(prob.lam),
                                   rule=O.OurRule(gamma=gam), tol=1e-9, maxit=200_000)
    assert abs(numit - len(nres)) <= 0.1 * numit


def test_c_generator_is_the_numpy_generator(clib):
    """bench.py's CPU sample is built by the C generator: same matrix bit for bit, same problem to rounding."""
    a = P.random_lasso(150, 320, pfactor=10, seed=5)
    b = lasso_omp.random_lasso(150, 320, pfactor=10, seed=5, lib=clib)
    # (alpha comes from C'y, whose summation order differs: the scaled matrix agrees to rounding, the raw fill exactly)
    raw = np.empty((150, 320), order="F")
    clib.apx_oracle_fill_uniform(raw.ctypes.data, 150, 320, 150, 5, -1.0, 1.0, 0, 150)
    assert np.array_equal(raw, P.uniform_matrix(5, 150, 320, -1.0, 1.0))
    assert np.array_equal(a.perm, b.perm)
    np.testing.assert_allclose(b.A, a.A, rtol=1e-12, atol=0)
    np.testing.assert_allclose(b.b, a.b, rtol=1e-11, atol=1e-13)
    np.testing.assert_allclose(b.x_star, a.x_star, rtol=1e-12, atol=0)
    assert abs(a.optimum - b.optimum) <= 1e-12 * a.optimum
