"""An INDEPENDENT check of the end state: scikit-learn's coordinate-descent Lasso (no code in
common with the oracle or the HIP path) solves the same problem; the adaptive proximal-gradient
solution must coincide.  sklearn minimises (1/(2m))||y - Xw||^2 + alpha*||w||_1, i.e.
alpha = lambda/m for the reference's 0.5*||Aw - b||^2 + lambda*||w||_1 (experiments/lasso/runme.jl:83-84)."""
import numpy as np
import pytest
from sklearn.linear_model import Lasso

from oracle import adaprox_oracle as O
from oracle import problems as P


def _sklearn_lasso(A, b, lam):
    m = A.shape[0]
    mdl = Lasso(alpha=lam / m, fit_intercept=False, tol=1e-14, max_iter=200_000, selection="cyclic")
    mdl.fit(A, b)
    return mdl.coef_


def test_oracle_vs_sklearn_lasso():
    rng = np.random.default_rng(0)
    A = rng.standard_normal((120, 60))                     # overdetermined: unique minimiser
    b = A @ (rng.standard_normal(60) * (rng.uniform(size=60) < 0.3)) + 0.1 * rng.standard_normal(120)
    lam = 3.0
    w_ref = _sklearn_lasso(A, b, lam)
    gam = 1.0 / np.linalg.norm(A, 2) ** 2
    w, it = O.adaptive_proxgrad(np.zeros(60), f=O.LinearLeastSquares(A, b), g=O.NormL1(lam),
                                rule=O.OurRule(gamma=gam), tol=1e-10, maxit=100_000)
    assert it < 100_000
    assert np.max(np.abs(w - w_ref)) < 1e-8
    assert np.array_equal(w == 0, np.abs(w_ref) < 1e-12)   # same support


@pytest.mark.gpu
def test_hip_vs_sklearn_lasso():
    import adaprox_amd as apx
    rng = np.random.default_rng(1)
    A = np.asfortranarray(rng.standard_normal((400, 150)))
    b = A @ (rng.standard_normal(150) * (rng.uniform(size=150) < 0.2)) + 0.1 * rng.standard_normal(400)
    lam = 5.0
    w_ref = _sklearn_lasso(A, b, lam)
    f, g = apx.LinearLeastSquares(A, b), apx.NormL1(lam)
    gam = 1.0 / apx.DeviceMatrix.from_numpy(A).opnorm_sq(iters=200)
    for solve in (lambda: apx.adaptive_proxgrad(np.zeros(150), f=f, g=g, rule=apx.OurRule(gamma=gam), tol=1e-10,
                                                maxit=100_000),
                  lambda: apx.backtracking_proxgrad(np.zeros(150), f=f, g=g, gamma0=gam, xi=1.5, tol=1e-10,
                                                    maxit=100_000),
                  lambda: apx.fixed_nesterov(np.zeros(150), f=f, g=g, gamma=gam, tol=1e-10, maxit=200_000)):
        w, it = solve()
        assert np.max(np.abs(w - w_ref)) < 1e-7
