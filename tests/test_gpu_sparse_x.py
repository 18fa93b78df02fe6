"""Opt-in sparse-x mode (apx_ctx_exploit_sparsity): `A*x` streams only the columns with
x_j != 0.  Same product up to summation order for every sparsity pattern (none, all, ragged),
and the lasso iteration is unchanged."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import adaprox_oracle as O
from oracle import problems as P


@pytest.fixture()
def ctx_sparse():
    import adaprox_amd as apx
    ctx = apx.Context(0)
    ctx.exploit_sparsity(True)
    return apx, ctx


@pytest.mark.parametrize("m,n", [(1, 1), (33, 65), (500, 1000), (1031, 5000), (4096, 37)])
@pytest.mark.parametrize("density", [0.0, 0.003, 0.1, 0.5, 1.0])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_product_skipping_zero_columns(ctx_sparse, m, n, density, dtype):
    apx, ctx = ctx_sparse
    rng = np.random.default_rng(m * 13 + n)
    A = np.asfortranarray(rng.uniform(-1, 1, (m, n)).astype(dtype))
    x = (rng.standard_normal(n) * (rng.uniform(size=n) < density)).astype(dtype)
    dA = apx.DeviceMatrix.from_numpy(A, ctx)
    ref = A.astype(np.float64) @ x.astype(np.float64)
    tol = 1e-12 if dtype == np.float64 else 2e-6
    assert np.max(np.abs(dA @ x - ref)) <= tol * max(1.0, np.max(np.abs(ref)))
    x[-1] = 1.5                                            # last column alone / appended
    ref = A.astype(np.float64) @ x.astype(np.float64)
    assert np.max(np.abs(dA @ x - ref)) <= tol * max(1.0, np.max(np.abs(ref)))


def test_lasso_iterates_unchanged(ctx_sparse):
    apx, ctx = ctx_sparse
    prob = P.random_lasso(500, 1000, pfactor=10, seed=0)
    gam = 1.0 / P.opnorm_sq_power(prob.A, iters=50)
    K = 300
    orecs, drecs = [], []
    xo, _ = O.adaptive_proxgrad(np.zeros(1000), f=O.LinearLeastSquares(prob.A, prob.b), g=O.NormL1(1.0),
                                rule=O.OurRule(gamma=gam), tol=0.0, maxit=K, record=orecs.append)
    x, _ = apx.adaptive_proxgrad(np.zeros(1000), f=apx.LinearLeastSquares(prob.A, prob.b, ctx=ctx), g=apx.NormL1(1.0),
                                 rule=apx.OurRule(gamma=gam), tol=0.0, maxit=K, record=drecs.append, ctx=ctx)
    assert np.max(np.abs(x - xo)) <= 1e-10 * np.max(np.abs(xo))
    assert [r["gamma"] for r in drecs] == pytest.approx([r["gamma"] for r in orecs], rel=1e-8)
    assert np.count_nonzero(x) < 400                       # the iterate IS sparse: the mode has something to skip
