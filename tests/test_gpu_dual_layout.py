"""Optional dual layout (apx_matrix_build_transpose): `A*x` computed by the column-streaming
kernel over a transposed copy must equal the single-layout product up to summation order, the
copy must be dropped when A is edited in place, and solvers must be unaffected."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import adaprox_oracle as O
from oracle import problems as P


@pytest.fixture(scope="module")
def apx():
    import adaprox_amd as apx
    apx.load()
    return apx


@pytest.mark.parametrize("m,n", [(1, 1), (33, 65), (500, 1000), (4097, 129), (70, 5000), (3000, 2)])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_products_with_transposed_copy(apx, m, n, dtype):
    rng = np.random.default_rng(m + 3 * n)
    A = np.asfortranarray(rng.uniform(-1, 1, (m, n)).astype(dtype))
    x = rng.standard_normal(n).astype(dtype)
    r = rng.standard_normal(m).astype(dtype)
    dA = apx.DeviceMatrix.from_numpy(A)
    y0, g0 = dA @ x, dA.T @ r
    dA.build_transpose()
    y1, g1 = dA @ x, dA.T @ r
    tol = 1e-12 if dtype == np.float64 else 2e-6
    ref = A.astype(np.float64) @ x.astype(np.float64)
    assert np.max(np.abs(y1 - ref)) <= tol * max(1.0, np.max(np.abs(ref)))
    assert np.max(np.abs(y1 - y0)) <= tol * max(1.0, np.max(np.abs(ref)))
    assert np.array_equal(g1, g0)                      # A'*r does not use the copy
    # editing A in place drops the copy: the product follows the edit
    alpha = rng.uniform(0.5, 2.0, n)
    dA.scale_columns(alpha)
    ref2 = (A.astype(np.float64) * alpha) @ x.astype(np.float64)
    assert np.max(np.abs((dA @ x) - ref2)) <= 10 * tol * max(1.0, np.max(np.abs(ref2)))


def test_solvers_with_dual_layout(apx):
    """The GENERIC loop on a matrix with a transposed copy (`A*x` as the column-streaming kernel over the copy): a context
    with the small-size policy off — with it on, 500 x 1000 and 600 x 240 would run small.hip / small_pd.hip on private
    copies and never reach the dual-layout products this test is about."""
    ctx = apx.Context(0)
    ctx.fused_small(False)
    prob = P.random_lasso(500, 1000, pfactor=10, seed=0)
    gam = 1.0 / P.opnorm_sq_power(prob.A, iters=50)
    dA = apx.DeviceMatrix.from_numpy(prob.A, ctx).build_transpose()
    f, g = apx.LinearLeastSquares(dA, prob.b), apx.NormL1(1.0)
    K = 200
    xo, _ = O.adaptive_proxgrad(np.zeros(1000), f=O.LinearLeastSquares(prob.A, prob.b), g=O.NormL1(1.0),
                                rule=O.OurRule(gamma=gam), tol=0.0, maxit=K)
    ctx.kernel_stats(reset=True)
    ctx.profile(True)
    x, _ = apx.adaptive_proxgrad(np.zeros(1000), f=f, g=g, rule=apx.OurRule(gamma=gam), tol=0.0, maxit=K, ctx=ctx)
    st = ctx.kernel_stats(reset=True)
    ctx.profile(False)
    # both products of every iteration were GEMV launches (the small paths book none)
    assert st["gemv_n"]["launches"] == K + 1 and st["gemv_t"]["launches"] == K + 1
    assert np.max(np.abs(x - xo)) <= 1e-10 * np.max(np.abs(xo))
    xg, _ = apx.adaptive_proxgrad(np.zeros(1000), f=f, g=g, rule=apx.OurRule(gamma=gam), tol=0.0, maxit=K, use_graph=True,
                                  ctx=ctx)
    assert np.array_equal(xg, x)
    lad = P.random_lad(600, 240, seed=1, lam=10.0)
    dL = apx.DeviceMatrix.from_numpy(lad.A, ctx).build_transpose()
    kw = dict(Lf=0.0, norm_A=lad.norm_A, tol=0.0, maxit=150)
    xo, yo, _ = O.condat_vu(np.zeros(240), np.zeros(600), f=O.Zero(), g=O.NormL1(10.0),
                            h=O.Translate(O.NormL1(1.0), -lad.y), A=lad.A, **kw)
    x, y, _ = apx.condat_vu(np.zeros(240), np.zeros(600), f=apx.Zero(), g=apx.NormL1(10.0),
                            h=apx.Translate(apx.NormL1(1.0), -lad.y), A=dL, ctx=ctx, **kw)
    assert np.max(np.abs(x - xo)) <= 1e-10 * np.max(np.abs(xo))
    assert np.max(np.abs(y - yo)) <= 1e-10 * np.max(np.abs(yo))
