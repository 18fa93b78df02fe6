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
