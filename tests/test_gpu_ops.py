"""GPU parity, op level: every C-ABI operator against the CPU oracle on the same seeded
inputs.  fp64 tolerance 1e-12 relative (summation
order differs from numpy's), fp32 storage 2e-6."""
import math

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


def _rel(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    den = max(float(np.max(np.abs(b))), 1e-300)
    return float(np.max(np.abs(a - b))) / den


SHAPES = [(1, 1), (2, 3), (33, 65), (500, 1000), (1031, 517), (4096, 64), (64, 4099), (3000, 2)]


@pytest.mark.parametrize("m,n", SHAPES)
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_gemv_pair_matches_numpy(apx, m, n, dtype):
    rng = np.random.default_rng(m * 7919 + n)
    A = np.asfortranarray(rng.uniform(-1, 1, (m, n)).astype(dtype))
    x = rng.standard_normal(n).astype(dtype)
    r = rng.standard_normal(m).astype(dtype)
    dA = apx.DeviceMatrix.from_numpy(A)
    tol = 1e-12 if dtype == np.float64 else 2e-6
    y_ref = A.astype(np.float64) @ x.astype(np.float64)
    g_ref = A.astype(np.float64).T @ r.astype(np.float64)
    assert _rel(dA @ x, y_ref) < tol
    assert _rel(dA.T @ r, g_ref) < tol
    assert np.array_equal(dA.to_numpy(), A)


def _variant_count(transposed: bool) -> int:
    """The compiled variant tables are queried from the library itself (no GPU needed for that)."""
    import importlib
    L = importlib.import_module("adaptive-proximal-algorithms_amd._lib")
    return int(L.load().apx_gemv_variant_count(1 if transposed else 0))


@pytest.mark.parametrize("variant", range(_variant_count(False)))
def test_every_gemv_n_variant(apx, variant):
    ctx = apx.Context(0)          # private context: the default one keeps its tuned configuration
    rng = np.random.default_rng(variant)
    m, n = 1537, 2111
    A = np.asfortranarray(rng.uniform(-1, 1, (m, n)))
    x = rng.standard_normal(n)
    dA = apx.DeviceMatrix.from_numpy(A, ctx)
    for blocks in (64, 4096):
        ctx.set_gemv_config(n_variant=variant, n_blocks=blocks)
        assert _rel(dA @ x, A @ x) < 1e-12


@pytest.mark.parametrize("variant", range(_variant_count(True)))
def test_every_gemv_t_variant(apx, variant):
    ctx = apx.Context(0)          # private context: the default one keeps its tuned configuration
    rng = np.random.default_rng(100 + variant)
    m, n = 9001, 333
    A = np.asfortranarray(rng.uniform(-1, 1, (m, n)))
    r = rng.standard_normal(m)
    dA = apx.DeviceMatrix.from_numpy(A, ctx)
    for blocks in (1, 8192):
        ctx.set_gemv_config(t_variant=variant, t_blocks=blocks)
        assert _rel(dA.T @ r, A.T @ r) < 1e-12


def test_device_generator_is_bit_exact(apx):
    for dtype in (np.float64, np.float32):
        m, n = 257, 131
        ref = P.uniform_matrix(5, m, n, -1.0, 1.0, dtype=dtype)
        dev = apx.DeviceMatrix.uniform(m, n, seed=5, dtype=dtype).to_numpy()
        assert np.array_equal(dev, ref)
    # a row shard of a global matrix is generated independently
    full = P.uniform_matrix(9, 300, 40, -1.0, 1.0)
    shard = apx.DeviceMatrix.uniform(100, 40, seed=9, row_offset=150, m_total=300).to_numpy()
    assert np.array_equal(shard, full[150:250])


def test_matrix_helpers(apx):
    rng = np.random.default_rng(3)
    A = np.asfortranarray(rng.uniform(-1, 1, (77, 45)))
    dA = apx.DeviceMatrix.from_numpy(A)
    assert dA.norm() == pytest.approx(np.linalg.norm(A), rel=1e-13)          # Frobenius (Appendix A.1)
    alpha = rng.uniform(0.5, 2, 45)
    dA.scale_columns(alpha)
    assert np.array_equal(dA.to_numpy(), A * alpha[None, :])
    dA.fill_column(44, 1.0)
    assert np.all(dA.to_numpy()[:, 44] == 1.0)
    B = np.asfortranarray(rng.uniform(-1, 1, (300, 120)))
    dB = apx.DeviceMatrix.from_numpy(B)
    assert dB.opnorm_sq(iters=60) == pytest.approx(P.opnorm_sq_power(B, iters=60), rel=1e-10)


PROX_CASES = [
    ("l1", lambda b: (O.NormL1(0.7), "NormL1", dict(lam=0.7))),
    ("box", lambda b: (O.IndBox(-0.3, 0.9), "IndBox", dict(lb=-0.3, ub=0.9))),
    ("zero", lambda b: (O.Zero(), "Zero", {})),
    ("indzero", lambda b: (O.IndZero(), "IndZero", {})),
    ("l2", lambda b: (O.NormL2(1.3), "NormL2", dict(lam=1.3))),
]


@pytest.mark.parametrize("name,mk", PROX_CASES)
@pytest.mark.parametrize("shifted", [False, True])
@pytest.mark.parametrize("conj", [False, True])
def test_prox_matches_oracle(apx, name, mk, shifted, conj):
    rng = np.random.default_rng(11)
    n = 1000
    v = rng.standard_normal(n) * 2
    b = rng.standard_normal(n)
    og, cls, kw = mk(b)
    dg = getattr(apx, cls)(**kw)
    if shifted:
        og, dg = O.Translate(og, b), apx.Translate(dg, b)
    if conj:
        og, dg = O.convex_conjugate(og), apx.convex_conjugate(dg)
    for gamma in (0.05, 1.0, 3.7):
        y_ref, val_ref = O.prox(og, v, gamma)
        y, val = apx.prox(dg, v, gamma)
        assert np.allclose(y, y_ref, rtol=0, atol=1e-14 * max(1.0, np.max(np.abs(y_ref))))
        if not conj:
            assert val == pytest.approx(val_ref, rel=1e-12, abs=1e-300)


def test_prox_box_vector_bounds_and_values(apx):
    rng = np.random.default_rng(2)
    v = rng.standard_normal(50) * 3
    lb = np.full(50, -np.inf)
    ub = np.full(50, np.inf)
    lb[0], ub[0] = -2.9, 2.9                      # the Simple2DBox of test/runtests.jl:15-23, widened
    y, val = apx.prox(apx.IndBox(lb, ub), v, 1.0)
    y_ref, _ = O.prox(O.IndBox(lb, ub), v, 1.0)
    assert np.array_equal(y, y_ref) and val == 0.0
    assert apx.NormL1(2.0)(v) == pytest.approx(2.0 * np.sum(np.abs(v)), rel=1e-13)
    assert apx.IndBox(-1.0, 1.0)(np.array([0.5, 2.0])) == math.inf
    assert apx.IndBox(-1.0, 1.0)(np.array([0.5, -1.0])) == 0.0
    assert apx.IndZero()(np.zeros(4)) == 0.0 and apx.IndZero()(np.array([0.0, 1e-300])) == math.inf


@pytest.mark.parametrize("m,n", [(500, 1000), (37, 5), (1, 9), (2000, 300)])
def test_least_squares_eval_pullback(apx, m, n):
    rng = np.random.default_rng(m + n)
    A = np.asfortranarray(rng.uniform(-1, 1, (m, n)))
    b = rng.standard_normal(m)
    w = rng.standard_normal(n)
    f_ref, pb_ref = O.LinearLeastSquares(A, b).eval_with_pullback(w)
    f = apx.LinearLeastSquares(A, b)
    fx, pb = apx.eval_with_pullback(f, w)
    assert fx == pytest.approx(f_ref, rel=1e-13)
    assert _rel(pb(), pb_ref()) < 1e-12
    # pb() uses the residual of the LATEST eval (closure semantics)
    w2 = rng.standard_normal(n)
    fx2, pb2 = apx.eval_with_pullback(f, w2)
    assert _rel(pb2(), O.LinearLeastSquares(A, b).eval_with_pullback(w2)[1]()) < 1e-12


@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-12), (np.float32, 5e-6)])
@pytest.mark.parametrize("m,n", [(400, 31), (1000, 32), (64, 200)])
def test_logistic_eval_pullback(apx, dtype, tol, m, n):
    # n columns of X -> w has n+1 entries, bias last (experiments/sparse_logreg/runme.jl:24)
    rng = np.random.default_rng(m * 3 + n)
    X = np.asfortranarray(rng.uniform(-1, 1, (m, n)).astype(dtype))
    y = (rng.uniform(size=m) < 0.4).astype(dtype)
    w = (rng.standard_normal(n + 1) * 0.5).astype(dtype)
    f_ref, pb_ref = O.LogisticLoss(X.astype(np.float64), y.astype(np.float64)).eval_with_pullback(w.astype(np.float64))
    f = apx.LogisticLoss(X, y)
    fx, pb = apx.eval_with_pullback(f, w)
    assert fx == pytest.approx(f_ref, rel=max(tol, 1e-12))
    assert _rel(pb(), pb_ref()) < tol


def test_invalid_arguments_fail_loudly(apx):
    A = np.zeros((4, 3), order="F")
    f = apx.LinearLeastSquares(A, np.zeros(4))
    with pytest.raises(ValueError):
        apx.adaptive_proxgrad(np.zeros(5), f=f, g=apx.NormL1(1.0), rule=apx.OurRule(gamma=1.0))
    with pytest.raises(ValueError):
        apx.OurRule(gamma=0, norm_A=0)            # src/AdaProx.jl:246
    with pytest.raises(ValueError):
        apx.OurRulePlus(gamma=0)                  # src/AdaProx.jl:288
    with pytest.raises(TypeError):
        apx.adaptive_proxgrad(np.zeros(3), f=object(), g=apx.NormL1(1.0), rule=apx.OurRule(gamma=1.0))
