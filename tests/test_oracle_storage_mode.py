"""The oracle's Float32-STORAGE mode (oracle/adaprox_oracle.py storage_rounding, `emulate=np.float32`): the arithmetic the
product implements for Float32 data — vectors stored Float32, element arithmetic and sums in Float64, the logistic link in
Float32.  The reference never runs Float32 (`zeros(n)` is Float64 whatever `T`), so this mode pins nothing of the
reference ; it exists so that the Float32 GPU tests compare like with like (tests/test_gpu_slab.py).  Here: it is
identity by default, every stored vector is Float32-representable with it, and it stays within Float32's epsilon of the Float64 run."""
import numpy as np

from oracle import adaprox_oracle as O


def _problem(logistic):
    rng = np.random.default_rng(0)
    m, n = 300, 40
    A = rng.uniform(-1, 1, (m, n)).astype(np.float32).astype(np.float64)
    if logistic:
        b = (rng.uniform(size=m) < 0.5).astype(np.float64)
        return (lambda em: O.LogisticLoss(A, b, emulate=em)), n + 1, 4.0 * m / (np.linalg.norm(A) ** 2 + m), 0.01
    b = rng.standard_normal(m).astype(np.float32).astype(np.float64)
    return (lambda em: O.LinearLeastSquares(A, b, emulate=em)), n, 1.0 / np.linalg.norm(A, 2) ** 2, 1.0


def test_default_is_identity_and_the_mode_rounds_where_the_product_stores():
    for logistic in (False, True):
        mk, nx, gam, lam = _problem(logistic)
        x0 = np.full(nx, 0.01)
        runs = {}
        for em in (None, np.float32):
            tr = []
            x, it = O.adaptive_proxgrad(x0, f=mk(em), g=O.NormL1(lam), rule=O.OurRule(gamma=gam), tol=0.0, maxit=25, trace=tr, storage=em)
            runs[em] = (x, tr)
        x64, t64 = runs[None]
        x32, t32 = runs[np.float32]
        # identity by default: the plain call is the same computation
        xp, _ = O.adaptive_proxgrad(x0, f=mk(None), g=O.NormL1(lam), rule=O.OurRule(gamma=gam), tol=0.0, maxit=25)
        assert np.array_equal(xp, x64)
        # every stored iterate of the mode is a Float32 number ; the Float64 run's are not
        assert all(np.array_equal(q["x"], q["x"].astype(np.float32).astype(np.float64)) for q in t32)
        assert not np.array_equal(x64, x64.astype(np.float32).astype(np.float64))
        # and the two runs agree to Float32's epsilon while the step size is not yet a ratio of rounding errors
        assert np.max(np.abs(x32 - x64)) <= 5e-6 * np.max(np.abs(x64))
        # (once the rule's ratios of differences take over — iteration 4 here — Float32 storage moves the step size by ~1e-6, a few
        #  iterations later by ~2e-4: why a Float32 run cannot be held to a Float64 oracle tighter than that, and is to this mode)
        assert all(abs(a["gamma"] - b["gamma"]) <= 1e-3 * b["gamma"] for a, b in zip(t32[:12], t64[:12]))


def test_the_link_of_the_mode_is_float32():
    mk, nx, _, _ = _problem(True)
    w = np.linspace(-0.5, 0.5, nx)
    f64, pb64 = mk(None).eval_with_pullback(w)
    f32, pb32 = mk(np.float32).eval_with_pullback(w)
    assert f32 != f64 and abs(f32 - f64) <= 2e-6 * abs(f64)
    g64, g32 = pb64(), pb32()
    assert np.max(np.abs(g32 - g64)) <= 2e-6 * np.max(np.abs(g64)) and not np.array_equal(g32, g64)
