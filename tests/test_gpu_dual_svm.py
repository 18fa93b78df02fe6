"""experiments/dual_svm/runme.jl:47-76 — the only direct caller of `adaptive_primal_dual` with
`OurRule(t, norm_A)` in the reference: f = Quadratic(Q, q) (Q = Dy X X' Dy dense N x N),
g = IndBox(0, C), h = IndZero(), A = y' (1 x N) — natively on the HIP path against the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import adaprox_oracle as O


@pytest.fixture(scope="module")
def apx():
    import adaprox_amd as apx
    apx.load()
    return apx


def _rel_inf(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b)))) / max(float(np.max(np.abs(b))), 1e-300)


def _dual_svm(N=400, d=30, seed=0):
    rng = np.random.default_rng(seed)
    X = rng.uniform(-1, 1, (N, d))
    y = np.sign(X @ rng.standard_normal(d) + 0.1 * rng.standard_normal(N))
    Dy = np.diag(y)
    Q = np.asfortranarray(Dy @ X @ X.T @ Dy)                    # :48
    q = -np.ones(N)                                             # :49
    A = np.asfortranarray(y[None, :])                           # :54  A = Matrix(y')
    return Q, q, A


def test_quadratic_eval_pullback(apx):
    Q, q, _ = _dual_svm()
    rng = np.random.default_rng(1)
    x = rng.standard_normal(Q.shape[0])
    f_ref, pb_ref = O.Quadratic(Q, q).eval_with_pullback(x)
    fx, pb = apx.eval_with_pullback(apx.Quadratic(Q, q), x)
    assert fx == pytest.approx(f_ref, rel=1e-12)
    assert _rel_inf(pb(), pb_ref()) <= 1e-12


@pytest.mark.parametrize("t", [0.01, 0.2, 1.0, 10.0])
def test_dual_svm_adaptive_primal_dual(apx, t):
    Q, q, A = _dual_svm()
    N = Q.shape[0]
    C = 0.1                                                      # :35
    norm_A = float(np.linalg.norm(A))
    K = 200
    orecs, drecs = [], []
    of, df = O.Counting(O.Quadratic(Q, q)), apx.Counting(apx.Quadratic(Q, q))
    xo, yo, _ = O.adaptive_primal_dual(np.zeros(N), np.zeros(1), f=of, g=O.IndBox(0.0, C), h=O.IndZero(), A=A,
                                       rule=O.OurRule(t=t, norm_A=norm_A), maxit=K, tol=0.0, record=orecs.append)   # :63-76
    x, y, _ = apx.adaptive_primal_dual(np.zeros(N), np.zeros(1), f=df, g=apx.IndBox(0.0, C), h=apx.IndZero(), A=A,
                                       rule=apx.OurRule(t=t, norm_A=norm_A), maxit=K, tol=0.0, record=drecs.append)
    assert _rel_inf(x, xo) <= 1e-10 and _rel_inf(y, yo) <= 1e-9
    for ro, rd in zip(orecs, drecs):
        assert rd["gamma"] == pytest.approx(ro["gamma"], rel=1e-8)
        assert rd["sigma"] == pytest.approx(ro["sigma"], rel=1e-8)
        assert rd["objective"] == pytest.approx(ro["objective"], rel=1e-10, abs=1e-12)
    assert (df.eval_count, df.grad_count) == (of.eval_count, of.grad_count)
    assert np.all(x >= 0) and np.all(x <= C)


@pytest.mark.parametrize("N,t", [(400, 0.01), (400, 0.2), (400, 1.0), (400, 10.0), (270, 0.5), (1243, 2.0), (4096, 1.0), (5200, 1.0)])
def test_dual_svm_malitsky_pock_device_resident(apx, N, t):
    """experiments/dual_svm/runme.jl:78-92: malitsky_pock(t, sigma = 1/norm_A) on the dual SVM — the device-resident
    linesearch (csrc/small_mpq.hip) against the oracle and against the host-driven loop, incl. the evaluation counts."""
    Q, q, A = _dual_svm(N=N, d=22)
    norm_A = float(np.linalg.norm(A))
    K = 40 if N <= 1243 else (15 if N <= 4096 else 8)
    kw = dict(sigma=1.0 / norm_A, t=t, tol=0.0, maxit=K)
    orecs, drecs = [], []
    of, df = O.Counting(O.Quadratic(Q, q)), apx.Counting(apx.Quadratic(Q, q))
    oA, dA = O.Counting(A), apx.Counting(apx.DeviceMatrix.from_numpy(A))
    xo, yo, ito = O.malitsky_pock(np.zeros(N), np.zeros(1), f=of, g=O.IndBox(0.0, 0.1), h=O.IndZero(), A=oA,
                                  record=orecs.append, **kw)
    x, y, it = apx.malitsky_pock(np.zeros(N), np.zeros(1), f=df, g=apx.IndBox(0.0, 0.1), h=apx.IndZero(), A=dA,
                                 record=drecs.append, **kw)
    assert it == ito == K and _rel_inf(x, xo) <= 1e-10 and _rel_inf(y, yo) <= 1e-9
    assert len(drecs) == K
    for ro, rd in zip(orecs, drecs):
        assert list(rd) == list(ro)
        for k in ("gamma", "sigma", "norm_res", "objective"):
            assert rd[k] == pytest.approx(ro[k], rel=1e-8), k
        for k in ro:
            if k.endswith("_evals"):
                assert rd[k] == ro[k], k
    assert (df.eval_count, df.grad_count, dA.mul_count, dA.amul_count) == (of.eval_count, of.grad_count, oA.mul_count, oA.amul_count)
    assert any(r["sigma"] < o["sigma"] * 2 ** 0.5 for r, o in zip(drecs[1:], drecs[:-1]))      # the linesearch did backtrack
    # the host-driven loop (a context without the small-size policy) gives the same iterate
    g = apx.Context(0)
    g.fused_small(False)
    xg, yg, _ = apx.malitsky_pock(np.zeros(N), np.zeros(1), f=apx.Quadratic(apx.DeviceMatrix.from_numpy(Q, g), q, ctx=g),
                                  g=apx.IndBox(0.0, 0.1), h=apx.IndZero(), A=apx.DeviceMatrix.from_numpy(A, g), ctx=g, **kw)
    assert _rel_inf(x, xg) <= 1e-10


def test_dual_svm_malitsky_pock_termination_and_maxit(apx):
    Q, q, A = _dual_svm(N=300, d=12)
    N, norm_A = 300, float(np.linalg.norm(A))
    kw = dict(sigma=1.0 / norm_A, t=1.0)
    mk = lambda M: dict(f=M.Quadratic(Q, q), g=M.IndBox(0.0, 0.1), h=M.IndZero(), A=A)      # noqa: E731
    ro = []
    O.malitsky_pock(np.zeros(N), np.zeros(1), tol=0.0, maxit=60, record=ro.append, **mk(O), **kw)
    tol = min(r["norm_res"] for r in ro[:40]) * (1 + 1e-7)
    xo, yo, ito = O.malitsky_pock(np.zeros(N), np.zeros(1), tol=tol, maxit=5000, **mk(O), **kw)
    x, y, it = apx.malitsky_pock(np.zeros(N), np.zeros(1), tol=tol, maxit=5000, **mk(apx), **kw)
    assert it == ito <= 40 and _rel_inf(x, xo) <= 1e-10 and _rel_inf(y, yo) <= 1e-9
    for K in (0, 1, 2, 70):
        xo, yo, _ = O.malitsky_pock(np.zeros(N), np.zeros(1), tol=0.0, maxit=K, **mk(O), **kw)
        x, y, it = apx.malitsky_pock(np.zeros(N), np.zeros(1), tol=0.0, maxit=K, **mk(apx), **kw)
        assert it == K and _rel_inf(x, xo) <= 1e-10 if K else np.array_equal(x, np.zeros(N))


@pytest.mark.parametrize("N", [1243, 3000, 4096])
def test_dual_svm_adaptive_primal_dual_larger_sizes(apx, N):
    """The two-kernel primal-dual path (csrc/small_pd.hip) where its in-block dot products need several passes per
    block: Q is N x N with N > 1024."""
    Q, q, A = _dual_svm(N=N, d=22)
    norm_A = float(np.linalg.norm(A))
    K = 25
    xo, yo, _ = O.adaptive_primal_dual(np.zeros(N), np.zeros(1), f=O.Quadratic(Q, q), g=O.IndBox(0.0, 0.1), h=O.IndZero(), A=A,
                                       rule=O.OurRule(t=1.0, norm_A=norm_A), maxit=K, tol=0.0)
    x, y, it = apx.adaptive_primal_dual(np.zeros(N), np.zeros(1), f=apx.Quadratic(Q, q), g=apx.IndBox(0.0, 0.1), h=apx.IndZero(),
                                        A=A, rule=apx.OurRule(t=1.0, norm_A=norm_A), maxit=K, tol=0.0)
    assert it == K and _rel_inf(x, xo) <= 1e-10 and _rel_inf(y, yo) <= 1e-9
    xo, yo, _ = O.condat_vu(np.zeros(N), np.zeros(1), f=O.Quadratic(Q, q), g=O.IndBox(0.0, 0.1), h=O.IndZero(), A=A,
                            Lf=float(np.linalg.norm(Q)), norm_A=norm_A, maxit=K, tol=0.0)
    x, y, it = apx.condat_vu(np.zeros(N), np.zeros(1), f=apx.Quadratic(Q, q), g=apx.IndBox(0.0, 0.1), h=apx.IndZero(), A=A,
                             Lf=float(np.linalg.norm(Q)), norm_A=norm_A, maxit=K, tol=0.0)
    assert it == K and _rel_inf(x, xo) <= 1e-10 and _rel_inf(y, yo) <= 1e-9
