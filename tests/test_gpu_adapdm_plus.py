"""GPU parity for the first "next" row of SURVEY §8f: adaptive_linesearch_primal_dual
("AdaPDM+", src/AdaProx.jl:463-550) — what experiments/least_absolute_deviation/runme.jl:80-94
and experiments/square_root_lasso actually showcase — against the oracle."""
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


def _rel_inf(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b)))) / max(float(np.max(np.abs(b))), 1e-300)


@pytest.mark.parametrize("t", [0.01, 0.1, 1.0, 10.0])
def test_adapdm_plus_lad(apx, t):
    """LAD: f = Zero, g = NormL1(lambda), h = Translate(NormL1(), -y), eta = norm(A)
    (experiments/least_absolute_deviation/runme.jl:80-94)."""
    m, n = 600, 240
    prob = P.random_lad(m, n, seed=1, lam=10.0)
    K = 150
    oA = O.Counting(prob.A)
    oh = O.Counting(O.Translate(O.NormL1(1.0), -prob.y))
    orecs = []
    xo, yo, ito = O.adaptive_linesearch_primal_dual(np.zeros(n), np.zeros(m), f=O.Zero(), g=O.NormL1(prob.lam),
                                                    h=oh, A=oA, eta=prob.norm_A, t=t, tol=0.0, maxit=K,
                                                    record=orecs.append)
    dA = apx.Counting(apx.DeviceMatrix.from_numpy(prob.A))
    dh = apx.Counting(apx.Translate(apx.NormL1(1.0), -prob.y))
    drecs = []
    x, y, it = apx.adaptive_linesearch_primal_dual(np.zeros(n), np.zeros(m), f=apx.Zero(), g=apx.NormL1(prob.lam),
                                                   h=dh, A=dA, eta=prob.norm_A, t=t, tol=0.0, maxit=K,
                                                   record=drecs.append)
    assert it == ito == K
    assert _rel_inf(x, xo) <= 1e-10 and _rel_inf(y, yo) <= 1e-10
    for ro, rd in zip(orecs, drecs):
        assert rd["gamma"] == pytest.approx(ro["gamma"], rel=1e-9)
        assert rd["sigma"] == pytest.approx(ro["sigma"], rel=1e-9)
        assert rd["norm_res"] == pytest.approx(ro["norm_res"], rel=1e-8)
        assert rd["objective"] == pytest.approx(ro["objective"], rel=1e-11)
        assert (rd["A_evals"], rd["At_evals"], rd["prox_h_evals"]) == (ro["A_evals"], ro["At_evals"], ro["prox_h_evals"])
    # every linesearch trial is one A'*y and one conjugate prox
    assert (dA.mul_count, dA.amul_count, dh.prox_count) == (oA.mul_count, oA.amul_count, oh.prox_count)


def test_adapdm_plus_converges_like_the_oracle(apx):
    m, n = 300, 120
    prob = P.random_lad(m, n, seed=2, lam=5.0)
    kw = dict(eta=prob.norm_A, t=1.0, tol=1e-2, maxit=20_000)      # oracle: 1711 iterations
    xo, yo, ito = O.adaptive_linesearch_primal_dual(np.zeros(n), np.zeros(m), f=O.Zero(), g=O.NormL1(prob.lam),
                                                    h=O.Translate(O.NormL1(1.0), -prob.y), A=prob.A, **kw)
    x, y, it = apx.adaptive_linesearch_primal_dual(np.zeros(n), np.zeros(m), f=apx.Zero(), g=apx.NormL1(prob.lam),
                                                   h=apx.Translate(apx.NormL1(1.0), -prob.y), A=prob.A, **kw)
    assert it < 20_000 and abs(it - ito) <= max(3, ito // 50)
    obj = lambda xx: prob.lam * np.sum(np.abs(xx)) + np.sum(np.abs(prob.A @ xx - prob.y))
    assert obj(x) == pytest.approx(obj(xo), rel=1e-6)


def test_adapdm_plus_least_squares_f_and_sqrt_lasso_h(apx):
    """A smooth f together with h = Translate(NormL2(), -y) (square_root_lasso's h): all
    branches of the step-size formula (:517-521) are exercised (C, L != 0)."""
    rng = np.random.default_rng(8)
    m, n, k = 150, 80, 60
    A = np.asfortranarray(rng.uniform(-1, 1, (m, n)))
    Q = np.asfortranarray(rng.uniform(-1, 1, (k, n)) / np.sqrt(n))
    q, yv = rng.standard_normal(k), rng.standard_normal(m)
    nA = float(np.linalg.norm(A))
    kw = dict(eta=nA, t=0.5, tol=0.0, maxit=80)
    xo, yo, _ = O.adaptive_linesearch_primal_dual(np.zeros(n), np.zeros(m), f=O.LinearLeastSquares(Q, q),
                                                  g=O.NormL1(0.3), h=O.Translate(O.NormL2(1.0), -yv), A=A, **kw)
    x, y, _ = apx.adaptive_linesearch_primal_dual(np.zeros(n), np.zeros(m), f=apx.LinearLeastSquares(Q, q),
                                                  g=apx.NormL1(0.3), h=apx.Translate(apx.NormL2(1.0), -yv), A=A, **kw)
    assert _rel_inf(x, xo) <= 1e-9 and _rel_inf(y, yo) <= 1e-9


def test_adapdm_plus_asserts(apx):
    A = np.eye(3, order="F")
    kw = dict(f=apx.Zero(), g=apx.Zero(), h=apx.Zero(), A=A)
    with pytest.raises(AssertionError, match="eta must be positive"):          # src/AdaProx.jl:481
        apx.adaptive_linesearch_primal_dual(np.zeros(3), np.zeros(3), eta=0.0, **kw)
    with pytest.raises(AssertionError, match="Theta"):                          # :482
        apx.adaptive_linesearch_primal_dual(np.zeros(3), np.zeros(3), Theta=1.0, **kw)
    with pytest.raises(AssertionError, match="gamma is too large"):             # :488
        apx.adaptive_linesearch_primal_dual(np.zeros(3), np.zeros(3), gamma=10.0, **kw)


# ---------------------------------------------------------------------------------------
# tall, skinny A: the device-resident linesearch of csrc/skinny_ls.hip (the shapes of the reference's own
# LAD / square-root-lasso data: housing 506 x 14, abalone 4177 x 9, cpusmall 8192 x 13)
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("m,n,hkind,t", [(506, 14, "l1", 0.01), (506, 14, "l1", 1.0), (506, 14, "l2", 0.1), (506, 14, "l2", 10.0),
                                         (4177, 9, "l1", 0.5), (8192, 13, "l2", 2.0), (300, 1, "l1", 1.0), (1000, 120, "l1", 0.2)])
def test_adapdm_plus_skinny_vs_oracle(apx, m, n, hkind, t):
    # (l2: a small lambda keeps x away from 0 — with x == 0 and y on the ball the linesearch ratio is 0/0 noise)
    prob = P.random_lad(m, n, seed=7, lam=10.0 if hkind == "l1" else 0.05)
    mkh = (lambda M: M.Translate(M.NormL1(1.0), -prob.y)) if hkind == "l1" else (lambda M: M.Translate(M.NormL2(1.0), -prob.y))
    K = 120
    oA, oh = O.Counting(prob.A), O.Counting(mkh(O))
    orecs, drecs = [], []
    xo, yo, ito = O.adaptive_linesearch_primal_dual(np.zeros(n), np.zeros(m), f=O.Zero(), g=O.NormL1(prob.lam), h=oh, A=oA,
                                                    eta=prob.norm_A, t=t, tol=0.0, maxit=K, record=orecs.append)
    dA, dh = apx.Counting(apx.DeviceMatrix.from_numpy(prob.A)), apx.Counting(mkh(apx))
    x, y, it = apx.adaptive_linesearch_primal_dual(np.zeros(n), np.zeros(m), f=apx.Zero(), g=apx.NormL1(prob.lam), h=dh, A=dA,
                                                   eta=prob.norm_A, t=t, tol=0.0, maxit=K, record=drecs.append)
    assert it == ito == K and len(drecs) == K
    assert _rel_inf(x, xo) <= 1e-9 and _rel_inf(y, yo) <= 1e-9
    for ro, rd in zip(orecs, drecs):
        assert rd["gamma"] == pytest.approx(ro["gamma"], rel=1e-8)
        assert rd["sigma"] == pytest.approx(ro["sigma"], rel=1e-8)
        assert rd["norm_res"] == pytest.approx(ro["norm_res"], rel=1e-7, abs=1e-12)
        assert rd["objective"] == pytest.approx(ro["objective"], rel=1e-10)
        assert (rd["A_evals"], rd["At_evals"], rd["prox_h_evals"]) == (ro["A_evals"], ro["At_evals"], ro["prox_h_evals"])
    assert (dA.mul_count, dA.amul_count, dh.prox_count) == (oA.mul_count, oA.amul_count, oh.prox_count)
    assert oA.amul_count > K + 1 or t >= 1.0          # (small t: the linesearch does reject trials here)


def test_adapdm_plus_skinny_termination_maxit_and_fp32(apx):
    prob = P.random_lad(2000, 10, seed=9, lam=5.0)
    terms = lambda M: dict(f=M.Zero(), g=M.NormL1(prob.lam), h=M.Translate(M.NormL1(1.0), -prob.y))   # noqa: E731
    # a tolerance reached at a known iteration: same stop, same iterates as the oracle
    orecs = []
    O.adaptive_linesearch_primal_dual(np.zeros(10), np.zeros(2000), A=prob.A, eta=prob.norm_A, t=1.0, tol=0.0, maxit=200,
                                      record=orecs.append, **terms(O))
    tol = orecs[149]["norm_res"] * (1 + 1e-7)
    xo, yo, ito = O.adaptive_linesearch_primal_dual(np.zeros(10), np.zeros(2000), A=prob.A, eta=prob.norm_A, t=1.0, tol=tol,
                                                    maxit=5000, **terms(O))
    x, y, it = apx.adaptive_linesearch_primal_dual(np.zeros(10), np.zeros(2000), A=prob.A, eta=prob.norm_A, t=1.0, tol=tol,
                                                   maxit=5000, **terms(apx))
    assert it == ito <= 150 and _rel_inf(x, xo) <= 1e-9 and _rel_inf(y, yo) <= 1e-9
    # maxit = 0, 1, and a count that is not a multiple of the enqueue chunk
    for K in (0, 1, 67, 100):      # (drift horizon of this problem: 8e-8 at 130 iterations)
        xo, yo, _ = O.adaptive_linesearch_primal_dual(np.zeros(10), np.zeros(2000), A=prob.A, eta=prob.norm_A, t=0.3, tol=0.0,
                                                      maxit=K, **terms(O))
        x, y, it = apx.adaptive_linesearch_primal_dual(np.zeros(10), np.zeros(2000), A=prob.A, eta=prob.norm_A, t=0.3,
                                                       tol=0.0, maxit=K, **terms(apx))
        assert it == K and _rel_inf(x, xo) <= 1e-8 and _rel_inf(y, yo) <= 1e-8
    # fp32 storage
    A32, y32 = prob.A.astype(np.float32), prob.y.astype(np.float32)
    x, y, it = apx.adaptive_linesearch_primal_dual(np.zeros(10, dtype=np.float32), np.zeros(2000, dtype=np.float32),
                                                   f=apx.Zero(), g=apx.NormL1(prob.lam), h=apx.Translate(apx.NormL1(1.0), -y32),
                                                   A=A32, eta=prob.norm_A, t=1.0, tol=0.0, maxit=40)
    xo, yo, _ = O.adaptive_linesearch_primal_dual(np.zeros(10), np.zeros(2000), A=prob.A, eta=prob.norm_A, t=1.0, tol=0.0,
                                                  maxit=40, **terms(O))
    assert x.dtype == np.float32 and it == 40 and _rel_inf(x, xo) <= 5e-3
    # a linesearch that cannot terminate is reported, not spun on
    with pytest.raises(apx.ApxError, match="linesearch did not terminate"):
        apx.adaptive_linesearch_primal_dual(np.zeros(10), np.zeros(2000), A=prob.A, eta=1e-12, t=1.0, r=1.0, tol=0.0,
                                            maxit=5, max_linesearch=7, gamma=1e-3, **terms(apx))
