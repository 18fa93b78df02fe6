"""hipGraph replay of the iteration: identical kernels, so results are bit-identical to the
eager stream-ordered launches (adaptive PG — the two-kernel small-problem path here —, PD with
a dual kernel, logistic), including the device-side termination freeze; the same once more with
the generic kernels (fused_small off)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import problems as P


@pytest.fixture(scope="module")
def apx():
    import adaprox_amd as apx
    apx.load()
    return apx


def test_graph_replay_is_bit_identical(apx):
    prob = P.random_lasso(500, 1000, pfactor=10, seed=0)
    gam = 1.0 / P.opnorm_sq_power(prob.A, iters=30)
    f, g = apx.LinearLeastSquares(prob.A, prob.b), apx.NormL1(1.0)
    for tol, maxit in ((0.0, 200), (0.5, 5000)):
        ra, rb = [], []
        xa, ia = apx.adaptive_proxgrad(np.zeros(1000), f=f, g=g, rule=apx.OurRule(gamma=gam), tol=tol, maxit=maxit,
                                       record=ra.append, use_graph=False)
        xb, ib = apx.adaptive_proxgrad(np.zeros(1000), f=f, g=g, rule=apx.OurRule(gamma=gam), tol=tol, maxit=maxit,
                                       record=rb.append, use_graph=True)
        assert ia == ib and np.array_equal(xa, xb) and ra == rb
    lad = P.random_lad(400, 150, seed=1, lam=5.0)
    gctx = apx.Context()
    gctx.fused_small(False)       # (the small primal-dual path rotates its buffers on the host: no replay)
    kw = dict(f=apx.Zero(), g=apx.NormL1(lad.lam), h=apx.Translate(apx.NormL2(1.0), -lad.y),
              A=apx.DeviceMatrix.from_numpy(lad.A, gctx), rule=apx.OurRule(t=0.7, norm_A=lad.norm_A), tol=0.0,
              maxit=120, ctx=gctx)
    xa, ya, _ = apx.adaptive_primal_dual(np.zeros(150), np.zeros(400), use_graph=False, **kw)
    xb, yb, _ = apx.adaptive_primal_dual(np.zeros(150), np.zeros(400), use_graph=True, **kw)
    assert np.array_equal(xa, xb) and np.array_equal(ya, yb)
    lr = P.random_logreg(800, 60, seed=2, dtype=np.float32)
    fl = apx.LogisticLoss(lr.X, lr.y)
    xa, _ = apx.adaptive_proxgrad(np.zeros(61, dtype=np.float32), f=fl, g=g, rule=apx.OurRule(gamma=lr.gamma0), tol=0.0, maxit=25,
                                  use_graph=False)
    xb, _ = apx.adaptive_proxgrad(np.zeros(61, dtype=np.float32), f=fl, g=g, rule=apx.OurRule(gamma=lr.gamma0), tol=0.0, maxit=25,
                                  use_graph=True)
    assert np.array_equal(xa, xb)


def test_graph_replay_generic_kernels(apx):
    ctx = apx.Context()
    ctx.fused_small(False)
    prob = P.random_lasso(300, 700, pfactor=10, seed=3)
    gam = 1.0 / P.opnorm_sq_power(prob.A, iters=30)
    f, g = apx.LinearLeastSquares(apx.DeviceMatrix.from_numpy(prob.A, ctx), prob.b), apx.NormL1(1.0)
    out = [apx.adaptive_proxgrad(np.zeros(700), f=f, g=g, rule=apx.OurRule(gamma=gam), tol=0.0, maxit=150, ctx=ctx,
                                 use_graph=ug) for ug in (False, True, None)]
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][0], out[2][0])


def test_graph_refuses_host_round_trips(apx):
    class F:
        def eval_with_pullback(self, x):
            return float(x @ x), (lambda: 2 * x)

    with pytest.raises(apx.ApxError, match="device-only"):
        apx.adaptive_proxgrad(np.ones(4), f=F(), g=apx.Zero(), rule=apx.OurRule(gamma=0.1), maxit=3, use_graph=True)
