"""HIP path against the committed golden fixtures (oracle-generated; see
tests/golden/make_golden.py for provenance).  The problem DATA is regenerated from the seeds
(the generator is itself checked bit-exact against the device generator in test_gpu_ops)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import problems as P

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "adaprox_golden.npz"))


@pytest.fixture(scope="module")
def apx():
    import adaprox_amd as apx
    apx.load()
    return apx


def _close(a, b, tol=1e-10):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b)))) <= tol * max(1.0, float(np.max(np.abs(b))))


def test_lasso_rules_against_golden(apx):
    prob = P.random_lasso(60, 120, pfactor=10, seed=3)
    gam = float(G["lasso_gamma0"])
    f, g = apx.LinearLeastSquares(prob.A, prob.b), apx.NormL1(prob.lam)
    rules = {"ours": apx.OurRule(gamma=gam), "mm": apx.MalitskyMishchenkoRule(gamma=gam),
             "fixed": apx.FixedStepsize(gam, 1.0), "plus": apx.OurRulePlus(gamma=gam)}
    for name, rule in rules.items():
        recs = []
        x, _ = apx.adaptive_proxgrad(np.zeros(120), f=f, g=g, rule=rule, tol=0.0, maxit=12, record=recs.append)
        assert _close(x, G[f"lasso_{name}_x"]), name
        assert np.allclose([r["gamma"] for r in recs], G[f"lasso_{name}_gamma"], rtol=1e-8), name
        assert np.allclose([r["norm_res"] for r in recs], G[f"lasso_{name}_norm_res"], rtol=1e-8), name
        assert np.allclose([r["objective"] for r in recs], G[f"lasso_{name}_objective"], rtol=1e-11), name
    recs = []
    z, _ = apx.backtracking_proxgrad(np.zeros(120), f=f, g=g, gamma0=gam, xi=1.5, tol=0.0, maxit=12, record=recs.append)
    assert _close(z, G["lasso_bt_z"])
    assert np.allclose([r["gamma"] for r in recs], G["lasso_bt_gamma"], rtol=1e-13)


def test_pd_logreg_prox_against_golden(apx):
    lad = P.random_lad(80, 30, seed=4, lam=2.0)
    recs = []
    x, y, _ = apx.condat_vu(np.zeros(30), np.zeros(80), f=apx.Zero(), g=apx.NormL1(lad.lam),
                            h=apx.Translate(apx.NormL1(1.0), -lad.y), A=lad.A, Lf=0.0, norm_A=lad.norm_A, tol=0.0,
                            maxit=40, record=recs.append)
    assert _close(x, G["lad_x"]) and _close(y, G["lad_y"])
    assert np.allclose([r["norm_res"] for r in recs], G["lad_norm_res"], rtol=1e-9)
    assert (recs[0]["gamma"], recs[0]["sigma"]) == pytest.approx(tuple(G["lad_gamma_sigma"]), rel=1e-15)
    lr = P.random_logreg(150, 20, seed=5, lam=0.01, dtype=np.float64)
    recs = []
    x, _ = apx.adaptive_proxgrad(np.zeros(21), f=apx.LogisticLoss(lr.X, lr.y), g=apx.NormL1(lr.lam),
                                 rule=apx.OurRule(gamma=lr.gamma0), tol=0.0, maxit=15, record=recs.append)
    assert _close(x, G["logreg_x"])
    assert np.allclose([r["objective"] for r in recs], G["logreg_objective"], rtol=1e-11)
    v, b = G["prox_v"], G["prox_b"]
    assert np.array_equal(apx.prox(apx.NormL1(0.7), v, 0.9)[0], G["prox_l1"])
    assert np.array_equal(apx.prox(apx.IndBox(-0.5, 1.25), v, 0.9)[0], G["prox_box"])
    assert _close(apx.prox(apx.convex_conjugate(apx.Translate(apx.NormL1(1.0), -b)), v, 0.37)[0],
                  G["prox_l1_translate_conj"], 1e-15)
    assert _close(apx.prox(apx.convex_conjugate(apx.Translate(apx.NormL2(1.0), -b)), v, 0.37)[0],
                  G["prox_l2_translate_conj"], 1e-14)
