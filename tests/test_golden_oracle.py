"""The committed golden fixtures (tests/golden/adaprox_golden.npz, produced by the oracle —
see make_golden.py for provenance) are reproduced by the oracle on this machine."""
import os

import numpy as np
import pytest

from oracle import adaprox_oracle as O
from oracle import problems as P

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "adaprox_golden.npz"))


def test_oracle_reproduces_lasso_goldens():
    prob = P.random_lasso(int(G["lasso_m"]), int(G["lasso_n"]), pfactor=10, seed=int(G["lasso_seed"]))
    gam = float(G["lasso_gamma0"])
    assert 1.0 / P.opnorm_sq_power(prob.A, iters=50) == pytest.approx(gam, rel=1e-12)
    rules = {"ours": O.OurRule(gamma=gam), "mm": O.MalitskyMishchenkoRule(gamma=gam),
             "fixed": O.FixedStepsize(gam, 1.0), "plus": O.OurRulePlus(gamma=gam)}
    for name, rule in rules.items():
        recs = []
        x, _ = O.adaptive_proxgrad(np.zeros(120), f=O.LinearLeastSquares(prob.A, prob.b), g=O.NormL1(prob.lam),
                                   rule=rule, tol=0.0, maxit=12, record=recs.append)
        assert np.allclose(x, G[f"lasso_{name}_x"], rtol=0, atol=1e-12 * np.max(np.abs(x)))
        assert np.allclose([r["gamma"] for r in recs], G[f"lasso_{name}_gamma"], rtol=1e-10)


def test_oracle_reproduces_pd_and_logreg_goldens():
    lad = P.random_lad(int(G["lad_m"]), int(G["lad_n"]), seed=int(G["lad_seed"]), lam=float(G["lad_lam"]))
    x, y, _ = O.condat_vu(np.zeros(30), np.zeros(80), f=O.Zero(), g=O.NormL1(lad.lam),
                          h=O.Translate(O.NormL1(1.0), -lad.y), A=lad.A, Lf=0.0, norm_A=lad.norm_A, tol=0.0, maxit=40)
    assert np.allclose(x, G["lad_x"], rtol=0, atol=1e-12) and np.allclose(y, G["lad_y"], rtol=0, atol=1e-12)
    lr = P.random_logreg(150, 20, seed=5, lam=0.01, dtype=np.float64)
    x, _ = O.adaptive_proxgrad(np.zeros(21), f=O.LogisticLoss(lr.X, lr.y), g=O.NormL1(lr.lam),
                               rule=O.OurRule(gamma=lr.gamma0), tol=0.0, maxit=15)
    assert np.allclose(x, G["logreg_x"], rtol=0, atol=1e-12)
