#!/usr/bin/env python3
"""Generates tests/golden/adaprox_golden.npz.

PROVENANCE: these vectors are produced by THIS REPOSITORY'S CPU ORACLE
(oracle/adaprox_oracle.py, a numpy restatement of AdaProx.jl) — NOT by the Julia reference,
which cannot run in this image (no Julia runtime) and whose own tests hold no golden
vectors.  They freeze the oracle so that (a) a change of numpy/BLAS or an edit of the oracle
is noticed by the CPU suite and (b) the GPU suite can check the HIP path against fixed
numbers.  Iteration counts are kept inside the window where fp64 summation-order noise is
< 1e-12 (tests/test_oracle_sensitivity.py).

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import adaprox_oracle as O   # noqa: E402
from oracle import problems as P         # noqa: E402


def main():
    out = {}
    # ---- lasso 60 x 120 (experiments/lasso/runme.jl construction), three rules
    prob = P.random_lasso(60, 120, pfactor=10, seed=3)
    gam = 1.0 / P.opnorm_sq_power(prob.A, iters=50)
    out["lasso_seed"], out["lasso_m"], out["lasso_n"], out["lasso_gamma0"] = 3, 60, 120, gam
    for name, rule in (("ours", O.OurRule(gamma=gam)), ("mm", O.MalitskyMishchenkoRule(gamma=gam)),
                       ("fixed", O.FixedStepsize(gam, 1.0)), ("plus", O.OurRulePlus(gamma=gam))):
        recs = []
        K = 12
        x, _ = O.adaptive_proxgrad(np.zeros(120), f=O.LinearLeastSquares(prob.A, prob.b), g=O.NormL1(prob.lam),
                                   rule=rule, tol=0.0, maxit=K, record=recs.append)
        out[f"lasso_{name}_x"] = x
        out[f"lasso_{name}_gamma"] = np.array([r["gamma"] for r in recs])
        out[f"lasso_{name}_norm_res"] = np.array([r["norm_res"] for r in recs])
        out[f"lasso_{name}_objective"] = np.array([r["objective"] for r in recs])
    # ---- backtracking
    recs = []
    z, _ = O.backtracking_proxgrad(np.zeros(120), f=O.LinearLeastSquares(prob.A, prob.b), g=O.NormL1(prob.lam),
                                   gamma0=gam, xi=1.5, tol=0.0, maxit=12, record=recs.append)
    out["lasso_bt_z"] = z
    out["lasso_bt_gamma"] = np.array([r["gamma"] for r in recs])
    out["lasso_bt_norm_res"] = np.array([r["norm_res"] for r in recs])
    # ---- LAD via Condat-Vu 80 x 30
    lad = P.random_lad(80, 30, seed=4, lam=2.0)
    recs = []
    x, y, _ = O.condat_vu(np.zeros(30), np.zeros(80), f=O.Zero(), g=O.NormL1(lad.lam),
                          h=O.Translate(O.NormL1(1.0), -lad.y), A=lad.A, Lf=0.0, norm_A=lad.norm_A, tol=0.0,
                          maxit=40, record=recs.append)
    out["lad_seed"], out["lad_m"], out["lad_n"], out["lad_lam"] = 4, 80, 30, 2.0
    out["lad_x"], out["lad_y"] = x, y
    out["lad_norm_res"] = np.array([r["norm_res"] for r in recs])
    out["lad_gamma_sigma"] = np.array([recs[0]["gamma"], recs[0]["sigma"]])
    # ---- logistic 150 x 20 fp64
    lr = P.random_logreg(150, 20, seed=5, lam=0.01, dtype=np.float64)
    recs = []
    x, _ = O.adaptive_proxgrad(np.zeros(21), f=O.LogisticLoss(lr.X, lr.y), g=O.NormL1(lr.lam),
                               rule=O.OurRule(gamma=lr.gamma0), tol=0.0, maxit=15, record=recs.append)
    out["logreg_seed"], out["logreg_m"], out["logreg_n"] = 5, 150, 20
    out["logreg_x"] = x
    out["logreg_gamma"] = np.array([r["gamma"] for r in recs])
    out["logreg_objective"] = np.array([r["objective"] for r in recs])
    # ---- prox vectors
    v = np.linspace(-3, 3, 41)
    b = np.cos(np.arange(41))
    out["prox_v"], out["prox_b"] = v, b
    out["prox_l1"] = O.prox(O.NormL1(0.7), v, 0.9)[0]
    out["prox_box"] = O.prox(O.IndBox(-0.5, 1.25), v, 0.9)[0]
    out["prox_l1_translate_conj"] = O.prox(O.convex_conjugate(O.Translate(O.NormL1(1.0), -b)), v, 0.37)[0]
    out["prox_l2_translate_conj"] = O.prox(O.convex_conjugate(O.Translate(O.NormL2(1.0), -b)), v, 0.37)[0]
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "adaprox_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
