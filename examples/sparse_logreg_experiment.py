#!/usr/bin/env python3
"""experiments/sparse_logreg/runme.jl:41-146 on the HIP path: l1-regularised logistic regression
on a sparse X (a LIBSVM file if given — mushrooms / a5a / phishing in the reference — or a synthetic
matrix of mushrooms' shape), with the solvers the reference compares.

    python examples/sparse_logreg_experiment.py [--data path/to/libsvm_file] [--lam 0.01] [--maxit 2000] [--tol 1e-7]
"""
import argparse
import importlib
import os
import sys

import numpy as np
import scipy.sparse as sp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import adaprox_amd as apx  # noqa: E402

libsvm = importlib.import_module("adaptive-proximal-algorithms_amd.libsvm")
logging_ = importlib.import_module("adaptive-proximal-algorithms_amd.logging")


def run_logreg_l1_data(X, y, lam, tol=1e-5, maxit=1000, record=None):
    m, n = X.shape
    n = n + 1                                                               # :52-53 (bias = last coordinate)
    dX = apx.DeviceMatrix.from_sparse(X)
    f = apx.LogisticLoss(dX, y)
    g = apx.NormL1(lam)
    X1 = sp.hstack([X, np.ones((m, 1))]).tocsr()
    Lf = sp.linalg.norm(X1 @ X1.T) / 4 / m                                  # :58-59
    x0 = np.zeros(n)
    gam_init = 1.0 / Lf
    kw = dict(g=g, record=record)
    out = {}
    # :65-74 the "accurate" reference run at tol/10, twice maxit
    out["AdaPGM (accurate)"] = apx.adaptive_proxgrad(x0, f=f, rule=apx.OurRule(gamma=gam_init), tol=tol / 10,
                                                     maxit=maxit * 2, name="AdaPGM (accurate)", **kw)
    out["PGM (fixed)"] = apx.fixed_proxgrad(x0, f=apx.Counting(f), gamma=gam_init, tol=tol, maxit=maxit,
                                            name="PGM (fixed)", **kw)
    for xi in (1, 1.5, 2):
        name = f"PGM (backtracking)-(xi={xi})"
        out[name] = apx.backtracking_proxgrad(x0, f=apx.Counting(f), gamma0=gam_init, xi=xi, tol=tol, maxit=maxit / 2,
                                              name=name, **kw)                # maxit/2 is a Float64 in the reference
    out["Nesterov (backtracking)"] = apx.backtracking_nesterov(x0, f=apx.Counting(f), gamma0=gam_init, tol=tol,
                                                               maxit=maxit / 2, name="Nesterov (backtracking)", **kw)
    out["AdaPGM (MM)"] = apx.adaptive_proxgrad(x0, f=apx.Counting(f), rule=apx.MalitskyMishchenkoRule(gam_init), tol=tol,
                                               maxit=maxit, name="AdaPGM (MM)", **kw)
    out["AdaPGM (Ours)"] = apx.adaptive_proxgrad(x0, f=apx.Counting(f), rule=apx.OurRule(gamma=gam_init), tol=tol,
                                                 maxit=maxit, name="AdaPGM (Ours)", **kw)
    out["aGRAAL"] = apx.agraal(x0, f=apx.Counting(f), gamma0=gam_init, tol=tol, maxit=maxit, name="aGRAAL",
                               rng=np.random.default_rng(0), **kw)
    return f, g, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--data", default=None)
    ap.add_argument("--lam", type=float, default=0.01)
    ap.add_argument("--maxit", type=int, default=2000)
    ap.add_argument("--tol", type=float, default=1e-7)
    ap.add_argument("--out", default="sparse_logreg.jsonl")
    a = ap.parse_args()
    if a.data:
        X, y = libsvm.load_libsvm_dataset(a.data, np.float64, labels=[0.0, 1.0])   # :50
    else:
        rng = np.random.default_rng(0)
        X = sp.random(8124, 112, density=0.19, format="csr", random_state=rng, data_rvs=lambda k: np.ones(k))
        w = rng.standard_normal(112) * (rng.uniform(size=112) < 0.2)
        y = (1 / (1 + np.exp(-(X @ w))) > rng.uniform(size=8124)).astype(np.float64)
    if os.path.exists(a.out):
        os.remove(a.out)
    sink = logging_.get_logger(a.out, console=False)
    f, g, out = run_logreg_l1_data(X, y, a.lam, a.tol, a.maxit, record=sink)
    sink.close()
    for name, (x, it) in out.items():
        print(f"{name:34s} it={it:6d}  objective={f(x) + g(x):.12g}")
    print("records ->", a.out)


if __name__ == "__main__":
    main()
