#!/usr/bin/env python3
"""experiments/dual_svm/runme.jl:30-135 on the HIP path with synthetic labelled data (the reference loads
LIBSVM files): f = Quadratic(Q, q) with Q = Dy X X' Dy, q = -1, g = IndBox(0, C), h = IndZero(), A = y';
AdaPDM over the reference's t_values (the only direct use of `adaptive_primal_dual` with
OurRule(t, norm_A) in the reference), then Malitsky-Pock and AdaPDM+; records as JSONL.

    python examples/dual_svm_experiment.py --N 1243 --d 22 --maxit 1000
"""
import argparse
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import adaprox_amd as apx  # noqa: E402

logging_ = importlib.import_module("adaptive-proximal-algorithms_amd.logging")

T_VALUES = [0.01, 0.15, 0.02, 0.025, 0.05, 0.1, 0.2, 0.5, 1, 2, 5, 10]                    # runme.jl:59


def make_data(N, d, seed=0):
    rng = np.random.default_rng(seed)
    X = rng.uniform(-1, 1, (N, d))
    y = np.sign(X @ rng.standard_normal(d) + 0.3 * rng.standard_normal(N))
    y[y == 0] = 1.0
    return X, y


def run_dsvm(X, y, C=1e-1, tol=1e-5, maxit=1000, record=None, t_values=T_VALUES):
    N = y.size
    Q = np.asfortranarray((y[:, None] * X) @ (y[:, None] * X).T)                           # :47-48  Dy X X' Dy
    q = -np.ones(N)                                                                       # :49
    A = apx.DeviceMatrix.from_numpy(np.asfortranarray(y[None, :]))                         # :54  Matrix(y')
    f = apx.Quadratic(Q, q)
    g, h = apx.IndBox(0.0, C), apx.IndZero()                                              # :52-53
    norm_A = float(np.linalg.norm(y))                                                     # :57
    x0, y0 = np.zeros(N), np.zeros(1)
    kw = dict(g=g, h=h, tol=tol, maxit=maxit, record=record)
    out = {}
    for t in t_values:                                                                    # :61-76
        out[f"AdaPDM (t={t})"] = apx.adaptive_primal_dual(x0, y0, f=apx.Counting(f), A=A,
                                                          rule=apx.OurRule(t=t, norm_A=norm_A), name=f"AdaPDM (t={t})", **kw)
    for t in t_values:                                                                    # :78-93
        out[f"Malitsky-Pock (t={t})"] = apx.malitsky_pock(x0, y0, f=apx.Counting(f), A=apx.Counting(A), sigma=1.0, t=t,
                                                          name=f"Malitsky-Pock (t={t})", **kw)
    for t in t_values:                                                                    # :95-110
        out[f"AdaPDM+ (t={t})"] = apx.adaptive_linesearch_primal_dual(x0, y0, f=apx.Counting(f), A=apx.Counting(A),
                                                                      eta=norm_A, t=t, name=f"AdaPDM+ (t={t})", **kw)
    return (Q, q, y, C), out


def dual_objective(Q, q, x):
    return 0.5 * x @ (Q @ x) + q @ x


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--N", type=int, default=1243)
    ap.add_argument("--d", type=int, default=22)
    ap.add_argument("--C", type=float, default=0.1)
    ap.add_argument("--maxit", type=int, default=1000)
    ap.add_argument("--tol", type=float, default=1e-5)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    path = a.out or f"dual_svm_{a.N}_{a.d}.jsonl"
    if os.path.exists(path):
        os.remove(path)
    sink = logging_.get_logger(path, console=False)
    X, y = make_data(a.N, a.d)
    (Q, q, yl, C), out = run_dsvm(X, y, a.C, tol=a.tol, maxit=a.maxit, record=sink)
    sink.close()
    for name, (x, yd, it) in out.items():
        print(f"{name:26s} it={it:5d}  dual objective={dual_objective(Q, q, x):.8g}  |y'x|={abs(yl @ x):.2e}")
    print("records ->", path)


if __name__ == "__main__":
    main()
