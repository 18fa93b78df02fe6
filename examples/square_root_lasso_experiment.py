#!/usr/bin/env python3
"""experiments/square_root_lasso/runme.jl:23-94 on the HIP path with a synthetic dense X (the reference
loads cpusmall_scale through LIBSVM): f = Zero, g = NormL1(lambda), h = Translate(NormL2(), -y),
A = [X 1]; Condat-Vu, Malitsky-Pock and AdaPDM+ over the reference's t_values, records as JSONL.

    python examples/square_root_lasso_experiment.py --m 2000 --n 12 --maxit 5000
"""
import argparse
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import adaprox_amd as apx  # noqa: E402

problems = importlib.import_module("adaptive-proximal-algorithms_amd.problems")
logging_ = importlib.import_module("adaptive-proximal-algorithms_amd.logging")

T_VALUES = [0.01, 0.15, 0.02, 0.025, 0.05, 0.1, 0.2, 0.5, 1, 2, 5, 10, 20, 50, 100]     # runme.jl:46


def run_square_root_lasso(m, n, lam=1e-1, seed=0, tol=1e-5, maxit=500, record=None, t_values=T_VALUES):
    p = problems.device_lad(m, n, seed=seed, lam=lam)            # same data model: A = [X 1], y = A x_true + noise  (:33-41)
    f, g, h = apx.Zero(), apx.NormL1(lam), apx.Translate(apx.NormL2(), -p.y)              # :37-39
    x0, y0 = np.zeros(n), np.zeros(m)
    kw = dict(f=f, g=g, h=h, tol=tol, maxit=maxit, record=record)
    out = {"Condat-Vu": apx.condat_vu(x0, y0, A=apx.Counting(p.A), Lf=0.0, norm_A=p.norm_A, name="Condat-Vu", **kw)}   # :48-60
    for t in t_values:                                                                    # :62-76
        out[f"Malitsky-Pock (t={t})"] = apx.malitsky_pock(x0, y0, A=apx.Counting(p.A), sigma=1.0, t=t,
                                                          name=f"Malitsky-Pock (t={t})", **kw)
    for t in t_values:                                                                    # :78-92
        out[f"AdaPDM+ (t={t})"] = apx.adaptive_linesearch_primal_dual(x0, y0, A=apx.Counting(p.A), eta=p.norm_A, t=t,
                                                                      name=f"AdaPDM+ (t={t})", **kw)
    return p, out


def objective(p, lam, x):
    return lam * np.sum(np.abs(x)) + np.linalg.norm(p.A.matvec(x) - p.y)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, default=2000)
    ap.add_argument("--n", type=int, default=12)
    ap.add_argument("--lam", type=float, default=10.0)        # main(): lambda = 1e1
    ap.add_argument("--maxit", type=int, default=5000)
    ap.add_argument("--tol", type=float, default=1e-5)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    path = a.out or f"sqrt_lasso_{a.m}_{a.n}.jsonl"
    if os.path.exists(path):
        os.remove(path)
    sink = logging_.get_logger(path, console=False)
    p, out = run_square_root_lasso(a.m, a.n, a.lam, tol=a.tol, maxit=a.maxit, record=sink)
    sink.close()
    for name, (x, y, it) in out.items():
        print(f"{name:28s} it={it:5d}  objective={objective(p, a.lam, x):.8g}")
    print("records ->", path)


if __name__ == "__main__":
    main()
