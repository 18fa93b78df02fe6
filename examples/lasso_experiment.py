#!/usr/bin/env python3
"""experiments/lasso/runme.jl on the HIP path: `run_random_lasso` (:30-160) with every solver
the reference runs (:86-160), records written as JSONL by the mirror of experiments/logging.jl.
Plotting (:163-189) stays with the reference's own scripts, which read these files unchanged.

    python examples/lasso_experiment.py --m 500 --n 1000 --pfactor 10 --maxit 2000 --tol 1e-7
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


def run_random_lasso(m=400, n=1000, pfactor=5, seed=0, tol=1e-5, maxit=10_000, record=None):
    prob = problems.device_lasso(m, n, pfactor=pfactor, seed=seed, power_iters=100)      # :40-84
    if record is not None:
        record(dict(method=None, it=1, objective=prob.optimum))                           # :79
    gam_init = 1 / prob.Lf                                                                # :81-82
    f = apx.LinearLeastSquares(prob.A, prob.b)
    g = apx.NormL1(prob.lam)
    kw = dict(g=g, tol=tol, maxit=maxit, record=record)
    x0 = np.zeros(n)
    out = {}
    out["PGM (fixed)"] = apx.fixed_proxgrad(x0, f=apx.Counting(f), gamma=gam_init, name="PGM (fixed)", **kw)   # :86-94
    for xi in (1, 1.5, 2):                                                                # :96-108
        name = f"PGM (backtracking)-(xi={xi})"
        out[name] = apx.backtracking_proxgrad(x0, f=apx.Counting(f), gamma0=gam_init, xi=xi, name=name, **kw)
    out["Nesterov (backtracking)"] = apx.backtracking_nesterov(x0, f=apx.Counting(f), gamma0=gam_init,
                                                               name="Nesterov (backtracking)", **kw)   # :110-118
    out["Nesterov (fixed)"] = apx.fixed_nesterov(x0, f=apx.Counting(f), gamma=gam_init, name="Nesterov (fixed)", **kw)
    out["AdaPGM (MM)"] = apx.adaptive_proxgrad(x0, f=apx.Counting(f), rule=apx.MalitskyMishchenkoRule(gam_init),
                                               name="AdaPGM (MM)", **kw)                  # :131-139
    out["AdaPGM (Ours)"] = apx.adaptive_proxgrad(x0, f=apx.Counting(f), rule=apx.OurRule(gamma=gam_init),
                                                 name="AdaPGM (Ours)", **kw)              # :141-149
    out["aGRAAL"] = apx.agraal(x0, f=apx.Counting(f), gamma0=gam_init, name="aGRAAL",
                               rng=np.random.default_rng(seed), **kw)                     # :151-159
    return prob, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, default=500)
    ap.add_argument("--n", type=int, default=1000)
    ap.add_argument("--pfactor", type=float, default=10)
    ap.add_argument("--maxit", type=int, default=2000)
    ap.add_argument("--tol", type=float, default=1e-7)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--out", default=None, help="JSONL path (default: lasso_<m>_<n>_<pfactor>.jsonl)")
    a = ap.parse_args()
    path = a.out or f"lasso_{a.m}_{a.n}_{a.pfactor:g}.jsonl"
    if os.path.exists(path):
        os.remove(path)
    sink = logging_.get_logger(path, console=False)
    prob, out = run_random_lasso(a.m, a.n, a.pfactor, a.seed, a.tol, a.maxit, record=sink)
    sink.close()
    print(f"optimum {prob.optimum:.12g}   records -> {path}")
    for name, (x, it) in out.items():
        obj = 0.5 * np.linalg.norm(prob.A.matvec(x) - prob.b) ** 2 + prob.lam * np.sum(np.abs(x))
        print(f"{name:34s} it={it:6d}  objective-optimum={obj - prob.optimum:.3e}")


if __name__ == "__main__":
    main()
