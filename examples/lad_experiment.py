#!/usr/bin/env python3
"""experiments/least_absolute_deviation/runme.jl:23-97 on the HIP path with a synthetic dense X
(the LIBSVM ingest is out of scope): Condat-Vu, Malitsky-Pock and AdaPDM+ over the reference's
t_values, records as JSONL.

    python examples/lad_experiment.py --m 2000 --n 400 --maxit 1000
    python examples/lad_experiment.py --m 50000 --n 20000 --maxit 200      # BASELINE config C5's size: one pass over A per iteration / trial
"""
import argparse
import importlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import adaprox_amd as apx  # noqa: E402

problems = importlib.import_module("adaptive-proximal-algorithms_amd.problems")
logging_ = importlib.import_module("adaptive-proximal-algorithms_amd.logging")

T_VALUES = [0.01, 0.15, 0.02, 0.025, 0.05, 0.1, 0.2, 0.5, 1, 2, 5, 10, 20, 50, 100]     # runme.jl:48


def run_least_absolute_deviation(m, n, lam=10.0, seed=0, tol=1e-5, maxit=1000, record=None, t_values=T_VALUES):
    p = problems.device_lad(m, n, seed=seed, lam=lam)                                     # :39-46
    f, g, h = apx.Zero(), apx.NormL1(lam), apx.Translate(apx.NormL1(), -p.y)
    x0, y0 = np.zeros(n), np.zeros(m)
    kw = dict(f=f, g=g, h=h, tol=tol, maxit=maxit, record=record)
    out, secs = {}, {}

    def timed(name, call):
        t0 = time.perf_counter()
        out[name] = call()
        secs[name] = time.perf_counter() - t0

    timed("Condat-Vu", lambda: apx.condat_vu(x0, y0, A=apx.Counting(p.A), Lf=0.0, norm_A=p.norm_A, name="Condat-Vu", **kw))
    for t in t_values:                                                                    # :64-78
        timed(f"Malitsky-Pock (t={t})", lambda: apx.malitsky_pock(x0, y0, A=apx.Counting(p.A), sigma=1.0, t=t,
                                                                   name=f"Malitsky-Pock (t={t})", **kw))
    for t in t_values:                                                                    # :80-94
        timed(f"AdaPDM+ (t={t})", lambda: apx.adaptive_linesearch_primal_dual(x0, y0, A=apx.Counting(p.A), eta=p.norm_A, t=t,
                                                                              name=f"AdaPDM+ (t={t})", **kw))
    run_least_absolute_deviation.seconds = secs          # (wall time per method: at BASELINE config C5's size — --m 50000 --n 20000 —
    return p, out                                        #  every method reads A once per iteration / per linesearch trial)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, default=2000)
    ap.add_argument("--n", type=int, default=400)
    ap.add_argument("--lam", type=float, default=10.0)
    ap.add_argument("--maxit", type=int, default=1000)
    ap.add_argument("--tol", type=float, default=1e-5)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    path = a.out or f"lad_{a.m}_{a.n}.jsonl"
    if os.path.exists(path):
        os.remove(path)
    sink = logging_.get_logger(path, console=False)
    p, out = run_least_absolute_deviation(a.m, a.n, a.lam, tol=a.tol, maxit=a.maxit, record=sink)
    sink.close()
    for name, (x, y, it) in out.items():
        obj = a.lam * np.sum(np.abs(x)) + np.sum(np.abs(p.A.matvec(x) - p.y))
        sec = run_least_absolute_deviation.seconds[name]
        print(f"{name:28s} it={it:5d}  objective={obj:.8g}  {sec:7.2f} s  {it / sec:8.1f} it/s")
    print("records ->", path)


if __name__ == "__main__":
    main()
