#!/usr/bin/env python3
"""Interleaved A/B of library options on one BASELINE config, in ONE process on ONE matrix.

Runs of the same binary minutes apart differ by +-3 % on this part (clocks), boxes by up to 8 %: the only comparison that means
anything is one that alternates the variants — R rounds, each variant S timed steps per round, HIP-event time of the pass over A
(the library's own events: apx_ctx_profile) and wall time per iteration.  A variant is a set of context options
(include/adaprox_hip.h APX_OPT_*), applied before its solver is CREATED (shapes are chosen at set-up) and again before each of
its timed blocks (a few options are read at launch).

    python tools/ab.py c3 base "MULTI_ROWS_MASK=7" "MULTI_WAVES=11" --rounds 6 --steps 40
    python tools/ab.py c5 base "TEAM_MEMBERS=4" "TEAM_MEMBERS=4,TEAM_SLOTS=62" "MULTI_WAVES=11"
    python tools/ab.py c2 base "MULTI_WAVES=11" "TEAM_COOPERATIVE=1"
"""
import argparse
import json
import os
import statistics
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import adaprox_amd as apx  # noqa: E402
from importlib import import_module  # noqa: E402

L = import_module("adaptive-proximal-algorithms_amd._lib")
problems = import_module("adaptive-proximal-algorithms_amd.problems")

DEFAULTS = {"MULTI_ROWS_MASK": 15, "SLAB_GENERATION": 0, "TEAM_COOPERATIVE": 0, "FUSED_TAIL": 1, "MULTI_WAVES": 0, "TEAM_MEMBERS": 0,
            "TEAM_SLOTS": 0, "ROW_DEAL": -1, "TEAM_INTERLEAVE": 1, "TEAM_PD_DEFER": 1}


def parse_variant(text):
    opts = dict(DEFAULTS)
    if text != "base":
        for kv in text.split(","):
            k, v = kv.split("=")
            if k == "ONE_PASS":
                opts["ONE_PASS"] = int(v)

            else:
                assert k in DEFAULTS, k
                opts[k] = int(v)
    return opts


def apply(ctx, opts):
    ctx.one_pass(opts.get("ONE_PASS", 1))
    for k, v in opts.items():
        if k != "ONE_PASS":
            ctx.set_option(getattr(L, "APX_OPT_" + k), v)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("config", choices=["c2", "c3", "c3f64", "c5", "c5f32"])
    ap.add_argument("variants", nargs="+")
    ap.add_argument("--rounds", type=int, default=6)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--rows", type=int, default=0)
    ap.add_argument("--cols", type=int, default=0)
    args = ap.parse_args()
    ctx = apx.Context(0)
    apx.set_default_context(ctx)
    if args.config == "c2":
        m, n, es = args.rows or 50_000, args.cols or 100_000, 8
        p = problems.device_lasso(m, n, pfactor=10, seed=0, power_iters=20, ctx=ctx, dual_layout=True)
        mk = lambda: apx.Solver(np.zeros(n), f=apx.LinearLeastSquares(p.A, p.b), g=apx.NormL1(p.lam), rule=apx.OurRule(gamma=1.0 / p.Lf),      # noqa: E731
                                tol=0.0, ctx=ctx, poll_every=1 << 30)
        restart_every = 1 << 30
    elif args.config in ("c3", "c3f64"):
        dt = np.float32 if args.config == "c3" else np.float64
        m, n, es = args.rows or 100_000, 10_000, np.dtype(dt).itemsize
        p = problems.device_logreg(m, n, seed=0, lam=0.01, dtype=dt, ctx=ctx)
        p.X.build_transpose()
        mk = lambda: apx.Solver(np.zeros(n + 1, dtype=dt), f=apx.LogisticLoss(p.X, p.y), g=apx.NormL1(p.lam), rule=apx.OurRule(gamma=p.gamma0),      # noqa: E731
                                tol=0.0, ctx=ctx, poll_every=1 << 30)
        restart_every = 36 if dt == np.float32 else 40        # (the loop runs past its fixed point: DESIGN.md section 9)
    else:
        dt = np.float64 if args.config == "c5" else np.float32
        m, n, es = args.rows or 50_000, args.cols or (20_000 if args.config == "c5" else 10_000), np.dtype(dt).itemsize
        p = problems.device_lad(m, n, seed=0, lam=10.0, ctx=ctx) if dt == np.float64 else None
        if p is None:
            raise SystemExit("c5f32: not wired")
        p.A.build_transpose()
        gamma, sigma = 1.0 / p.norm_A, 0.99 / p.norm_A
        mk = lambda: apx.Solver(np.zeros(n), np.zeros(m), f=apx.Zero(), g=apx.NormL1(p.lam), h=apx.Translate(apx.NormL1(1.0), -p.y), A=p.A,      # noqa: E731
                                rule=apx.FixedStepsize(gamma, np.sqrt(sigma / gamma)), tol=0.0, ctx=ctx, poll_every=1 << 30)
        restart_every = 1 << 30
    variants = [(v, parse_variant(v)) for v in args.variants]
    solvers, names, done = {}, {}, {}
    for v, o in variants:
        apply(ctx, o)
        s = mk()
        s.step(5, sync=True)
        solvers[v], names[v], done[v] = s, ctx.kernel_name("one_pass") or ctx.kernel_name("gemv_t"), 5
    kern = {v: [] for v, _ in variants}
    wall = {v: [] for v, _ in variants}
    for r in range(args.rounds):
        order = variants if r % 2 == 0 else variants[::-1]
        for v, o in order:
            apply(ctx, o)
            if done[v] + args.steps > restart_every:
                solvers[v].close()
                solvers[v] = mk()
                solvers[v].step(5, sync=True)
                done[v] = 5
            s = solvers[v]
            ctx.kernel_stats(reset=True)
            ctx.profile(True)
            ctx.sync()
            t0 = time.perf_counter()
            s.step(args.steps, sync=True)
            dtw = time.perf_counter() - t0
            ctx.profile(False)
            st = ctx.kernel_stats(reset=True)
            done[v] += args.steps
            ms = sum(k["total_ms"] for k in st.values()) / args.steps
            kern[v].append(ms)
            wall[v].append(dtw / args.steps * 1e3)
    finals, tw = {}, {}
    for v, _ in variants:
        x, _, res = solvers[v].result()
        finals[v] = (int(res.numit), float(res.objective), bool(np.all(np.isfinite(x))))
        try:
            tw[v] = solvers[v].team_waits()
        except Exception:      # noqa: BLE001 (not a team kernel)
            tw[v] = None
    base = statistics.median(kern[variants[0][0]])
    for v, _ in variants:
        k, w = statistics.median(kern[v]), statistics.median(wall[v])
        print(json.dumps({"config": args.config, "variant": v, "kernel": names[v], "pass_ms_median": round(k, 4), "pass_ms_min": round(min(kern[v]), 4),
                          "pass_ms_max": round(max(kern[v]), 4), "iter_ms_median": round(w, 4), "TBps_of_bytes_read": round(m * n * es / k / 1e9, 3),
                          "vs_first": round(k / base, 4), "rounds": args.rounds, "steps": args.steps,
                          "last_solver": {"iterations": finals[v][0], "objective": finals[v][1], "finite": finals[v][2]}, "team_waits": tw[v]}), flush=True)


if __name__ == "__main__":
    main()
