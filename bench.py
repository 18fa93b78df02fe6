#!/usr/bin/env python3
"""bench.py — headline benchmark of the adaptive proximal-gradient hot path on MI355X.

Metric (BASELINE.json): prox-grad iterations/sec of `adaptive_proxgrad` (OurRule) on the
dense lasso of experiments/lasso/runme.jl with A 50 000 x 100 000 fp64 (40.0 GB resident in
HBM), plus the fraction of the HBM roofline of the dominant GEMV kernel.

A "step" = ONE iteration of src/AdaProx.jl:334-362 = GEMV-N (A x - b, ||r||^2) + GEMV-T
(A' r) + the fused residual / step-size / soft-threshold kernels, on data already resident
in HBM.  N > 1 (launched by torch.distributed.run, one rank per GPU): weak scaling — every
rank holds a 50 000-row shard of the (50 000 N) x 100 000 matrix (BASELINE config C4 at
N = 8) and the partial gradient A_k' r_k is all-reduced with RCCL over xGMI each iteration.
`value` = shard-iterations per second summed over the ranks (= iterations/sec at N = 1).

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md "HBM3E peak BW")
HBM_COPY_GBS = 6290.0      # same table: measured float4-copy ceiling (SURVEY.md 8d asks for both)
HBM_READ_GBS = 7211.0      # fastest pure 40 GB read found on this part over access pattern / unroll / grid / cache
                           # policy (tools/read_ceiling.hip, profiles/r01e_read_ceiling_40GB.log)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults = SURVEY.md 8(d): 200 timed iterations after 10 warm-up iterations (2.4 s at 11.3 ms each)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    # (long names: torch.distributed.run's own parser treats a bare `--m` after the script as an
    # ambiguous abbreviation of its options)
    ap.add_argument("--rows", dest="m", type=int, default=50_000, help="rows per GPU")
    ap.add_argument("--cols", dest="n", type=int, default=100_000)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--exploit-sparsity", action="store_true",
                    help="NOT the BASELINE metric: let A*x skip the columns with x_j == 0 (flagged in the JSON line)")
    ap.add_argument("--single-layout", action="store_true",
                    help="do not keep the transposed copy of A (40 GB instead of 80 GB; A*x ~5 %% slower)")
    ap.add_argument("--strong", action="store_true",
                    help="strong scaling: --rows is the GLOBAL row count, split over the ranks (default: weak, rows per GPU)")
    # CPU baseline sample: 4000 x 100000 fp64 (3.2 GB), 50 iterations ~ 10 s of dgemv on the box's host cores
    ap.add_argument("--cpu-rows", type=int, default=4000, help="rows of the CPU-baseline sample")
    ap.add_argument("--cpu-iters", type=int, default=50)
    return ap.parse_args()


def cpu_baseline(n: int, m_full: int, rows: int, iters: int, seed: int) -> dict:
    """The oracle (numpy -> OpenBLAS dgemv, the BLAS family Julia's `A*x` reaches) timed on the
    host cores on a bounded row-sample of the same workload, scaled per byte to the full
    matrix.  A reported baseline, not the target."""
    from oracle import adaprox_oracle as O
    from oracle import problems as P
    try:
        from threadpoolctl import threadpool_info
        threads = max([int(i.get("num_threads", 1)) for i in threadpool_info()] or [1])
    except Exception:
        threads = os.cpu_count() or 1
    prob = P.random_lasso(rows, n, pfactor=10, seed=seed)
    gam = 1.0 / P.opnorm_sq_power(prob.A, iters=5)
    f, g = O.LinearLeastSquares(prob.A, prob.b), O.NormL1(prob.lam)
    O.adaptive_proxgrad(np.zeros(n), f=f, g=g, rule=O.OurRule(gamma=gam), tol=0.0, maxit=2)       # warm-up
    t0 = time.perf_counter()
    O.adaptive_proxgrad(np.zeros(n), f=f, g=g, rule=O.OurRule(gamma=gam), tol=0.0, maxit=iters)
    dt = time.perf_counter() - t0
    its_sample = (iters + 1) / dt                       # iters loop passes + the pre-loop gradient
    return {
        "value": its_sample * rows / m_full, "unit": "iters/sec", "cores": threads, "kind": "port",
        "sample": f"oracle (numpy/OpenBLAS dgemv, {threads} threads) on a {rows}x{n} fp64 lasso of the same "
                  f"generator, {iters} iterations in {dt:.2f} s = {its_sample:.2f} it/s on the sample, scaled by "
                  f"bytes ({rows}/{m_full}) to the {m_full}x{n} matrix; Julia runtime absent on this image",
    }


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    # APX_BENCH_SHARE_GPU=1: rehearsal of the N-rank path on a ONE-GPU box — every rank uses
    # device 0 and the collectives go through gloo (RCCL refuses two ranks on one device).  Checks
    # the sharding / all-reduce logic with real ranks; its timings mean nothing.
    share_gpu = os.environ.get("APX_BENCH_SHARE_GPU") == "1"
    dev = 0 if share_gpu else local_rank
    torch.cuda.set_device(dev)
    # APX_BENCH_FORCE_DIST=1 runs the collective code path with a single rank (the 1-GPU box
    # rehearsal of what the driver launches with N ranks)
    use_dist = world > 1 or os.environ.get("APX_BENCH_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if share_gpu:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev))

    if not os.path.exists(os.path.join(ROOT, "adaptive-proximal-algorithms_amd", "csrc", "libadaprox_hip.so")):
        if local_rank == 0:
            import __graft_entry__
            __graft_entry__.build()          # in-tree hipcc build (the .so is git-ignored)
        if use_dist:
            dist.barrier()
    import adaprox_amd as apx
    from importlib import import_module
    problems = import_module("adaptive-proximal-algorithms_amd.problems")
    sharded = import_module("adaptive-proximal-algorithms_amd.sharded")

    m_global = args.m if args.strong else args.m * world
    row0 = rank * args.m
    if args.strong:
        row0, args.m = sharded.shard_rows(m_global, rank, world)   # this rank's first row and row count
    stream = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(stream):
        ctx = apx.Context(device=dev, stream=stream.cuda_stream)
        apx.set_default_context(ctx)
        ar_host = sharded.host_allreduce(dist, dev) if use_dist else None
        ar_dev = sharded.device_allreduce(dist, dev) if use_dist else None

        t_setup = time.perf_counter()
        prob = problems.device_lasso(args.m, args.n, pfactor=10, seed=args.seed, rank=rank, world=world,
                                     power_iters=30, ctx=ctx, allreduce_host=ar_host,
                                     dual_layout=not args.single_layout,   # dual: both passes stream contiguously (80 of 288 GB)
                                     m_total=m_global, row_offset=row0)
        t_setup = time.perf_counter() - t_setup

        if args.exploit_sparsity:
            ctx.exploit_sparsity(True)
        f = apx.LinearLeastSquares(prob.A, prob.b, allreduce=ar_dev)
        g = apx.NormL1(prob.lam)
        solver = apx.Solver(np.zeros(args.n), f=f, g=g, rule=apx.OurRule(gamma=1.0 / prob.Lf), tol=0.0,
                            ctx=ctx, poll_every=1 << 30)

        def barrier():
            if use_dist:
                dist.barrier()
            torch.cuda.synchronize()

        solver.step(args.warmup, sync=True)
        ctx.kernel_stats(reset=True)
        ctx.profile(True)                       # HIP events around every GEMV launch, on the launch stream
        barrier()
        t0 = time.perf_counter()
        solver.step(args.steps, sync=False)
        barrier()
        dt = time.perf_counter() - t0
        ctx.profile(False)
        stats = ctx.kernel_stats(reset=True)

        x, _, res = solver.result()
        if not np.all(np.isfinite(x)):
            sys.exit("bench.py: non-finite iterate")
        if int(res.numit) != args.warmup + args.steps or res.converged:
            sys.exit("bench.py: the loop stopped after %d iterations: the timed steps were not all real" % int(res.numit))
        solver.close()

    if use_dist:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cpu" if share_gpu else f"cuda:{dev}")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    if rank == 0:
        es = 8
        per_kernel = {}
        for k, st in stats.items():
            if st["launches"]:
                avg_ms = st["total_ms"] / st["launches"]
                per_kernel[k] = {"launches": st["launches"], "avg_ms": avg_ms,
                                 "GBps": st["bytes"] / st["launches"] / (avg_ms * 1e-3) / 1e9}
        dom = max(per_kernel, key=lambda k: per_kernel[k]["avg_ms"] * per_kernel[k]["launches"])
        algo_bytes = float(args.m) * args.n * es                 # one streaming read of A per launch
        achieved = per_kernel[dom]["GBps"]
        iters_per_s = args.steps / dt
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        # PMC traffic was collected (separate rocprofv3 --pmc passes, profiles/README.md) for the
        # default per-GPU shape only
        if os.path.exists(tpath) and (args.m, args.n) == (50_000, 100_000) and not share_gpu:
            try:
                # with the dual layout both passes run the column-streaming kernel (the "gemv_t" entry)
                tkey = "gemv_t" if prob.dual_layout else dom
                traffic = json.load(open(tpath)).get(tkey, {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "prox-grad iters/sec on dense lasso A 50k x 100k (adaptive_proxgrad, OurRule)",
            "value": (1 if args.strong else world) * iters_per_s, "unit": "iters/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong" if args.strong else "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"lasso, dense A {args.m}x{args.n} fp64 per GPU, adaptive_proxgrad on "
                                   f"{world} MI355X (GEMV + soft-threshold + stepsize reductions)"
                                   + ("" if world == 1 else f"; row-sharded {m_global}x{args.n}, RCCL all-reduce of A'r"),
                       "rows_per_gpu": args.m, "cols": args.n, "rule": "OurRule(gamma=1/Lf)",
                       "layout": ("single: A column-major, 40 GB" if not prob.dual_layout else
                                  "dual: A and its transposed copy resident (2 x 40 GB of the 288 GB), "
                                  "each pass still reads 40 GB"),
                       "parallelism": "single" if world == 1 else f"row-shard x{world}"},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "frac_of_measured_copy_ceiling": achieved / HBM_COPY_GBS,
                         "frac_of_measured_read_ceiling": achieved / HBM_READ_GBS,
                         "traffic": traffic,
                         "algorithmic_bytes_per_launch": algo_bytes, "per_kernel": per_kernel,
                         "iteration_frac": 2 * algo_bytes * iters_per_s / 1e9 / HBM_PEAK_GBS},
            "setup_s": t_setup, "final_objective": res.objective, "lasso_optimum": prob.optimum,
        }
        if args.exploit_sparsity:
            out["note"] = ("x-sparsity exploited (A*x skips columns with x_j == 0; nnz(x) = %d of %d at the end): this "
                           "is NOT the BASELINE dense metric; roofline figures assume the dense byte count"
                           % (int(np.count_nonzero(x)), args.n))
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args.n, args.m, args.cpu_rows, args.cpu_iters, args.seed)
        print(json.dumps(out), flush=True)

    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
