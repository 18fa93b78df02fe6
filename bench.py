#!/usr/bin/env python3
"""bench.py — headline benchmark of the adaptive proximal-gradient hot path on MI355X.

Metric (BASELINE.json): prox-grad iterations/sec of `adaptive_proxgrad` (OurRule) on the
dense lasso of experiments/lasso/runme.jl with A 50 000 x 100 000 fp64 (40.0 GB resident in
HBM), plus the fraction of the HBM roofline of the dominant GEMV kernel.

A "step" = ONE iteration of src/AdaProx.jl:334-362 = GEMV-N (A x - b, ||r||^2) + GEMV-T
(A' r) + the fused residual / step-size / soft-threshold kernels, on data already resident
in HBM.  N > 1 (one rank per GPU — launched by torch.distributed.run, or by THIS script: `python bench.py --gpus N`
without a launcher's environment starts the N ranks itself, before anything has touched a GPU): weak scaling — every
rank holds a 50 000-row shard of the (50 000 N) x 100 000 matrix (BASELINE config C4 at
N = 8) and the partial gradient A_k' r_k is all-reduced with RCCL over xGMI each iteration: ONE
ncclAllReduce(n + 2 fp64) per iteration, enqueued by the native loop itself on its stream through the communicator
the library owns (apx_comm_init_rank, csrc/comm.hip).  torch.distributed is the launcher only: a gloo group carries
the 128-byte RCCL id from rank 0, the barriers of the timing protocol and the max over ranks of the elapsed time.
`value` = shard-iterations per second summed over the ranks (= iterations/sec at N = 1).

At N = 1 the same JSON line also carries `other_configs`: BASELINE configs C3 (sparse_logreg 100 000 x 10 000 fp32)
and C5 (LAD 50 000 x 20 000 fp64, Condat-Vu) timed with the same protocol after the headline has released its memory.

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
    ap.add_argument("--layout", choices=("rows", "dual", "cols"), default="rows",
                    help="storage of A: rows = ONE resident copy, row-contiguous (40 GB ; default) ; dual = column-major + the row copy "
                         "(2 x 40 GB: rounds 1-4) ; cols = column-major only (the two GEMV passes)")
    ap.add_argument("--single-layout", action="store_true", help="= --layout cols (kept for older command lines)")
    ap.add_argument("--strong", action="store_true",
                    help="strong scaling: --rows is the GLOBAL row count, split over the ranks (default: weak, rows per GPU)")
    # CPU baseline sample: 10 000 x 100 000 fp64 (8 GB) when the host has >= 32 GB of RAM, else 4000 rows
    ap.add_argument("--cpu-rows", type=int, default=0, help="rows of the CPU-baseline sample (0: 10000, or 4000 on a small host)")
    ap.add_argument("--cpu-seconds", type=float, default=8.0, help="time budget of each CPU-baseline leg")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the C3 / C5 legs of the N = 1 run")
    ap.add_argument("--two-pass", action="store_true", help="NOT the default product path: the two GEMV passes of round 1 (apx_ctx_one_pass(ctx, 0))")
    ap.add_argument("--gemv-t-variant", type=int, default=-1, help="tile shape of the column-streaming kernel of the two-pass iteration (tools/bench_variants.sh)")
    ap.add_argument("--option", action="append", default=[], metavar="KEY=VALUE",
                    help="NOT the default product path: a context option by name (include/adaprox_hip.h APX_OPT_*, e.g. FUSED_TAIL=0) — A/B runs ; flagged in the JSON line")
    return ap.parse_args()


def host_ram_gb() -> float:
    try:
        return os.sysconf("SC_PAGE_SIZE") * os.sysconf("SC_PHYS_PAGES") / 2**30
    except (ValueError, OSError):
        return 0.0


def host_cpu_share() -> int:
    """Cores this process may actually use: the scheduler affinity, capped by the cgroup CPU quota (a GPU box gives a
    one-GPU job a share of its host — 128 threads on a 16-core quota only fight each other)."""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, q // p))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def cpu_baseline(n: int, m_full: int, rows: int, seconds: float, seed: int) -> dict:
    """The CPU restatement of the reference iteration timed on the host cores on a bounded row sample of the same
    workload (same generator), scaled per byte to the full matrix.  Two legs: C + OpenMP on every host thread
    (oracle/lasso_omp.c — the figure reported as `value`) and the numpy oracle (-> OpenBLAS dgemv, the BLAS family
    Julia's `A*x` reaches; its dgemv barely threads).  A reported baseline, not the target."""
    import tempfile
    from oracle import adaprox_oracle as O
    from oracle import lasso_omp
    if rows <= 0:
        rows = 10_000 if host_ram_gb() >= 32.0 else 4_000
    share = host_cpu_share()
    os.environ["OMP_NUM_THREADS"] = str(share)           # (libgomp reads it when the library below is loaded)
    # rebuilt HERE with -march=native: this leg is about this host's cores
    lib = lasso_omp.load(lasso_omp.build(native=True, out=os.path.join(tempfile.mkdtemp(prefix="apx_omp_"), "liblasso_omp.so")))
    lib.apx_oracle_omp_set_threads(share)
    threads = int(lib.apx_oracle_omp_threads())
    t0 = time.perf_counter()
    prob = lasso_omp.random_lasso(rows, n, pfactor=10, seed=seed, lib=lib)
    t_gen = time.perf_counter() - t0
    sample_bytes = float(rows) * n * 8
    # Lf by power iteration on A'A (reference: opnorm(A)^2, experiments/lasso/runme.jl:81)
    v = np.ones(n) / np.sqrt(n)
    for _ in range(5):
        w = lasso_omp.gemv_t(prob.A, lasso_omp.gemv_n(prob.A, v, lib), lib)
        Lf = float(np.linalg.norm(w))
        v = w / Lf
    gam = 1.0 / Lf

    def leg(run) -> tuple:
        run(2)                                             # warm-up (page faults, thread pool)
        t0 = time.perf_counter()
        run(3)
        per_it = (time.perf_counter() - t0) / 4            # 3 loop passes + the pre-loop gradient
        iters = max(3, min(200, int(seconds / per_it)))
        t0 = time.perf_counter()
        run(iters)
        dt = time.perf_counter() - t0
        return iters, dt, (iters + 1) / dt                 # iterations/s on the sample

    it_c, dt_c, ips_c = leg(lambda k: lasso_omp.adaptive_proxgrad_lasso(prob.A, prob.b, prob.lam, gam, k, lib=lib))
    f, g = O.LinearLeastSquares(prob.A, prob.b), O.NormL1(prob.lam)
    try:
        from threadpoolctl import threadpool_info, threadpool_limits
        with threadpool_limits(limits=share):              # (OpenBLAS sized for the host, not for this job's share, otherwise)
            it_p, dt_p, ips_p = leg(lambda k: O.adaptive_proxgrad(np.zeros(n), f=f, g=g, rule=O.OurRule(gamma=gam), tol=0.0, maxit=k))
            blas_threads = max([int(i.get("num_threads", 1)) for i in threadpool_info() if i.get("user_api") == "blas"] or [1])
    except ImportError:
        it_p, dt_p, ips_p = leg(lambda k: O.adaptive_proxgrad(np.zeros(n), f=f, g=g, rule=O.OurRule(gamma=gam), tol=0.0, maxit=k))
        blas_threads = os.cpu_count() or 1
    scale = rows / m_full
    return {
        "value": ips_c * scale, "unit": "iters/sec", "cores": threads, "kind": "port",
        "host_GBps": 2 * sample_bytes * ips_c / 1e9, "host_cpu_share": share, "host_logical_cpus": os.cpu_count(),
        "sample": f"C + OpenMP restatement of the reference iteration (oracle/lasso_omp.c, {threads} threads = this job's CPU share of "
                  f"the host's {os.cpu_count()} logical CPUs) on a {rows}x{n} "
                  f"fp64 lasso of the same generator ({sample_bytes / 1e9:.1f} GB, built in {t_gen:.1f} s): {it_c} iterations "
                  f"in {dt_c:.2f} s = {ips_c:.2f} it/s on the sample = {2 * sample_bytes * ips_c / 1e9:.0f} GB/s of A streamed, "
                  f"scaled by bytes ({rows}/{m_full}) to the {m_full}x{n} matrix; Julia runtime absent on this image",
        "numpy_openblas": {"value": ips_p * scale, "cores": blas_threads, "host_GBps": 2 * sample_bytes * ips_p / 1e9,
                           "sample": f"numpy oracle (OpenBLAS dgemv, {blas_threads} threads) on the same sample: {it_p} "
                                     f"iterations in {dt_p:.2f} s"},
    }


def relaunch_command(argv: list, gpus: int, port: int) -> list:
    """`python bench.py --gpus N` outside a launcher: the command that starts the N ranks (one per GPU, RCCL between them)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def free_port() -> int:
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return int(s.getsockname()[1])


def maybe_relaunch(args, argv: list, run=None) -> bool:
    """With --gpus N > 1 and no launcher's environment (WORLD_SIZE unset) this process becomes the LAUNCHER: it starts
    `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a child, passes its output through and exits with its
    code.  It happens before torch is imported and before any HIP call — a process that has initialised the GPU must neither
    fork ranks nor be replaced by another program on this pool — and the children find WORLD_SIZE set, so they do not come
    through here again.  Returns False when there is nothing to launch (this process is a rank, or N = 1)."""
    if args.gpus <= 1 or "WORLD_SIZE" in os.environ:
        return False
    assert "torch" not in sys.modules, "bench.py: the launcher must not have imported torch"
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # (dmabuf IPC: RCCL between processes needs it on this pool)
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = relaunch_command(argv, args.gpus, free_port())
    print("bench.py: launching %d ranks: %s" % (args.gpus, " ".join(cmd)), file=sys.stderr, flush=True)
    rc = (run or subprocess.call)(cmd, env=env)
    sys.exit(int(rc))


class _StdoutToStderr:
    """Rank 0 prints ONE JSON line on stdout.  Gloo and RCCL announce themselves from C (`[Gloo] Rank 0 is connected ...`,
    `RCCL version : ...`) on file descriptor 1 when their groups are created: while that happens, descriptor 1 points at
    descriptor 2."""

    def __enter__(self):
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self._saved, 1)
        os.close(self._saved)
        return False


def kernel_table(stats: dict) -> dict:
    out = {}
    for k, st in stats.items():
        if st["launches"]:
            avg_ms = st["total_ms"] / st["launches"]
            out[k] = {"launches": st["launches"], "avg_ms": avg_ms,
                      "GBps": st["bytes"] / st["launches"] / (avg_ms * 1e-3) / 1e9}
    return out


def timed_steps(ctx, solver, steps: int, warmup: int, barrier) -> tuple:
    """W untimed steps, then exactly K steps between two barrier + device-synchronise brackets, with HIP events
    around every pass over A on the launch stream (apx_ctx_profile)."""
    solver.step(warmup, sync=True)
    ctx.kernel_stats(reset=True)
    ctx.profile(True)
    barrier()
    t0 = time.perf_counter()
    solver.step(steps, sync=False)
    barrier()
    dt = time.perf_counter() - t0
    ctx.profile(False)
    return dt, ctx.kernel_stats(reset=True)


def backtracking_leg(apx, prob, ctx, barrier, m: int, n: int, iters: int = 100, xi: float = 1.5) -> dict:
    """`backtracking_proxgrad` (src/AdaProx.jl:34-64) on the headline's matrix while it is resident: xi = 1.5, the middle one of
    the reference's settings (experiments/lasso/runme.jl:96-108), gamma0 = 1/Lf.  ONE call of `iters` iterations — its upload
    of x0, the evaluation in front of the loop and the read-back included ; the linesearch is host-driven (four scalars read
    back per trial: the accept test is a data-dependent branch).  Every trial evaluates f at the trial point, and
    on the one-pass kernels that evaluation leaves the gradient too, so an accepted trial's gradient is free: passes over A
    per iteration = trials per iteration, bytes read per iteration = trials x m n sizeof(T)."""
    f = apx.Counting(apx.LinearLeastSquares(prob.A, prob.b))
    g = apx.NormL1(prob.lam)
    kw = dict(f=f, g=g, gamma0=1.0 / prob.Lf, xi=xi, tol=0.0, ctx=ctx)
    apx.backtracking_proxgrad(np.zeros(n), maxit=3, **kw)                 # warm-up
    e0 = f.eval_count
    barrier()
    ctx.kernel_stats(reset=True)
    ctx.profile(True)
    recs = []
    t0 = time.perf_counter()
    z, it = apx.backtracking_proxgrad(np.zeros(n), maxit=iters, record=recs.append, **kw)
    barrier()
    dt = time.perf_counter() - t0
    ctx.profile(False)
    k = kernel_table(ctx.kernel_stats(reset=True))
    trials = f.eval_count - e0 - 1                                        # (one evaluation at x0 in front of the loop, :52)
    launches = sum(v["launches"] for v in k.values())
    dom = max(k, key=lambda q: k[q]["avg_ms"] * k[q]["launches"])
    its = iters / dt
    real = (int(it) == iters and bool(np.all(np.isfinite(z))) and trials >= iters and len(recs) == iters
            and recs[-1]["objective"] < recs[0]["objective"])
    return {"workload": f"lasso, dense A {m}x{n} fp64, backtracking_proxgrad (xi = {xi}, shrink = 0.5, gamma0 = 1/Lf) on 1 MI355X",
            "iters_per_s": its, "ms_per_iter": dt / iters * 1e3, "steps": int(it), "linesearch_trials": int(trials),
            "trials_per_iter": trials / iters, "passes_over_A_per_iter": launches / iters, "kernels": k,
            "roofline": {"bound": "hbm", "kernel": ctx.kernel_name(dom), "achieved": k[dom]["GBps"], "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": k[dom]["GBps"] / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": float(m) * n * 8,
                         # of the bytes the whole call reads: (1 + n_bt) m n sizeof(T) per iteration, n_bt rejected trials
                         "bytes_read_iteration_frac": launches * float(m) * n * 8 / dt / 1e9 / HBM_PEAK_GBS,
                         # SURVEY 8d's own model for this loop: (2 + n_bt) m n sizeof(T) per iteration
                         "two_pass_equivalent_frac": (launches + iters) * float(m) * n * 8 / dt / 1e9 / HBM_PEAK_GBS},
            "objective": recs[-1]["objective"], "gamma": recs[-1]["gamma"], "timed_steps_all_real": real,
            "note": "one CALL of %d iterations (upload, the evaluation in front of the loop and the read-back included) ; every "
                    "linesearch trial is one launch of the one-pass kernel, which leaves f AND grad f at the trial point: "
                    "an accepted trial's gradient costs nothing more" % iters}


def other_configs(apx, problems, ctx, barrier) -> dict:
    """BASELINE configs C3 and C5 at full size, same protocol as the headline (data resident, warm-up, K timed
    iterations, HIP-event durations of the passes over A).  Each reports the fraction of the HBM peak twice: of the
    bytes the dominant kernel actually READS per launch, and of the two-pass algorithmic bytes SURVEY 8d defines
    (2 m n sizeof(T) per iteration: above 1 where one pass over A serves both products)."""
    out = {}

    def report(m, n, es, steps, dt, stats, res, one_pass_note):
        k = kernel_table(stats)
        dom = max(k, key=lambda q: k[q]["avg_ms"] * k[q]["launches"])
        its = steps / dt
        passes = sum(v["launches"] for v in k.values()) / steps       # passes over A per iteration, as launched
        r = {"iters_per_s": its, "ms_per_iter": dt / steps * 1e3, "steps": steps,
             "passes_over_A_per_iter": passes, "kernels": k,
             "roofline": {"bound": "hbm", "kernel": dom, "achieved": k[dom]["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                          "frac": k[dom]["GBps"] / HBM_PEAK_GBS,
                          "algorithmic_bytes_per_launch": float(m) * n * es,
                          "two_pass_equivalent_frac": 2.0 * m * n * es * its / 1e9 / HBM_PEAK_GBS,
                          "bytes_read_iteration_frac": passes * m * n * es * its / 1e9 / HBM_PEAK_GBS},
             "objective": res.objective, "norm_res": res.norm_res}
        if passes < 1.5:
            r["note"] = one_pass_note
        r["roofline"]["kernel"] = ctx.kernel_name(dom)           # (the kernels as the library launched them)
        return r

    # C3: sparse_logreg, dense synthetic X 100 000 x 10 000 fp32, logistic gradient + NormL1 prox
    m, n = 100_000, 10_000
    p = problems.device_logreg(m, n, seed=0, lam=0.01, dtype=np.float32, ctx=ctx)
    p.X.drop_primary()                                          # one resident copy: the row storage the one-pass kernels read
    # This problem — SURVEY 8d's lambda, step size and generator — reaches the residual 2e-7 at iteration 27 and is solved to sixteen
    # digits by iteration 40 ; past that the iterations do the same work on the same bytes but their objective says nothing (VERDICT
    # r4).  So the 150 timed iterations are SIX solves of 25 iterations from x0 = 0, every one of them ahead of convergence: six
    # solvers created beforehand (set-up and the evaluation in front of the loop are outside the timed region, as for every leg),
    # 10 warm-up iterations on a seventh, then step(25) on each between the two barriers.
    mk = lambda: apx.Solver(np.zeros(n + 1, dtype=np.float32), f=apx.LogisticLoss(p.X, p.y), g=apx.NormL1(p.lam),      # noqa: E731
                            rule=apx.OurRule(gamma=p.gamma0), tol=0.0, ctx=ctx, poll_every=1 << 30)
    warm = mk()
    warm.step(10, sync=True)
    warm.close()
    solvers = [mk() for _ in range(6)]
    ctx.kernel_stats(reset=True)
    ctx.profile(True)
    barrier()
    t0 = time.perf_counter()
    for s in solvers:
        s.step(25, sync=False)
    barrier()
    dt = time.perf_counter() - t0
    ctx.profile(False)
    st = ctx.kernel_stats(reset=True)
    ends = [s.result() for s in solvers]
    x, _, res = ends[-1]
    real = all(np.all(np.isfinite(xx)) and int(rr.numit) == 25 and not rr.converged for xx, _, rr in ends) and \
        len({float(rr.objective) for _, _, rr in ends}) == 1 and res.norm_res > 0.0
    if not real:
        raise RuntimeError("C3 leg: a solve stopped early, left the finite numbers or differs from the others (%s): the timed steps were not all real"
                           % [(int(rr.numit), bool(rr.converged), float(rr.objective), float(rr.norm_res)) for _, _, rr in ends])
    out["C3"] = report(m, n, 4, 150, dt, st, res,
                       "f and grad f in ONE pass over X per iteration (csrc/slab.hip): "
                       "two_pass_equivalent_frac prices SURVEY 8d's 2 m n sizeof(T) and may exceed 1; frac is of the bytes read")
    out["C3"]["workload"] = "sparse_logreg, dense synthetic A 100000x10000 fp32, logistic grad + NormL1 prox, adaptive_proxgrad (OurRule)"
    out["C3"]["timed_steps_all_real"] = True
    out["C3"]["protocol"] = "six solves of 25 iterations from x0 = 0 (all ahead of convergence: residual %.2e at their end), bit-identical objectives" % res.norm_res
    for s in solvers:
        s.close()
    p.X.free()

    # C5: least_absolute_deviation via adaptive_primal_dual (Condat-Vu), A 50 000 x 20 000 fp64
    m, n = 50_000, 20_000
    p = problems.device_lad(m, n, seed=0, lam=10.0, ctx=ctx)
    p.A.drop_primary()
    gamma = 1.0 / (0.0 / 2 + p.norm_A / 1.0)                    # condat_vu with Lf = 0 (src/AdaProx.jl:405-411)
    sigma = 0.99 / (p.norm_A * 1.0)
    s = apx.Solver(np.zeros(n), np.zeros(m), f=apx.Zero(), g=apx.NormL1(p.lam), h=apx.Translate(apx.NormL1(1.0), -p.y),
                   A=p.A, rule=apx.FixedStepsize(gamma, np.sqrt(sigma / gamma)), tol=0.0, ctx=ctx, poll_every=1 << 30)
    dt, st = timed_steps(ctx, s, 200, 10, barrier)
    x, _, res = s.result()
    if not np.all(np.isfinite(x)) or int(res.numit) != 210 or res.converged:
        raise RuntimeError("C5 leg: the loop stopped after %d iterations: the timed steps were not all real" % int(res.numit))
    out["C5"] = report(m, n, 8, 200, dt, st, res,
                       "A x, the dual step and A' y in ONE pass over A per iteration (csrc/slab_pd.hip, csrc/team_rows.h)")
    out["C5"]["workload"] = "least_absolute_deviation via adaptive_primal_dual (Condat-Vu), A 50000x20000 fp64 (Ax / A'y pair + conj-prox)"
    s.close()
    # Two more loops on C5's matrix while it is resident (not BASELINE configs: the north star's third prox and the method the
    # reference's LAD / square-root-lasso experiments showcase ; a failure here is reported, not raised):
    try:
        # the l2 ball (experiments/square_root_lasso/runme.jl:39-41) in one pass over A — lambda 0.5: at LAD's 10 x would stay 0
        s = apx.Solver(np.zeros(n), np.zeros(m), f=apx.Zero(), g=apx.NormL1(0.5), h=apx.Translate(apx.NormL2(1.0), -p.y), A=p.A,
                       rule=apx.OurRule(t=1.0, norm_A=p.norm_A), tol=0.0, ctx=ctx, poll_every=1 << 30)
        dt, st = timed_steps(ctx, s, 100, 10, barrier)
        x, _, res = s.result()
        ok = bool(np.all(np.isfinite(x))) and int(res.numit) == 110 and not res.converged and bool(np.any(x != 0.0))
        out["C5_matrix_l2_ball"] = report(m, n, 8, 100, dt, st, res, "the l2 ball's conjugate prox is linear in the rows once the norm is known: "
                                          "A' z and sum z^2 in ONE pass over A, scaled by the closing kernel (DESIGN 4, round 4)")
        out["C5_matrix_l2_ball"]["workload"] = "square-root-lasso terms (h = l2 ball) on C5's matrix, adaptive_primal_dual (OurRule)"
        out["C5_matrix_l2_ball"]["timed_steps_all_real"] = ok
        s.close()
        # AdaPDM+ (src/AdaProx.jl:463-550): host-driven linesearch, the first trial of an iteration takes A x along
        kw = dict(f=apx.Zero(), g=apx.NormL1(p.lam), h=apx.Translate(apx.NormL1(1.0), -p.y), A=apx.Counting(p.A), eta=p.norm_A, t=1.0,
                  tol=0.0, ctx=ctx)
        apx.adaptive_linesearch_primal_dual(np.zeros(n), np.zeros(m), maxit=5, **kw)
        barrier()
        ctx.kernel_stats(reset=True)
        ctx.profile(True)
        recs = []
        t0 = time.perf_counter()
        x, _, it = apx.adaptive_linesearch_primal_dual(np.zeros(n), np.zeros(m), maxit=100, record=recs.append, **kw)
        barrier()
        dt = time.perf_counter() - t0
        ctx.profile(False)
        k = kernel_table(ctx.kernel_stats(reset=True))
        out["C5_adapdm_plus"] = {"workload": "least_absolute_deviation via adaptive_linesearch_primal_dual (AdaPDM+), A 50000x20000 fp64",
                                 "iters_per_s": 100 / dt, "ms_per_iter": dt * 10.0, "steps": int(it),
                                 "linesearch_trials": int(recs[-1]["At_evals"]) - 1, "kernels": k, "objective": recs[-1]["objective"],
                                 "note": "one CALL of 100 iterations, its upload / two products in front of the loop / read-back included ; "
                                         "the first linesearch trial of an iteration leaves A x and A' y_next from one read of A, "
                                         "a rejected trial costs one GEMV-T more (DESIGN 4, round 4)"}
    except (Exception, SystemExit) as e:           # noqa: BLE001
        out["C5_matrix_extra_error"] = f"{type(e).__name__}: {e}"
    p.A.free()
    try:
        out["sparse_logreg_1Mx100k"] = sparse_leg(apx, ctx, barrier)
    except (Exception, SystemExit) as e:           # noqa: BLE001
        out["sparse_logreg_1Mx100k"] = {"error": f"{type(e).__name__}: {e}"}
    return out


def sparse_leg(apx, ctx, barrier, m: int = 1_000_000, n: int = 100_000, per_row: int = 60, steps: int = 100) -> dict:
    """The orientation the reference's real datasets use (experiments/sparse_logreg/runme.jl:24,31: `f.X * w`, `f.X' * r` on a
    SparseMatrixCSC, Float64): l1-logistic regression on a SPARSE X, here 1 000 000 x 100 000 with ~60 entries per row (60 M stored
    entries: 720 MB of values + indices per orientation — beyond the Infinity Cache, so both products stream from HBM ; their operand
    is gathered from 64 KB windows staged in LDS, csrc/sparse.hip spmv_tiles).  Not a BASELINE config.  Algorithmic bytes per product:
    nnz * (sizeof(T) + 4)."""
    import scipy.sparse as sp
    rng = np.random.default_rng(0)
    nnz_row = np.clip(rng.poisson(per_row, m), 1, n)
    indptr = np.concatenate([[0], np.cumsum(nnz_row)]).astype(np.int64)
    X = sp.csr_matrix((rng.uniform(-1, 1, indptr[-1]), rng.integers(0, n, indptr[-1]).astype(np.int32), indptr), shape=(m, n))
    X.sum_duplicates()
    w = np.random.default_rng(3).standard_normal(n)
    y = (1.0 / (1.0 + np.exp(-(X @ w))) > np.random.default_rng(4).uniform(size=m)).astype(np.float64)
    gam = 4.0 * m / (float(X.multiply(X).sum()) + m)
    dX = apx.DeviceMatrix.from_sparse(X, ctx)
    nnz = int(X.nnz)
    del X
    # (lambda: below the gradient's entries at x = 0, ~2e-5 here — with 1e-4 only the intercept ever left zero: the same kernels at the
    #  same cost, but `X w` of a zero w)
    s = apx.Solver(np.zeros(n + 1), f=apx.LogisticLoss(dX, y, ctx=ctx), g=apx.NormL1(1e-6), rule=apx.OurRule(gamma=gam), tol=0.0, ctx=ctx,
                   poll_every=1 << 30)
    dt, st = timed_steps(ctx, s, steps, 10, barrier)
    x, _, res = s.result()
    ok = bool(np.all(np.isfinite(x))) and int(res.numit) == steps + 10 and not res.converged and bool(np.any(x[:-1] != 0.0))
    k = kernel_table(st)
    dom = max(k, key=lambda q: k[q]["avg_ms"] * k[q]["launches"])
    names = {q: ctx.kernel_name(q) for q in k}
    s.close()
    dX.free()
    per_pass = float(nnz) * 12.0
    traffic = None      # (the committed PMC passes of this command, like the main line's: profiles/pmc_traffic.json)
    try:
        if (m, n, per_row) == (1_000_000, 100_000, 60) and "spmv_tiles" in names[dom]:
            traffic = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json"))).get("spmv_tiles", {}).get("hbm_bytes_per_launch")
    except Exception:      # noqa: BLE001
        traffic = None
    return {"workload": f"sparse_logreg on a sparse X {m}x{n}, {nnz} stored entries, fp64 values + int32 indices, adaptive_proxgrad (OurRule)",
            "iters_per_s": steps / dt, "ms_per_iter": dt / steps * 1e3, "steps": steps, "kernels": k, "kernel_names": names,
            "roofline": {"bound": "hbm", "kernel": names[dom], "achieved": k[dom]["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": k[dom]["GBps"] / HBM_PEAK_GBS, "traffic": traffic, "algorithmic_bytes_per_launch": per_pass,
                         "iteration_frac": 2.0 * per_pass * steps / dt / 1e9 / HBM_PEAK_GBS,
                         "note": "values + indices per product (the entries' 12 bytes ; the tiles' padding, the operand's windows — read "
                                 "once per tile, from L2 — and the splits written are excluded).  The ceiling of the access pattern "
                                 "(stream + one gather per entry from LDS) is 5.4 - 5.9 TB/s: profiles/r05_gather_probe.log"},
            "objective": res.objective, "timed_steps_all_real": ok}


def main():
    args = parse()
    if args.single_layout:
        args.layout = "cols"
    maybe_relaunch(args, sys.argv[1:])               # (--gpus N > 1 outside a launcher: this process starts the ranks and exits)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        args.gpus = world
    # ---- the transport of the row-sharded run: decided from the environment alone, BEFORE anything touches the GPU ----
    #   "rccl"   (default with more than one rank; APX_BENCH_FORCE_DIST=1 runs it with a single rank — the 1-GPU box
    #             rehearsal): the library's own communicator, ncclAllReduce enqueued by the native loop (csrc/comm.hip);
    #             torch.distributed (gloo, CPU) only launches, carries the id and synchronises the clocks
    #   "gloo"   (APX_BENCH_SHARE_GPU=1): every rank on device 0, the all-reduce CALLBACK over gloo — rehearsal of the
    #             N-rank logic on a one-GPU box (RCCL refuses two ranks on one device); its timings mean nothing
    #   "torch-nccl" (APX_BENCH_TRANSPORT=torch-nccl): the round-1 path, the callback over torch's nccl backend
    share_gpu = os.environ.get("APX_BENCH_SHARE_GPU") == "1"
    use_dist = world > 1 or os.environ.get("APX_BENCH_FORCE_DIST") == "1"
    transport = "none"
    if use_dist:
        transport = "gloo" if share_gpu else os.environ.get("APX_BENCH_TRANSPORT", "rccl")
        if transport not in ("rccl", "gloo", "torch-nccl"):
            sys.exit(f"bench.py: unknown APX_BENCH_TRANSPORT {transport!r}")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # (dmabuf IPC: RCCL between processes needs it here)

    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    dev = 0 if share_gpu else local_rank
    torch.cuda.set_device(dev)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        with _StdoutToStderr():
            if transport == "torch-nccl":
                dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev))
            else:
                dist.init_process_group(backend="gloo")

    # a library built from other sources than the ones in the tree is rebuilt (content hash), never benchmarked
    import __graft_entry__
    if local_rank == 0 and __graft_entry__.is_stale():
        __graft_entry__.build()              # in-tree hipcc build (the .so is git-ignored)
    if use_dist:
        dist.barrier()
    if __graft_entry__.is_stale():
        sys.exit("bench.py: libadaprox_hip.so does not match the sources in the tree")
    import adaprox_amd as apx
    from importlib import import_module
    problems = import_module("adaptive-proximal-algorithms_amd.problems")
    sharded = import_module("adaptive-proximal-algorithms_amd.sharded")

    m_global = args.m if args.strong else args.m * world
    row0 = rank * args.m
    if args.strong:
        row0, args.m = sharded.shard_rows(m_global, rank, world)   # this rank's first row and row count
    stream = torch.cuda.Stream(device=dev)
    rccl_ranks = None
    other = None
    with torch.cuda.stream(stream):
        ctx = apx.Context(device=dev, stream=stream.cuda_stream)
        apx.set_default_context(ctx)
        comm = None
        ar_host = ar_dev = None
        if transport == "rccl":
            # If the in-library communicator cannot be set up on ANY rank (librccl not loadable, ncclCommInitRank refusing
            # the environment), every rank falls back to the callback transport over the launcher's gloo group — slow (the
            # 0.8 MB travel through the host) but a valid row-sharded iteration, and the JSON line says so — instead of
            # the job dying without a number.  (A rank that HANGS inside RCCL's bootstrap cannot be helped from here.)
            rccl_error = ""
            try:
                with _StdoutToStderr():
                    # (librccl is 0.3 GB of device code: on a box whose image is not cached yet the loader demand-pages it at
                    #  ~2 MB/s — minutes ; a sequential read first takes seconds.  sharded.Comm does it: said here because it is
                    #  the first thing a fresh N-GPU node spends time on)
                    if os.environ.get("APX_BENCH_SIMULATE_RCCL_FAILURE") == "1":      # (test hook: tests/test_gpu_comm.py)
                        raise RuntimeError("simulated failure of the communicator set-up")
                    uid = [sharded.Comm.unique_id() if rank == 0 else None]      # ncclGetUniqueId on rank 0 ...
                    dist.broadcast_object_list(uid, src=0)                       # ... carried to every rank by the launcher's group
                    comm = sharded.Comm.init_rank(ctx, world, rank, uid[0])      # ncclCommInitRank on this rank's device
                    rccl_ranks = comm.nranks                                     # (ncclCommCount of the live communicator)
                    comm(np.zeros(1))                                            # (the first collective: RCCL's banner comes with it)
                if rccl_ranks != world:
                    raise RuntimeError(f"the communicator has {rccl_ranks} ranks, the job {world}")
            except Exception as e:      # noqa: BLE001 — whatever went wrong, the ranks must agree on what to do next
                rccl_error = f"{type(e).__name__}: {e}"
                print(f"bench.py rank {rank}: RCCL inside the library is not available: {rccl_error}", file=sys.stderr)
            bad = torch.tensor([1 if rccl_error else 0], dtype=torch.int32)
            dist.all_reduce(bad, op=dist.ReduceOp.MAX)
            if int(bad.item()):
                if comm is not None:
                    comm.close()
                comm, rccl_ranks = None, None
                transport = "gloo-fallback"
                ar_host = sharded.host_allreduce(dist, None)
                ar_dev = sharded.device_allreduce(dist, dev)
            else:
                ar_host = ar_dev = comm
        elif use_dist:
            ar_host = sharded.host_allreduce(dist, dev)
            ar_dev = sharded.device_allreduce(dist, dev)

        t_setup = time.perf_counter()
        prob = problems.device_lasso(args.m, args.n, pfactor=10, seed=args.seed, rank=rank, world=world,
                                     power_iters=30, ctx=ctx, allreduce_host=ar_host,
                                     dual_layout=args.layout == "dual", layout="rows" if args.layout == "rows" else "cols",
                                     m_total=m_global, row_offset=row0)
        t_setup = time.perf_counter() - t_setup

        if args.exploit_sparsity:
            ctx.exploit_sparsity(True)
        if args.two_pass:
            ctx.one_pass(0)
        if args.gemv_t_variant >= 0:
            ctx.set_gemv_config(t_variant=args.gemv_t_variant)
        for kv in args.option:
            key, val = kv.split("=")
            ctx.set_option(getattr(import_module("adaptive-proximal-algorithms_amd._lib"), "APX_OPT_" + key), int(val))
        # (test hook, tests/test_gpu_two_ranks.py: options on ONE rank only — "1:TEAM_FAULT_GRID=248,TEAM_WAIT_MS=50" makes rank 1's
        #  team hand-offs fail deterministically, so that the ranks' agreement on the fallback is exercised, not hoped for)
        rank_opts = os.environ.get("APX_BENCH_RANK_OPTIONS", "")
        if rank_opts and int(rank_opts.split(":", 1)[0]) == rank:
            for kv in rank_opts.split(":", 1)[1].split(","):
                key, val = kv.split("=")
                ctx.set_option(getattr(import_module("adaptive-proximal-algorithms_amd._lib"), "APX_OPT_" + key), int(val))
        g = apx.NormL1(prob.lam)

        def make_solver():
            f = apx.LinearLeastSquares(prob.A, prob.b, allreduce=ar_dev)
            return apx.Solver(np.zeros(args.n), f=f, g=g, rule=apx.OurRule(gamma=1.0 / prob.Lf), tol=0.0, ctx=ctx,
                              poll_every=1 << 30)

        def barrier():
            torch.cuda.synchronize()
            if use_dist:
                dist.barrier()
            torch.cuda.synchronize()

        # W warm-up steps, then the K timed ones.  The one-pass kernels for rows of this length (teams of workgroups) need the GPU
        # to themselves — all 256 workgroups of a launch resident together ; a hand-off that fails (a GPU shared with other
        # work) invalidates the run.  On a row shard the library sums the error word over the ranks wherever it reads it, so
        # every rank gets the failure from the same call ; the ranks then agree ONCE MORE over the launcher's group (any kind
        # of failure, on any rank) before anybody raises or moves on: nobody is left inside a collective.  After a team
        # failure every rank repeats warm-up and timed steps on the two GEMV passes, and the JSON line says so
        # (`note_team_kernels`, with the diagnostic record of the wave that gave up) instead of the job dying without a number.
        def agree(flag: int) -> int:
            if not use_dist:
                return flag
            t = torch.tensor([flag], dtype=torch.int32, device=f"cuda:{dev}" if transport == "torch-nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return int(t.item())

        def attempt():
            """-> (status, payload): 0 = timed (dt, stats, x, res) ; 1 = a team hand-off failed (message) ; 2 = another error"""
            solver = None
            try:
                solver = make_solver()
                solver.step(args.warmup, sync=True)
                solver.result()                              # (reads the error word behind the warm-up)
                dt_, stats_ = timed_steps(ctx, solver, args.steps, 0, barrier)
                x_, _, res_ = solver.result()
                return 0, (dt_, stats_, x_, res_)
            except apx.ApxError as e:
                return (1 if "one-pass kernel (teams)" in str(e) or "failed team hand-off" in str(e) else 2), str(e)
            finally:
                if solver is not None:
                    try:
                        solver.close()
                    except apx.ApxError:
                        pass

        team_failure, team_incident = 0, ""
        status, payload = attempt()
        worst = agree(status)
        if worst == 1:
            team_failure = 1
            team_incident = payload if status == 1 else apx.last_incident()
            print(f"bench.py rank {rank}: {team_incident or 'a team hand-off failed on another rank'}", file=sys.stderr)
            # (every rank is at the same point: the library's own agreement made the failure surface from the same call on all
            #  of them — after the warm-up, before the first barrier, or after the timed steps, behind the second)
            ctx.one_pass(0)                                  # every rank: the two GEMV passes
            status, payload = attempt()
            worst = agree(status)
        if worst != 0:
            sys.exit(f"bench.py rank {rank}: {payload if status != 0 else 'another rank failed'}")
        dt, stats, x, res = payload
        if not np.all(np.isfinite(x)):
            sys.exit("bench.py: non-finite iterate")
        if int(res.numit) != args.warmup + args.steps or res.converged:
            sys.exit("bench.py: the loop stopped after %d iterations: the timed steps were not all real" % int(res.numit))
        dual_layout = prob.dual_layout
        layout = prob.layout
        kernel_names = {k: ctx.kernel_name(k) for k in ("gemv_n", "gemv_t", "one_pass")}
        final_objective, optimum = res.objective, prob.optimum
        if world == 1 and not use_dist and not args.no_other_configs and (args.m, args.n) == (50_000, 100_000):
            bt = None
            try:                                           # the second of north_star's three entry points, on the headline's matrix
                bt = backtracking_leg(apx, prob, ctx, barrier, args.m, args.n)
            except (Exception, SystemExit) as e:           # noqa: BLE001
                bt = {"error": f"{type(e).__name__}: {e}"}
                print(f"bench.py: the backtracking leg failed: {bt['error']}", file=sys.stderr)
            prob.A.free()                                  # give the matrix back before the other configs
            try:                                           # (a side leg that fails must not take the headline line with it)
                other = other_configs(apx, problems, ctx, barrier)
            except (Exception, SystemExit) as e:           # noqa: BLE001
                other = {"error": f"{type(e).__name__}: {e}"}
                print(f"bench.py: other_configs failed: {other['error']}", file=sys.stderr)
            other["C2_backtracking"] = bt

    if use_dist:
        tmax = torch.tensor([dt], dtype=torch.float64, device=f"cuda:{dev}" if transport == "torch-nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    if rank == 0:
        es = 8
        per_kernel = kernel_table(stats)
        dom = max(per_kernel, key=lambda k: per_kernel[k]["avg_ms"] * per_kernel[k]["launches"])
        algo_bytes = float(args.m) * args.n * es                 # one streaming read of A per launch
        achieved = per_kernel[dom]["GBps"]
        iters_per_s = args.steps / dt
        # passes over A per iteration as launched: 2 = the two GEMVs of the reference's cost model (lasso/runme.jl:179-180) ;
        # 1 = f and its gradient from ONE read of A (csrc/slab.hip: k_team_rows at this row length)
        passes = sum(v["launches"] for v in per_kernel.values()) / args.steps
        one_pass = passes < 1.5
        traffic = traffic_source = None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        # HBM bytes per launch from the PMC passes (separate `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` runs of THIS
        # command, FETCH_SIZE doubled per the gfx950 correction of MI355X_MICROARCH.md; profiles/README.md): a counter
        # run cannot be nested inside the timed run, so the line quotes the committed passes and says which
        if os.path.exists(tpath) and (args.m, args.n) == (50_000, 100_000) and not share_gpu:
            try:
                # with the dual layout both passes run the column-streaming kernel (the "gemv_t" entry)
                tj = json.load(open(tpath))
                tkey = "k_team_rows" if one_pass else ("gemv_t" if dual_layout or layout == "rows" else dom)      # (keys of profiles/pmc_traffic.json)
                traffic = tj.get(tkey, {}).get("hbm_bytes_per_launch")
                traffic_source = tj.get("source", "profiles/pmc_traffic.json") + f" [{tkey}]"
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
                       "layout": (("single, %.0f GB: A resident ONCE, in the row-contiguous storage the one-pass kernels read"
                                   % (algo_bytes / 1e9)
                                   + ("" if one_pass else " ; two GEMV passes per iteration, both served from it")) if layout == "rows" else
                                  "single: A column-major, 40 GB" if not dual_layout else
                                  "dual: A and its transposed copy resident (2 x 40 GB of the 288 GB)"
                                  + (", the iteration reads the transposed copy ONCE (40 GB)" if one_pass
                                     else ", each pass still reads 40 GB")),
                       "parallelism": "single" if world == 1 else f"row-shard x{world}"},
            "roofline": {"bound": "hbm", "kernel": kernel_names.get(dom) or dom,      # (as the library launched it: apx_ctx_kernel_name)
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "frac_of_measured_copy_ceiling": achieved / HBM_COPY_GBS,
                         "frac_of_measured_read_ceiling": achieved / HBM_READ_GBS,
                         "traffic": traffic, "traffic_source": traffic_source,
                         "algorithmic_bytes_per_launch": algo_bytes, "per_kernel": per_kernel,
                         "passes_over_A_per_iter": passes,
                         # SURVEY 8d prices an iteration at B_iter = 2 m n sizeof(T) (two streaming reads of A): of THAT figure
                         "iteration_frac": 2 * algo_bytes * iters_per_s / 1e9 / HBM_PEAK_GBS,
                         "bytes_read_iteration_frac": passes * algo_bytes * iters_per_s / 1e9 / HBM_PEAK_GBS},
            "setup_s": t_setup, "final_objective": final_objective, "lasso_optimum": optimum,
            "lib_srchash": __graft_entry__.source_hash()[:16],
        }
        if team_failure:
            out["note_team_kernels"] = ("a hand-off of the one-pass team kernels failed on at least one rank (the GPU was shared with other work): warm-up "
                                        "and timed steps were repeated on the two GEMV passes by every rank. " + team_incident)
        if one_pass:
            out["roofline"]["note"] = (
                "one launch forms A x, the residual AND A' r from ONE read of A (teams of 16 workgroups per slab of rows): "
                "`achieved` / `frac` price the bytes the launch reads (m n sizeof(T) = algorithmic_bytes_per_launch, confirmed by "
                "`traffic`); the two-GEMV cost model of SURVEY 8d (2 m n sizeof(T) per iteration) is `iteration_frac`, which "
                "exceeds 1 because half of those bytes are no longer read — not because the HBM peak was beaten")
        if use_dist:
            out["collective"] = {
                "transport": transport, "rccl_ranks": rccl_ranks,
                "what": ("ncclAllReduce(sum, %d fp64) per iteration, enqueued by the native loop on its stream "
                         "(apx_comm_init_rank, csrc/comm.hip)" % (args.n + 2)) if transport == "rccl" else
                        ("FALLBACK: the communicator inside the library could not be set up (see stderr); all-reduce callback "
                         "over the launcher's gloo group, through the host — these timings are not the RCCL path's"
                         if transport == "gloo-fallback" else
                         "all-reduce callback of the test transport (not RCCL inside the library)")}
            out["rccl_ranks"] = rccl_ranks
        if args.option or args.two_pass or args.gemv_t_variant >= 0:
            out["non_default_options"] = {"option": args.option, "two_pass": args.two_pass, "gemv_t_variant": args.gemv_t_variant}
        if args.exploit_sparsity:
            out["note"] = ("x-sparsity exploited (A*x skips columns with x_j == 0; nnz(x) = %d of %d at the end): this "
                           "is NOT the BASELINE dense metric; roofline figures assume the dense byte count"
                           % (int(np.count_nonzero(x)), args.n))
        if other is not None:
            out["other_configs"] = other
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args.n, args.m, args.cpu_rows, args.cpu_seconds, args.seed)
        print(json.dumps(out), flush=True)

    if comm is not None:
        comm.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
