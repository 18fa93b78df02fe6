"""What ONE rank of the row-sharded run costs at the shard sizes of the 1 -> 8 GPU curve: bench.py through the
N-rank code path with a single rank (APX_BENCH_FORCE_DIST=1: the in-library communicator, one ncclAllReduce(n + 2) per iteration,
the sharded loop's kernels) at rows-per-rank 50 000 / 25 000 / 12 500 / 6 250 x 100 000 — strong scaling of C2 over P = 1, 2, 4, 8 ;
50 000 is also the weak-scaling shard (C4).  Beside each: the plain single-GPU run of the same shard.  One JSON object per line
in gpurun_out/<tag>_shard_sizes.jsonl and the summary profiles/... is made from it (tools/shard_sizes.py --summarize)."""
import json
import os
import subprocess
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
out_path = os.path.join(ROOT, "gpurun_out", f"{tag}_shard_sizes.jsonl")
os.makedirs(os.path.dirname(out_path), exist_ok=True)
rows_list = [int(v) for v in os.environ.get("SHARD_ROWS", "50000,25000,12500,6250").split(",")]
with open(out_path, "w") as fo:
    for rows in rows_list:
        for dist in (True, False):
            env = dict(os.environ)
            if dist:
                env["APX_BENCH_FORCE_DIST"] = "1"
            else:
                env.pop("APX_BENCH_FORCE_DIST", None)
            cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--rows", str(rows), "--cols", "100000", "--steps", "200", "--warmup", "10",
                   "--no-cpu-baseline", "--no-other-configs"]
            r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
            lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
            if r.returncode != 0 or len(lines) != 1:
                rec = dict(rows_per_rank=rows, n_rank_path=dist, error=r.stderr[-800:])
            else:
                d = json.loads(lines[0])
                rec = dict(rows_per_rank=rows, n_rank_path=dist, iters_per_s=d["value"], ms_per_step=d["ms_per_step"],
                           kernel=d["roofline"]["kernel"], kernel_GBps=d["roofline"]["achieved"], frac=d["roofline"]["frac"],
                           passes=d["roofline"]["passes_over_A_per_iter"], per_kernel=d["roofline"]["per_kernel"],
                           collective=d.get("collective"), layout=d["config"]["layout"], lib_srchash=d["lib_srchash"])
            print(json.dumps(rec), flush=True)
            fo.write(json.dumps(rec) + "\n")
            fo.flush()
