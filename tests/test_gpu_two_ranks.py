"""The N > 1 code path with TWO REAL RANKS on a one-GPU box: both ranks share device 0 and the
collectives go through gloo (RCCL refuses two ranks on one device), everything else is the
production path of bench.py — per-rank device generation of the row shard, all-reduced set-up,
the native loop calling the all-reduce hook once per gradient, replicated x / prox / step-size
state.  The sharded run must reproduce the single-rank run of the same global matrix."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(args, env_extra, nproc):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **env_extra)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    if nproc == 1:
        cmd = [sys.executable, "bench.py"] + args
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
               "--master-addr", "127.0.0.1", "--master-port", "29571", "bench.py", "--gpus", str(nproc)] + args
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]          # ONE JSON line, printed by rank 0 only
    return json.loads(lines[0])


def test_two_ranks_reproduce_one_rank():
    common = ["--cols", "6000", "--steps", "25", "--warmup", "5", "--no-cpu-baseline"]
    one = _bench(["--rows", "4000"] + common, {}, 1)
    two = _bench(["--rows", "2000"] + common, {"APX_BENCH_SHARE_GPU": "1"}, 2)
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2 and two["scaling"] == "weak"
    assert two["lasso_optimum"] == pytest.approx(one["lasso_optimum"], rel=1e-12)       # same global problem
    assert two["final_objective"] == pytest.approx(one["final_objective"], rel=1e-10)   # same 30 iterations
    assert two["config"]["parallelism"] == "row-shard x2"


def test_strong_scaling_uneven_split():
    """`--strong`: --rows is the global row count; 4001 rows over two ranks = 2001 + 2000."""
    common = ["--rows", "4001", "--cols", "6000", "--steps", "25", "--warmup", "5", "--no-cpu-baseline"]
    one = _bench(common, {}, 1)
    two = _bench(common + ["--strong"], {"APX_BENCH_SHARE_GPU": "1"}, 2)
    assert two["scaling"] == "strong" and two["n_gpus"] == 2
    assert two["lasso_optimum"] == pytest.approx(one["lasso_optimum"], rel=1e-12)
    assert two["final_objective"] == pytest.approx(one["final_objective"], rel=1e-10)
    assert two["value"] == pytest.approx(1e3 / two["ms_per_step"], rel=1e-9)     # job iterations/s, not x world


def test_row_sharded_primal_dual():
    """A_lin row-sharded over two real ranks (SURVEY §8e "PD (C5-style)"): y, A_x and the dual
    residual are shards, A'y and the residual norms are all-reduced inside the native loop; the
    l1 (clamp) and l2-ball conjugate proxes and both the fixed and the adaptive rule must reproduce
    the single-rank iterates."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
           "127.0.0.1", "--master-port", "29573", os.path.join(ROOT, "tests", "two_rank_pd_worker.py")]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-1500:])
    assert out.stdout.count("PD-SHARDED") == 2
