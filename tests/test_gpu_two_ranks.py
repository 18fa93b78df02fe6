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


def test_four_ranks_reproduce_one_rank():
    """"until then validate numerics with P logical shards on one device" — four real ranks, each with its own
    context, stream and row shard on device 0 (the box allows six processes on the card ; eight is the driver's to run)."""
    common = ["--cols", "6000", "--steps", "25", "--warmup", "5", "--no-cpu-baseline"]
    one = _bench(["--rows", "4400"] + common, {}, 1)
    four = _bench(["--rows", "1100"] + common, {"APX_BENCH_SHARE_GPU": "1"}, 4)
    assert four["n_gpus"] == 4 and four["config"]["parallelism"] == "row-shard x4"
    assert four["lasso_optimum"] == pytest.approx(one["lasso_optimum"], rel=1e-12)
    assert four["final_objective"] == pytest.approx(one["final_objective"], rel=1e-10)


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
    assert out.stdout.count("PD-SHARDED failed set-up on rank 1") == 3 and "one-pass launches [0, 0]" in out.stdout, out.stdout[-2500:]
    assert out.stdout.count("PD-SHARDED") == 9 and "PD-SHARDED AdaPDM+ launches [12, 12]" in out.stdout, out.stdout[-1500:]
    assert "PD-SHARDED Malitsky-Pock" in out.stdout, out.stdout[-1500:]
    assert "PD-SHARDED one-pass l1 launches [15, 15]" in out.stdout and "PD-SHARDED one-pass l2 launches [15, 15]" in out.stdout, out.stdout[-1500:]


def _own_ranks(rows, cols, env_extra):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", APX_BENCH_SHARE_GPU="1", **env_extra)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    common = ["--cols", str(cols), "--steps", "10", "--warmup", "3", "--no-cpu-baseline"]
    out = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--rows", str(rows)] + common, cwd=ROOT, env=env,
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    two = json.loads(lines[0])
    assert two["n_gpus"] == 2 and two["config"]["parallelism"] == "row-shard x2"
    assert "launching 2 ranks" in out.stderr
    one = _bench(["--rows", str(2 * rows)] + common, {}, 1)
    assert two["lasso_optimum"] == pytest.approx(one["lasso_optimum"], rel=1e-12)
    assert two["final_objective"] == pytest.approx(one["final_objective"], rel=1e-10)
    return two, out.stderr


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it (the shape of the driver's command): the script starts its two
    ranks itself, before anything has touched the GPU, and rank 0 prints ONE JSON line with n_gpus = 2 — the same 13
    iterations as one rank on the whole matrix.  Launcher only: the shards (4200 x 6000 fp64, one workgroup per row) are NOT on
    the team kernels' path, which two processes sharing one GPU cannot have (the outcome must not depend on how
    their launches happen to interleave)."""
    two, _ = _own_ranks(4200, 6000, {})
    assert two["roofline"]["passes_over_A_per_iter"] == 1.0 and "note_team_kernels" not in two, two


def test_a_team_failure_on_one_rank_is_a_fallback_on_both():
    """The multi-rank fallback, deterministically: rank 1 launches 248 of its team kernels' 256 workgroups (the test
    hook of tests/test_gpu_team_fault.py, through APX_BENCH_RANK_OPTIONS) with a 50 ms bound, so ITS hand-offs fail whatever
    rank 0's do.  The library sums the error word over the ranks where it reads it: both ranks get the failure from the same
    call, both report it, both repeat warm-up and timed steps on the two GEMV passes — the collectives stay matched, the job
    ends with a number and the line says which path it timed."""
    two, err = _own_ranks(4200, 12000, {"APX_BENCH_RANK_OPTIONS": "1:TEAM_FAULT_GRID=248,TEAM_WAIT_MS=50"})
    assert two["roofline"]["passes_over_A_per_iter"] == 2.0 and "note_team_kernels" in two, two
    assert "bench.py rank 0:" in err and "bench.py rank 1:" in err, err[-3000:]
    assert "a hand-off wait expired" in err, err[-3000:]


def test_ranks_agree_on_the_path_when_only_one_shard_is_eligible():
    """8191 rows over two ranks = 4096 + 4095: the first shard is eligible for the one-pass kernels (>= 4096 rows), the second is
    not.  The two paths enqueue different collectives, so each rank deciding for itself would hang or corrupt the run: the
    ranks agree on the first evaluation (csrc/slab.hip slab_eval_with_gradient) and both take the two passes."""
    common = ["--rows", "8191", "--cols", "12000", "--steps", "10", "--warmup", "3", "--no-cpu-baseline"]
    one = _bench(common, {}, 1)
    two = _bench(common + ["--strong"], {"APX_BENCH_SHARE_GPU": "1"}, 2)
    assert two["n_gpus"] == 2 and two["roofline"]["passes_over_A_per_iter"] == 2.0, two["roofline"]
    assert two["final_objective"] == pytest.approx(one["final_objective"], rel=1e-10)
