"""`python bench.py --gpus N` outside a launcher must start its own N ranks — and must do so before torch (and with it the HIP
runtime) has been imported: a process that has initialised the GPU may neither fork ranks nor be replaced by another program
on the GPU pool.  CPU-only: the launch itself is intercepted."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import os, sys
sys.path.insert(0, %r)
sys.argv = ["bench.py", "--gpus", "%d", "--rows", "4000", "--cols", "12000", "--steps", "3"]
import bench
args = bench.parse()
seen = {}
def fake_run(cmd, env=None):
    seen["cmd"] = cmd
    seen["torch_imported"] = "torch" in sys.modules
    seen["legacy_ipc"] = env.get("HSA_ENABLE_IPC_MODE_LEGACY")
    return 0
try:
    launched = bench.maybe_relaunch(args, sys.argv[1:], run=fake_run)
    print("RETURNED", launched)
except SystemExit as e:
    print("EXIT", e.code)
print("CMD", " ".join(seen.get("cmd", [])))
print("TORCH", seen.get("torch_imported"), "IPC", seen.get("legacy_ipc"))
"""


def _run(gpus, env=None):
    e = dict(os.environ)
    e.pop("WORLD_SIZE", None)
    e.update(env or {})
    return subprocess.run([sys.executable, "-c", CHILD % (ROOT, gpus)], env=e, capture_output=True, text=True, timeout=120)


def test_gpus_n_launches_its_own_ranks_before_anything_touches_the_gpu():
    out = _run(2)
    assert out.returncode == 0, out.stderr[-1500:]
    assert "EXIT 0" in out.stdout, out.stdout                       # the launcher exits with the children's code
    cmd = [l for l in out.stdout.splitlines() if l.startswith("CMD")][0]
    assert "-m torch.distributed.run" in cmd and "--nproc-per-node=2" in cmd and "--master-addr 127.0.0.1" in cmd, cmd
    assert cmd.rstrip().endswith("bench.py --gpus 2 --rows 4000 --cols 12000 --steps 3"), cmd
    assert "TORCH False IPC 0" in out.stdout, out.stdout            # torch not imported when the ranks are started ; dmabuf IPC set


def test_one_gpu_and_ranks_do_not_relaunch():
    out = _run(1)
    assert out.returncode == 0 and "RETURNED False" in out.stdout, out.stdout + out.stderr[-800:]
    out = _run(2, {"WORLD_SIZE": "2"})                              # a rank of a launched job: straight on, no second launch
    assert out.returncode == 0 and "RETURNED False" in out.stdout and "CMD \n" in out.stdout + "\n", out.stdout + out.stderr[-800:]
