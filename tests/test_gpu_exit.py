"""The destroy entry points called by finalizers WHILE THE PROCESS EXITS (Julia runs every `finalizer` at exit:
julia/AdaProxHIP.jl registers apx_ctx_destroy / apx_matrix_free / apx_comm_destroy): before the library's exit handler
they free for real — in any order, the context is refcounted — and after it (when the HIP runtime may already be torn
down) they are no-ops.  Round 1 has an abort on record from exactly this pattern.  One fresh child
process per case; it must exit 0."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "adaptive-proximal-algorithms_amd", "csrc", "libadaprox_hip.so")


@pytest.fixture(scope="module")
def probe(tmp_path_factory):
    exe = tmp_path_factory.mktemp("exit") / "exit_probe"
    subprocess.run(["gcc", "-O1", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "exit_probe.c"), "-o",
                    str(exe), "-ldl"], check=True)
    return str(exe)


@pytest.mark.parametrize("mode,extra", [("atexit", []), ("destructor", []), ("atexit", ["comm"]), ("destructor", ["comm"])])
def test_finalizers_at_exit_are_safe(probe, mode, extra):
    if extra:      # the C program binds the SYSTEM's librccl (0.57 GB): read it sequentially first — on a box whose image is not cached
        import importlib      # the loader demand-pages it at ~2 MB/s (this case once took more than its 300 s: profiles/incidents/README.md)
        sharded = importlib.import_module("adaptive-proximal-algorithms_amd.sharded")
        took = sharded.prefetch_library_image(os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib", "librccl.so"))
        print("prefetched the system's librccl in %.1f s" % took)
    out = subprocess.run([probe, LIB, mode] + extra, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, f"{mode} {extra}: exit {out.returncode}\n{out.stdout}\n{out.stderr[-2000:]}"
    assert "exiting" in out.stdout


def test_python_atexit_destroys(probe):
    """the same from Python: the three destroys registered with `atexit` after the GPU was used"""
    code = r"""
import atexit, sys
sys.path.insert(0, %r)
import numpy as np
import adaprox_amd as apx
ctx = apx.Context(0)
A = apx.DeviceMatrix.uniform(300, 200, seed=1, ctx=ctx)
f = apx.LinearLeastSquares(A, np.zeros(300), ctx=ctx)
fx, pb = apx.eval_with_pullback(f, np.full(200, 0.01))
lib = ctx._lib
atexit.register(lib.apx_problem_destroy, f._op.h)     # (atexit is LIFO: the context goes FIRST, its children after it)
atexit.register(lib.apx_matrix_free, A._h)
atexit.register(lib.apx_ctx_destroy, ctx._h)
ctx._h = A._h = None; f._op.h = None                   # (the atexit handlers own the handles now)
print("ok", fx)
""" % ROOT
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-2000:]
