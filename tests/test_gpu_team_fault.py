"""The team kernels need all 256 workgroups of a launch resident together (csrc/slab.hip k_team_rows).  When they are not —
a GPU shared with other work — some teams miss members.  The design's promise: every wait is bounded (20 ms of wall clock),
the grid DRAINS, and the host gets APX_ERR_HIP instead of a wrong iterate.  APX_TEAM_FAULT_GRID launches 248 of the 256
workgroups (31 per XCD: one team of each XCD is a member short); one fresh child process, which must finish in seconds."""
import os
import subprocess
import sys
import time

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys, time
sys.path.insert(0, %r)
import numpy as np
import adaprox_amd as apx
ctx = apx.Context(0)
ctx.one_pass_threshold(16 << 20)
m, n = 4300, 12000
A = apx.DeviceMatrix.uniform(m, n, seed=5, ctx=ctx).build_transpose()
b = np.random.default_rng(0).standard_normal(m)
f = apx.LinearLeastSquares(A, b, ctx=ctx)
t0 = time.perf_counter()
try:
    apx.adaptive_proxgrad(np.zeros(n), f=f, g=apx.NormL1(1.0), rule=apx.OurRule(gamma=1e-7), tol=0.0, maxit=3, ctx=ctx)
    print("NO ERROR")
except apx.ApxError as e:
    print("ERROR:", e)
print("seconds %%.2f" %% (time.perf_counter() - t0))
"""


def test_missing_team_members_time_out_and_report():
    env = dict(os.environ, APX_TEAM_FAULT_GRID="248")
    t0 = time.perf_counter()
    out = subprocess.run([sys.executable, "-c", CHILD % ROOT], env=env, capture_output=True, text=True, timeout=120)
    dt = time.perf_counter() - t0
    assert out.returncode == 0, out.stderr[-2000:]
    assert "ERROR:" in out.stdout and "waited 20 ms" in out.stdout, out.stdout
    assert dt < 60.0                                     # (bounded waits: the grid drained)


def test_the_same_child_without_the_fault_is_fine():
    out = subprocess.run([sys.executable, "-c", CHILD % ROOT], env=dict(os.environ), capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "NO ERROR" in out.stdout, out.stdout + out.stderr[-1500:]
