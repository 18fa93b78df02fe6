"""How the rows of A are dealt to the workgroups of a one-pass launch (csrc/team_dev.h: slabs, or chunks of consecutive rows
dealt cyclically) — the functions the kernels call, compiled for the CPU with g++ and checked exhaustively: every row to exactly
one block, in order, clamped row numbers inside the block, at most one chunk of imbalance, and the chunk rule's values at the
BASELINE configs.  No GPU."""
import os
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_row_is_dealt_exactly_once():
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, "deal_check")
        subprocess.run(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "deal_check.cpp")], check=True)
        out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
        assert out.returncode == 0 and out.stdout.startswith("OK"), out.stdout[-2000:] + out.stderr[-500:]
