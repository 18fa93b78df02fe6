#!/usr/bin/env python3
"""One small-path configuration for rocprofv3 --kernel-trace --stats: `python3 tools/small_prof.py M N [graph]`."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import adaprox_amd as apx  # noqa: E402

m, n = int(sys.argv[1]), int(sys.argv[2])
graph = len(sys.argv) > 3 and sys.argv[3] == "graph"
rng = np.random.default_rng(0)
A = rng.uniform(-1, 1, (m, n))
b = A @ np.where(rng.uniform(size=n) < 0.1, rng.standard_normal(n), 0.0) + 0.1 * rng.standard_normal(m)
gam = 1.0 / np.linalg.norm(A, 2) ** 2
s = apx.Solver(np.zeros(n), f=apx.LinearLeastSquares(A, b), g=apx.NormL1(1.0), rule=apx.OurRule(gamma=gam),
               tol=0.0, use_graph=graph, poll_every=1 << 30)
s.step(100, sync=True)
t0 = time.perf_counter()
s.step(1000, sync=True)
print("us/iteration", (time.perf_counter() - t0) / 1000 * 1e6)
s.close()
