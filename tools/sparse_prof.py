#!/usr/bin/env python3
"""One sparse shape for rocprofv3 --kernel-trace --stats: `python3 tools/sparse_prof.py M N PER_ROW`."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import adaprox_amd as apx  # noqa: E402
from sparse_bench import planted_labels, rand_csr  # noqa: E402

m, n, per_row = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
A = rand_csr(m, n, per_row, 0, np.float64)
dA = apx.DeviceMatrix.from_sparse(A)
y = planted_labels(A, np.float64)      # (coin-flip labels make x = 0 optimal at this lambda: every iteration after the first a no-op)
f = apx.LogisticLoss(dA, y)
s = apx.Solver(np.zeros(n + 1), f=f, g=apx.NormL1(0.01), rule=apx.OurRule(gamma=float(4.0 * m / (A.multiply(A).sum() + m))),
               tol=0.0, poll_every=1 << 30)
s.step(50, sync=True)
s.close()
