#!/usr/bin/env python3
"""The dual SVM of experiments/dual_svm/runme.jl:47-76 at one dataset size (N x N dense Q) for rocprofv3:
`python3 tools/dual_svm_prof.py N [iterations]` (mushrooms: N = 8124, Q = 528 MB)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import adaprox_amd as apx  # noqa: E402

N = int(sys.argv[1])
K = int(sys.argv[2]) if len(sys.argv) > 2 else 300
rng = np.random.default_rng(0)
X = rng.uniform(-1, 1, (N, 112))
yl = np.sign(X @ rng.standard_normal(112) + 0.1 * rng.standard_normal(N))
Q, q, Ay = (yl[:, None] * X) @ (yl[:, None] * X).T, -np.ones(N), yl[None, :].copy()
s = apx.Solver(np.zeros(N), np.zeros(1), f=apx.Quadratic(Q, q), g=apx.IndBox(0.0, 0.1), h=apx.IndZero(), A=Ay,
               rule=apx.OurRule(t=1.0, norm_A=float(np.sqrt(N))), tol=0.0, poll_every=1 << 30)
s.step(K, sync=True)
s.close()
