#!/usr/bin/env python3
"""BASELINE config C3 (logistic 100 000 x 10 000 fp32) on the one-pass kernel (csrc/slab.hip) for rocprofv3:
`rocprofv3 --kernel-trace --stats -- python3 tools/slab_prof.py [iterations]`, or `--pmc FETCH_SIZE` for the bytes read."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import adaprox_amd as apx  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
m, n = 100_000, 10_000
ctx = apx.Context(0)
A = apx.DeviceMatrix.uniform(m, n, seed=1, dtype=np.float32, ctx=ctx).build_transpose()
y = (np.random.default_rng(0).uniform(size=m) < 0.5).astype(np.float32)
s = apx.Solver(np.zeros(n + 1, dtype=np.float32), f=apx.LogisticLoss(A, y, ctx=ctx), g=apx.NormL1(1e-4),
               rule=apx.OurRule(gamma=4.0 / (n / 3.0 + 1.0)), tol=0.0, ctx=ctx, poll_every=1 << 30)
s.step(K, sync=True)
s.close()
