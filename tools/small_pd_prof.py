#!/usr/bin/env python3
"""One LAD shape on the two-kernel primal-dual path for rocprofv3: `python3 tools/small_pd_prof.py M N l1|l2`."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import adaprox_amd as apx  # noqa: E402

m, n, hk = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
rng = np.random.default_rng(1)
A = np.hstack([rng.uniform(-1, 1, (m, n - 1)), np.ones((m, 1))])
y = A @ rng.standard_normal(n) + rng.laplace(size=m)
h = apx.Translate(apx.NormL1(1.0) if hk == "l1" else apx.NormL2(1.0), -y)
s = apx.Solver(np.zeros(n), np.zeros(m), f=apx.Zero(), g=apx.NormL1(5.0), h=h, A=A,
               rule=apx.OurRule(t=1.0, norm_A=float(np.linalg.norm(A))), tol=0.0, poll_every=1 << 30)
s.step(1000, sync=True)
s.close()
