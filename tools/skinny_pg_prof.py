"""Kernel-level view of the tall-skinny proximal-gradient iteration (csrc/skinny_pg.hip): run under
`rocprofv3 --kernel-trace --stats -- python3 tools/skinny_pg_prof.py`."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import __graft_entry__ as ge  # noqa: E402

ge.build()
import adaprox_amd as apx  # noqa: E402


def main():
    rng = np.random.default_rng(0)
    for m, n, dt in ((8124, 112, np.float32), (11055, 68, np.float32), (100000, 20, np.float64)):
        X = rng.uniform(-1, 1, (m, n)).astype(dt)
        y = (rng.uniform(size=m) < 0.5).astype(dt)
        gam = 4.0 * m / (np.linalg.norm(X) ** 2 + m)
        s = apx.Solver(np.zeros(n + 1, dtype=dt), f=apx.LogisticLoss(X, y), g=apx.NormL1(0.01), rule=apx.OurRule(gamma=gam),
                       tol=0.0, poll_every=1 << 30)
        s.step(3000, sync=True)
        s.close()


if __name__ == "__main__":
    main()
