"""Checker script (not collected by pytest): drift of the two-kernel small path and of the generic kernels from the
oracle as the iteration count grows — `python tests/drift_probe.py` on an MI355X (see DESIGN.md, fp64 horizon)."""
import sys; sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import numpy as np
import adaprox_amd as apx
from oracle import adaprox_oracle as O, problems as P
prob = P.random_lasso(300, 500, pfactor=10, seed=2)
Lf = P.opnorm_sq_power(prob.A, iters=50)
g = apx.Context(0); g.fused_small(False)
Ag = apx.DeviceMatrix.from_numpy(prob.A, g)
for K in (10,20,30,40,50,60,80):
    xo,_ = O.adaptive_proxgrad(np.zeros(500), f=O.LinearLeastSquares(prob.A, prob.b), g=O.NormL1(1.0), rule=O.OurRule(gamma=1/Lf), tol=0.0, maxit=K)
    xf,_ = apx.adaptive_proxgrad(np.zeros(500), f=apx.LinearLeastSquares(prob.A, prob.b), g=apx.NormL1(1.0), rule=apx.OurRule(gamma=1/Lf), tol=0.0, maxit=K)
    xg,_ = apx.adaptive_proxgrad(np.zeros(500), f=apx.LinearLeastSquares(Ag, prob.b), g=apx.NormL1(1.0), rule=apx.OurRule(gamma=1/Lf), tol=0.0, maxit=K, ctx=g)
    r=lambda a,b: np.max(np.abs(a-b))/np.max(np.abs(b))
    print(K, "fused-oracle %.2e generic-oracle %.2e fused-generic %.2e"%(r(xf,xo), r(xg,xo), r(xf,xg)))
