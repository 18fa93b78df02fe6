import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import adaprox_amd as apx
from oracle import adaprox_oracle as O, problems as P
prob = P.random_lasso(500, 1000, pfactor=10, seed=0)
gam = 1.0/P.opnorm_sq_power(prob.A, iters=50)
orecs, drecs, tr = [], [], []
O.adaptive_proxgrad(np.zeros(1000), f=O.LinearLeastSquares(prob.A,prob.b), g=O.NormL1(1.0), rule=O.MalitskyMishchenkoRule(gamma=gam), tol=0.0, maxit=60, record=orecs.append, trace=tr)
f = apx.LinearLeastSquares(prob.A, prob.b)
apx.adaptive_proxgrad(np.zeros(1000), f=f, g=apx.NormL1(1.0), rule=apx.MalitskyMishchenkoRule(gamma=gam), tol=0.0, maxit=60, record=drecs.append)
for o,d in zip(orecs,drecs):
    print(o['it'], '%.16e %.16e'%(o['gamma'],d['gamma']), 'dg %.2e dres %.2e dobj %.2e'%(abs(o['gamma']-d['gamma'])/o['gamma'], abs(o['norm_res']-d['norm_res'])/o['norm_res'], abs(o['objective']-d['objective'])/abs(o['objective'])))
