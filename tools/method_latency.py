#!/usr/bin/env python3
"""µs per iteration of every solver of the reference's lasso experiment (experiments/lasso/runme.jl:86-160)
at its own size 500 x 1000 and of the LAD experiment's methods at 506 x 14 — which loops are device-resident
and which are host-driven.  `python tools/method_latency.py` on an MI355X; one JSON line per method."""
import importlib
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import adaprox_amd as apx  # noqa: E402

problems = importlib.import_module("adaptive-proximal-algorithms_amd.problems")


def timed(name, fn, K):
    fn(20)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        out = fn(K)
        best = min(best, time.perf_counter() - t0)
        assert int(out[-1]) == K, (name, out[-1])       # (tol = 0: every iteration ran)
    print(json.dumps({"method": name, "us_per_iteration": round(best / K * 1e6, 1)}), flush=True)


def main():
    K = 400
    prob = problems.device_lasso(500, 1000, pfactor=10, seed=0, power_iters=100)
    gam = 1 / prob.Lf
    f, g, x0 = apx.LinearLeastSquares(prob.A, prob.b), apx.NormL1(prob.lam), np.zeros(1000)
    kw = dict(f=f, g=g, tol=0.0)
    timed("lasso 500x1000  PGM (fixed)", lambda k: apx.fixed_proxgrad(x0, gamma=gam, maxit=k, **kw), K)
    timed("lasso 500x1000  AdaPGM (Ours)", lambda k: apx.adaptive_proxgrad(x0, rule=apx.OurRule(gamma=gam), maxit=k, **kw), K)
    timed("lasso 500x1000  AdaPGM (MM)", lambda k: apx.adaptive_proxgrad(x0, rule=apx.MalitskyMishchenkoRule(gam), maxit=k, **kw), K)
    for xi in (1, 1.5, 2):
        timed(f"lasso 500x1000  PGM (backtracking, xi={xi})",
              lambda k: apx.backtracking_proxgrad(x0, gamma0=gam, xi=xi, maxit=k, **kw), K)
    timed("lasso 500x1000  Nesterov (backtracking)", lambda k: apx.backtracking_nesterov(x0, gamma0=gam, maxit=k, **kw), K)
    timed("lasso 500x1000  Nesterov (fixed)", lambda k: apx.fixed_nesterov(x0, gamma=gam, maxit=k, **kw), K)
    timed("lasso 500x1000  aGRAAL", lambda k: apx.agraal(x0, gamma0=gam, maxit=k, rng=np.random.default_rng(0), **kw), K)

    rng = np.random.default_rng(1)
    m, n = 506, 14
    A = np.hstack([rng.uniform(-1, 1, (m, n - 1)), np.ones((m, 1))])
    y = A @ rng.standard_normal(n) + rng.laplace(size=m)
    nrm = float(np.linalg.norm(A))
    dA = apx.DeviceMatrix.from_numpy(A)
    t = dict(f=apx.Zero(), g=apx.NormL1(10.0), h=apx.Translate(apx.NormL1(1.0), -y), A=dA, tol=0.0)
    x0, y0 = np.zeros(n), np.zeros(m)
    timed("LAD 506x14  Condat-Vu", lambda k: apx.condat_vu(x0, y0, Lf=0.0, norm_A=nrm, maxit=k, **t), K)
    timed("LAD 506x14  AdaPDM", lambda k: apx.adaptive_primal_dual(x0, y0, rule=apx.OurRule(t=1.0, norm_A=nrm), maxit=k, **t), K)
    timed("LAD 506x14  AdaPDM+", lambda k: apx.adaptive_linesearch_primal_dual(x0, y0, eta=nrm, maxit=k, **t), K)
    for mm, nn in ((4177, 9), (8192, 13)):
        A2 = np.hstack([rng.uniform(-1, 1, (mm, nn - 1)), np.ones((mm, 1))])
        y2 = A2 @ rng.standard_normal(nn) + rng.laplace(size=mm)
        t2 = dict(f=apx.Zero(), g=apx.NormL1(10.0), h=apx.Translate(apx.NormL1(1.0), -y2), A=apx.DeviceMatrix.from_numpy(A2),
                  tol=0.0)
        n2 = float(np.linalg.norm(A2))
        timed(f"LAD {mm}x{nn}  AdaPDM+", lambda k: apx.adaptive_linesearch_primal_dual(np.zeros(nn), np.zeros(mm), eta=n2,
                                                                                        maxit=k, **t2), K)
    timed("LAD 506x14  Malitsky-Pock", lambda k: apx.malitsky_pock(x0, y0, sigma=1.0 / nrm, maxit=k, **t), K)

    # the dual SVM's three methods (experiments/dual_svm/runme.jl:62-104) at svmguide3's size: Q is 1243 x 1243, A = y'
    N, nf = 1243, 22
    Xs = rng.uniform(-1, 1, (N, nf))
    ys = np.sign(Xs @ rng.standard_normal(nf) + 0.1 * rng.standard_normal(N))
    Q, qv, Ay = (ys[:, None] * Xs) @ (ys[:, None] * Xs).T, -np.ones(N), ys[None, :].copy()
    ts = dict(f=apx.Quadratic(Q, qv), g=apx.IndBox(0.0, 0.1), h=apx.IndZero(), A=apx.DeviceMatrix.from_numpy(Ay), tol=0.0)
    nA, xs0, ys0 = float(np.sqrt(N)), np.zeros(N), np.zeros(1)
    timed("dual SVM N=1243  AdaPDM", lambda k: apx.adaptive_primal_dual(xs0, ys0, rule=apx.OurRule(t=1.0, norm_A=nA), maxit=k, **ts), 300)
    timed("dual SVM N=1243  Malitsky-Pock", lambda k: apx.malitsky_pock(xs0, ys0, t=1.0, sigma=1 / nA, maxit=k, **ts), 300)
    timed("dual SVM N=1243  Condat-Vu", lambda k: apx.condat_vu(xs0, ys0, Lf=float(np.linalg.norm(Q)), norm_A=nA, maxit=k, **ts), 300)

    # the logistic experiment's methods at mushrooms' shape (experiments/sparse_logreg/runme.jl:66-124), dense copy
    m, n = 8124, 112
    X = (rng.uniform(size=(m, n)) < 0.19).astype(np.float64)
    w = rng.standard_normal(n) * (rng.uniform(size=n) < 0.3)
    yl = (1 / (1 + np.exp(-(X @ w))) > rng.uniform(size=m)).astype(np.float64)
    gam = 4.0 * m / (float(np.sum(X * X)) + m)
    f, g, x0 = apx.LogisticLoss(X, yl), apx.NormL1(0.01), np.zeros(n + 1)
    kw = dict(f=f, g=g, tol=0.0)
    K = 60
    timed("logreg 8124x112  AdaPGM (Ours)", lambda k: apx.adaptive_proxgrad(x0, rule=apx.OurRule(gamma=gam), maxit=k, **kw), K)
    for xi in (1, 2):
        timed(f"logreg 8124x112  PGM (backtracking, xi={xi})",
              lambda k: apx.backtracking_proxgrad(x0, gamma0=gam, xi=xi, maxit=k, **kw), K)
    timed("logreg 8124x112  Nesterov (backtracking)", lambda k: apx.backtracking_nesterov(x0, gamma0=gam, maxit=k, **kw), K)
    timed("logreg 8124x112  Nesterov (fixed)", lambda k: apx.fixed_nesterov(x0, gamma=gam, maxit=k, **kw), K)
    timed("logreg 8124x112  aGRAAL", lambda k: apx.agraal(x0, gamma0=gam, maxit=k, rng=np.random.default_rng(0), **kw), K)


if __name__ == "__main__":
    main()
