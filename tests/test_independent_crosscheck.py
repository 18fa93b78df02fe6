"""INDEPENDENT checks of end states and of the third-party prox arithmetic (ProximalOperators /
ProximalCore are un-vendored and unpinned, the reference holds no iterate vectors, Julia is absent — so neither the
oracle nor the HIP path can be compared with the reference's own numbers).  What CAN be done without the reference:

  * every term family the experiments use is solved by a THIRD-PARTY solver that shares no code with the oracle or the
    HIP path, and the solvers here must reach that solution:
      - lasso                              sklearn coordinate descent     (experiments/lasso/runme.jl:83-84)
      - least absolute deviation           scipy linprog / HiGHS, as an LP (experiments/least_absolute_deviation/runme.jl:39-94)
      - l1-logistic, PENALISED bias last   sklearn liblinear on [X 1], no intercept (experiments/sparse_logreg/runme.jl:18-39,56)
      - square-root lasso                  scipy L-BFGS-B on the split x = p - q  (experiments/square_root_lasso/runme.jl:39-94)
      - dual-SVM box QP + equality         scipy SLSQP                    (experiments/dual_svm/runme.jl:47-76)
  * and a solver-free certificate: the KKT conditions / the duality gap of the returned (x, y), evaluated in numpy
    from the problem data alone (-A'y in dg(x), y in dh(Ax) ; P(x) - D(y) >= 0 with equality only at a solution);
  * each prox the oracle restates "from its mathematical definition" is checked against a brute-force minimisation of
    that definition, argmin_y g(y) + |y - x|^2 / (2 gamma) (scipy, one coordinate at a time or over the whole vector).

The same bodies run on the CPU oracle (`-m "not gpu"`) and on the HIP path (`-m gpu`): the small cases take the
device-resident small-size kernels, the `one_pass` cases the row-streaming kernels the BASELINE configs run on."""
import math

import numpy as np
import pytest
from scipy.optimize import linprog, minimize, minimize_scalar
from sklearn.linear_model import Lasso, LogisticRegression

from oracle import adaprox_oracle as O

END_TOL = 1e-6          # each family reaches the third-party solution to <= 1e-6


@pytest.fixture(scope="module")
def apx():
    import adaprox_amd as apx
    apx.load()
    return apx


def _backend(request, which):
    return O if which == "oracle" else request.getfixturevalue("apx")


BACKENDS = [pytest.param("oracle", id="oracle"), pytest.param("hip", id="hip", marks=pytest.mark.gpu)]


# ---------------------------------------------------------------------------------------------------------------------
# lasso — sklearn coordinate descent
# sklearn minimises (1/(2m))||y - Xw||^2 + alpha*||w||_1, i.e. alpha = lambda/m for the reference's
# 0.5*||Aw - b||^2 + lambda*||w||_1 (experiments/lasso/runme.jl:83-84).
def _sklearn_lasso(A, b, lam):
    m = A.shape[0]
    mdl = Lasso(alpha=lam / m, fit_intercept=False, tol=1e-14, max_iter=200_000, selection="cyclic")
    mdl.fit(A, b)
    return mdl.coef_


def lasso_kkt(A, b, lam, w):
    """max violation of  -A'(Aw - b) in lam * d||w||_1."""
    g = A.T @ (A @ w - b)
    nz = w != 0
    v_nz = float(np.max(np.abs(g[nz] + lam * np.sign(w[nz])))) if nz.any() else 0.0
    v_z = float(np.max(np.maximum(np.abs(g[~nz]) - lam, 0.0))) if (~nz).any() else 0.0
    return max(v_nz, v_z)


def test_oracle_vs_sklearn_lasso():
    rng = np.random.default_rng(0)
    A = rng.standard_normal((120, 60))                     # overdetermined: unique minimiser
    b = A @ (rng.standard_normal(60) * (rng.uniform(size=60) < 0.3)) + 0.1 * rng.standard_normal(120)
    lam = 3.0
    w_ref = _sklearn_lasso(A, b, lam)
    gam = 1.0 / np.linalg.norm(A, 2) ** 2
    w, it = O.adaptive_proxgrad(np.zeros(60), f=O.LinearLeastSquares(A, b), g=O.NormL1(lam),
                                rule=O.OurRule(gamma=gam), tol=1e-10, maxit=100_000)
    assert it < 100_000
    assert np.max(np.abs(w - w_ref)) < 1e-8
    assert np.array_equal(w == 0, np.abs(w_ref) < 1e-12)   # same support
    assert lasso_kkt(A, b, lam, w) < 1e-8


@pytest.mark.gpu
def test_hip_vs_sklearn_lasso(apx):
    rng = np.random.default_rng(1)
    A = np.asfortranarray(rng.standard_normal((400, 150)))
    b = A @ (rng.standard_normal(150) * (rng.uniform(size=150) < 0.2)) + 0.1 * rng.standard_normal(400)
    lam = 5.0
    w_ref = _sklearn_lasso(A, b, lam)
    f, g = apx.LinearLeastSquares(A, b), apx.NormL1(lam)
    gam = 1.0 / apx.DeviceMatrix.from_numpy(A).opnorm_sq(iters=200)
    for solve in (lambda: apx.adaptive_proxgrad(np.zeros(150), f=f, g=g, rule=apx.OurRule(gamma=gam), tol=1e-10,
                                                maxit=100_000),
                  lambda: apx.backtracking_proxgrad(np.zeros(150), f=f, g=g, gamma0=gam, xi=1.5, tol=1e-10,
                                                    maxit=100_000),
                  lambda: apx.fixed_nesterov(np.zeros(150), f=f, g=g, gamma=gam, tol=1e-10, maxit=200_000),
                  lambda: apx.agraal(np.zeros(150), f=f, g=g, x0=np.full(150, 0.01), tol=1e-10, maxit=200_000)):
        w, it = solve()
        assert np.max(np.abs(w - w_ref)) < 1e-7
        assert lasso_kkt(A, b, lam, w) < 1e-6


# ---------------------------------------------------------------------------------------------------------------------
# least absolute deviation  min ||Ax - b||_1 + lam ||x||_1  — an LP (HiGHS) + the duality gap
# (experiments/least_absolute_deviation/runme.jl:39-41: f = Zero, g = NormL1(lam), h = Translate(NormL1(), -b),
#  A = [X 1]).  Dual:  max -b'y  s.t.  ||y||_inf <= 1, ||A'y||_inf <= lam  (h*(y) = b'y + ind, g*(u) = ind).
def lad_problem(m, n, seed, lam, noise="laplace"):
    """noise = "outliers": a tenth of the rows carry a gross error, the rest none — robust regression, what LAD is for ; the
    primal-dual methods identify that active set in a few thousand iterations, where Laplace noise on every row takes them
    1e5 - 1e6 at a few thousand rows (measured on the oracle: a rigorous gap of 1e-9 in 350 - 4100 iterations against 3e-4
    after 20 000).  Both states of the clamp occur: saturated on the outlier rows, interior on the clean ones."""
    rng = np.random.default_rng(seed)
    A = np.asfortranarray(np.hstack([rng.standard_normal((m, n - 1)), np.ones((m, 1))]))      # runme.jl:42
    xt = rng.standard_normal(n) * (rng.uniform(size=n) < 0.5)
    e = rng.laplace(size=m) if noise == "laplace" else 10.0 * rng.standard_normal(m) * (rng.uniform(size=m) < 0.1)
    return A, A @ xt + e, lam


def lad_linprog(A, b, lam):
    m, n = A.shape
    c = np.concatenate([lam * np.ones(2 * n), np.ones(2 * m)])          # x = xp - xn, Ax - b = rp - rn, all >= 0
    Aeq = np.hstack([A, -A, -np.eye(m), np.eye(m)])
    r = linprog(c, A_eq=Aeq, b_eq=b, bounds=(0, None), method="highs")
    assert r.status == 0
    return r.x[:n] - r.x[n:2 * n], float(r.fun)


def lad_certificate(A, b, lam, x, y):
    """(primal objective, duality gap, dual infeasibility) from the data alone."""
    P = float(np.abs(A @ x - b).sum() + lam * np.abs(x).sum())
    D = float(-b @ y)
    infeas = max(float(np.max(np.abs(y))) - 1.0, float(np.max(np.abs(A.T @ y))) - lam, 0.0)
    return P, P - D, infeas


def rigorous_relative_gap(A, b, lam, x, y, h_kind):
    """y scaled INTO the dual feasible set (||y|| <= 1 in h's dual norm, ||A'y||_inf <= lam): D(y~) = -b'y~ is then a true lower
    bound of the optimal value whatever the solver did, and 0 <= (P(x) - D(y~)) / P(x) bounds the relative suboptimality of x."""
    r = A @ x - b
    P = float((np.abs(r).sum() if h_kind == "l1" else np.linalg.norm(r)) + lam * np.abs(x).sum())
    s = max(1.0, float(np.max(np.abs(y))) if h_kind == "l1" else float(np.linalg.norm(y)), float(np.max(np.abs(A.T @ y))) / lam)
    return P, (P + float(b @ y) / s) / P


def _pd_methods(M, A, f, g, h, n, m, norm2, tol, maxit):
    z = lambda k: np.zeros(k)       # noqa: E731
    return {
        "condat_vu": lambda: M.condat_vu(z(n), z(m), f=f, g=g, h=h, A=A, Lf=0.0, norm_A=norm2, tol=tol, maxit=maxit),
        "adaptive_primal_dual": lambda: M.adaptive_primal_dual(z(n), z(m), f=f, g=g, h=h, A=A,
                                                               rule=M.OurRule(t=1.0, norm_A=norm2), tol=tol, maxit=maxit),
        "adaptive_linesearch_primal_dual": lambda: M.adaptive_linesearch_primal_dual(z(n), z(m), f=f, g=g, h=h, A=A,
                                                                                     eta=norm2, t=1.0, tol=tol,
                                                                                     maxit=maxit),
        "malitsky_pock": lambda: M.malitsky_pock(z(n), z(m), f=f, g=g, h=h, A=A, sigma=1.0, t=1.0, tol=tol, maxit=maxit),
    }


PD_METHODS = ["condat_vu", "adaptive_primal_dual", "adaptive_linesearch_primal_dual", "malitsky_pock"]


@pytest.fixture(scope="module")
def lad_small():
    A, b, lam = lad_problem(60, 12, seed=3, lam=2.0)
    x_lp, obj_lp = lad_linprog(A, b, lam)
    return A, b, lam, x_lp, obj_lp


@pytest.mark.parametrize("which", BACKENDS)
@pytest.mark.parametrize("method", PD_METHODS)
def test_lad_reaches_the_linear_programs_solution(request, which, method, lad_small):
    """P4 (the clamp = conjugate prox of the translated l1 norm), P1 and every primal-dual loop: the end state is the
    LP's vertex (unique here: generic data), the objective the LP's optimum, the returned y a dual certificate."""
    M = _backend(request, which)
    A, b, lam, x_lp, obj_lp = lad_small
    m, n = A.shape
    # adaptive_primal_dual's plain rule needs ~1e5 iterations here for 1e-9 ; the others a few thousand
    solve = _pd_methods(M, A, M.Zero(), M.NormL1(lam), M.Translate(M.NormL1(), -b), n, m, float(np.linalg.norm(A, 2)),
                        tol=1e-9, maxit=300_000)[method]
    x, y, it = solve()
    assert it < 300_000
    P, gap, infeas = lad_certificate(A, b, lam, x, y)
    assert abs(P - obj_lp) <= END_TOL * obj_lp * 1e-2            # objective: 1e-8 relative
    assert np.max(np.abs(x - x_lp)) <= END_TOL * 1e-1            # iterate: 1e-7
    assert abs(gap) <= 1e-7 and infeas <= 1e-8                   # and the solver-free certificate


# ---------------------------------------------------------------------------------------------------------------------
# square-root lasso  min ||Ax - b||_2 + lam ||x||_1  — scipy L-BFGS-B on x = p - q, p, q >= 0 (smooth away from Ax = b)
# (experiments/square_root_lasso/runme.jl:39-41: h = Translate(NormL2(), -b)).
# Dual:  max -b'y  s.t.  ||y||_2 <= 1, ||A'y||_inf <= lam.
def sqrt_lasso_scipy(A, b, lam):
    n = A.shape[1]

    def fs(pq):
        r = A @ (pq[:n] - pq[n:]) - b
        nr = np.linalg.norm(r)
        gg = A.T @ (r / nr)
        return nr + lam * pq.sum(), np.concatenate([gg + lam, -gg + lam])

    r = minimize(fs, np.zeros(2 * n), jac=True, bounds=[(0, None)] * (2 * n), method="L-BFGS-B",
                 options=dict(ftol=1e-16, gtol=1e-13, maxiter=100_000, maxfun=1_000_000))
    return r.x[:n] - r.x[n:], float(r.fun)


def sqrt_lasso_certificate(A, b, lam, x, y):
    P = float(np.linalg.norm(A @ x - b) + lam * np.abs(x).sum())
    D = float(-b @ y)
    infeas = max(float(np.linalg.norm(y)) - 1.0, float(np.max(np.abs(A.T @ y))) - lam, 0.0)
    return P, P - D, infeas


@pytest.fixture(scope="module")
def sqrt_lasso_small():
    rng = np.random.default_rng(7)
    m, n = 80, 14
    A = np.asfortranarray(np.hstack([rng.standard_normal((m, n - 1)), np.ones((m, 1))]))
    b = A @ (rng.standard_normal(n) * (rng.uniform(size=n) < 0.5)) + 0.5 * rng.standard_normal(m)
    lam = 1.5
    x_ref, obj_ref = sqrt_lasso_scipy(A, b, lam)
    return A, b, lam, x_ref, obj_ref


@pytest.mark.parametrize("which", BACKENDS)
@pytest.mark.parametrize("method", PD_METHODS)
def test_square_root_lasso_reaches_scipys_solution(request, which, method, sqrt_lasso_small):
    """P4, l2 case: the conjugate prox of the translated l2 norm is the projection on the unit ball."""
    M = _backend(request, which)
    A, b, lam, x_ref, obj_ref = sqrt_lasso_small
    m, n = A.shape
    solve = _pd_methods(M, A, M.Zero(), M.NormL1(lam), M.Translate(M.NormL2(), -b), n, m, float(np.linalg.norm(A, 2)),
                        tol=1e-10, maxit=200_000)[method]
    x, y, it = solve()
    assert it < 200_000
    P, gap, infeas = sqrt_lasso_certificate(A, b, lam, x, y)
    assert abs(P - obj_ref) <= 1e-9 * obj_ref
    assert np.max(np.abs(x - x_ref)) <= END_TOL * 1e-1           # (L-BFGS-B itself stops ~7e-9 from the optimum)
    assert abs(gap) <= 1e-8 and infeas <= 1e-9


# ---------------------------------------------------------------------------------------------------------------------
# l1-logistic regression with the bias as LAST coordinate, PENALISED — liblinear on [X 1] without an intercept
# (experiments/sparse_logreg/runme.jl:18-39: f = mean logistic loss of X w[1:end-1] .+ w[end] ; :56: g = NormL1(lam) on
#  all n + 1 coordinates).  liblinear minimises ||w||_1 + C * sum_i log(1 + exp(-s_i x_i'w)), s = 2y - 1:  C = 1/(m lam).
def logistic_problem(m, n, seed, lam):
    rng = np.random.default_rng(seed)
    X = np.asfortranarray(rng.standard_normal((m, n)))
    wt = np.concatenate([rng.standard_normal(n) * (rng.uniform(size=n) < 0.4), [0.3]])
    y = (1 / (1 + np.exp(-(X @ wt[:-1] + wt[-1]))) > rng.uniform(size=m)).astype(np.float64)
    return X, y, lam


def logistic_liblinear(X, y, lam):
    m = X.shape[0]
    clf = LogisticRegression(penalty="l1", solver="liblinear", fit_intercept=False, C=1.0 / (m * lam), tol=1e-10,
                             max_iter=100_000)
    clf.fit(np.hstack([X, np.ones((m, 1))]), 2 * y - 1)
    return clf.coef_.ravel()


def logistic_objective_and_gradient(X, y, lam, w):
    """Independent numpy statement of the objective: mean_i [log(1 + exp(z_i)) - y_i z_i] + lam ||w||_1, z = X w' + bias
    (= the reference's -mean((y-1) z - log(1 + exp(-z))), sparse_logreg/runme.jl:36)."""
    z = X @ w[:-1] + w[-1]
    p = 1 / (1 + np.exp(-z))
    val = float(np.mean(np.logaddexp(0.0, z) - y * z)) + lam * float(np.abs(w).sum())
    grad = np.concatenate([X.T @ (p - y), [np.sum(p - y)]]) / X.shape[0]
    return val, grad


def l1_kkt(grad, lam, w):
    nz = w != 0
    v_nz = float(np.max(np.abs(grad[nz] + lam * np.sign(w[nz])))) if nz.any() else 0.0
    v_z = float(np.max(np.maximum(np.abs(grad[~nz]) - lam, 0.0))) if (~nz).any() else 0.0
    return max(v_nz, v_z)


@pytest.fixture(scope="module")
def logistic_small():
    X, y, lam = logistic_problem(200, 15, seed=5, lam=0.02)
    return X, y, lam, logistic_liblinear(X, y, lam)


@pytest.mark.parametrize("which", BACKENDS)
@pytest.mark.parametrize("method", ["adaptive_proxgrad", "backtracking_proxgrad", "fixed_proxgrad"])
def test_l1_logistic_with_penalised_bias_reaches_liblinears_solution(request, which, method, logistic_small):
    """F2 + P1: the bias is coordinate `end` and IS soft-thresholded (g acts on all n + 1)."""
    M = _backend(request, which)
    X, y, lam, w_ref = logistic_small
    m, n = X.shape
    f, g = M.LogisticLoss(X, y), M.NormL1(lam)
    Lf = np.linalg.norm(np.hstack([X, np.ones((m, 1))]), 2) ** 2 / (4 * m)
    w0 = np.zeros(n + 1)
    w, it = {"adaptive_proxgrad": lambda: M.adaptive_proxgrad(w0, f=f, g=g, rule=M.OurRule(gamma=1 / Lf), tol=1e-11,
                                                              maxit=200_000),
             "backtracking_proxgrad": lambda: M.backtracking_proxgrad(w0, f=f, g=g, gamma0=1 / Lf, xi=1.5, tol=1e-11,
                                                                      maxit=200_000),
             "fixed_proxgrad": lambda: M.fixed_proxgrad(w0, f=f, g=g, gamma=1 / Lf, tol=1e-11, maxit=200_000)}[method]()
    assert it < 200_000
    assert np.max(np.abs(w - w_ref)) <= END_TOL * 1e-2                       # 1e-8 to liblinear
    assert abs(w[-1]) > 0.1 and abs(w[-1] - w_ref[-1]) <= 1e-8               # a live, penalised bias
    val, grad = logistic_objective_and_gradient(X, y, lam, w)
    val_ref, _ = logistic_objective_and_gradient(X, y, lam, w_ref)
    assert abs(val - val_ref) <= 1e-12
    assert l1_kkt(grad, lam, w) <= 1e-9
    assert abs(grad[-1] + lam * np.sign(w[-1])) <= 1e-9                      # the bias pays lam like every other weight
    # an unpenalised bias would end elsewhere: its gradient would vanish instead
    assert abs(grad[-1]) >= 0.5 * lam


# ---------------------------------------------------------------------------------------------------------------------
# dual SVM:  min 0.5 x'Qx + q'x  s.t. 0 <= x <= C, y'x = 0  — scipy SLSQP + KKT
# (experiments/dual_svm/runme.jl:47-56: f = Quadratic(Q, q), g = IndBox(0, C), h = IndZero(), A = Matrix(y'))
def dual_svm_problem(N, d, seed, C):
    rng = np.random.default_rng(seed)
    Xs = rng.standard_normal((N, d))
    lab = np.sign(Xs[:, 0] + 0.3 * rng.standard_normal(N))
    lab[lab == 0] = 1.0
    Z = lab[:, None] * Xs
    return np.asfortranarray(Z @ Z.T), -np.ones(N), np.asfortranarray(lab[None, :]), C


def box_qp_slsqp(Q, q, Aeq, C):
    N = q.size
    r = minimize(lambda x: (0.5 * x @ Q @ x + q @ x, Q @ x + q), np.zeros(N), jac=True, bounds=[(0, C)] * N,
                 constraints=[dict(type="eq", fun=lambda x: Aeq @ x, jac=lambda x: Aeq)], method="SLSQP",
                 options=dict(ftol=1e-15, maxiter=1000))
    assert r.success
    return r.x, float(r.fun)


def box_qp_kkt(Q, q, Aeq, C, x, y):
    """max violation of: x in the box, Aeq x = 0, (Qx + q + Aeq'y)_i >= 0 at the lower bound, <= 0 at the upper, = 0 inside."""
    gr = Q @ x + q + Aeq.T @ y
    lo, hi = x <= 0, x >= C
    mid = ~lo & ~hi
    v = [float(np.max(np.abs(Aeq @ x))), float(np.max(np.maximum(-x, 0))), float(np.max(np.maximum(x - C, 0)))]
    if mid.any():
        v.append(float(np.max(np.abs(gr[mid]))))
    if lo.any():
        v.append(float(np.max(np.maximum(-gr[lo], 0))))
    if hi.any():
        v.append(float(np.max(np.maximum(gr[hi], 0))))
    return max(v)


@pytest.fixture(scope="module")
def dual_svm_small():
    Q, q, Aeq, C = dual_svm_problem(40, 5, seed=11, C=0.1)
    x_ref, obj_ref = box_qp_slsqp(Q, q, Aeq, C)
    return Q, q, Aeq, C, x_ref, obj_ref


@pytest.mark.parametrize("which", BACKENDS)
@pytest.mark.parametrize("method", ["adaptive_primal_dual", "condat_vu", "malitsky_pock"])
def test_dual_svm_box_qp_reaches_slsqps_solution(request, which, method, dual_svm_small):
    """P2 (IndBox), P3 (IndZero: the conjugate prox is the identity — y is the equality constraint's multiplier)."""
    M = _backend(request, which)
    Q, q, Aeq, C, x_ref, obj_ref = dual_svm_small
    N = q.size
    f, g, h = M.Quadratic(Q, q), M.IndBox(0.0, C), M.IndZero()
    nA = float(np.linalg.norm(Aeq))
    x0, y0 = np.zeros(N), np.zeros(1)
    # (Malitsky-Pock's linesearch reaches a residual of 1e-7 in ~150 iterations here and then stalls in rounding ; the
    #  reference runs it to 1e-5)
    tol = 3e-7 if method == "malitsky_pock" else 1e-10
    x, y, it = {"adaptive_primal_dual": lambda: M.adaptive_primal_dual(x0, y0, f=f, g=g, h=h, A=Aeq,
                                                                       rule=M.OurRule(t=1.0, norm_A=nA), tol=tol,
                                                                       maxit=200_000),
                "condat_vu": lambda: M.condat_vu(x0, y0, f=f, g=g, h=h, A=Aeq, Lf=float(np.linalg.norm(Q)), tol=tol,
                                                 maxit=200_000),
                "malitsky_pock": lambda: M.malitsky_pock(x0, y0, f=f, g=g, h=h, A=Aeq, sigma=1 / nA, t=1.0, tol=tol,
                                                         maxit=50_000)}[method]()
    assert it < 50_000
    obj = 0.5 * x @ Q @ x + q @ x
    slack = 1e3 if method == "malitsky_pock" else 1.0
    assert abs(obj - obj_ref) <= 1e-9 * slack
    assert np.max(np.abs(x - x_ref)) <= END_TOL * 1e-1 * (10 if method == "malitsky_pock" else 1)
    assert box_qp_kkt(Q, q, Aeq, C, x, y) <= 1e-8 * slack * 10


# ---------------------------------------------------------------------------------------------------------------------
# the prox operators against a brute-force minimisation of their DEFINITION  prox_{gamma g}(x) = argmin g(y) + |y-x|^2/(2 gamma)
def _prox_1d(fun, x, gamma, lo=-50.0, hi=50.0):
    r = minimize_scalar(lambda t: fun(t) + (t - x) ** 2 / (2 * gamma), bounds=(lo, hi), method="bounded",
                        options=dict(xatol=1e-13, maxiter=10_000))
    return r.x


@pytest.mark.parametrize("which", BACKENDS)
def test_prox_operators_against_their_definition(request, which):
    """P1 NormL1, P2 IndBox, NormL2, and P4's two conjugate proxes through the Moreau identity, each against scipy's
    minimiser of the defining problem ; returned values g(prox) against numpy."""
    M = _backend(request, which)
    rng = np.random.default_rng(2)
    x = np.concatenate([rng.standard_normal(24) * 3, [0.0, 0.6, -0.6, 0.6000001, 1e-300]])      # (gamma lam = 0.6 below)
    n = x.size
    # P1: lam |.| coordinate-wise
    lam, gamma = 1.5, 0.4
    y, val = M.NormL1(lam).prox(x, gamma)
    ref = np.array([_prox_1d(lambda t: lam * abs(t), xi, gamma) for xi in x])
    assert np.max(np.abs(y - ref)) <= 1e-7
    assert val == pytest.approx(lam * np.abs(y).sum(), rel=1e-14)
    assert np.all(y[np.abs(x) <= gamma * lam] == 0.0)                              # exact zeros inside the threshold
    # P2: the box
    y, val = M.IndBox(-0.5, 1.25).prox(x, 0.7)
    ref = np.array([_prox_1d(lambda t: 0.0, xi, 0.7, lo=-0.5, hi=1.25) for xi in x])
    assert np.max(np.abs(y - ref)) <= 1e-7 and val == 0.0
    # NormL2: lam ||.||_2 — the whole vector at once (smooth away from 0 ; also the case that collapses to 0)
    for lam2, gam2 in ((2.0, 0.9), (50.0, 1.0)):
        y, val = M.NormL2(lam2).prox(x, gam2)
        r = minimize(lambda v: lam2 * np.linalg.norm(v) + np.sum((v - x) ** 2) / (2 * gam2), x * 0.5, method="BFGS",
                     options=dict(gtol=1e-12))
        assert np.max(np.abs(y - r.x)) <= 1e-6
        assert val == pytest.approx(lam2 * np.linalg.norm(y), rel=1e-13, abs=1e-300)
    # P4: prox of sigma h*, h = ||. - b||_1 (LAD) and h = ||. - b||_2 (square-root lasso):
    #     h*(y) = b'y + indicator(dual ball)  =>  prox = projection of (w - sigma b) on the dual ball
    b = rng.standard_normal(n)
    w = rng.standard_normal(n) * 2
    sigma = 0.8
    for base, proj in ((M.NormL1(), lambda u: np.clip(u, -1.0, 1.0)),
                       (M.NormL2(), lambda u: u / max(1.0, np.linalg.norm(u)))):
        hc = M.convex_conjugate(M.Translate(base, -b))
        y, _ = hc.prox(w, sigma)
        assert np.max(np.abs(y - proj(w - sigma * b))) <= 1e-13
        # and the definition itself, brute force on the l1 case (separable): argmin b_i t + (t - w_i)^2/(2 sigma), |t| <= 1
        if isinstance(base, M.NormL1):
            ref = np.array([_prox_1d(lambda t, bi=bi: bi * t, wi, sigma, lo=-1.0, hi=1.0) for wi, bi in zip(w, b)])
            assert np.max(np.abs(y - ref)) <= 1e-7
    # P3: Zero / IndZero and their conjugates (ProximalCore: convex_conjugate(Zero) = IndZero and back)
    y, _ = M.Zero().prox(x, 3.0)
    assert np.array_equal(y, x)
    y, _ = M.IndZero().prox(x, 3.0)
    assert not np.any(y)
    y, _ = M.convex_conjugate(M.IndZero()).prox(x, 3.0)
    assert np.array_equal(y, x)
    y, _ = M.convex_conjugate(M.Zero()).prox(x, 3.0)
    assert not np.any(y)


# ---------------------------------------------------------------------------------------------------------------------
# the same certificates on the ROW-STREAMING kernels the BASELINE configs run on (one pass over A per iteration):
# sizes a third-party solver cannot take in seconds, so the solver-free certificate is the check.
@pytest.mark.gpu
@pytest.mark.parametrize("h_kind", ["l1", "l2"])
@pytest.mark.parametrize("method", ["condat_vu", "adaptive_linesearch_primal_dual", "malitsky_pock"])
def test_one_pass_primal_dual_kernels_end_in_a_certified_optimum(apx, h_kind, method):
    """k_multi_rows / k_slabpd in the primal-dual form (the kernels of C5, at a size numpy can check): the clamp and the l2
    ball inside the row kernels, certified by a rigorous duality gap."""
    A, b, lam = lad_problem(4200, 700, seed=21, lam=5.0 if h_kind == "l1" else 4.0, noise="outliers" if h_kind == "l1" else "laplace")
    m, n = A.shape
    ctx = apx.Context(0)
    ctx.fused_small(False, one_pass=True)                  # (2.9 M entries would still run on the small-size kernels)
    ctx.one_pass_threshold(16 << 20)
    dA = apx.DeviceMatrix.from_numpy(A, ctx).build_transpose()
    h = apx.Translate(apx.NormL1() if h_kind == "l1" else apx.NormL2(), -b)
    nrm = float(np.linalg.norm(A, 2))
    z = np.zeros
    ctx.profile(True)
    ctx.kernel_stats(reset=True)
    kw = dict(f=apx.Zero(), g=apx.NormL1(lam), h=h, A=dA, tol=1e-8, maxit=60_000, ctx=ctx)
    x, y, it = {"condat_vu": lambda: apx.condat_vu(z(n), z(m), Lf=0.0, norm_A=nrm, **kw),
                "adaptive_linesearch_primal_dual": lambda: apx.adaptive_linesearch_primal_dual(z(n), z(m), eta=nrm, t=1.0,
                                                                                               **kw),
                "malitsky_pock": lambda: apx.malitsky_pock(z(n), z(m), sigma=1.0, t=1.0, **kw)}[method]()
    st = ctx.kernel_stats()
    assert st["one_pass"]["launches"] >= it, st               # the iterations ran on the one-pass kernels
    assert it < 60_000
    P, relgap = rigorous_relative_gap(A, b, lam, x, y, h_kind)
    assert 0.0 <= relgap + 1e-13 and relgap <= END_TOL * 1e-1, (P, relgap, it)
    assert np.count_nonzero(x) > 10
    dA.free()


@pytest.mark.gpu
@pytest.mark.parametrize("logistic", [False, True])
def test_one_pass_gradient_kernels_end_at_a_kkt_point(apx, logistic):
    """k_multi_rows / k_slab rows (the kernels of C2 / C3 at a size numpy can check): lasso and l1-logistic."""
    rng = np.random.default_rng(31)
    m, n = 4300, 600
    A = np.asfortranarray(rng.standard_normal((m, n)))
    ctx = apx.Context(0)
    ctx.fused_small(False, one_pass=True)
    ctx.one_pass_threshold(16 << 20)
    dA = apx.DeviceMatrix.from_numpy(A, ctx).build_transpose()
    if logistic:
        wt = np.concatenate([rng.standard_normal(n) * (rng.uniform(size=n) < 0.1), [0.3]])
        yy = (1 / (1 + np.exp(-(A @ wt[:-1] + wt[-1]))) > rng.uniform(size=m)).astype(np.float64)
        lam = 0.01
        f, nx = apx.LogisticLoss(dA, yy, ctx=ctx), n + 1
        gam = 4.0 * m / (np.linalg.norm(A) ** 2 + m)
    else:
        yy = A @ (rng.standard_normal(n) * (rng.uniform(size=n) < 0.1)) + 0.1 * rng.standard_normal(m)
        lam = 20.0
        f, nx = apx.LinearLeastSquares(dA, yy, ctx=ctx), n
        gam = 1.0 / np.linalg.norm(A, 2) ** 2
    ctx.profile(True)
    ctx.kernel_stats(reset=True)
    w, it = apx.adaptive_proxgrad(np.zeros(nx), f=f, g=apx.NormL1(lam), rule=apx.OurRule(gamma=gam), tol=1e-10,
                                  maxit=100_000, ctx=ctx)
    assert ctx.kernel_stats()["one_pass"]["launches"] >= it
    assert it < 100_000
    if logistic:
        _, grad = logistic_objective_and_gradient(A, yy, lam, w)
        assert l1_kkt(grad, lam, w) <= 1e-9                  # (liblinear at this size and accuracy takes minutes: the small case has it)
        assert abs(w[-1]) > 0 and abs(grad[-1] + lam * np.sign(w[-1])) <= 1e-9
    else:
        assert lasso_kkt(A, yy, lam, w) <= 1e-6
        assert np.max(np.abs(w - _sklearn_lasso(A, yy, lam))) <= END_TOL
    dA.free()
