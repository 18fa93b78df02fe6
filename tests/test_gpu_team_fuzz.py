"""Ragged shapes for the team kernel (csrc/team_rows.h k_team_rows: one workgroup, or 2 / 4 / 8 / 16 per slab of rows — f and
its gradient, csrc/slab.hip ; the primal-dual loop, csrc/slab_pd.hip): row counts that do not divide by the number of teams (down to 64 rows per team), odd column counts, slice boundaries at the edges of the KV vectors per thread, the boundaries between team sizes.
Matrices are generated on the device; the one-pass run is compared with the two GEMV passes on the same data (fp64: the two
differ by summation order only) and its first evaluation with values recomputed on the CPU from spot rows and columns."""
import importlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import problems as P


@pytest.fixture(scope="module")
def apx():
    import adaprox_amd as apx
    apx.load()
    return apx


def _cpu_cols(seed, cols, m):
    out = np.empty((m, len(cols)))
    i = np.arange(m, dtype=np.uint64)
    for k, jj in enumerate(cols):
        out[:, k] = -1.0 + 2.0 * P.hash_uniform(seed, np.uint64(jj) * np.uint64(m) + i)
    return out


def _cpu_rows(seed, rows, m, n):
    out = np.empty((len(rows), n))
    j = np.arange(n, dtype=np.uint64)
    for k, i in enumerate(rows):
        out[k] = -1.0 + 2.0 * P.hash_uniform(seed, j * np.uint64(m) + np.uint64(i))
    return out


# (m, n): the boundaries of the team sizes and slice widths of round 2's first policy (teams of 4 up to 28 160 columns, of 8 up
# to 56 320, of 16 beyond) — still boundaries of KV = ceil(n / 2 / P / 704) under the present one (csrc/slab.hip slab_setup)
PG_SHAPES = [(4096, 10753), (4097, 11263), (4099, 11265), (5000, 16897), (4111, 28160), (4096, 28161), (4098, 33793),
             (4101, 56320), (4096, 56321), (4100, 67585), (4097, 112640), (4300, 99999),
             # ... and of the present one: one workgroup / teams of two, five / six vectors, two / four, four / eight, the longest row
             (4100, 14080), (4099, 14081), (4100, 16896), (4101, 16897), (4100, 22528), (4097, 22529), (4098, 33793), (4098, 45056), (4096, 45057),
             (4100, 67585), (4100, 112641), (4099, 135169), (4097, 180223),
             # round 4: teams of eight in the three-slot shape from 67 585 to 90 112 columns (seven / eight vectors), sixteen beyond
             (4100, 67584), (4099, 78849), (4098, 90112), (4097, 90113)]


# (the logistic link is the same code at every team size: the shapes beyond 40 000 columns run least squares only — said in the
#  parameter ids instead of by a skip ; the link on wide teams: tests/test_gpu_slab.py, 88 001 and 180 224 columns)
PG_CASES = [pytest.param(m, n, False, id=f"{m}-{n}-least_squares" + ("-only" if n > 40000 else "")) for m, n in PG_SHAPES] + \
           [pytest.param(m, n, True, id=f"{m}-{n}-logistic") for m, n in PG_SHAPES if n <= 40000]


@pytest.mark.parametrize("m,n,logistic", PG_CASES)
def test_team_rows_ragged_shapes(apx, m, n, logistic):
    seed = m + n
    out = []
    rng = np.random.default_rng(seed)
    nx = n + 1 if logistic else n
    x0 = (0.02 * rng.standard_normal(nx)) * (rng.uniform(size=nx) < 0.5)
    b = (rng.uniform(size=m) < 0.5).astype(np.float64) if logistic else rng.standard_normal(m)
    gam = 4.0 / (n / 3.0 + 1.0) if logistic else 1.0 / (m * n / 3.0)
    for one_pass in (True, False):
        ctx = apx.Context(0)
        ctx.fused_small(one_pass)
        ctx.one_pass_threshold(16 << 20)
        A = apx.DeviceMatrix.uniform(m, n, seed=seed, ctx=ctx)
        if one_pass:
            A.build_transpose()
        f = apx.LogisticLoss(A, b, ctx=ctx) if logistic else apx.LinearLeastSquares(A, b, ctx=ctx)
        ctx.kernel_stats(reset=True)
        ctx.profile(True)
        fx, pb = apx.eval_with_pullback(f, x0)
        g0 = pb()
        st = ctx.kernel_stats(reset=True)
        ctx.profile(False)
        assert (st["gemv_t"]["launches"] == 0) == one_pass, (m, n, st)          # the team kernel really ran (or did not)
        recs = []
        x, it = apx.adaptive_proxgrad(x0, f=f, g=apx.NormL1(1e-3 if logistic else 1.0), rule=apx.OurRule(gamma=gam), tol=0.0,
                                      maxit=6, ctx=ctx, record=recs.append)
        out.append((fx, g0, x, [r["objective"] for r in recs]))
        A.free()
    (f1, g1, x1, o1), (f2, g2, x2, o2) = out
    assert f1 == pytest.approx(f2, rel=1e-12)
    assert np.max(np.abs(g1 - g2)) <= 1e-11 * np.max(np.abs(g2))
    assert np.max(np.abs(x1 - x2)) <= 1e-10 * max(np.max(np.abs(x2)), 1e-300)
    assert o1 == pytest.approx(o2, rel=1e-11)
    # spot check against CPU arithmetic on regenerated columns: grad_j = A[:, j]' res (/ N), res from regenerated rows is
    # too expensive for every row, so: res through the GPU's own A x (the two-pass product, checked on spot rows)
    rows = [0, 1, m // 2, m - 1]
    Ax_spot = _cpu_rows(seed, rows, m, n) @ x0[:n]
    ctx = apx.Context(0)
    ctx.fused_small(False)
    A = apx.DeviceMatrix.uniform(m, n, seed=seed, ctx=ctx)
    Ax = A @ x0[:n]
    assert np.max(np.abs(Ax[rows] - Ax_spot)) <= 1e-12 * max(np.max(np.abs(Ax_spot)), 1.0)
    if logistic:
        res = 1.0 / (1.0 + np.exp(-(Ax + x0[-1]))) - b
        scale = 1.0 / m
    else:
        res = Ax - b
        scale = 1.0
    cols = [0, 1, n // 4 - 1, n // 4, n // 2, n - 2, n - 1]
    g_ref = _cpu_cols(seed, cols, m).T @ res * scale
    assert np.max(np.abs(g1[cols] - g_ref)) <= 1e-11 * np.max(np.abs(g1[:n]))
    A.free()


# two-member teams: 10 753 ... 21 504 columns, KV = ceil(n / 4 / 768)
PD_SHAPES = [(4096, 10753), (4097, 12289), (4099, 15361), (4098, 18433), (4500, 21503), (4096, 21504), (6007, 19999),
             # the primal-dual loop on k_team_rows (round 2, last): every boundary of csrc/team_rows.h team_plan, and the longest row
             (4100, 11264), (4099, 11265), (4100, 16896), (4101, 16897), (4100, 22528), (4097, 22529), (4098, 33793), (4098, 45057),
             (4100, 67585), (4100, 135169), (4097, 180223),
             # round 4: teams of two with five vectors in the three-slot shape (11 265 ... 14 080), teams of eight in it (67 585 ... 90 112)
             (4100, 12001), (4099, 14080), (4098, 14081), (4100, 67584), (4099, 78849), (4098, 90112), (4097, 90113)]


@pytest.mark.parametrize("m,n", PD_SHAPES)
def test_teampd_rows_ragged_shapes(apx, m, n):
    seed = 3 * m + n
    rng = np.random.default_rng(seed)
    yv = rng.laplace(size=m)
    out = []
    for one_pass in (True, False):
        ctx = apx.Context(0)
        ctx.fused_small(one_pass)
        ctx.one_pass_threshold(16 << 20)
        A = apx.DeviceMatrix.uniform(m, n, seed=seed, ctx=ctx)
        A.fill_column(n - 1, 1.0)
        nA = A.norm()
        recs = []
        x, y, it = apx.adaptive_primal_dual(np.zeros(n), np.zeros(m), f=apx.Zero(), g=apx.NormL1(3.0),
                                            h=apx.Translate(apx.NormL1(1.0), -yv), A=A, rule=apx.OurRule(t=1.0, norm_A=nA),
                                            tol=0.0, maxit=8, ctx=ctx, record=recs.append)
        out.append((x, y, [r["norm_res"] for r in recs], [r["objective"] for r in recs]))
        A.free()
    (x1, y1, r1, o1), (x2, y2, r2, o2) = out
    assert np.max(np.abs(x1 - x2)) <= 1e-10 * max(np.max(np.abs(x2)), 1e-300)
    assert np.max(np.abs(y1 - y2)) <= 1e-10 * max(np.max(np.abs(y2)), 1e-300)
    assert r1 == pytest.approx(r2, rel=1e-9) and o1 == pytest.approx(o2, rel=1e-10)
    assert np.max(np.abs(y1)) <= 1.0                     # y stays in the dual ball of the l1 norm
