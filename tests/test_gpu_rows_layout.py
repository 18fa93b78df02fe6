"""ONE resident copy of A: a dense matrix that lives in the row-contiguous storage alone
(APX_LAYOUT_ROWS / apx_matrix_drop_primary, include/adaprox_hip.h).  No reference counterpart — a Julia `Matrix` is
column-major ; what the reference fixes is the RESULT of `A*x`, `A'*y` (src/counting.jl:65-82) and of the loops on them
(src/AdaProx.jl:312-364), and those must not depend on where the library keeps the bytes:

  * the operator entry points, the element-wise helpers and upload / download on the row storage against numpy ;
  * the one-pass kernels read that storage in either layout: BIT-identical iterates against the dual layout of rounds 2-4 ;
  * the two-pass fallback (and every GEMV outside the one-pass launches) served from the one copy, against the oracle ;
  * what cannot work there says so (drop_transpose) or steps aside (the small-size kernels read column-major storage)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import adaprox_oracle as O


@pytest.fixture(scope="module")
def apx():
    import adaprox_amd as apx
    apx.load()
    return apx


def _rel(a, b):
    return float(np.max(np.abs(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64)))) / max(float(np.max(np.abs(b))), 1e-300)


@pytest.mark.parametrize("m,n,dtype", [(1, 1, np.float64), (33, 65, np.float64), (500, 1000, np.float64), (1000, 37, np.float32),
                                       (4099, 2051, np.float64), (300, 70001, np.float64), (70001, 130, np.float32),
                                       (2, 200000, np.float64)])
def test_operators_and_helpers_on_the_row_storage(apx, m, n, dtype):
    rng = np.random.default_rng(m * 7 + n)
    A = np.asfortranarray(rng.uniform(-1, 1, (m, n)).astype(dtype))
    ctx = apx.Context(0)
    R = apx.DeviceMatrix.from_numpy(A, ctx, layout="rows")          # upload: column slabs transposed through the staging buffer
    assert R.layout == "rows"
    assert np.array_equal(R.to_numpy(), A)                         # download: row slabs transposed back
    x, r = rng.standard_normal(n).astype(dtype), rng.standard_normal(m).astype(dtype)
    A64 = A.astype(np.float64)
    tol = 1e-12 if dtype == np.float64 else 2e-5
    assert _rel(R.matvec(x), A64 @ x.astype(np.float64)) <= tol * max(1.0, np.sqrt(n) / 30)
    assert _rel(R.rmatvec(r), A64.T @ r.astype(np.float64)) <= tol * max(1.0, np.sqrt(m) / 30)
    assert R.sumsq() == pytest.approx(float(np.sum(A64 * A64)), rel=1e-12)
    # the counter-based generator writes the same element values along the rows as along the columns
    U_rows = apx.DeviceMatrix.uniform(m, n, seed=3, dtype=dtype, row_offset=5, m_total=m + 11, ctx=ctx, layout="rows")
    U_cols = apx.DeviceMatrix.uniform(m, n, seed=3, dtype=dtype, row_offset=5, m_total=m + 11, ctx=ctx)
    assert np.array_equal(U_rows.to_numpy(), U_cols.to_numpy())
    alpha = rng.uniform(0.5, 2.0, n)
    U_rows.scale_columns(alpha)
    U_cols.scale_columns(alpha)
    j = n // 2
    U_rows.fill_column(j, 1.0)
    U_cols.fill_column(j, 1.0)
    assert U_rows.layout == "rows"                                 # edits work on the one copy in place
    assert np.array_equal(U_rows.to_numpy(), U_cols.to_numpy())
    # a column-major matrix turned rows-only: the same bytes as one uploaded that way
    D = apx.DeviceMatrix.from_numpy(A, ctx)
    assert D.layout == "cols" and D.drop_primary().layout == "rows" and D.drop_primary().layout == "rows"
    assert np.array_equal(D.to_numpy(), A)
    assert np.array_equal(D.matvec(x), R.matvec(x)) and np.array_equal(D.rmatvec(r), R.rmatvec(r))
    D.build_transpose()                                            # nothing to build
    assert D.layout == "rows"
    with pytest.raises(apx.ApxInvalid):
        D.drop_transpose()                                         # it is the matrix
    for M in (R, U_rows, U_cols, D):
        M.free()


CASES = [(8500, 1000, "ls", np.float64),       # k_multi_rows
         (8700, 600, "ls", np.float32),
         (9000, 900, "logistic", np.float32),
         (4101, 8000, "logistic", np.float64),  # k_team_rows, one workgroup
         (4098, 10000, "ls", np.float64),
         (4100, 26000, "ls", np.float64),       # teams of four
         (4099, 100000, "ls", np.float64),      # C2's row: teams of sixteen
         (4200, 1100, "lad", np.float64),       # the primal-dual loop: k_multi_rows
         (4100, 20000, "lad", np.float64),      # C5's row: teams of two
         (4100, 12000, "l2ball", np.float64)]


@pytest.mark.parametrize("m,n,kind,dtype", CASES)
def test_one_pass_iterates_are_bit_identical_to_the_dual_layout(apx, m, n, kind, dtype):
    rng = np.random.default_rng(m + n)
    A = np.asfortranarray(rng.uniform(-1, 1, (m, n)).astype(dtype))
    K = 12
    out = {}
    for layout in ("dual", "rows"):
        ctx = apx.Context(0)
        ctx.fused_small(False, one_pass=True)
        ctx.one_pass_threshold(16 << 20)
        dA = apx.DeviceMatrix.from_numpy(A, ctx, layout="rows") if layout == "rows" else apx.DeviceMatrix.from_numpy(A, ctx).build_transpose()
        assert dA.layout == layout
        ctx.profile(True)
        ctx.kernel_stats(reset=True)
        recs = []
        if kind in ("ls", "logistic"):
            if kind == "ls":
                b = rng.standard_normal(m).astype(dtype) if "b" not in out else out["b"]
                f, nx, gam, lam = apx.LinearLeastSquares(dA, b, ctx=ctx), n, 1.0 / (m * n / 3.0), 1.0
            else:
                b = (rng.uniform(size=m) < 0.5).astype(dtype) if "b" not in out else out["b"]
                f, nx, gam, lam = apx.LogisticLoss(dA, b, ctx=ctx), n + 1, 4.0 * m / (np.linalg.norm(A.astype(np.float64)) ** 2 + m), 0.01
            out["b"] = b
            x0 = np.full(nx, 0.01, dtype=dtype)
            x, it = apx.adaptive_proxgrad(x0, f=f, g=apx.NormL1(lam), rule=apx.OurRule(gamma=gam), tol=0.0, maxit=K, ctx=ctx,
                                          record=recs.append)
            y = None
        else:
            b = rng.standard_normal(m) if "b" not in out else out["b"]
            out["b"] = b
            h = apx.Translate(apx.NormL1(1.0) if kind == "lad" else apx.NormL2(1.0), -b)
            x, y, it = apx.adaptive_primal_dual(np.full(n, 0.01), np.zeros(m), f=apx.Zero(), g=apx.NormL1(0.01), h=h, A=dA,
                                                rule=apx.OurRule(t=1.0, norm_A=float(np.sqrt(m * n / 3.0))), tol=0.0, maxit=K, ctx=ctx,
                                                record=recs.append)
        st = ctx.kernel_stats(reset=True)
        assert it == K and st["one_pass"]["launches"] >= K, st
        out[layout] = (x, y, [(r["gamma"], r["norm_res"], r["objective"]) for r in recs], ctx.kernel_name("one_pass"))
        dA.free()
    (x1, y1, r1, k1), (x2, y2, r2, k2) = out["dual"], out["rows"]
    assert k1 == k2                                                # the same kernel instance
    assert np.array_equal(x1, x2) and r1 == r2 and np.any(x1 != 0)
    if y1 is not None:
        assert np.array_equal(y1, y2)


@pytest.mark.parametrize("m,n,dtype", [(200, 400, np.float64), (4300, 1200, np.float64), (3000, 20000, np.float64),
                                       (4300, 1200, np.float32)])
def test_two_pass_loops_on_one_copy_match_the_oracle(apx, m, n, dtype):
    """The fallback of the one-pass kernels (and every loop with apx_ctx_one_pass(ctx, 0)): both GEMV orientations from the row
    storage — `A*x` by the column-streaming kernel over it, `A'*r` by the row-tiled kernel with the roles swapped.  A SMALL
    rows-only matrix runs here too: the small-size kernels read column-major storage and step aside."""
    rng = np.random.default_rng(m + n + 1)
    A = np.asfortranarray(rng.uniform(-1, 1, (m, n)).astype(dtype))
    A64 = A.astype(np.float64)
    b = rng.standard_normal(m).astype(dtype)
    K = 25
    gam = 1.0 / (m * n / 3.0)                                      # (1 / ||A||_F^2 in expectation: a valid step, no SVD of A needed)
    ro = []
    xo, _ = O.adaptive_proxgrad(np.zeros(n), f=O.LinearLeastSquares(A64, b.astype(np.float64)), g=O.NormL1(1.0), rule=O.OurRule(gamma=gam),
                                tol=0.0, maxit=K, record=ro.append)
    ctx = apx.Context(0)
    ctx.one_pass(0)
    dA = apx.DeviceMatrix.from_numpy(A, ctx, layout="rows")
    ctx.profile(True)
    ctx.kernel_stats(reset=True)
    rd = []
    x, it = apx.adaptive_proxgrad(np.zeros(n, dtype=dtype), f=apx.LinearLeastSquares(dA, b, ctx=ctx), g=apx.NormL1(1.0),
                                  rule=apx.OurRule(gamma=gam), tol=0.0, maxit=K, ctx=ctx, record=rd.append)
    st = ctx.kernel_stats(reset=True)
    assert st["one_pass"]["launches"] == 0 and st["gemv_n"]["launches"] == K + 1 and st["gemv_t"]["launches"] == K + 1, st
    tol = 1e-10 if dtype == np.float64 else 2e-3
    assert _rel(x, xo) <= tol
    for a, c in zip(rd, ro):
        assert a["gamma"] == pytest.approx(c["gamma"], rel=1e-7 if dtype == np.float64 else 5e-2)
        assert a["objective"] == pytest.approx(c["objective"], rel=1e-10 if dtype == np.float64 else 1e-4)
    # the primal-dual loop and a backtracking linesearch on the same copy
    yb = rng.standard_normal(m)
    if dtype == np.float64:
        nA = float(np.sqrt(m * n / 3.0))
        xo, yo, _ = O.adaptive_primal_dual(np.zeros(n), np.zeros(m), f=O.Zero(), g=O.NormL1(0.5), h=O.Translate(O.NormL1(), -yb), A=A64,
                                           rule=O.OurRule(t=1.0, norm_A=nA), tol=0.0, maxit=K)
        xd, yd, _ = apx.adaptive_primal_dual(np.zeros(n), np.zeros(m), f=apx.Zero(), g=apx.NormL1(0.5), h=apx.Translate(apx.NormL1(), -yb),
                                             A=dA, rule=apx.OurRule(t=1.0, norm_A=nA), tol=0.0, maxit=K, ctx=ctx)
        assert _rel(xd, xo) <= 1e-10 and _rel(yd, yo) <= 1e-10
        zo, _ = O.backtracking_proxgrad(np.zeros(n), f=O.LinearLeastSquares(A64, b.astype(np.float64)), g=O.NormL1(1.0), gamma0=gam, xi=1.5,
                                        tol=0.0, maxit=K)
        zd, _ = apx.backtracking_proxgrad(np.zeros(n), f=apx.LinearLeastSquares(dA, b, ctx=ctx), g=apx.NormL1(1.0), gamma0=gam, xi=1.5,
                                          tol=0.0, maxit=K, ctx=ctx)
        assert _rel(zd, zo) <= 1e-10
    dA.free()


def test_a_live_solver_survives_an_edit_of_a_rows_only_matrix(apx):
    """An in-place edit of a dual-layout matrix drops the row copy and a live solver falls back to the two passes
    (tests/test_gpu_slab.py) ; a rows-only matrix is edited where it is: the solver keeps its kernels, on the new values."""
    rng = np.random.default_rng(9)
    m, n = 8500, 1000
    A = np.asfortranarray(rng.uniform(-1, 1, (m, n)))
    b = rng.standard_normal(m)
    ctx = apx.Context(0)
    ctx.fused_small(False, one_pass=True)
    ctx.one_pass_threshold(16 << 20)
    dA = apx.DeviceMatrix.from_numpy(A, ctx, layout="rows")
    s = apx.Solver(np.zeros(n), f=apx.LinearLeastSquares(dA, b, ctx=ctx), g=apx.NormL1(1.0), rule=apx.OurRule(gamma=3.0 / (m * n)), tol=0.0,
                   ctx=ctx)
    s.step(5, sync=True)
    alpha = np.full(n, 0.5)
    dA.scale_columns(alpha)
    ctx.profile(True)
    ctx.kernel_stats(reset=True)
    s.step(5, sync=True)
    st = ctx.kernel_stats(reset=True)
    assert st["one_pass"]["launches"] == 5 and dA.layout == "rows", st
    x, _, res = s.result()
    assert np.all(np.isfinite(x)) and int(res.numit) == 10
    # the last evaluation really used the scaled matrix: f(x9) = 0.5 ||0.5 A x9 - b||^2 is what the record says
    s.close()
    dA.free()


def test_drop_primary_waits_for_the_problems_on_the_matrix(apx):
    """A solver on a SMALL problem holds pointers into the column-major storage (the small-size kernels read it): the primary copy
    cannot be taken away under it — APX_ERR_INVALID while problems created on the matrix exist, fine once they are gone."""
    rng = np.random.default_rng(4)
    A = np.asfortranarray(rng.uniform(-1, 1, (300, 200)))
    b = rng.standard_normal(300)
    ctx = apx.Context(0)
    dA = apx.DeviceMatrix.from_numpy(A, ctx)
    s = apx.Solver(np.zeros(200), f=apx.LinearLeastSquares(dA, b, ctx=ctx), g=apx.NormL1(1.0), rule=apx.OurRule(gamma=1e-3), tol=0.0, ctx=ctx)
    s.step(3, sync=True)
    with pytest.raises(apx.ApxInvalid, match="problems created on this matrix"):
        dA.drop_primary()
    s.step(3, sync=True)
    x_small, _, _ = s.result()
    s.close()
    assert dA.drop_primary().layout == "rows"
    s2 = apx.Solver(np.zeros(200), f=apx.LinearLeastSquares(dA, b, ctx=ctx), g=apx.NormL1(1.0), rule=apx.OurRule(gamma=1e-3), tol=0.0, ctx=ctx)
    s2.step(6, sync=True)
    x_rows, _, _ = s2.result()
    s2.close()
    assert _rel(x_rows, x_small) <= 1e-12 and np.any(x_small != 0)      # (the generic kernels on the row storage: the same six iterations)
    dA.free()

