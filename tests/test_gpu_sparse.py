"""Sparse design matrices: the products and the solvers on compressed
storage against scipy / the oracle; the LIBSVM reader mirror of experiments/libsvm.jl."""
import importlib

import numpy as np
import pytest
import scipy.sparse as sp

pytestmark = pytest.mark.gpu

from oracle import adaprox_oracle as O


@pytest.fixture(scope="module")
def apx():
    import adaprox_amd as apx
    apx.load()
    return apx


@pytest.fixture(scope="module")
def generic_ctx(apx):
    ctx = apx.Context(0)
    ctx.fused_small(False)
    return ctx


def _rel(a, b):
    return float(np.max(np.abs(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64)))) / \
        max(float(np.max(np.abs(b))), 1e-300)


@pytest.mark.parametrize("m,n,density", [(1, 1, 1.0), (7, 5, 0.5), (300, 1000, 0.01), (5000, 120, 0.15),
                                         (2000, 3000, 0.002), (64, 64, 0.0), (800, 70, 0.9),
                                         (30000, 8, 0.6), (6, 40000, 0.4), (2049, 3, 1.0)])   # long rows of A' / of A
@pytest.mark.parametrize("fmt", ["csr", "csc", "coo"])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_sparse_products(apx, m, n, density, fmt, dtype):
    rng = np.random.default_rng(m * 31 + n)
    A = sp.random(m, n, density=density, format=fmt, random_state=rng, dtype=np.float64).astype(dtype)
    x = rng.standard_normal(n).astype(dtype)
    r = rng.standard_normal(m).astype(dtype)
    dA = apx.DeviceMatrix.from_sparse(A)
    tol = 1e-13 if dtype == np.float64 else 2e-6
    A64 = A.astype(np.float64).tocsr()
    y_ref, g_ref = A64 @ x.astype(np.float64), A64.T @ r.astype(np.float64)
    assert np.max(np.abs(dA @ x - y_ref)) <= tol * max(1.0, np.max(np.abs(y_ref)))
    assert np.max(np.abs(dA.T @ r - g_ref)) <= tol * max(1.0, np.max(np.abs(g_ref)))
    assert dA.norm() == pytest.approx(np.sqrt(A64.multiply(A64).sum()), rel=1e-6 if dtype == np.float32 else 1e-13)
    with pytest.raises(apx.ApxError, match="dense matrices only"):
        dA.to_numpy()


@pytest.mark.parametrize("m,n,per_row,dtype", [(450_000, 10_000, 6, np.float64),      # X' r gathers from 450 000 doubles: 2 windows
                                               (1_100_000, 20_000, 3, np.float32),    # ... from 1.1 M floats: 3 windows
                                               (3_000, 500_000, 300, np.float64),     # X w gathers from 500 000 doubles: 2 windows
                                               (800_000, 400_000, 2, np.float64),     # both products windowed (4 and 2)
                                               (450_001, 9_000, 7, np.float64)])      # ... and a column / a row too heavy for the streaming tile
def test_products_in_l2_windows_of_the_gathered_operand(apx, m, n, per_row, dtype):
    """Round 5 (csrc/sparse.hip launch_windows): a product whose gathered operand exceeds 3 MB runs window by window of that
    operand — a partial per row and window, summed in window order.  Against scipy, against the one-launch form (the A/B option, read
    at upload), reproducible bit for bit, and through a solver (the fused epilogues take the windows as their splits)."""
    L = importlib.import_module("adaptive-proximal-algorithms_amd._lib")
    rng = np.random.default_rng(m + n)
    nnz_row = np.clip(rng.poisson(per_row, m), 0, n)
    indptr = np.concatenate([[0], np.cumsum(nnz_row)]).astype(np.int64)
    A = sp.csr_matrix((rng.uniform(-1, 1, indptr[-1]).astype(dtype), rng.integers(0, n, indptr[-1]).astype(np.int32), indptr), shape=(m, n))
    if m == 450_001:      # column 0 with 6000 entries (a row of X' beyond the tile of 2048 in both of its windows), row 5 of X with 3000
        heavy = sp.csr_matrix((rng.uniform(-1, 1, 9000), (np.concatenate([rng.choice(m, 6000, replace=False), np.full(3000, 5)]),
                                                           np.concatenate([np.zeros(6000, dtype=np.int64), rng.choice(n, 3000, replace=False)]))),
                              shape=(m, n))
        A = (A + heavy).tocsr()
    A.sum_duplicates()
    x = rng.standard_normal(n).astype(dtype)
    r = rng.standard_normal(m).astype(dtype)
    A64 = A.astype(np.float64)
    y_ref, g_ref = A64 @ x.astype(np.float64), A64.T @ r.astype(np.float64)
    tol = 1e-13 if dtype == np.float64 else 3e-6
    out = {}
    for windows in (1, 0):
        ctx = apx.Context(0)
        ctx.fused_small(False)
        ctx.set_option(L.APX_OPT_SPARSE_WINDOWS, windows)
        dA = apx.DeviceMatrix.from_sparse(A, ctx)
        ctx.profile(True)
        y, g = dA @ x, dA.T @ r
        ctx.kernel_stats(reset=True)
        names = (ctx.kernel_name("gemv_n"), ctx.kernel_name("gemv_t"))
        assert np.max(np.abs(y - y_ref)) <= tol * max(1.0, np.max(np.abs(y_ref)))
        assert np.max(np.abs(g - g_ref)) <= tol * max(1.0, np.max(np.abs(g_ref)))
        assert np.array_equal(dA @ x, y) and np.array_equal(dA.T @ r, g)
        out[windows] = (y, g, names)
        if windows and m >= 400_000 and dtype == np.float64:      # a few logistic iterations on the windowed storage against the plain one
            yy = (rng.uniform(size=m) < 0.5).astype(np.float64)
            gam = 4.0 * m / (float(A64.multiply(A64).sum()) + m)
            out["x_win"] = apx.adaptive_proxgrad(np.full(n + 1, 0.01), f=apx.LogisticLoss(dA, yy, ctx=ctx), g=apx.NormL1(1e-7), rule=apx.OurRule(gamma=gam),
                                                 tol=0.0, maxit=8, ctx=ctx)[0]
            out["yy"], out["gam"] = yy, gam
        elif not windows and "yy" in out:
            x_plain = apx.adaptive_proxgrad(np.full(n + 1, 0.01), f=apx.LogisticLoss(dA, out["yy"], ctx=ctx), g=apx.NormL1(1e-7),
                                            rule=apx.OurRule(gamma=out["gam"]), tol=0.0, maxit=8, ctx=ctx)[0]
            assert _rel(out["x_win"], x_plain) <= 1e-10 and np.any(x_plain != 0)
        dA.free()
    assert "L2 windows" in out[1][2][0] or "L2 windows" in out[1][2][1], out[1][2]
    # the streaming form where every row (of a window) fits its tile — and the lanes-per-row kernel where one does not
    if m == 450_000:
        assert out[1][2][0] == "spmv_stream<double>" and out[1][2][1].startswith("spmv_stream<double> in 2 L2 windows"), out[1][2]
    if m == 450_001:      # (X' row 0 has ~3000 entries in each of its two windows, X row 5 has 3000: blocks of their own in the streaming form)
        assert out[1][2][0] == "spmv_stream<double>" and out[1][2][1].startswith("spmv_stream<double> in 2 L2 windows"), out[1][2]
    assert "L2 windows" not in out[0][2][0] and "L2 windows" not in out[0][2][1], out[0][2]
    assert ("L2 windows" in out[1][2][1]) == (m * np.dtype(dtype).itemsize > 3 << 20)
    assert ("L2 windows" in out[1][2][0]) == (n * np.dtype(dtype).itemsize > 3 << 20)


def test_sparse_products_fuzz(apx):
    """Shapes, fills and irregularities drawn at random against scipy: empty rows and columns, a row / a column that holds a large
    share of the entries (a block of its own in the streaming form, or the long-row form), operands on both sides of the 3 MB
    from which a product runs in L2 windows, CSR and CSC input, both element types, the LDS-tile form forced or not ; every product twice
    (bit-reproducible)."""
    from hypothesis import HealthCheck, given, settings, strategies as st

    big = st.sampled_from([393_215, 393_217, 450_000, 700_001])
    small = st.one_of(st.integers(1, 3000), st.sampled_from([1, 2, 31, 32, 33, 255, 256, 257, 4097]))

    @settings(max_examples=100, deadline=None, derandomize=True, suppress_health_check=list(HealthCheck))
    @given(m=st.one_of(small, big), n=st.one_of(small, big), per_row=st.sampled_from([0.3, 1.0, 3.0, 9.0, 40.0]), heavy=st.sampled_from([0, 1, 2]),
           fmt=st.sampled_from(["csr", "csc"]), f32=st.booleans(), seed=st.integers(0, 2**31 - 1), tiles=st.booleans())
    def body(m, n, per_row, heavy, fmt, f32, seed, tiles):
        if m * per_row > 3_000_000:
            per_row = 3_000_000 / m
        rng = np.random.default_rng(seed)
        dtype = np.float32 if f32 else np.float64
        cnt = np.minimum(rng.poisson(per_row, m), n)
        rows = np.repeat(np.arange(m), cnt)
        cols = rng.integers(0, n, rows.size)
        if heavy == 1 and m > 1:      # one row with up to 5000 entries
            k = min(n, 5000)
            rows = np.concatenate([rows, np.full(k, m // 2)])
            cols = np.concatenate([cols, rng.choice(n, k, replace=False)])
        if heavy == 2 and n > 1:      # one column with up to 5000 entries
            k = min(m, 5000)
            rows = np.concatenate([rows, rng.choice(m, k, replace=False)])
            cols = np.concatenate([cols, np.full(k, n // 3)])
        A = sp.coo_matrix((rng.uniform(-1, 1, rows.size).astype(dtype), (rows, cols)), shape=(m, n)).asformat(fmt)
        A.sum_duplicates()
        x, r = rng.standard_normal(n).astype(dtype), rng.standard_normal(m).astype(dtype)
        ctx = apx.Context(0)
        ctx.fused_small(False)
        # the LDS-tile form on whatever is drawn, or never (by itself it takes matrices of a million entries or more)
        ctx.set_option(importlib.import_module("adaptive-proximal-algorithms_amd._lib").APX_OPT_SPARSE_WINDOWS, 3 if tiles else 1)
        dA = apx.DeviceMatrix.from_sparse(A, ctx)
        A64 = A.astype(np.float64)
        y_ref, g_ref = A64 @ x.astype(np.float64), A64.T @ r.astype(np.float64)
        tol = 1e-13 if dtype == np.float64 else 3e-6
        ctx.profile(True)
        y, g = dA @ x, dA.T @ r
        if A.nnz:
            assert ("spmv_tiles" in ctx.kernel_name("gemv_n")) == tiles and ("spmv_tiles" in ctx.kernel_name("gemv_t")) == tiles
        scale_y = max(1.0, float(np.max(np.abs(y_ref))) if m else 1.0)
        scale_g = max(1.0, float(np.max(np.abs(g_ref))) if n else 1.0)
        assert np.max(np.abs(y - y_ref)) <= tol * scale_y * max(1.0, np.sqrt(A.nnz / max(m, 1)) / 4), (m, n, per_row, heavy, fmt, f32)
        assert np.max(np.abs(g - g_ref)) <= tol * scale_g * max(1.0, np.sqrt(A.nnz / max(n, 1)) / 4), (m, n, per_row, heavy, fmt, f32)
        assert np.array_equal(dA @ x, y) and np.array_equal(dA.T @ r, g)
        assert dA.norm() == pytest.approx(np.sqrt(A64.multiply(A64).sum()), rel=1e-6 if f32 else 1e-12, abs=1e-300)
        dA.free()

    body()


def _tile_case(rng, m, n, per_row, dtype, heavy):
    cnt = np.minimum(rng.poisson(per_row, m), n)
    rows = np.repeat(np.arange(m), cnt)
    cols = rng.integers(0, n, rows.size)
    if heavy and m > 40 and n > 40:      # a full row, a full column, ten empty rows
        rows = np.concatenate([rows, np.full(n, 3), np.arange(m)])
        cols = np.concatenate([cols, np.arange(n), np.full(m, 7)])
    A = sp.coo_matrix((rng.uniform(-1, 1, rows.size).astype(dtype), (rows, cols)), shape=(m, n)).tocsr()
    A.sum_duplicates()
    if heavy and m > 40 and n > 40:
        A = A.tolil()
        A[10:20, :] = 0
        A = A.tocsr()
        A.eliminate_zeros()
    return A


@pytest.mark.parametrize("m,n,per_row,heavy", [(5, 5, 3, False), (300, 200, 10, True), (9000, 20_000, 40, True), (20_000, 9000, 40, True),
                                               (8192, 8192, 10, False), (8193, 16_385, 14, True),      # (a row block / a window of one row / column)
                                               (1, 50_000, 15_000, False), (50_000, 1, 0.5, False),     # one row ; one column
                                               (8192, 8192, 90, False),                                # 740 000 entries in ONE tile: two batches of chunks
                                               (40_000, 17_000, 0.3, True),                             # fewer entries than rows
                                               (70_000, 500, 8, False), (500, 70_000, 900, True)])      # one window: row blocks of 288 rows ; 500 rows: blocks of ~20
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_products_on_lds_tiles(apx, m, n, per_row, heavy, dtype):
    """Round 5 (csrc/sparse.hip spmv_tiles): the operand's 64 KB windows in LDS, the entries tile by tile with their local row and index
    in one word, the rows' sums in LDS across the windows — forced on every shape here (APX_OPT_SPARSE_WINDOWS 3 ; by itself only where
    it pays).  Against scipy to a few units of the last place of sum |a_ij x_j|, against the streaming / window forms, and twice (a
    row's sum has one owner and a fixed shape: bit-reproducible)."""
    L = importlib.import_module("adaptive-proximal-algorithms_amd._lib")
    rng = np.random.default_rng(m * 31 + n)
    A = _tile_case(rng, m, n, per_row, dtype, heavy)
    x, r = rng.standard_normal(n).astype(dtype), rng.standard_normal(m).astype(dtype)
    A64 = A.astype(np.float64)
    y_ref, g_ref = A64 @ x.astype(np.float64), A64.T @ r.astype(np.float64)
    sy = abs(A64) @ np.abs(x.astype(np.float64)) + 1e-300
    sg = abs(A64).T @ np.abs(r.astype(np.float64)) + 1e-300
    eps = np.finfo(dtype).eps
    out = {}
    for mode in (3, 1):
        ctx = apx.Context(0)
        ctx.fused_small(False)
        ctx.set_option(L.APX_OPT_SPARSE_WINDOWS, mode)
        dA = apx.DeviceMatrix.from_sparse(A if mode == 3 else A.tocsc(), ctx)
        ctx.profile(True)
        y, g = dA @ x, dA.T @ r
        names = (ctx.kernel_name("gemv_n"), ctx.kernel_name("gemv_t"))
        assert ("spmv_tiles" in names[0]) == (mode == 3) and ("spmv_tiles" in names[1]) == (mode == 3), names
        if mode == 3 and (m, n) == (70_000, 500):      # (blocks x windows must give every CU an item: 70 016 rows / 256)
            assert "row blocks of up to 288," in names[0] and 200 <= int(names[1].split(": ")[1].split()[0]) <= 256, names
        assert np.max(np.abs(y - y_ref) / sy) <= 4 * eps and np.max(np.abs(g - g_ref) / sg) <= 4 * eps
        assert np.array_equal(dA @ x, y) and np.array_equal(dA.T @ r, g)
        assert dA.norm() == pytest.approx(np.sqrt(A64.multiply(A64).sum()), rel=1e-6 if dtype == np.float32 else 1e-12)
        out[mode] = (y, g)
        dA.free()
    assert np.max(np.abs(out[3][0] - out[1][0]) / sy) <= 4 * eps and np.max(np.abs(out[3][1] - out[1][1]) / sg) <= 4 * eps


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_lds_tiles_on_a_power_law_matrix(apx, dtype):
    """The head of a power-law X' — a few hundred features holding half of the entries — is cut into row blocks of its own (a block ends
    where its entries exceed what its windows can be divided into: csrc/sparse.hip plan_tiles) ; the products against scipy, twice."""
    L = importlib.import_module("adaptive-proximal-algorithms_amd._lib")
    rng = np.random.default_rng(8)
    m, n = 60_000, 3000
    cnt = rng.poisson(20, m)
    rows = np.repeat(np.arange(m), cnt)
    cols = np.minimum((n * rng.uniform(size=rows.size) ** 3).astype(np.int64), n - 1)
    A = sp.coo_matrix((rng.uniform(-1, 1, rows.size).astype(dtype), (rows, cols)), shape=(m, n)).tocsr()
    A.sum_duplicates()
    ctx = apx.Context(0)
    ctx.fused_small(False)
    ctx.set_option(L.APX_OPT_SPARSE_WINDOWS, 3)
    dA = apx.DeviceMatrix.from_sparse(A, ctx)
    ctx.profile(True)
    x, r = rng.standard_normal(n).astype(dtype), rng.standard_normal(m).astype(dtype)
    y, g = dA @ x, dA.T @ r
    assert "spmv_tiles" in ctx.kernel_name("gemv_n") and "spmv_tiles" in ctx.kernel_name("gemv_t")
    A64 = A.astype(np.float64)
    eps = np.finfo(dtype).eps
    assert np.max(np.abs(y - A64 @ x.astype(np.float64)) / (abs(A64) @ np.abs(x.astype(np.float64)) + 1e-300)) <= 4 * eps
    assert np.max(np.abs(g - A64.T @ r.astype(np.float64)) / (abs(A64).T @ np.abs(r.astype(np.float64)) + 1e-300)) <= 4 * eps
    assert np.array_equal(dA @ x, y) and np.array_equal(dA.T @ r, g)
    dA.free()


@pytest.mark.parametrize("layout", [0, 1])
def test_lds_tiles_take_unsorted_rows_and_duplicates_through_the_c_abi(apx, layout):
    """apx_matrix_upload_sparse asks for neither sorted indices nor unique ones (the Python wrapper sorts ; a ccall from Julia hands over
    whatever `SparseMatrixCSC` holds): the tiles bucket the entries by window in the caller's order, duplicates add up like in the rows."""
    import ctypes as C
    L = importlib.import_module("adaptive-proximal-algorithms_amd._lib")
    rng = np.random.default_rng(11)
    m, n, nnz = 9000, 21_000, 400_000
    nr, nc = (m, n) if layout == 0 else (n, m)      # layout 0: rows of A (CSR) ; 1: columns of A (CSC)
    cnt = rng.multinomial(nnz, np.full(nr, 1.0 / nr))
    ptr = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64)
    idx = rng.integers(0, nc, nnz).astype(np.int32)          # unsorted within a row, with repeats
    idx[1::7] = idx[0::7][: idx[1::7].size]                  # (many immediate duplicates)
    val = rng.uniform(-1, 1, nnz)
    D = np.zeros((m, n))
    rows = np.repeat(np.arange(nr), cnt)
    np.add.at(D, (rows, idx) if layout == 0 else (idx, rows), val)
    ctx = apx.Context(0)
    ctx.fused_small(False)
    ctx.set_option(L.APX_OPT_SPARSE_WINDOWS, 3)
    h = C.c_void_p()
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    L.check(ctx._lib.apx_matrix_upload_sparse(ctx._h, m, n, nnz, layout, p(ptr), p(idx), p(val), L.APX_F64, C.byref(h)))
    x, r = rng.standard_normal(n), rng.standard_normal(m)
    y, g = np.empty(m), np.empty(n)
    ctx.profile(True)
    L.check(ctx._lib.apx_gemv_n(h, p(x), p(y)))
    L.check(ctx._lib.apx_gemv_t(h, p(r), p(g)))
    assert "spmv_tiles" in ctx.kernel_name("gemv_n") and "spmv_tiles" in ctx.kernel_name("gemv_t")
    assert np.max(np.abs(y - D @ x)) <= 1e-12 * np.max(np.abs(D @ x)) and np.max(np.abs(g - D.T @ r)) <= 1e-12 * np.max(np.abs(D.T @ r))
    L.check(ctx._lib.apx_matrix_free(h))


def test_lds_tiles_by_default_and_through_a_solver(apx):
    """What a matrix gets BY ITSELF (APX_OPT_SPARSE_WINDOWS 2, the default): 100 000 x 60 000 with 5.5 M entries — 55 000 per tile — runs
    both products on the LDS tiles ; the same shape with 0.5 M entries (below the million) keeps the streaming / window forms.  And a sparse logistic
    regression on the tiles follows the one on the round-5 forms (the fused epilogues take the tiles' splits like the windows')."""
    L = importlib.import_module("adaptive-proximal-algorithms_amd._lib")
    rng = np.random.default_rng(5)
    m, n = 100_000, 60_000
    for per_row, want in ((55, True), (5, False)):
        A = _tile_case(rng, m, n, per_row, np.float64, False)
        ctx = apx.Context(0)
        assert ctx.get_option(L.APX_OPT_SPARSE_WINDOWS) == 2
        dA = apx.DeviceMatrix.from_sparse(A, ctx)
        ctx.profile(True)
        x, r = rng.standard_normal(n), rng.standard_normal(m)
        y, g = dA @ x, dA.T @ r
        names = (ctx.kernel_name("gemv_n"), ctx.kernel_name("gemv_t"))
        assert all(("spmv_tiles" in k) == want for k in names), names
        assert np.allclose(y, A @ x, rtol=0, atol=1e-12 * np.max(np.abs(y))) and np.allclose(g, A.T @ r, rtol=0, atol=1e-12 * np.max(np.abs(g)))
        if want:
            yy = (rng.uniform(size=m) < 0.5).astype(np.float64)
            gam = 4.0 * m / (float(A.multiply(A).sum()) + m)
            run = lambda c, d: apx.adaptive_proxgrad(np.full(n + 1, 0.01), f=apx.LogisticLoss(d, yy, ctx=c), g=apx.NormL1(1e-7),
                                                     rule=apx.OurRule(gamma=gam), tol=0.0, maxit=12, ctx=c)[0]
            x_tiles = run(ctx, dA)
            ctx1 = apx.Context(0)
            ctx1.set_option(L.APX_OPT_SPARSE_WINDOWS, 1)
            dA1 = apx.DeviceMatrix.from_sparse(A, ctx1)
            x_stream = run(ctx1, dA1)
            assert _rel(x_tiles, x_stream) <= 1e-10 and np.any(x_stream != 0.01)
            dA1.free()
        dA.free()


@pytest.mark.parametrize("m,n,spmv", [(3000, 112, True), (3000, 112, False), (6414, 123, False), (1500, 400, False),
                                      (20000, 12, True)])          # (long rows of X': the chunked SpMV)
def test_sparse_logreg_matches_oracle(apx, generic_ctx, m, n, spmv):
    """experiments/sparse_logreg/runme.jl:41-63 with a sparse X, Float64 (as the reference runs it): on the SpMV
    kernels (spmv: a context without the small-size policy) and, by default, on the dense copy a small sparse
    matrix is given (mushrooms- and a5a-like shapes: csrc/skinny_pg.hip ; 1500 x 400: csrc/small.hip)."""
    rng = np.random.default_rng(0)
    ctx = generic_ctx if spmv else apx.default_context()
    X = sp.random(m, n, density=0.19, format="csc", random_state=rng, data_rvs=lambda k: np.ones(k))
    w_true = rng.standard_normal(n) * (rng.uniform(size=n) < 0.2)
    y = (1 / (1 + np.exp(-(X @ w_true))) > rng.uniform(size=m)).astype(np.float64)
    X1 = sp.hstack([X, np.ones((m, 1))]).tocsr()
    if m < 10000:
        Lf = sp.linalg.norm(X1 @ X1.T) / 4 / m              # :58-59 (Frobenius of the Gram matrix)
    else:                                                   # (the m x m Gram matrix of the tall case is ten seconds: the bound
        Lf = X1.multiply(X1).sum() / 4 / m                  #  ||X1||_F^2 / 4m >= it, SURVEY 8d's choice for C3)
    gam = 1.0 / Lf
    K = 40
    orecs, drecs = [], []
    xo, _ = O.adaptive_proxgrad(np.zeros(n + 1), f=O.LogisticLoss(X.tocsr(), y), g=O.NormL1(0.01),
                                rule=O.OurRule(gamma=gam), tol=0.0, maxit=K, record=orecs.append)
    fX = apx.LogisticLoss(apx.DeviceMatrix.from_sparse(X, ctx), y, ctx=ctx)
    x, _ = apx.adaptive_proxgrad(np.zeros(n + 1), f=fX, g=apx.NormL1(0.01),
                                 rule=apx.OurRule(gamma=gam), tol=0.0, maxit=K, record=drecs.append, ctx=ctx)
    assert _rel(x, xo) <= 1e-10
    for ro, rd in zip(orecs, drecs):
        if ro["norm_res"] <= 1e-4 * orecs[0]["norm_res"]:      # (the step size is a ratio of differences of iterates:
            break                                              #  near convergence its digits are rounding)
        assert rd["gamma"] == pytest.approx(ro["gamma"], rel=1e-8)
        assert rd["objective"] == pytest.approx(ro["objective"], rel=1e-11)
    # backtracking and the PD loop run on the same storage
    Kb = 25 if m < 10000 else 8            # (tall problems contract fast: `f_z > ub_z` soon compares rounding errors)
    zo, _ = O.backtracking_proxgrad(np.zeros(n + 1), f=O.LogisticLoss(X.tocsr(), y), g=O.NormL1(0.01), gamma0=gam,
                                    xi=1.5, tol=0.0, maxit=Kb)
    z, _ = apx.backtracking_proxgrad(np.zeros(n + 1), f=fX, g=apx.NormL1(0.01), gamma0=gam, xi=1.5,
                                     tol=0.0, maxit=Kb, ctx=ctx)
    assert _rel(z, zo) <= 1e-10


@pytest.mark.parametrize("m,n,spmv", [(700, 90, True), (700, 90, False), (5000, 14, False)])
def test_sparse_lad_condat_vu(apx, generic_ctx, m, n, spmv):
    rng = np.random.default_rng(5)
    ctx = generic_ctx if spmv else apx.default_context()
    A = sp.hstack([sp.random(m, n - 1, density=0.1, random_state=rng), np.ones((m, 1))]).tocsr()
    yv = rng.standard_normal(m)
    nA = float(sp.linalg.norm(A))
    kw = dict(Lf=0.0, norm_A=nA, tol=0.0, maxit=150)
    xo, yo, _ = O.condat_vu(np.zeros(n), np.zeros(m), f=O.Zero(), g=O.NormL1(2.0), h=O.Translate(O.NormL1(1.0), -yv),
                            A=A, **kw)
    x, y, _ = apx.condat_vu(np.zeros(n), np.zeros(m), f=apx.Zero(), g=apx.NormL1(2.0),
                            h=apx.Translate(apx.NormL1(1.0), -yv), A=apx.DeviceMatrix.from_sparse(A, ctx), ctx=ctx, **kw)
    assert _rel(x, xo) <= 1e-10 and _rel(y, yo) <= 1e-10


@pytest.mark.parametrize("method", ["condat_vu", "adaptive_primal_dual", "malitsky_pock"])
def test_sparse_lad_on_lds_tiles_matches_the_oracle(apx, method):
    """experiments/least_absolute_deviation/runme.jl:41-94 on a sparse A of 2.4 M entries (150 000 x 2500): both products of the
    primal-dual loops run on the LDS tiles by default ; 60 iterations against the oracle on scipy's products."""
    rng = np.random.default_rng(6)
    m, n = 150_000, 2500
    cnt = np.maximum(rng.poisson(16, m), 1)
    A = sp.csr_matrix((rng.uniform(-1, 1, cnt.sum()), rng.integers(0, n, cnt.sum()), np.concatenate([[0], np.cumsum(cnt)])), shape=(m, n))
    A.sum_duplicates()
    yv = A @ (rng.standard_normal(n) * (rng.uniform(size=n) < 0.1)) + 0.1 * rng.standard_normal(m)
    nA = float(sp.linalg.svds(A, k=1, return_singular_vectors=False)[0])
    ctx = apx.Context(0)
    dA = apx.DeviceMatrix.from_sparse(A, ctx)
    ctx.profile(True)
    K = 60
    if method == "condat_vu":
        kw = dict(Lf=0.0, norm_A=nA, tol=0.0, maxit=K)
    elif method == "adaptive_primal_dual":
        kw = dict(rule=None, tol=0.0, maxit=K)
    else:
        kw = dict(sigma=1.0 / nA, tol=0.0, maxit=K)
    terms = lambda M: dict(f=M.Zero(), g=M.NormL1(0.5), h=M.Translate(M.NormL1(1.0), -yv))
    if method == "adaptive_primal_dual":
        kw["rule"] = O.OurRule(t=1.0, norm_A=nA)
        xo, yo, _ = O.adaptive_primal_dual(np.zeros(n), np.zeros(m), A=A, **terms(O), **kw)
        kw["rule"] = apx.OurRule(t=1.0, norm_A=nA)
        x, y, _ = apx.adaptive_primal_dual(np.zeros(n), np.zeros(m), A=dA, ctx=ctx, **terms(apx), **kw)
    else:
        xo, yo, _ = getattr(O, method)(np.zeros(n), np.zeros(m), A=A, **terms(O), **kw)
        x, y, _ = getattr(apx, method)(np.zeros(n), np.zeros(m), A=dA, ctx=ctx, **terms(apx), **kw)
    assert "spmv_tiles" in ctx.kernel_name("gemv_n") and "spmv_tiles" in ctx.kernel_name("gemv_t")
    assert _rel(x, xo) <= 1e-10 and _rel(y, yo) <= 1e-10 and np.any(x != 0)
    dA.free()


def test_libsvm_reader(apx, tmp_path):
    lib = importlib.import_module("adaptive-proximal-algorithms_amd.libsvm")
    path = tmp_path / "toy"
    path.write_text("1 1:0.5 3:2\n2 2:-1\n1 1:1 4:0.25\n2 \n")
    X, y = lib.load_libsvm_dataset(str(path), np.float64, labels=[0.0, 1.0])      # sparse_logreg/runme.jl:50
    assert X.shape == (4, 4) and X.nnz == 5
    assert np.array_equal(y, [0.0, 1.0, 0.0, 1.0])                                  # labels {1,2} -> {0,1}
    assert np.array_equal(X.toarray(), [[0.5, 0, 2, 0], [0, -1, 0, 0], [1, 0, 0, 0.25], [0, 0, 0, 0]])
    X2, y2 = lib.load_libsvm_dataset(str(path), np.float32)
    assert X2.dtype == np.float32 and np.array_equal(y2, [1, 2, 1, 2])
    dX = apx.DeviceMatrix.from_sparse(X)
    assert np.allclose(dX @ np.ones(4), X @ np.ones(4))
