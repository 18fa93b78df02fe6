"""The C-ABI called directly (ctypes, no Python mirror in between) with bad arguments: every entry point
reports a status + `apx_last_error()` instead of crashing or throwing across the boundary (SURVEY §8b
"Errors"): NULL handles, wrong dtype / shapes / contexts, the rule constructors' `error(...)` texts
(src/AdaProx.jl:246, :288), negative maxit, a second solver on a busy problem."""
import ctypes as C
import importlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

L = importlib.import_module("adaptive-proximal-algorithms_amd._lib")


@pytest.fixture(scope="module")
def lib():
    return L.load()


@pytest.fixture(scope="module")
def ctx(lib):
    h = C.c_void_p()
    rc = lib.apx_ctx_create(0, None, C.byref(h))
    assert rc == L.APX_OK, lib.apx_last_error().decode()
    yield h
    lib.apx_ctx_destroy(h)


def _err(lib):
    return lib.apx_last_error().decode()


def _matrix(lib, ctx, A):
    A = np.asfortranarray(A)
    h = C.c_void_p()
    code = L.APX_F32 if A.dtype == np.float32 else L.APX_F64
    assert lib.apx_matrix_upload(ctx, A.ctypes.data_as(C.c_void_p), A.shape[0], A.shape[1], A.shape[0], code,
                                 C.byref(h)) == L.APX_OK
    return h


def _ls_desc(A_h, b, n, dtype=L.APX_F64):
    d = L.ProblemDesc()
    d.f_kind, d.dtype, d.n = L.APX_F_LEAST_SQUARES, dtype, n
    d.A_f = A_h
    d.b = b.ctypes.data_as(C.c_void_p)
    d.g.kind, d.g.lam = L.APX_G_L1, 1.0
    d.h.kind = L.APX_G_ZERO
    return d


def test_null_and_bad_handles(lib, ctx):
    assert lib.apx_ctx_create(0, None, None) == L.APX_ERR_INVALID and "NULL" in _err(lib)
    assert lib.apx_ctx_create(99, None, C.byref(C.c_void_p())) == L.APX_ERR_INVALID and "device" in _err(lib)
    assert lib.apx_ctx_sync(None) == L.APX_ERR_INVALID
    assert lib.apx_matrix_create(ctx, -1, 3, L.APX_F64, C.byref(C.c_void_p())) == L.APX_ERR_INVALID
    assert lib.apx_matrix_create(ctx, 2, 3, 7, C.byref(C.c_void_p())) == L.APX_ERR_INVALID and "dtype" in _err(lib)
    assert lib.apx_matrix_upload(ctx, None, 2, 2, 2, L.APX_F64, C.byref(C.c_void_p())) == L.APX_ERR_INVALID
    A = np.zeros((4, 3))
    assert lib.apx_matrix_upload(ctx, A.ctypes.data_as(C.c_void_p), 4, 3, 2, L.APX_F64,
                                 C.byref(C.c_void_p())) == L.APX_ERR_INVALID and "lda" in _err(lib)
    assert lib.apx_gemv_n(None, None, None) == L.APX_ERR_INVALID
    assert lib.apx_problem_create(ctx, None, C.byref(C.c_void_p())) == L.APX_ERR_INVALID
    assert lib.apx_solver_step(None, 1, 1) == L.APX_ERR_INVALID
    assert lib.apx_solver_result(None, None, None, None) == L.APX_ERR_INVALID
    assert lib.apx_matrix_free(None) == L.APX_OK and lib.apx_problem_destroy(None) == L.APX_OK   # free(NULL) is fine
    assert lib.apx_solver_destroy(None) == L.APX_OK


def test_matrix_layout_entry_points_validate(lib, ctx):
    """ABI 5: the one-copy layout (apx_matrix_create_layout / upload_layout / drop_primary / layout) and apx_ctx_device_clock
    through the raw C-ABI — bad layouts and NULLs are status codes, a rows-only matrix refuses to give its only copy away."""
    h = C.c_void_p()
    assert lib.apx_matrix_create_layout(ctx, 4, 3, L.APX_F64, 2, C.byref(h)) == L.APX_ERR_INVALID and "layout" in _err(lib)
    assert lib.apx_matrix_create_layout(ctx, 4, 3, L.APX_F64, L.APX_LAYOUT_ROWS, None) == L.APX_ERR_INVALID
    assert lib.apx_matrix_upload_layout(ctx, None, 4, 3, 4, L.APX_F64, L.APX_LAYOUT_ROWS, C.byref(h)) == L.APX_ERR_INVALID
    A = np.asfortranarray(np.arange(12.0).reshape(4, 3))
    assert lib.apx_matrix_upload_layout(ctx, A.ctypes.data_as(C.c_void_p), 4, 3, 3, L.APX_F64, L.APX_LAYOUT_ROWS, C.byref(h)) == L.APX_ERR_INVALID   # lda < m
    assert lib.apx_matrix_upload_layout(ctx, A.ctypes.data_as(C.c_void_p), 4, 3, 4, L.APX_F64, L.APX_LAYOUT_ROWS, C.byref(h)) == L.APX_OK
    mask = C.c_int32(-1)
    assert lib.apx_matrix_layout(h, C.byref(mask)) == L.APX_OK and mask.value == 2          # the row storage alone
    assert lib.apx_matrix_layout(h, None) == L.APX_ERR_INVALID and lib.apx_matrix_layout(None, C.byref(mask)) == L.APX_ERR_INVALID
    assert lib.apx_matrix_drop_transpose(h) == L.APX_ERR_INVALID and "no other copy" in _err(lib)
    assert lib.apx_matrix_build_transpose(h) == L.APX_OK and lib.apx_matrix_drop_primary(h) == L.APX_OK      # nothing to do, twice
    out = np.zeros((4, 3), order="F")
    assert lib.apx_matrix_download(h, out.ctypes.data_as(C.c_void_p), 4) == L.APX_OK and np.array_equal(out, A)
    y = np.zeros(4)
    x = np.array([1.0, -1.0, 2.0])
    assert lib.apx_gemv_n(h, x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p)) == L.APX_OK and np.allclose(y, A @ x, rtol=1e-15)
    assert lib.apx_matrix_drop_primary(None) == L.APX_ERR_INVALID
    lib.apx_matrix_free(h)
    d, hh, u = C.c_int64(), C.c_int64(), C.c_int64()
    assert lib.apx_ctx_device_clock(ctx, None, C.byref(hh), C.byref(u)) == L.APX_ERR_INVALID
    assert lib.apx_ctx_device_clock(ctx, C.byref(d), C.byref(hh), None) == L.APX_OK and d.value > 0 and hh.value > 1_600_000_000 * 10**9
    d2 = C.c_int64()
    assert lib.apx_ctx_device_clock(ctx, C.byref(d2), C.byref(hh), C.byref(u)) == L.APX_OK and d2.value > d.value and 0 <= u.value < 10**9


def test_problem_shape_dtype_and_term_checks(lib, ctx):
    A = _matrix(lib, ctx, np.ones((4, 3)))
    b = np.zeros(4)
    out = C.c_void_p()
    d = _ls_desc(A, b, 5)
    assert lib.apx_problem_create(ctx, C.byref(d), C.byref(out)) == L.APX_ERR_INVALID and "columns" in _err(lib)
    d = _ls_desc(A, b, 3, dtype=L.APX_F32)
    assert lib.apx_problem_create(ctx, C.byref(d), C.byref(out)) == L.APX_ERR_INVALID and "dtype" in _err(lib)
    d = _ls_desc(A, b, 3)
    d.b = None
    assert lib.apx_problem_create(ctx, C.byref(d), C.byref(out)) == L.APX_ERR_INVALID
    d = _ls_desc(A, b, 3)
    d.h.kind = L.APX_G_L1                                  # h without a linear operator
    assert lib.apx_problem_create(ctx, C.byref(d), C.byref(out)) == L.APX_ERR_INVALID and "h must be Zero" in _err(lib)
    d = _ls_desc(A, b, 3)
    d.g.kind = L.APX_G_L2
    assert lib.apx_problem_create(ctx, C.byref(d), C.byref(out)) == L.APX_ERR_UNSUPPORTED
    d = _ls_desc(A, b, 3)
    d.g.kind = 42
    assert lib.apx_problem_create(ctx, C.byref(d), C.byref(out)) == L.APX_ERR_INVALID and "prox kind" in _err(lib)
    d = _ls_desc(A, b, 3)
    d.f_kind = 17
    assert lib.apx_problem_create(ctx, C.byref(d), C.byref(out)) == L.APX_ERR_INVALID
    d = _ls_desc(A, b, 3)
    d.f_kind = L.APX_F_CALLBACK                            # callbacks missing
    assert lib.apx_problem_create(ctx, C.byref(d), C.byref(out)) == L.APX_ERR_INVALID and "callback" in _err(lib)
    # a matrix of ANOTHER context
    other = C.c_void_p()
    assert lib.apx_ctx_create(0, None, C.byref(other)) == L.APX_OK
    A2 = _matrix(lib, other, np.ones((4, 3)))
    d = _ls_desc(A2, b, 3)
    assert lib.apx_problem_create(ctx, C.byref(d), C.byref(out)) == L.APX_ERR_INVALID and "context" in _err(lib)
    lib.apx_matrix_free(A2)
    lib.apx_ctx_destroy(other)
    # linear operator with the wrong number of columns
    Al = _matrix(lib, ctx, np.ones((2, 7)))
    d = _ls_desc(A, b, 3)
    d.A_lin = Al
    assert lib.apx_problem_create(ctx, C.byref(d), C.byref(out)) == L.APX_ERR_INVALID and "columns" in _err(lib)
    lib.apx_matrix_free(Al)
    lib.apx_matrix_free(A)


def test_rules_maxit_and_busy_problem(lib, ctx):
    rng = np.random.default_rng(0)
    A = _matrix(lib, ctx, rng.uniform(-1, 1, (6, 4)))
    b = rng.uniform(-1, 1, 6)
    d = _ls_desc(A, b, 4)
    p = C.c_void_p()
    assert lib.apx_problem_create(ctx, C.byref(d), C.byref(p)) == L.APX_OK
    x0, x = np.zeros(4), np.zeros(4)
    res = L.Result()
    px0, px = x0.ctypes.data_as(C.c_void_p), x.ctypes.data_as(C.c_void_p)

    r = L.Rule()
    r.kind, r.t, r.Theta = L.APX_RULE_OURS, 1.0, 1.2       # OurRule() : gamma = 0 and norm_A = 0
    assert lib.apx_adaptive_proxgrad(p, C.byref(r), px0, 0.0, 5, None, px, C.byref(res)) == L.APX_ERR_INVALID
    assert _err(lib) == "you must provide gamma > 0 if norm_A = 0"                  # src/AdaProx.jl:246
    r.kind = L.APX_RULE_OURS_PLUS
    assert lib.apx_adaptive_proxgrad(p, C.byref(r), px0, 0.0, 5, None, px, C.byref(res)) == L.APX_ERR_INVALID
    assert _err(lib) == "you must provide gamma > 0"                                # src/AdaProx.jl:288
    r.kind = 9
    assert lib.apx_adaptive_proxgrad(p, C.byref(r), px0, 0.0, 5, None, px, C.byref(res)) == L.APX_ERR_INVALID
    r.kind, r.gamma = L.APX_RULE_OURS, 0.05
    assert lib.apx_adaptive_proxgrad(p, C.byref(r), px0, 0.0, -1, None, px, C.byref(res)) == L.APX_ERR_INVALID
    assert lib.apx_adaptive_proxgrad(p, C.byref(r), None, 0.0, 5, None, px, C.byref(res)) == L.APX_ERR_INVALID
    assert lib.apx_adaptive_proxgrad(None, C.byref(r), px0, 0.0, 5, None, px, C.byref(res)) == L.APX_ERR_INVALID
    assert lib.apx_fixed_proxgrad(p, -1.0, px0, 0.0, 5, None, px, C.byref(res)) == L.APX_ERR_INVALID
    assert lib.apx_condat_vu(p, 0.0, 1.0, px0, px0, 0.0, 5, None, px, px, C.byref(res)) == L.APX_ERR_INVALID
    assert "linear operator" in _err(lib)
    # a valid call still works after all those failures, and leaves the problem reusable
    assert lib.apx_adaptive_proxgrad(p, C.byref(r), px0, 0.0, 5, None, px, C.byref(res)) == L.APX_OK
    assert res.numit == 5 and np.all(np.isfinite(x))

    # one solver at a time per problem; op-level calls are refused while it lives
    s1, s2 = C.c_void_p(), C.c_void_p()
    assert lib.apx_solver_create(p, C.byref(r), px0, None, 0.0, None, C.byref(s1)) == L.APX_OK
    assert lib.apx_solver_create(p, C.byref(r), px0, None, 0.0, None, C.byref(s2)) == L.APX_ERR_INVALID
    assert "another solver" in _err(lib)
    fx = C.c_double()
    assert lib.apx_eval(p, px0, C.byref(fx)) == L.APX_ERR_INVALID and "active" in _err(lib)
    assert lib.apx_solver_step(s1, -3, 1) == L.APX_ERR_INVALID
    assert lib.apx_solver_step(s1, 3, 1) == L.APX_OK
    assert lib.apx_solver_destroy(s1) == L.APX_OK
    assert lib.apx_eval(p, px0, C.byref(fx)) == L.APX_OK and fx.value == pytest.approx(0.5 * b @ b, rel=1e-14)
    assert lib.apx_pullback(p, px) == L.APX_OK and np.allclose(x, -(np.asarray(b) @ _download(lib, A, 6, 4)), rtol=1e-13)
    lib.apx_problem_destroy(p)
    lib.apx_matrix_free(A)


def _download(lib, A_h, m, n):
    out = np.zeros((m, n), order="F")
    assert lib.apx_matrix_download(A_h, out.ctypes.data_as(C.c_void_p), m) == L.APX_OK
    return out


def test_prox_entry_points_validate(lib, ctx):
    v, y = np.ones(5), np.zeros(5)
    pd = L.ProxDesc()
    pd.kind, pd.lam = L.APX_G_L1, 0.5
    val = C.c_double()
    pv, py = v.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p)
    assert lib.apx_prox(ctx, C.byref(pd), 0, L.APX_F64, pv, 0, 1.0, py, C.byref(val)) == L.APX_ERR_INVALID
    assert lib.apx_prox(ctx, C.byref(pd), 0, 5, pv, 5, 1.0, py, C.byref(val)) == L.APX_ERR_INVALID
    assert lib.apx_prox(ctx, None, 0, L.APX_F64, pv, 5, 1.0, py, C.byref(val)) == L.APX_ERR_INVALID
    pd.kind = L.APX_G_BOXV                                  # vector bounds missing
    assert lib.apx_prox(ctx, C.byref(pd), 0, L.APX_F64, pv, 5, 1.0, py, C.byref(val)) == L.APX_ERR_INVALID
    pd.kind = L.APX_G_L1
    assert lib.apx_prox(ctx, C.byref(pd), 0, L.APX_F64, pv, 5, 1.0, py, C.byref(val)) == L.APX_OK
    assert np.allclose(y, 0.5) and val.value == pytest.approx(0.5 * 2.5)
    assert lib.apx_prox_value(ctx, C.byref(pd), L.APX_F64, pv, 5, None) == L.APX_ERR_INVALID


def test_options_one_pass_switch_and_names(lib, ctx):
    """ABI 3: options by key instead of environment variables, the switch of the one-pass kernels, the timing slot's kernel name,
    the incident text and the hand-off statistics — all through the raw C-ABI."""
    v = C.c_int64(-7)
    for key, default in ((L.APX_OPT_MULTI_ROWS_MASK, 15), (L.APX_OPT_SLAB_GENERATION, 0), (L.APX_OPT_TEAM_WAIT_MS, 1000),
                         (L.APX_OPT_TEAM_COOPERATIVE, 0), (L.APX_OPT_TEAM_FAULT_GRID, 0), (L.APX_OPT_FUSED_TAIL, 1),
                         (L.APX_OPT_MULTI_WAVES, 0), (L.APX_OPT_TEAM_MEMBERS, 0), (L.APX_OPT_TEAM_SLOTS, 0), (L.APX_OPT_ROW_DEAL, -1),
                         (L.APX_OPT_TEAM_INTERLEAVE, 1), (L.APX_OPT_TEAM_PD_DEFER, 1), (L.APX_OPT_FAULT_SETUP, 0),
                         (L.APX_OPT_SPARSE_WINDOWS, 2), (L.APX_OPT_ONE_PASS_MIN_ROWS, 4096)):
        assert lib.apx_ctx_get_option(ctx, key, C.byref(v)) == L.APX_OK and v.value == default, (key, v.value)
    assert lib.apx_ctx_set_option(ctx, L.APX_OPT_TEAM_WAIT_MS, 250) == L.APX_OK
    assert lib.apx_ctx_get_option(ctx, L.APX_OPT_TEAM_WAIT_MS, C.byref(v)) == L.APX_OK and v.value == 250
    assert lib.apx_ctx_set_option(ctx, L.APX_OPT_TEAM_WAIT_MS, 1000) == L.APX_OK
    for key, bad in ((999, 1), (L.APX_OPT_MULTI_ROWS_MASK, 16), (L.APX_OPT_TEAM_WAIT_MS, 0), (L.APX_OPT_TEAM_MEMBERS, 3),
                     (L.APX_OPT_ROW_DEAL, 3), (L.APX_OPT_MULTI_WAVES, 9), (L.APX_OPT_TEAM_SLOTS, 43), (L.APX_OPT_TEAM_FAULT_GRID, 513),
                     (L.APX_OPT_TEAM_INTERLEAVE, 3), (L.APX_OPT_TEAM_PD_DEFER, 2), (L.APX_OPT_FAULT_SETUP, 2), (L.APX_OPT_SPARSE_WINDOWS, 4), (L.APX_OPT_SPARSE_WINDOWS, -1),
                     (L.APX_OPT_ONE_PASS_MIN_ROWS, 255), (L.APX_OPT_ONE_PASS_MIN_ROWS, 4097)):
        assert lib.apx_ctx_set_option(ctx, key, bad) == L.APX_ERR_INVALID and _err(lib), (key, bad)
    assert lib.apx_ctx_get_option(ctx, 999, C.byref(v)) == L.APX_ERR_INVALID
    assert lib.apx_ctx_get_option(ctx, L.APX_OPT_ROW_DEAL, None) == L.APX_ERR_INVALID
    assert lib.apx_ctx_set_option(None, L.APX_OPT_ROW_DEAL, 0) == L.APX_ERR_INVALID
    for mode in (0, 2, 1):
        assert lib.apx_ctx_one_pass(ctx, mode) == L.APX_OK
    assert lib.apx_ctx_one_pass(ctx, 3) == L.APX_ERR_INVALID and "mode" in _err(lib)
    assert lib.apx_ctx_one_pass(None, 1) == L.APX_ERR_INVALID
    assert lib.apx_ctx_kernel_name(ctx, 99) == b"" and lib.apx_ctx_kernel_name(None, 0) == b""
    assert isinstance(lib.apx_last_incident(), bytes)
    # a problem that runs no team kernels: all-zero statistics ; NULL arguments are refused
    A = _matrix(lib, ctx, np.random.default_rng(0).standard_normal((40, 30)))
    b = np.zeros(40)
    d = L.ProblemDesc()
    d.f_kind, d.dtype, d.n, d.A_f, d.b = L.APX_F_LEAST_SQUARES, L.APX_F64, 30, A, b.ctypes.data_as(C.c_void_p)
    d.g.kind = L.APX_G_L1
    d.g.lam = 1.0
    p = C.c_void_p()
    assert lib.apx_problem_create(ctx, C.byref(d), C.byref(p)) == L.APX_OK, _err(lib)
    out = (C.c_int64 * 4)(1, 2, 3, 4)
    assert lib.apx_problem_team_waits(p, out) == L.APX_OK and list(out) == [0, 0, 0, 0]
    assert lib.apx_problem_team_waits(None, out) == L.APX_ERR_INVALID and lib.apx_problem_team_waits(p, None) == L.APX_ERR_INVALID
    # ... and slot by slot (ABI 4): nothing to report for a problem without team kernels ; a count without a buffer is fine
    slots, cnt = (L.TeamWaitSlot * 4)(), C.c_int32(-1)
    assert lib.apx_problem_team_wait_slots(p, slots, 4, C.byref(cnt)) == L.APX_OK and cnt.value == 0
    assert lib.apx_problem_team_wait_slots(p, None, 0, C.byref(cnt)) == L.APX_OK and cnt.value == 0
    assert lib.apx_problem_team_wait_slots(p, None, 4, C.byref(cnt)) == L.APX_ERR_INVALID
    assert lib.apx_problem_team_wait_slots(None, slots, 4, C.byref(cnt)) == L.APX_ERR_INVALID
    assert lib.apx_problem_team_wait_slots(p, slots, 4, None) == L.APX_ERR_INVALID
    lib.apx_problem_destroy(p)
    lib.apx_matrix_free(A)
