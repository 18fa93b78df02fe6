"""f and grad f in one pass over A (csrc/slab.hip: rows held in the registers of a workgroup, used by the generic loop
when the matrix has its transposed copy) against the oracle and against the two GEMV passes."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import adaprox_oracle as O


@pytest.fixture(scope="module")
def apx():
    import adaprox_amd as apx
    apx.load()
    return apx


@pytest.fixture(scope="module", autouse=True)
def _multi_mask_hook():
    """test_first_generation_shapes re-runs parity cases of this file in a child process with APX_TEST_MULTI_MASK set: every
    context that child creates gets the option APX_OPT_MULTI_ROWS_MASK (the library itself reads no environment)."""
    import os
    mask = os.environ.get("APX_TEST_MULTI_MASK")
    if mask is None:
        yield
        return
    import adaprox_amd as apx
    from importlib import import_module
    L = import_module("adaptive-proximal-algorithms_amd._lib")
    orig = apx.Context.__init__

    def patched(self, *a, **kw):
        orig(self, *a, **kw)
        self.set_option(L.APX_OPT_MULTI_ROWS_MASK, int(mask))

    apx.Context.__init__ = patched
    yield
    apx.Context.__init__ = orig


def _rel(a, b):
    return float(np.max(np.abs(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64)))) / \
        max(float(np.max(np.abs(b))), 1e-300)


def _L():
    from importlib import import_module
    return import_module("adaptive-proximal-algorithms_amd._lib")


def _fewer_rows(m, n):
    """The parity cases of rows of 40 000 entries and more run on ~1000 rows instead of ~4100 (the test hook APX_OPT_ONE_PASS_MIN_ROWS
    lets the one-pass kernels take such a matrix): the same kernels, the same ragged row counts, a quarter of the numpy oracle's work
    beside them — that oracle on a 6 GB matrix was the suite's time."""
    return m if n < 40000 else 1024 + (m - 4096)


def _opnorm_sq(A, iters=30):
    """||A||_2^2 by power iteration on A'A, from below (an SVD of a 4300 x 12 000 matrix is ten seconds of a GPU box's test time)."""
    v = np.ones(A.shape[1]) / np.sqrt(A.shape[1])
    lam = 0.0
    for _ in range(iters):
        w = A.T @ (A @ v)
        lam = float(np.linalg.norm(w))
        v = w / lam
    return lam


# (the path starts at 192 MB of A by default — its measured break-even ; the contexts below lower that to 16 MB)
@pytest.mark.parametrize("m,n,logistic,dtype", [(8500, 1000, False, np.float64),      # 1 column per thread, 32 rows per step
                                                (8600, 1024, True, np.float64),
                                                (8700, 600, False, np.float32),
                                                (9000, 900, True, np.float32),
                                                (8500, 1300, False, np.float64),      # 2 columns per thread, 768 threads
                                                (8600, 1536, True, np.float32),
                                                (8500, 2000, False, np.float64),      # 2 columns per thread, 1024 threads
                                                (4097, 2049, True, np.float64),       # 5 columns per thread, ragged rows
                                                (4200, 5120, False, np.float64),
                                                (4100, 3000, False, np.float64),      # 3 columns per thread
                                                (4200, 3072, True, np.float32),
                                                (4099, 3500, False, np.float32),      # 4 columns per thread
                                                (4101, 4096, True, np.float64),
                                                (4098, 10000, False, np.float64),     # 14 columns per thread, two rows per step
                                                (4101, 8000, True, np.float64),       # ... logistic fp64 (round 2: k_team_rows, P = 1)
                                                (9000, 1900, True, np.float32),
                                                (4099, 10239, True, np.float32),      # 10 columns per thread (fp32 only)
                                                (4100, 10240, False, np.float32),
                                                # long slabs (>= 16 steps per block): two row groups in flight
                                                (140000, 200, False, np.float32),
                                                (140000, 300, False, np.float64),     # k_multi_rows: three groups of three waves
                                                (150000, 200, True, np.float64),      # ... five groups of two waves, logistic
                                                (130000, 255, False, np.float64),
                                                (90000, 500, True, np.float64),       # ... two groups of five waves
                                                (90001, 640, False, np.float64),
                                                (100000, 130, False, np.float64),
                                                (20000, 2500, True, np.float32),
                                                (17000, 6000, False, np.float32),     # ... and > 384 MB: non-temporal loads
                                                # BASELINE config C3's own kernel: fp32 logistic, 14 columns per thread, four rows
                                                # per step in two groups, >= 384 MB (non-temporal loads) — 9800 x 10000 x 4 B = 392 MB
                                                (9800, 10000, True, np.float32),
                                                # the second-generation long-row kernel (k_slab2_rows: 16-byte buffer loads, KV
                                                # vectors per thread by the row length): Float32 KV = 2, 3 ; Float64 KV = 4 ... 7
                                                (6000, 5500, True, np.float32),
                                                (5000, 8000, True, np.float32),
                                                (4300, 9216, False, np.float32),
                                                (4200, 5500, False, np.float64),
                                                (4100, 7001, False, np.float64),
                                                (4100, 9000, False, np.float64),
                                                (4103, 10752, False, np.float64),
                                                # rows of more than 10 752 Float64 entries: TEAMS of 4 / 8 / 16 workgroups, a
                                                # slice of every row each, eleven compute waves + a communication wave
                                                # (k_team_rows) ; 100 000 is BASELINE config C2's row
                                                (4100, 11000, False, np.float64),      # one workgroup, KV = 8 (least squares)
                                                (4100, 12000, False, np.float64),      # teams of two, KV = 5
                                                (4099, 14080, True, np.float64),       # teams of two, KV = 5, logistic
                                                (4100, 15000, False, np.float64),      # teams of two, 6 vectors (four slots)
                                                (4101, 22528, True, np.float64),       # teams of two, 8 vectors, logistic
                                                (4100, 26000, False, np.float64),      # P = 4, KV = 5
                                                (4098, 40000, False, np.float64),      # P = 4, 8 vectors, one row of slack
                                                (4097, 130001, False, np.float64),     # P = 16, 6 vectors (four slots)
                                                (4099, 150000, False, np.float64),     # P = 16, 7 vectors, one row of slack
                                                (4100, 19000, True, np.float64),       # teams of two, 7 vectors, one row of slack
                                                (4100, 180224, True, np.float64),      # P = 16, 8 vectors: the longest row
                                                (4100, 20000, True, np.float64),       # P = 4, KV = 4, logistic
                                                (4097, 30001, False, np.float64),      # P = 4, 6 vectors (four slots)
                                                (4096, 56320, False, np.float64),      # P = 8, KV = 5
                                                (4099, 100000, False, np.float64),     # P = 16, KV = 5
                                                (4100, 70000, False, np.float64),      # P = 8, 7 vectors, one row of slack (round 4)
                                                (4098, 88001, True, np.float64),       # P = 8, 8 vectors, logistic
                                                # Float64 logistic rows that fit ONE workgroup: the same kernel with a "team"
                                                # of one — the communication wave evaluates the link, three rows in registers
                                                # (KV = 4 ... 8 vectors per compute thread: 5121 ... 11 264 columns)
                                                (4100, 5121, True, np.float64),
                                                (4099, 7000, True, np.float64),
                                                (4098, 8448, True, np.float64),
                                                (4097, 9857, True, np.float64),
                                                (4101, 10000, True, np.float64),       # C3's shape in the reference's precision
                                                (4100, 11264, True, np.float64),
                                                # Float32 on k_multi_rows (round 3): four columns per 16-byte vector, ragged row
                                                # lengths around the shape boundaries — five / three / two row groups, one vector
                                                # per thread, 10 or 11 compute waves, the longest row (4 x 704 vectors)
                                                (60000, 130, False, np.float32),
                                                (50000, 601, True, np.float32),
                                                (30000, 1279, False, np.float32),
                                                (4099, 2563, True, np.float32),
                                                (4098, 5000, True, np.float32),
                                                (4100, 5130, False, np.float32),
                                                (4097, 7677, True, np.float32),
                                                (4101, 11264, True, np.float32),
                                                (4100, 11263, False, np.float32),
                                                # ... and one row per step, five to eight vectors per thread: Float32 rows beyond the
                                                # first-generation kernels' 10 752 entries read A once too, up to 22 528
                                                (4100, 12000, False, np.float32),
                                                (4099, 16001, True, np.float32),
                                                (4100, 20000, False, np.float32),
                                                (4101, 22528, True, np.float32),
                                                # ... and beyond one workgroup: Float32 TEAMS of 2 / 4 / 8 / 16 (two to five vectors per
                                                # thread, four slots) — rows up to 225 280 entries read A once
                                                (4100, 22529, False, np.float32),      # teams of two, five vectors
                                                (4099, 28160, True, np.float32),       # teams of two, the longest row of two members
                                                (4100, 50001, False, np.float32),      # teams of four
                                                (4098, 90112, True, np.float32),       # teams of eight, four vectors
                                                (4097, 200000, False, np.float32)])    # teams of sixteen
def test_one_pass_matches_oracle_and_two_passes(apx, m, n, logistic, dtype):
    m = _fewer_rows(m, n)
    rng = np.random.default_rng(m + n)
    A = np.asfortranarray(rng.uniform(-1, 1, (m, n)).astype(dtype))
    nx = n + 1 if logistic else n
    if logistic:
        w = rng.standard_normal(n) * (rng.uniform(size=n) < 0.05)
        b = (1 / (1 + np.exp(-(A.astype(np.float64) @ w))) > rng.uniform(size=m)).astype(dtype)
        gam, lam = 4.0 * m / (np.linalg.norm(A.astype(np.float64)) ** 2 + m), 0.01
        mko = lambda: O.LogisticLoss(A.astype(np.float64), b.astype(np.float64))      # noqa: E731
    else:
        b = rng.standard_normal(m).astype(dtype)
        gam, lam = 1.0 / (m * n / 3.0), 1.0
        mko = lambda: O.LinearLeastSquares(A.astype(np.float64), b.astype(np.float64))      # noqa: E731
    K = 8
    x0 = (0.01 * rng.standard_normal(nx)).astype(dtype)
    ro = []
    # Float32 data: the oracle in its Float32-STORAGE mode (vectors rounded to Float32 where the product stores them, the logistic
    # link in Float32, sums in Float64 — oracle storage_rounding / emulate ): the arithmetic the kernels implement,
    # so the comparison is as tight as the Float64 one up to Float32's own epsilon (measured over these cases, 12 iterations:
    # iterates <= 1.2e-7, step sizes <= 1.3e-7 — tests/fp32_parity_probe.py, profiles/r05_fp32_parity.jsonl)
    emu = np.float32 if dtype == np.float32 else None
    fo = mko()
    fo.emulate = emu
    xo, _ = O.adaptive_proxgrad(x0.astype(np.float64), f=fo, g=O.NormL1(lam), rule=O.OurRule(gamma=gam), tol=0.0, maxit=K,
                                record=ro.append, storage=emu)
    out = []
    for one_pass in (True, False):
        ctx = apx.Context(0)
        ctx.fused_small(one_pass)
        ctx.one_pass_threshold(16 << 20)          # (default 192 MB: the measured break-even ; here: reach the kernel)
        ctx.set_option(_L().APX_OPT_ONE_PASS_MIN_ROWS, 1024)
        dA = apx.DeviceMatrix.from_numpy(A, ctx).build_transpose()
        f = apx.Counting(apx.LogisticLoss(dA, b, ctx=ctx) if logistic else apx.LinearLeastSquares(dA, b, ctx=ctx))
        recs = []
        x, it = apx.adaptive_proxgrad(x0, f=f, g=apx.NormL1(lam), rule=apx.OurRule(gamma=gam), tol=0.0, maxit=K, ctx=ctx,
                                      record=recs.append)
        assert it == K and (f.eval_count, f.grad_count) == (K + 1, K + 1)
        out.append((x, recs))
        dA.free()
    (x1, r1), (x2, r2) = out
    f64 = dtype == np.float64
    assert _rel(x1, xo) <= (1e-10 if f64 else 1e-6)
    assert _rel(x1, x2) <= (1e-10 if f64 else 5e-6)                # (the two GEMV passes round their Float32 partial sums elsewhere)
    for a, b_, c in zip(r1, ro, r2):
        assert a["objective"] == pytest.approx(b_["objective"], rel=1e-10 if f64 else 2e-6)
        assert a["gamma"] == pytest.approx(b_["gamma"], rel=1e-7 if f64 else 1e-6)
        assert a["gamma"] == pytest.approx(c["gamma"], rel=1e-7 if f64 else 1e-5)


def test_one_pass_builds_its_transposed_copy_and_survives_losing_it(apx):
    """A matrix without the transposed copy is given one when a solver is created on it; editing the matrix in place
    drops the copy, and a solver that is alive then falls back to the two passes."""
    rng = np.random.default_rng(5)
    m, n = 9000, 1000
    A = np.asfortranarray(rng.uniform(-1, 1, (m, n)))
    b = rng.standard_normal(m)
    gam = 1.0 / (m * n / 3.0)
    xo, _ = O.adaptive_proxgrad(np.zeros(n), f=O.LinearLeastSquares(A, b), g=O.NormL1(1.0), rule=O.OurRule(gamma=gam), tol=0.0,
                                maxit=12)
    ctx = apx.Context(0)
    ctx.one_pass_threshold(16 << 20)
    dA = apx.DeviceMatrix.from_numpy(A, ctx)                       # no build_transpose()
    f = apx.LinearLeastSquares(dA, b, ctx=ctx)
    x, _ = apx.adaptive_proxgrad(np.zeros(n), f=f, g=apx.NormL1(1.0), rule=apx.OurRule(gamma=gam), tol=0.0, maxit=12, ctx=ctx)
    assert _rel(x, xo) <= 1e-10
    s = apx.Solver(np.zeros(n), f=f, g=apx.NormL1(1.0), rule=apx.OurRule(gamma=gam), tol=0.0, ctx=ctx)
    s.step(5)
    dA.scale_columns(np.ones(n))                                   # an in-place edit (by 1.0): the copy is gone
    s.step(7)
    xs, _, res = s.result()
    s.close()
    assert res.numit == 12 and _rel(xs, xo) <= 1e-10
    dA.free()


def test_one_pass_every_rule_termination_and_stepping(apx):
    rng = np.random.default_rng(3)
    m, n = 9000, 1200
    A = np.asfortranarray(rng.uniform(-1, 1, (m, n)))
    b = rng.standard_normal(m)
    gam = 1.0 / (m * n / 3.0)
    ctx = apx.Context(0)
    ctx.one_pass_threshold(16 << 20)
    dA = apx.DeviceMatrix.from_numpy(A, ctx).build_transpose()
    f, g = apx.LinearLeastSquares(dA, b, ctx=ctx), apx.NormL1(1.0)
    fo, go = O.LinearLeastSquares(A, b), O.NormL1(1.0)
    for mk in (lambda M: M.FixedStepsize(gam, 1.0), lambda M: M.MalitskyMishchenkoRule(gamma=gam), lambda M: M.OurRulePlus(gamma=gam)):
        xo, _ = O.adaptive_proxgrad(np.zeros(n), f=fo, g=go, rule=mk(O), tol=0.0, maxit=15)
        x, _ = apx.adaptive_proxgrad(np.zeros(n), f=f, g=g, rule=mk(apx), tol=0.0, maxit=15, ctx=ctx)
        assert _rel(x, xo) <= 1e-10
    rule = apx.OurRule(gamma=gam)
    x1, _ = apx.adaptive_proxgrad(np.zeros(n), f=f, g=g, rule=rule, tol=0.0, maxit=30, ctx=ctx)
    x2, _ = apx.adaptive_proxgrad(np.zeros(n), f=f, g=g, rule=rule, tol=0.0, maxit=30, ctx=ctx, use_graph=True)
    assert np.array_equal(x1, x2)
    s = apx.Solver(np.zeros(n), f=f, g=g, rule=rule, tol=0.0, ctx=ctx)
    for k in (1, 9, 20):
        s.step(k)
        s.result()
    xs, _, res = s.result()
    s.close()
    assert res.numit == 30 and np.array_equal(xs, x1)
    recs = []
    apx.adaptive_proxgrad(np.zeros(n), f=f, g=g, rule=rule, tol=0.0, maxit=30, ctx=ctx, record=recs.append)
    tol = recs[19]["norm_res"] * (1 + 1e-9)
    kstar = 1 + next(k for k, r in enumerate(recs) if r["norm_res"] <= tol)
    xo, ito = O.adaptive_proxgrad(np.zeros(n), f=fo, g=go, rule=O.OurRule(gamma=gam), tol=tol, maxit=1000)
    for poll in (1, 64):
        x, it = apx.adaptive_proxgrad(np.zeros(n), f=f, g=g, rule=rule, tol=tol, maxit=1000, ctx=ctx, poll_every=poll)
        assert it == kstar == ito and _rel(x, xo) <= 1e-10
    dA.free()


# ---------------------------------------------------------------------------------------
# the primal-dual loop: A x, the dual step and A' y in one pass over A (csrc/slab_pd.hip)
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("m,n,hkind,dtype", [(9000, 1000, "tl1", np.float64),      # 1 column per thread
                                             (9000, 700, "tbox", np.float32),
                                             (100000, 250, "tl1", np.float64),     # k_multi_rows: five groups of two waves
                                             (90000, 380, "tbox", np.float64),     # ... three groups
                                             (80000, 600, "tl1", np.float64),      # ... two groups
                                             (9100, 1300, "tl1", np.float64),      # 2 columns per thread, 768 threads
                                             (9000, 1800, "tl1", np.float64),      # 2 columns per thread, 1024 threads
                                             (4099, 2100, "tbox", np.float64),     # 3
                                             (4100, 4000, "tl1", np.float32),      # 4
                                             (4098, 5000, "tl1", np.float64),      # 5
                                             (4100, 5300, "tl1", np.float64),      # 7 (768 threads)
                                             (4100, 6000, "tl1", np.float64),      # k_team_rows, one workgroup, 5 vectors
                                             (4099, 9000, "tbox", np.float64),     # ... 7 vectors
                                             (4099, 7000, "tbox", np.float32),     # 10
                                             (4101, 7680, "tl1", np.float64),
                                             (4097, 9000, "tl1", np.float64),      # 14, two rows per step
                                             (4100, 10752, "tl1", np.float32),     # 14, four rows per step
                                             (9001, 1900, "tbox", np.float32),
                                             (140000, 200, "tl1", np.float32),     # long slabs: two row groups in flight
                                             (17000, 6000, "tl1", np.float32),     # > 384 MB: non-temporal loads
                                             # rows of more than 10 752 Float64 entries: k_team_rows with the dual step in its
                                             # communication wave (one workgroup to 11 264 columns, then teams) ; 20 000 is
                                             # BASELINE config C5's row (teams of two, eight vectors, one row of slack)
                                             (4100, 11000, "tl1", np.float64),
                                             (4200, 15001, "tbox", np.float64),
                                             (4099, 18000, "tl1", np.float64),
                                             (4100, 20000, "tl1", np.float64),
                                             (4097, 21504, "tl1", np.float64),
                                             # ... and beyond, since the loop shares k_team_rows with f's gradient: teams of 4, 8, 16
                                             (4100, 30000, "tl1", np.float64),
                                             (4099, 60001, "tbox", np.float64),
                                             (4098, 150000, "tl1", np.float64),
                                             (4100, 12000, "tl1", np.float64),     # teams of two, five vectors, three slots (round 4)
                                             (4099, 80000, "tbox", np.float64),    # teams of eight, eight vectors, three slots (round 4)
                                             # Float32 teams in the primal-dual form (round 5: rows of 11 265 ... 225 280 Float32 entries)
                                             (4100, 11265, "tl1", np.float32),     # teams of two, two vectors
                                             (4099, 28160, "tbox", np.float32),    # teams of two, five vectors
                                             (4098, 50001, "tl2", np.float32),     # teams of four, the l2 ball
                                             (4097, 90112, "tl1", np.float32),     # teams of eight
                                             (4100, 200000, "tl1", np.float32),    # teams of sixteen
                                             # Float32 on k_multi_rows (round 3)
                                             (70000, 300, "tl1", np.float32),
                                             (4099, 5001, "tbox", np.float32),
                                             (4100, 11264, "tl1", np.float32),
                                             # h = the square-root lasso's l2 ball (round 4): not separable, but linear in the rows once
                                             # the norm is known — A' z and sum z^2 in one pass, scaled by the closing kernel, y+ and the
                                             # dual residual by k_dual_post (csrc/team_rows.h team_pd_dual, csrc/slab_pd.hip)
                                             (9000, 1000, "tl2", np.float64),      # k_multi_rows
                                             (80000, 600, "tl2", np.float64),      # ... two groups of compute waves
                                             (4100, 6000, "tl2", np.float64),      # k_team_rows, one workgroup
                                             (4100, 20000, "tl2", np.float64),     # ... teams of two (C5's row)
                                             (4099, 30001, "tl2", np.float64),     # ... teams of four
                                             (4098, 100000, "tl2", np.float64),    # ... teams of sixteen (C2's row)
                                             (4099, 5001, "tl2", np.float32)])     # k_multi_rows, Float32
def test_one_pass_primal_dual_matches_oracle_and_two_passes(apx, m, n, hkind, dtype):
    m = _fewer_rows(m, n)
    rng = np.random.default_rng(m + n)
    A = np.asfortranarray(np.hstack([rng.uniform(-1, 1, (m, n - 1)), np.ones((m, 1))]).astype(dtype))
    yv = (A.astype(np.float64) @ (rng.standard_normal(n) / np.sqrt(n)) + rng.laplace(size=m)).astype(dtype)
    nA = float(np.linalg.norm(A.astype(np.float64)))
    mk_h = {"tl1": lambda M: M.Translate(M.NormL1(1.0), -yv.astype(np.float64) if M is O else -yv),
            "tbox": lambda M: M.Translate(M.IndBox(-0.5, 0.8), -yv.astype(np.float64) if M is O else -yv),
            "tl2": lambda M: M.Translate(M.NormL2(1.0), -yv.astype(np.float64) if M is O else -yv)}[hkind]
    K = 10 if hkind == "tl2" else 8          # (the l2 ball's x leaves 0 only around iteration 8)
    lam = 0.05 if hkind == "tl2" else 5.0      # (the l2 ball: |A' y| <= 2 ||y|| or so, and y grows from 0 in steps of sigma b — a larger lambda keeps x at 0 for these ten iterations)
    x0, y0 = np.zeros(n, dtype=dtype), np.zeros(m, dtype=dtype)
    ro = []
    xo, yo, _ = O.adaptive_primal_dual(x0.astype(np.float64), y0.astype(np.float64), f=O.Zero(), g=O.NormL1(lam), h=mk_h(O),
                                       A=A.astype(np.float64), rule=O.OurRule(t=1.0, norm_A=nA), tol=0.0, maxit=K, record=ro.append)
    out = []
    for one_pass in (True, False):
        ctx = apx.Context(0)
        ctx.fused_small(one_pass)
        ctx.one_pass_threshold(16 << 20)          # (default 192 MB: the measured break-even ; here: reach the kernel)
        ctx.set_option(_L().APX_OPT_ONE_PASS_MIN_ROWS, 1024)
        dA = apx.DeviceMatrix.from_numpy(A, ctx)
        recs = []
        ctx.kernel_stats(reset=True)
        ctx.profile(True)
        x, y, it = apx.adaptive_primal_dual(x0, y0, f=apx.Zero(), g=apx.NormL1(lam), h=mk_h(apx), A=dA,
                                            rule=apx.OurRule(t=1.0, norm_A=nA), tol=0.0, maxit=K, ctx=ctx, record=recs.append)
        st = ctx.kernel_stats(reset=True)
        ctx.profile(False)
        assert it == K and len(recs) == K
        # A x and A' y of the K iterations from K passes over A (the loop's two products in front of it: two GEMVs), or from 2 K + 2
        assert st["one_pass"]["launches"] == (K if one_pass else 0) and (st["gemv_t"]["launches"] <= 2 if one_pass else st["gemv_t"]["launches"] >= K), st
        out.append((x, y, recs))
        dA.free()
    (x1, y1, r1), (x2, y2, r2) = out
    tol = 1e-10 if dtype == np.float64 else 2e-3
    assert (hkind != "tl2" or np.count_nonzero(xo) > 0) and ro[-1]["norm_res"] > 1e-6      # (a loop that is still moving)
    assert _rel(x1, xo) <= tol and _rel(y1, yo) <= 10 * tol and _rel(x1, x2) <= tol and _rel(y1, y2) <= 10 * tol
    for a, b_ in zip(r1, ro):
        assert a["norm_res"] == pytest.approx(b_["norm_res"], rel=1e-7 if dtype == np.float64 else 1e-2)
        assert a["objective"] == pytest.approx(b_["objective"], rel=1e-9 if dtype == np.float64 else 1e-3)


def test_one_pass_primal_dual_condat_vu_termination_and_stepping(apx):
    rng = np.random.default_rng(11)
    m, n = 9000, 1000
    A = np.asfortranarray(np.hstack([rng.uniform(-1, 1, (m, n - 1)), np.ones((m, 1))]))
    yv = A @ (rng.standard_normal(n) / np.sqrt(n)) + rng.laplace(size=m)
    nA = float(np.linalg.norm(A))
    ctx = apx.Context(0)
    ctx.one_pass_threshold(16 << 20)
    dA = apx.DeviceMatrix.from_numpy(A, ctx)
    t = dict(f=apx.Zero(), g=apx.NormL1(5.0), h=apx.Translate(apx.NormL1(1.0), -yv), A=dA, ctx=ctx)
    to = dict(f=O.Zero(), g=O.NormL1(5.0), h=O.Translate(O.NormL1(1.0), -yv), A=A)
    x0, y0 = np.zeros(n), np.zeros(m)
    recs = []
    xo, yo, _ = O.condat_vu(x0, y0, Lf=0.0, norm_A=nA, tol=0.0, maxit=30, **to)
    x, y, it = apx.condat_vu(x0, y0, Lf=0.0, norm_A=nA, tol=0.0, maxit=30, record=recs.append, **t)
    assert it == 30 and _rel(x, xo) <= 1e-10 and _rel(y, yo) <= 1e-9
    tol = recs[19]["norm_res"] * (1 + 1e-9)
    kstar = 1 + next(k for k, r in enumerate(recs) if r["norm_res"] <= tol)
    xo, yo, ito = O.condat_vu(x0, y0, Lf=0.0, norm_A=nA, tol=tol, maxit=1000, **to)
    for poll in (1, 64):
        x, y, it = apx.condat_vu(x0, y0, Lf=0.0, norm_A=nA, tol=tol, maxit=1000, poll_every=poll, **t)
        assert it == kstar == ito and _rel(x, xo) <= 1e-10 and _rel(y, yo) <= 1e-9
    rule = apx.OurRule(t=1.0, norm_A=nA)
    x1, y1, _ = apx.adaptive_primal_dual(x0, y0, rule=rule, tol=0.0, maxit=20, **t)
    s = apx.Solver(x0, y0, rule=rule, tol=0.0, **t)
    for k in (1, 9, 10):
        s.step(k)
        s.result()
    xs, ys, res = s.result()
    s.close()
    assert res.numit == 20 and np.array_equal(xs, x1) and np.array_equal(ys, y1)
    dA.free()


def test_one_pass_serves_every_loop_and_the_operator_calls(apx):
    """eval_f uses the one-pass kernel for any evaluation of an eligible f: the linesearch and momentum methods (host-driven
    at this size) and eval_with_pullback get f and the gradient from the same pass; counts stay those of the oracle."""
    rng = np.random.default_rng(21)
    m, n = 9000, 1100
    A = np.asfortranarray(rng.uniform(-1, 1, (m, n)))
    w = rng.standard_normal(n) * (rng.uniform(size=n) < 0.05)
    b = (1 / (1 + np.exp(-(A @ w))) > rng.uniform(size=m)).astype(np.float64)
    gam = 4.0 * m / (float(np.sum(A * A)) + m)
    ctx = apx.Context(0)
    ctx.one_pass_threshold(16 << 20)
    dA = apx.DeviceMatrix.from_numpy(A, ctx)
    mk = lambda M: (M.Counting(M.LogisticLoss(dA, b, ctx=ctx)) if M is apx else M.Counting(M.LogisticLoss(A, b)))      # noqa: E731
    x0 = 0.01 * rng.standard_normal(n + 1)
    fx_o, pb_o = O.LogisticLoss(A, b).eval_with_pullback(x0)
    fx, pb = apx.eval_with_pullback(apx.LogisticLoss(dA, b, ctx=ctx), x0)
    assert fx == pytest.approx(fx_o, rel=1e-12) and _rel(pb(), pb_o()) <= 1e-11
    K = 8
    for name, kw in (("backtracking_proxgrad", dict(gamma0=4 * gam, xi=1.5)), ("backtracking_nesterov", dict(gamma0=4 * gam)),
                     ("fixed_nesterov", dict(gamma=gam)), ("agraal", dict(x0=x0 + 0.01, gamma0=gam))):
        of, df = mk(O), mk(apx)
        zo, ito = getattr(O, name)(x0, f=of, g=O.NormL1(0.01), tol=0.0, maxit=K, **kw)
        z, it = getattr(apx, name)(x0, f=df, g=apx.NormL1(0.01), tol=0.0, maxit=K, ctx=ctx, **kw)
        assert it == ito == K and _rel(z, zo) <= 1e-10, name
        assert (df.eval_count, df.grad_count) == (of.eval_count, of.grad_count), name
    dA.free()


def test_linesearch_trials_rejected_on_the_one_pass_path(apx):
    """backtracking_proxgrad / backtracking_nesterov where EVERY iteration rejects trials (gamma0 = 1000 / L): each trial
    evaluates f at z, and on this path that evaluation also forms grad f(z); the next shrink trial must still be built
    from grad f(x) (src/AdaProx.jl:43-45).  Iterates, step sizes and evaluation counts against the oracle."""
    rng = np.random.default_rng(33)
    m, n = 9000, 1100
    A = np.asfortranarray(rng.uniform(-1, 1, (m, n)))
    b = rng.standard_normal(m)
    L = 1.02 * _opnorm_sq(A)
    K = 10
    x0 = 0.01 * rng.standard_normal(n)
    for one_pass in (True, False):
        ctx = apx.Context(0)
        ctx.fused_small(one_pass)
        ctx.one_pass_threshold(16 << 20)
        dA = apx.DeviceMatrix.from_numpy(A, ctx).build_transpose()
        for name, kw in (("backtracking_proxgrad", dict(gamma0=1e3 / L, xi=2.0)), ("backtracking_proxgrad", dict(gamma0=1e3 / L, xi=1.0)),
                         ("backtracking_nesterov", dict(gamma0=1e3 / L))):
            of, df = O.Counting(O.LinearLeastSquares(A, b)), apx.Counting(apx.LinearLeastSquares(dA, b, ctx=ctx))
            ro, rd = [], []
            zo, ito = getattr(O, name)(x0, f=of, g=O.NormL1(1.0), tol=0.0, maxit=K, record=ro.append, **kw)
            z, it = getattr(apx, name)(x0, f=df, g=apx.NormL1(1.0), tol=0.0, maxit=K, ctx=ctx, record=rd.append, **kw)
            assert it == ito == K
            assert of.eval_count > K + 1 + (K if name.endswith("nesterov") else 0), "no trial was rejected: the test tests nothing"
            assert (df.eval_count, df.grad_count) == (of.eval_count, of.grad_count), (name, kw, one_pass)
            assert _rel(z, zo) <= 1e-10, (name, kw, one_pass)
            assert [r["gamma"] for r in rd] == pytest.approx([r["gamma"] for r in ro], rel=1e-12)
            assert [r["objective"] for r in rd] == pytest.approx([r["objective"] for r in ro], rel=1e-10)
        dA.free()


def test_transposed_copy_is_opt_out(apx):
    """apx_ctx_auto_transpose(0): the library never allocates the second copy by itself; the one-pass kernels
    then run only where the caller built it, and apx_matrix_drop_transpose gives the memory back."""
    rng = np.random.default_rng(8)
    m, n = 9000, 1000
    A = np.asfortranarray(rng.uniform(-1, 1, (m, n)))
    b = rng.standard_normal(m)
    gam = 1.0 / (m * n / 3.0)
    xo, _ = O.adaptive_proxgrad(np.zeros(n), f=O.LinearLeastSquares(A, b), g=O.NormL1(1.0), rule=O.OurRule(gamma=gam), tol=0.0, maxit=10)
    ctx = apx.Context(0)
    ctx.one_pass_threshold(16 << 20)
    ctx.auto_transpose(False)
    dA = apx.DeviceMatrix.from_numpy(A, ctx)
    kw = dict(g=apx.NormL1(1.0), rule=apx.OurRule(gamma=gam), tol=0.0, maxit=10, ctx=ctx)
    ctx.kernel_stats(reset=True)
    ctx.profile(True)
    x, _ = apx.adaptive_proxgrad(np.zeros(n), f=apx.LinearLeastSquares(dA, b, ctx=ctx), **kw)
    st = ctx.kernel_stats(reset=True)
    assert st["gemv_t"]["launches"] == 11 and _rel(x, xo) <= 1e-10          # two passes: no copy was made
    dA.build_transpose()
    x, _ = apx.adaptive_proxgrad(np.zeros(n), f=apx.LinearLeastSquares(dA, b, ctx=ctx), **kw)
    st = ctx.kernel_stats(reset=True)
    assert st["gemv_t"]["launches"] == 0 and st["gemv_n"]["launches"] == 0 and st["one_pass"]["launches"] == 11 and _rel(x, xo) <= 1e-10      # one pass
    dA.drop_transpose()
    x, _ = apx.adaptive_proxgrad(np.zeros(n), f=apx.LinearLeastSquares(dA, b, ctx=ctx), **kw)
    st = ctx.kernel_stats(reset=True)
    assert st["gemv_t"]["launches"] == 11 and _rel(x, xo) <= 1e-10
    ctx.profile(False)
    dA.free()


def test_team_kernel_is_reproducible_steps_and_terminates(apx):
    """The two-workgroup teams (rows of C5's length): which physical workgroup becomes which member of which team is decided
    at run time (a ticket per XCD) — the results must not depend on it: two runs bit for bit, stepping in pieces = one
    call, termination at the iteration the oracle stops at, and the kernel really ran (one pass over A per iteration)."""
    rng = np.random.default_rng(12)
    m, n = 4500, 20000
    A = np.asfortranarray(np.hstack([rng.uniform(-1, 1, (m, n - 1)), np.ones((m, 1))]))
    yv = A @ (rng.standard_normal(n) / np.sqrt(n)) + rng.laplace(size=m)
    nA = float(np.linalg.norm(A))
    ctx = apx.Context(0)
    ctx.one_pass_threshold(16 << 20)
    dA = apx.DeviceMatrix.from_numpy(A, ctx)
    t = dict(f=apx.Zero(), g=apx.NormL1(5.0), h=apx.Translate(apx.NormL1(1.0), -yv), A=dA, ctx=ctx)
    to = dict(f=O.Zero(), g=O.NormL1(5.0), h=O.Translate(O.NormL1(1.0), -yv), A=A)
    x0, y0 = np.zeros(n), np.zeros(m)
    rule = apx.OurRule(t=1.0, norm_A=nA)
    ctx.kernel_stats(reset=True)
    ctx.profile(True)
    recs = []
    x1, y1, _ = apx.adaptive_primal_dual(x0, y0, rule=rule, tol=0.0, maxit=24, record=recs.append, **t)
    st = ctx.kernel_stats(reset=True)
    ctx.profile(False)
    assert st["gemv_t"]["launches"] == 1 and st["gemv_n"]["launches"] == 1 and st["one_pass"]["launches"] == 24       # the pre-loop products, then one pass each
    x2, y2, _ = apx.adaptive_primal_dual(x0, y0, rule=rule, tol=0.0, maxit=24, **t)
    assert np.array_equal(x1, x2) and np.array_equal(y1, y2)
    s = apx.Solver(x0, y0, rule=rule, tol=0.0, **t)
    for k in (1, 7, 16):
        s.step(k)
        s.result()
    xs, ys, res = s.result()
    s.close()
    assert res.numit == 24 and np.array_equal(xs, x1) and np.array_equal(ys, y1)
    xo, yo, _ = O.adaptive_primal_dual(x0, y0, rule=O.OurRule(t=1.0, norm_A=nA), tol=0.0, maxit=24, **to)
    assert _rel(x1, xo) <= 1e-10 and _rel(y1, yo) <= 1e-9
    tol = recs[15]["norm_res"] * (1 + 1e-9)
    kstar = 1 + next(k for k, r in enumerate(recs) if r["norm_res"] <= tol)
    xo, yo, ito = O.adaptive_primal_dual(x0, y0, rule=O.OurRule(t=1.0, norm_A=nA), tol=tol, maxit=1000, **to)
    for poll in (1, 64):
        x, y, it = apx.adaptive_primal_dual(x0, y0, rule=rule, tol=tol, maxit=1000, poll_every=poll, **t)
        assert it == kstar == ito and _rel(x, xo) <= 1e-10 and _rel(y, yo) <= 1e-9
    dA.free()


def test_team_kernel_proximal_gradient_stepping_termination_replay(apx):
    """k_team_rows (f and its gradient for rows beyond one workgroup; here teams of four): two runs bit for bit whatever
    workgroup became which member, stepping = one call, termination at the oracle's iteration with two polling intervals,
    the linesearch loops on it, and hipGraph replay refused (its launches carry host-side counters)."""
    rng = np.random.default_rng(41)
    m, n = 4300, 12000
    A = np.asfortranarray(rng.uniform(-1, 1, (m, n)))
    b = rng.standard_normal(m)
    L = 1.02 * _opnorm_sq(A)
    gam = 1.0 / L
    ctx = apx.Context(0)
    ctx.one_pass_threshold(16 << 20)
    dA = apx.DeviceMatrix.from_numpy(A, ctx).build_transpose()
    f, g = apx.LinearLeastSquares(dA, b, ctx=ctx), apx.NormL1(1.0)
    fo, go = O.LinearLeastSquares(A, b), O.NormL1(1.0)
    rule = apx.OurRule(gamma=gam)
    recs = []
    x1, _ = apx.adaptive_proxgrad(np.zeros(n), f=f, g=g, rule=rule, tol=0.0, maxit=30, ctx=ctx, record=recs.append)
    x2, _ = apx.adaptive_proxgrad(np.zeros(n), f=f, g=g, rule=rule, tol=0.0, maxit=30, ctx=ctx)
    assert np.array_equal(x1, x2)
    xo, _ = O.adaptive_proxgrad(np.zeros(n), f=fo, g=go, rule=O.OurRule(gamma=gam), tol=0.0, maxit=30)
    assert _rel(x1, xo) <= 1e-10
    s = apx.Solver(np.zeros(n), f=f, g=g, rule=rule, tol=0.0, ctx=ctx)
    for k in (1, 9, 20):
        s.step(k)
        s.result()
    xs, _, res = s.result()
    s.close()
    assert res.numit == 30 and np.array_equal(xs, x1)
    tol = recs[19]["norm_res"] * (1 + 1e-9)
    kstar = 1 + next(k for k, r in enumerate(recs) if r["norm_res"] <= tol)
    xo, ito = O.adaptive_proxgrad(np.zeros(n), f=fo, g=go, rule=O.OurRule(gamma=gam), tol=tol, maxit=1000)
    for poll in (1, 64):
        x, it = apx.adaptive_proxgrad(np.zeros(n), f=f, g=g, rule=rule, tol=tol, maxit=1000, ctx=ctx, poll_every=poll)
        assert it == kstar == ito and _rel(x, xo) <= 1e-10
    # the host-driven linesearch (every iteration rejects trials: gamma0 = 1000 / L, xi = 2) on the team kernel
    of, df = O.Counting(fo), apx.Counting(f)
    zo, _ = O.backtracking_proxgrad(np.zeros(n), f=of, g=go, gamma0=1e3 / L, xi=2.0, tol=0.0, maxit=8)
    z, _ = apx.backtracking_proxgrad(np.zeros(n), f=df, g=g, gamma0=1e3 / L, xi=2.0, tol=0.0, maxit=8, ctx=ctx)
    assert (df.eval_count, df.grad_count) == (of.eval_count, of.grad_count) and of.eval_count > 9
    assert _rel(z, zo) <= 1e-10
    with pytest.raises(apx.ApxError, match="hipGraph replay"):
        apx.adaptive_proxgrad(np.zeros(n), f=f, g=g, rule=rule, tol=0.0, maxit=5, ctx=ctx, use_graph=True)
    dA.free()


def test_first_generation_shapes():
    """The Float64 rows of up to 5632 entries run on k_multi_rows by default ; the first-generation shapes they replaced (one
    element per lane and load, 1 ... 5 columns per thread, the link next to the rows) remain behind the option APX_OPT_MULTI_ROWS_MASK = 0
    — the A/B switch — and are what Float32 runs: the same parity cases once more with the switch off, in a process of its own
    whose contexts are all created with it (APX_TEST_MULTI_MASK: read by THIS test file, not by the library — _multi_mask_hook)."""
    import os
    import subprocess
    import sys

    env = dict(os.environ, APX_TEST_MULTI_MASK="0")
    ids = "8500-1000-False-float64 or 8600-1024-True-float64 or 8500-1300-False-float64 or 4097-2049-True-float64 or " \
          "4100-3000-False-float64 or 4101-4096-True-float64 or 4200-5120-False-float64 or 9000-1000-tl1-float64 or " \
          "9100-1300-tl1-float64 or 4099-2100-tbox-float64 or 4098-5000-tl1-float64 or 8700-600-False-float32 or " \
          "9000-900-True-float32 or 8600-1536-True-float32 or 4200-3072-True-float32 or 4099-3500-False-float32 or " \
          "9800-10000-True-float32 or 4099-10239-True-float32 or 9000-700-tbox-float32 or 4100-4000-tl1-float32 or 4100-10752-tl1-float32"
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-m", "gpu", "-x", "-q", "-k", ids],
                         env=env, capture_output=True, text=True, timeout=900, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-1000:]
    assert " passed" in out.stdout and "21 passed" in out.stdout, out.stdout[-500:]


@pytest.mark.parametrize("m,n,logistic,dtype", [(9000, 1000, False, np.float64),      # k_multi_rows
                                                (8600, 1024, True, np.float32),       # first-generation shapes, the bias entry in a block of its own (1025 entries)
                                                (8300, 10000, True, np.float32),      # C3's kernel
                                                (4200, 9000, False, np.float64),      # k_slab2_rows
                                                (4400, 7000, True, np.float64),       # k_team_rows, one workgroup
                                                (4300, 30000, False, np.float64)])    # k_team_rows, teams of four
def test_fused_tail_is_bit_identical(apx, m, n, logistic, dtype):
    """The residual / step-size reductions and the rule ride in the closing kernel of the one-pass launch (csrc/slab.hip SlabTail:
    two kernels between consecutive passes over A instead of four).  They are summed in exactly the order of the separate
    kernels (k_reduce_n + k_rule), so switching the fusion off (APX_OPT_FUSED_TAIL = 0) must not change one bit: iterates,
    records, termination."""
    from importlib import import_module
    L = import_module("adaptive-proximal-algorithms_amd._lib")
    rng = np.random.default_rng(31)
    A = np.asfortranarray(rng.uniform(-1, 1, (m, n)).astype(dtype))
    b = (rng.uniform(size=m) < 0.5).astype(dtype) if logistic else rng.standard_normal(m).astype(dtype)
    nx = n + 1 if logistic else n
    gam = 4.0 / (n / 3.0 + 1.0) if logistic else 1.0 / (m * n / 3.0)
    x0 = (0.01 * rng.standard_normal(nx)).astype(dtype)
    runs = []
    for fused in (1, 0):
        ctx = apx.Context(0)
        ctx.one_pass_threshold(16 << 20)
        ctx.set_option(L.APX_OPT_FUSED_TAIL, fused)
        dA = apx.DeviceMatrix.from_numpy(A, ctx).build_transpose()
        f = apx.LogisticLoss(dA, b, ctx=ctx) if logistic else apx.LinearLeastSquares(dA, b, ctx=ctx)
        recs = []
        ctx.kernel_stats(reset=True)
        ctx.profile(True)
        x, it = apx.adaptive_proxgrad(x0, f=f, g=apx.NormL1(1e-3 if logistic else 1.0), rule=apx.OurRule(gamma=gam), tol=0.0, maxit=12,
                                      ctx=ctx, record=recs.append)
        st = ctx.kernel_stats(reset=True)
        ctx.profile(False)
        assert st["one_pass"]["launches"] == 13 and st["gemv_t"]["launches"] == 0, st
        runs.append((x, [(r["gamma"], r["norm_res"], r["objective"]) for r in recs]))
        dA.free()
    assert np.array_equal(runs[0][0], runs[1][0]) and runs[0][1] == runs[1][1]
    # ... and the termination test runs in the closing kernel too: the same iteration, the same iterate
    tol = runs[0][1][7][1] * 1.0000001
    outs = []
    for fused in (1, 0):
        ctx = apx.Context(0)
        ctx.one_pass_threshold(16 << 20)
        ctx.set_option(L.APX_OPT_FUSED_TAIL, fused)
        dA = apx.DeviceMatrix.from_numpy(A, ctx).build_transpose()
        f = apx.LogisticLoss(dA, b, ctx=ctx) if logistic else apx.LinearLeastSquares(dA, b, ctx=ctx)
        outs.append(apx.adaptive_proxgrad(x0, f=f, g=apx.NormL1(1e-3 if logistic else 1.0), rule=apx.OurRule(gamma=gam), tol=tol, maxit=50,
                                          ctx=ctx, poll_every=3))
        dA.free()
    assert outs[0][1] == outs[1][1] <= 8 and np.array_equal(outs[0][0], outs[1][0])


@pytest.mark.parametrize("m,n,one_pass", [(9000, 1000, True), (4100, 6000, True), (4100, 20000, True), (6000, 3000, False)])
def test_primal_dual_step_leaves_the_next_reductions_bit_identical(apx, m, n, one_pass):
    """f = Zero with a linear operator: the step kernel also leaves k_reduce_n's block partials of the NEXT iteration
    (csrc/vecops.hip k_step_red — one launch fewer per iteration of the least-absolute-deviation loops), summed in k_reduce_n's
    own order: switching it off (APX_OPT_FUSED_TAIL = 0) must not change one bit — iterates, records, termination, stepping in
    pieces — on the one-pass kernels and on the two GEMV passes."""
    from importlib import import_module
    L = import_module("adaptive-proximal-algorithms_amd._lib")
    rng = np.random.default_rng(m + n)
    A = np.asfortranarray(np.hstack([rng.uniform(-1, 1, (m, n - 1)), np.ones((m, 1))]))
    yv = A @ (rng.standard_normal(n) / np.sqrt(n)) + rng.laplace(size=m)
    nA = float(np.linalg.norm(A))
    x0, y0 = np.zeros(n), np.zeros(m)
    runs = []
    for fused in (1, 0):
        ctx = apx.Context(0)
        ctx.fused_small(one_pass)
        ctx.one_pass_threshold(16 << 20)
        ctx.set_option(L.APX_OPT_FUSED_TAIL, fused)
        dA = apx.DeviceMatrix.from_numpy(A, ctx)
        t = dict(f=apx.Zero(), g=apx.NormL1(5.0), h=apx.Translate(apx.NormL1(1.0), -yv), A=dA, ctx=ctx)
        recs = []
        x, y, it = apx.adaptive_primal_dual(x0, y0, rule=apx.OurRule(t=1.0, norm_A=nA), tol=0.0, maxit=14, record=recs.append, **t)
        s = apx.Solver(x0, y0, rule=apx.OurRule(t=1.0, norm_A=nA), tol=0.0, **t)
        for k in (1, 5, 8):
            s.step(k)
            s.result()
        xs, ys, _ = s.result()
        s.close()
        tol = recs[9]["norm_res"] * 1.0000001
        xt, yt, itt = apx.adaptive_primal_dual(x0, y0, rule=apx.OurRule(t=1.0, norm_A=nA), tol=tol, maxit=60, poll_every=3, **t)
        runs.append((x, y, [(r["gamma"], r["sigma"], r["norm_res"], r["objective"]) for r in recs], xs, ys, xt, yt, itt))
        dA.free()
    a, b = runs
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2]
    assert np.array_equal(a[3], a[0]) and np.array_equal(b[3], b[0]) and np.array_equal(a[4], b[4])      # stepping in pieces = one call
    assert a[7] == b[7] <= 10 and np.array_equal(a[5], b[5]) and np.array_equal(a[6], b[6])


@pytest.mark.parametrize("m,n,one_pass", [(4100, 6000, True), (6000, 3000, False)])
def test_fused_tail_toggled_under_a_live_primal_dual_solver(apx, m, n, one_pass):
    """APX_OPT_FUSED_TAIL is public and read per iteration: switched off under a live stepping solver with
    f = Zero, the iteration falls back to the plain step kernel, which leaves no reduction partials — the next iteration must
    take its reductions from the vectors again, not from the stale partials of the last fused step (csrc/solver.hip
    enqueue_iteration: red_ready).  Off, on again, off: every piece bit-equal to the solver that never fused."""
    from importlib import import_module
    L = import_module("adaptive-proximal-algorithms_amd._lib")
    rng = np.random.default_rng(m + n + 1)
    A = np.asfortranarray(np.hstack([rng.uniform(-1, 1, (m, n - 1)), np.ones((m, 1))]))
    yv = A @ (rng.standard_normal(n) / np.sqrt(n)) + rng.laplace(size=m)
    nA = float(np.linalg.norm(A))
    x0, y0 = np.zeros(n), np.zeros(m)
    pieces = [(3, 1), (4, 0), (2, 1), (3, 0), (1, 0)]         # (iterations, fused) — the toggling solver's schedule
    outs = []
    for toggling in (True, False):
        ctx = apx.Context(0)
        ctx.fused_small(one_pass)
        ctx.one_pass_threshold(16 << 20)
        ctx.set_option(L.APX_OPT_FUSED_TAIL, 1 if toggling else 0)
        dA = apx.DeviceMatrix.from_numpy(A, ctx)
        s = apx.Solver(x0, y0, f=apx.Zero(), g=apx.NormL1(5.0), h=apx.Translate(apx.NormL1(1.0), -yv), A=dA, ctx=ctx,
                       rule=apx.OurRule(t=1.0, norm_A=nA), tol=0.0)
        seen = []
        for k, fused in pieces:
            if toggling:
                ctx.set_option(L.APX_OPT_FUSED_TAIL, fused)
            s.step(k)
            x, y, res = s.result()
            seen.append((x.copy(), y.copy(), (res.numit, res.gamma, res.sigma, res.norm_res)))
        s.close()
        outs.append(seen)
        dA.free()
    assert outs[0][-1][2][0] == sum(k for k, _ in pieces)
    for (xa, ya, ra), (xb, yb, rb) in zip(*outs):
        assert np.array_equal(xa, xb) and np.array_equal(ya, yb) and ra == rb, (ra, rb)


@pytest.mark.parametrize("m,n,kind,dtype", [(4100, 12001, "ls", np.float64),        # teams of two (contiguous shares by default)
                                            (4099, 30001, "ls", np.float64),        # teams of four: the last 10 KB piece is partial
                                            (4098, 44803, "logistic", np.float64),  # teams of eight, the bias entry in a piece of its own
                                            (4097, 100000, "ls", np.float64),       # C2's row: sixteen members, pieces 78 and 79 of 80 short / empty
                                            (4100, 20000, "pd", np.float64),        # C5's row
                                            (4099, 30001, "pd", np.float64),
                                            (4100, 40002, "logistic", np.float32),  # Float32 teams
                                            (4098, 90001, "ls", np.float32)])
def test_team_members_shares_contiguous_or_interleaved(apx, m, n, kind, dtype):
    """A team's members divide a row either into P contiguous shares (round 2) or — teams of four or more, by default — take its
    10 KB pieces in turn (csrc/team_rows.h, APX_OPT_TEAM_INTERLEAVE: with every load instruction a team covers one contiguous
    stretch of the row ; C2's kernel 6.07 -> 5.87 ms).  Which columns a thread holds changes, so the sums associate differently:
    the three settings agree to rounding with each other and with the oracle, on rows whose last pieces are partial or empty."""
    from importlib import import_module
    L = import_module("adaptive-proximal-algorithms_amd._lib")
    m = _fewer_rows(m, n)
    rng = np.random.default_rng(m + n)
    K = 6
    x0 = (0.01 * rng.standard_normal(n + (kind == "logistic"))).astype(dtype)
    if kind == "pd":
        A = np.asfortranarray(np.hstack([rng.uniform(-1, 1, (m, n - 1)), np.ones((m, 1))]).astype(dtype))
        yv = (A.astype(np.float64) @ (rng.standard_normal(n) / np.sqrt(n)) + rng.laplace(size=m)).astype(dtype)
        nA = float(np.linalg.norm(A.astype(np.float64)))
        xo, yo, _ = O.adaptive_primal_dual(np.zeros(n), np.zeros(m), f=O.Zero(), g=O.NormL1(5.0), h=O.Translate(O.NormL1(1.0), -yv.astype(np.float64)),
                                           A=A.astype(np.float64), rule=O.OurRule(t=1.0, norm_A=nA), tol=0.0, maxit=K)
    else:
        A = np.asfortranarray(rng.uniform(-1, 1, (m, n)).astype(dtype))
        b = (rng.uniform(size=m) < 0.5).astype(dtype) if kind == "logistic" else rng.standard_normal(m).astype(dtype)
        gam = 4.0 / (n / 3.0 + 1.0) if kind == "logistic" else 1.0 / (m * n / 3.0)
        fo = O.LogisticLoss(A.astype(np.float64), b.astype(np.float64)) if kind == "logistic" else O.LinearLeastSquares(A.astype(np.float64), b.astype(np.float64))
        xo, _ = O.adaptive_proxgrad(x0.astype(np.float64), f=fo, g=O.NormL1(1e-3 if kind == "logistic" else 1.0), rule=O.OurRule(gamma=gam), tol=0.0, maxit=K)
    xs = []
    for mode in (1, 0, 2):
        ctx = apx.Context(0)
        ctx.one_pass_threshold(16 << 20)
        ctx.set_option(L.APX_OPT_ONE_PASS_MIN_ROWS, 1024)
        ctx.set_option(L.APX_OPT_TEAM_INTERLEAVE, mode)
        dA = apx.DeviceMatrix.from_numpy(A, ctx).build_transpose()
        ctx.kernel_stats(reset=True)
        ctx.profile(True)
        if kind == "pd":
            x, y, it = apx.adaptive_primal_dual(np.zeros(n, dtype=dtype), np.zeros(m, dtype=dtype), f=apx.Zero(), g=apx.NormL1(5.0),
                                                h=apx.Translate(apx.NormL1(1.0), -yv), A=dA, rule=apx.OurRule(t=1.0, norm_A=nA), tol=0.0, maxit=K, ctx=ctx)
        else:
            f = apx.LogisticLoss(dA, b, ctx=ctx) if kind == "logistic" else apx.LinearLeastSquares(dA, b, ctx=ctx)
            x, it = apx.adaptive_proxgrad(x0, f=f, g=apx.NormL1(1e-3 if kind == "logistic" else 1.0), rule=apx.OurRule(gamma=gam), tol=0.0, maxit=K, ctx=ctx)
        st = ctx.kernel_stats(reset=True)
        ctx.profile(False)
        assert it == K and st["one_pass"]["launches"] >= K and st["gemv_t"]["launches"] <= (1 if kind == "pd" else 0), st     # (pd: A' y0 once)
        assert "k_team_rows" in ctx.kernel_name("one_pass"), ctx.kernel_name("one_pass")
        xs.append(np.asarray(x, dtype=np.float64))
        dA.free()
    tol = 1e-10 if dtype == np.float64 else 2e-3
    for x in xs:
        assert _rel(x, xo) <= tol
    assert _rel(xs[1], xs[0]) <= tol and _rel(xs[2], xs[0]) <= tol


@pytest.mark.parametrize("n,members", [(17000, 2), (20000, 2), (22528, 2), (34000, 4), (40000, 4), (45056, 4), (60000, 8)])
def test_team_shapes_by_row_length_and_the_override(apx, n, members):
    """csrc/team_rows.h team_plan: the members' shares as large as the registers allow — up to 22 528 columns teams of two, up to
    45 056 teams of four, each with the three-slot shape (seven or eight vectors, one row of slack) beyond six vectors per thread —
    for least squares and for the primal-dual form alike (round 3 chose twice the members for the primal-dual form where they
    filled their lanes better, until the communication wave got its priority: csrc/team_rows.h) ; APX_OPT_TEAM_MEMBERS overrides.
    The shapes only — their parity is test_one_pass_primal_dual_matches_oracle_and_two_passes and
    test_team_members_shares_contiguous_or_interleaved."""
    from importlib import import_module
    L = import_module("adaptive-proximal-algorithms_amd._lib")
    m = 4100
    names = {}
    for forced in (0, 2 * members):
        ctx = apx.Context(0)
        ctx.one_pass_threshold(16 << 20)
        if forced:
            ctx.set_option(L.APX_OPT_TEAM_MEMBERS, forced)
        dA = apx.DeviceMatrix.uniform(m, n, seed=3, ctx=ctx).build_transpose()
        yv = np.random.default_rng(n).standard_normal(m)
        x, y, it = apx.adaptive_primal_dual(np.zeros(n), np.zeros(m), f=apx.Zero(), g=apx.NormL1(5.0), h=apx.Translate(apx.NormL1(1.0), -yv), A=dA,
                                            rule=apx.OurRule(t=1.0, norm_A=float(np.sqrt(m * n / 3.0))), tol=0.0, maxit=3, ctx=ctx)
        assert it == 3 and np.all(np.isfinite(x))
        names[forced] = ctx.kernel_name("one_pass")
        if not forced:
            b = np.random.default_rng(1).standard_normal(m)
            apx.adaptive_proxgrad(np.zeros(n), f=apx.LinearLeastSquares(dA, b, ctx=ctx), g=apx.NormL1(1.0), rule=apx.OurRule(gamma=1e-7), tol=0.0, maxit=2, ctx=ctx)
            ls_name = ctx.kernel_name("one_pass")
        dA.free()
    wide = "3 row slots" if n <= 45056 else "4 row slots"
    assert "primal_dual, %d members" % members in names[0] and wide in names[0], names
    assert "least_squares, %d members" % members in ls_name and wide in ls_name, ls_name
    assert "primal_dual, %d members" % (2 * members) in names[2 * members] and "4 row slots" in names[2 * members], names


@pytest.mark.parametrize("n,dtype", [(3000, np.float64), (9000, np.float32), (12001, np.float64), (40000, np.float64)])
def test_forced_row_deals_on_a_small_matrix(apx, n, dtype):
    """APX_OPT_ROW_DEAL is an A/B hook, but any value it accepts must be safe: slabs (0) and chunks far larger than the default
    (64, 1024 rows) leave some of the 256 workgroups (teams) WITHOUT rows on a matrix of 4097 rows — they run their dummy steps on
    the matrix's last row (weight 0) instead of pointing past it — and every deal gives the default's iterates to rounding."""
    from importlib import import_module
    L = import_module("adaptive-proximal-algorithms_amd._lib")
    m = 4097
    rng = np.random.default_rng(n)
    A = np.asfortranarray(rng.uniform(-1, 1, (m, n)).astype(dtype))
    b = rng.standard_normal(m).astype(dtype)
    xs = []
    for deal in (-1, 0, 1, 64, 1024):
        ctx = apx.Context(0)
        ctx.one_pass_threshold(16 << 20)
        ctx.set_option(L.APX_OPT_ROW_DEAL, deal)
        dA = apx.DeviceMatrix.from_numpy(A, ctx).build_transpose()
        ctx.kernel_stats(reset=True)
        ctx.profile(True)
        x, it = apx.adaptive_proxgrad(np.zeros(n, dtype=dtype), f=apx.LinearLeastSquares(dA, b, ctx=ctx), g=apx.NormL1(1.0),
                                      rule=apx.OurRule(gamma=1.0 / (m * n / 3.0)), tol=0.0, maxit=6, ctx=ctx)
        st = ctx.kernel_stats(reset=True)
        ctx.profile(False)
        assert it == 6 and st["one_pass"]["launches"] >= 6 and st["gemv_t"]["launches"] == 0, st
        xs.append(np.asarray(x, dtype=np.float64))
        dA.free()
    tol = 1e-10 if dtype == np.float64 else 2e-3
    for x in xs[1:]:
        assert np.all(np.isfinite(x)) and _rel(x, xs[0]) <= tol


def test_fused_small_keeps_one_pass_mode_two(apx):
    """Context.fused_small(True) switches the one-pass policy back on AS IT WAS: a context told never to use the team kernels
    (one_pass(2): a GPU shared with other work) stays off them — rows that need a team take the two passes."""
    rng = np.random.default_rng(2)
    m, n = 4100, 12000
    A = np.asfortranarray(rng.uniform(-1, 1, (m, n)))
    b = rng.standard_normal(m)
    launches = []
    for mode in (2, 1):
        ctx = apx.Context(0)
        ctx.one_pass_threshold(16 << 20)
        ctx.one_pass(mode)
        ctx.fused_small(False)
        ctx.fused_small(True)
        dA = apx.DeviceMatrix.from_numpy(A, ctx).build_transpose()
        ctx.kernel_stats(reset=True)
        ctx.profile(True)
        apx.adaptive_proxgrad(np.zeros(n), f=apx.LinearLeastSquares(dA, b, ctx=ctx), g=apx.NormL1(1.0), rule=apx.OurRule(gamma=1e-7),
                              tol=0.0, maxit=3, ctx=ctx)
        st = ctx.kernel_stats(reset=True)
        ctx.profile(False)
        launches.append((st["one_pass"]["launches"], st["gemv_t"]["launches"]))
        dA.free()
    assert launches[0][0] == 0 and launches[0][1] >= 4, launches      # mode 2: no team launch, the two passes
    assert launches[1] == (4, 0), launches                            # mode 1: the team kernels


@pytest.mark.parametrize("m,n,hkind,t", [(9000, 1000, "tl1", 1.0),      # k_multi_rows
                                         (9000, 1000, "tl2", 1.0),
                                         (4100, 6000, "tl1", 0.1),      # k_team_rows, one workgroup
                                         (4100, 20000, "tl1", 1.0),     # ... teams of two (C5's row)
                                         (4100, 20000, "tl2", 10.0),
                                         (4099, 30001, "tl2", 1.0),     # ... teams of four
                                         (100001, 250, "tl2", 1.0),     # k_multi_rows, five groups of two waves, a ragged last chunk
                                         (4097, 21505, "tl1", 1.0),     # teams of two, eight vectors, three slots, an odd row length
                                         (4099, 70001, "tl2", 1.0)])    # teams of eight, three slots (round 4's plan)
def test_adapdm_plus_first_trial_in_one_pass(apx, m, n, hkind, t):
    """adaptive_linesearch_primal_dual (AdaPDM+, src/AdaProx.jl:463-550): the first linesearch trial of an iteration takes
    A_x = A * x (:502) along — one launch of the one-pass primal-dual kernels leaves A x AND A' y_next (csrc/slab_pd.hip
    slab_pd_trial, round 4) ; a rejected trial continues on the stored A x with one GEMV-T each.  Against the oracle and against
    the context that runs every product as a GEMV: iterates, records, evaluation counts ; eta0 far below ||A||, so that trials
    ARE rejected."""
    m = _fewer_rows(m, n)
    rng = np.random.default_rng(m + n + 7)
    A = np.asfortranarray(np.hstack([rng.uniform(-1, 1, (m, n - 1)), np.ones((m, 1))]))
    yv = A @ (rng.standard_normal(n) / np.sqrt(n)) + rng.laplace(size=m)
    nA = float(np.linalg.norm(A))
    mk_h = {"tl1": lambda M: M.Translate(M.NormL1(1.0), -yv), "tl2": lambda M: M.Translate(M.NormL2(1.0), -yv)}[hkind]
    K = 12
    kw = dict(eta=nA / 2000.0, t=t, tol=0.0, maxit=K)
    x0, y0 = np.zeros(n), np.zeros(m)
    ro = []
    lam = 5.0 if hkind == "tl1" else 0.3       # (the l2 ball: |A' y| stays below 1.3 or so — a larger lambda keeps x at 0)
    xo, yo, _ = O.adaptive_linesearch_primal_dual(x0, y0, f=O.Zero(), g=O.NormL1(lam), h=O.Counting(mk_h(O)), A=O.Counting(A),
                                                  record=ro.append, **kw)
    assert ro[-1]["At_evals"] > K + 1          # (some trial was rejected)
    assert (hkind != "tl2" or np.count_nonzero(xo) > n // 10) and ro[-1]["norm_res"] > 1e-3       # (a loop that is still moving)
    out = []
    for one_pass in (True, False):
        ctx = apx.Context(0)
        ctx.fused_small(one_pass)
        ctx.one_pass_threshold(16 << 20)
        ctx.set_option(_L().APX_OPT_ONE_PASS_MIN_ROWS, 1024)
        dA = apx.DeviceMatrix.from_numpy(A, ctx)
        recs = []
        ctx.kernel_stats(reset=True)
        ctx.profile(True)
        x, y, it = apx.adaptive_linesearch_primal_dual(x0, y0, f=apx.Zero(), g=apx.NormL1(lam), h=apx.Counting(mk_h(apx)),
                                                       A=apx.Counting(dA), ctx=ctx, record=recs.append, **kw)
        st = ctx.kernel_stats(reset=True)
        ctx.profile(False)
        assert it == K and len(recs) == K
        assert st["one_pass"]["launches"] == (K if one_pass else 0), st      # one launch per iteration: the first trial's
        out.append((x, y, recs))
        dA.free()
    (x1, y1, r1), (x2, y2, r2) = out
    assert _rel(x1, xo) <= 1e-10 and _rel(y1, yo) <= 1e-9 and _rel(x1, x2) <= 1e-10 and _rel(y1, y2) <= 1e-9
    for a, b_ in zip(r1, ro):
        assert a["gamma"] == pytest.approx(b_["gamma"], rel=1e-9) and a["sigma"] == pytest.approx(b_["sigma"], rel=1e-9)
        assert a["norm_res"] == pytest.approx(b_["norm_res"], rel=1e-7) and a["objective"] == pytest.approx(b_["objective"], rel=1e-9)
        assert (a["A_evals"], a["At_evals"], a["prox_h_evals"]) == (b_["A_evals"], b_["At_evals"], b_["prox_h_evals"])


@pytest.mark.parametrize("m,n,hkind", [(9000, 1000, "tl1"),       # k_multi_rows
                                       (4100, 6000, "tl2"),       # k_team_rows, one workgroup ; the l2 ball
                                       (4100, 20000, "tl1"),      # ... teams of two (C5's row)
                                       (4099, 30001, "tl2"),      # ... teams of four
                                       (90001, 380, "tl1"),       # k_multi_rows, three groups of three waves
                                       (4099, 70001, "tl2")])     # teams of eight, three slots (round 4's plan)
def test_malitsky_pock_trials_in_one_pass(apx, m, n, hkind):
    """malitsky_pock (src/AdaProx.jl:555-629): every linesearch trial's A_x = A * x takes the NEXT iteration's dual step along
    (w = y + sigma A_x, y = prox, At_y = A' y with the trial's sigma) — one launch of the one-pass primal-dual kernels in trial
    mode per trial, no GEMV-T of the dual step in the loop (csrc/baselines.hip, round 4).  Against the oracle and against the
    context that runs every product as a GEMV ; sigma0 large enough that trials ARE rejected."""
    m = _fewer_rows(m, n)
    rng = np.random.default_rng(m + n + 13)
    A = np.asfortranarray(np.hstack([rng.uniform(-1, 1, (m, n - 1)), np.ones((m, 1))]))
    yv = A @ (rng.standard_normal(n) / np.sqrt(n)) + rng.laplace(size=m)
    nA = float(np.linalg.norm(A))
    mk_h = {"tl1": lambda M: M.Translate(M.NormL1(1.0), -yv), "tl2": lambda M: M.Translate(M.NormL2(1.0), -yv)}[hkind]
    lam = 5.0 if hkind == "tl1" else 0.3
    K = 12
    kw = dict(sigma=200.0 / nA, t=1.0, tol=0.0, maxit=K)
    x0, y0 = np.zeros(n), np.zeros(m)
    ro = []
    xo, yo, _ = O.malitsky_pock(x0, y0, f=O.Zero(), g=O.NormL1(lam), h=mk_h(O), A=O.Counting(A), record=ro.append, **kw)
    assert ro[-1]["A_evals"] > K + 1 and ro[-1]["norm_res"] > 1e-3          # (some trial was rejected ; a loop that is still moving)
    out = []
    for one_pass in (True, False):
        ctx = apx.Context(0)
        ctx.fused_small(one_pass)
        ctx.one_pass_threshold(16 << 20)
        ctx.set_option(_L().APX_OPT_ONE_PASS_MIN_ROWS, 1024)
        dA = apx.DeviceMatrix.from_numpy(A, ctx)
        recs = []
        ctx.kernel_stats(reset=True)
        ctx.profile(True)
        x, y, it = apx.malitsky_pock(x0, y0, f=apx.Zero(), g=apx.NormL1(lam), h=mk_h(apx), A=apx.Counting(dA), ctx=ctx, record=recs.append, **kw)
        st = ctx.kernel_stats(reset=True)
        ctx.profile(False)
        assert it == K and len(recs) == K
        if one_pass:      # one launch per TRIAL, and the only GEMV-Ts are the two in front of the loop's first trial
            assert st["one_pass"]["launches"] == recs[-1]["A_evals"] - 1 and st["gemv_t"]["launches"] <= 3, (st, recs[-1])
        else:
            assert st["one_pass"]["launches"] == 0, st
        out.append((x, y, recs))
        dA.free()
    (x1, y1, r1), (x2, y2, r2) = out
    assert _rel(x1, xo) <= 1e-10 and _rel(y1, yo) <= 1e-9 and _rel(x1, x2) <= 1e-10 and _rel(y1, y2) <= 1e-9
    for a, b_ in zip(r1, ro):
        assert a["gamma"] == pytest.approx(b_["gamma"], rel=1e-9) and a["sigma"] == pytest.approx(b_["sigma"], rel=1e-9)
        assert a["norm_res"] == pytest.approx(b_["norm_res"], rel=1e-7) and a["objective"] == pytest.approx(b_["objective"], rel=1e-9)
        assert (a["A_evals"], a["At_evals"]) == (b_["A_evals"], b_["At_evals"])


def test_one_pass_l2_ball_termination_and_stepping(apx):
    """The l2 ball on the one-pass kernels closes its iteration in a kernel BEHIND the closing kernel (k_dual_post + k_final:
    the convergence flag is set there): termination at the oracle's iteration for every polling interval, stepping in pieces
    equal to one call bit for bit, on k_multi_rows and on k_team_rows."""
    for m, n in ((9000, 1000), (4100, 12000)):
        rng = np.random.default_rng(m + n + 3)
        A = np.asfortranarray(np.hstack([rng.uniform(-1, 1, (m, n - 1)), np.ones((m, 1))]))
        yv = A @ (rng.standard_normal(n) / np.sqrt(n)) + rng.laplace(size=m)
        nA = float(np.linalg.norm(A))
        ctx = apx.Context(0)
        ctx.one_pass_threshold(16 << 20)
        ctx.set_option(_L().APX_OPT_ONE_PASS_MIN_ROWS, 1024)
        dA = apx.DeviceMatrix.from_numpy(A, ctx)
        t = dict(f=apx.Zero(), g=apx.NormL1(0.05), h=apx.Translate(apx.NormL2(1.0), -yv), A=dA, ctx=ctx)
        to = dict(f=O.Zero(), g=O.NormL1(0.05), h=O.Translate(O.NormL2(1.0), -yv), A=A)
        x0, y0 = np.zeros(n), np.zeros(m)
        rule, rule_o = apx.OurRule(t=1.0, norm_A=nA), O.OurRule(t=1.0, norm_A=nA)
        recs = []
        ctx.kernel_stats(reset=True)
        ctx.profile(True)
        x, y, it = apx.adaptive_primal_dual(x0, y0, rule=rule, tol=0.0, maxit=30, record=recs.append, **t)
        st = ctx.kernel_stats(reset=True)
        ctx.profile(False)
        assert it == 30 and st["one_pass"]["launches"] == 30, st
        tol = recs[19]["norm_res"] * (1 + 1e-9)
        kstar = 1 + next(k for k, r in enumerate(recs) if r["norm_res"] <= tol)
        xo, yo, ito = O.adaptive_primal_dual(x0, y0, rule=rule_o, tol=tol, maxit=1000, **to)
        for poll in (1, 7, 64):
            xt, yt, itt = apx.adaptive_primal_dual(x0, y0, rule=rule, tol=tol, maxit=1000, poll_every=poll, **t)
            assert itt == kstar == ito and _rel(xt, xo) <= 1e-10 and _rel(yt, yo) <= 1e-9, (m, n, poll, itt, kstar, ito)
        s = apx.Solver(x0, y0, rule=rule, tol=0.0, **t)
        for k in (1, 9, 20):
            s.step(k)
            s.result()
        xs, ys, res = s.result()
        s.close()
        assert res.numit == 30 and np.array_equal(xs, x) and np.array_equal(ys, y)
        dA.free()


@pytest.mark.parametrize("m,n,hkind", [(4100, 20000, "tl1"), (4099, 30001, "tl2"), (4098, 100000, "tbox")])
def test_deferred_residual_sums_leave_the_iterates_alone(apx, m, n, hkind):
    """APX_OPT_TEAM_PD_DEFER (round 4): the communication wave of the primal-dual team kernels works off a row's terms of the
    residual norm and of h(A x), and member 0's stores of y+ and A x, sixteen rows at a time in the lanes of the rows' numbers.
    Only the ORDER of the two sums changes: the iterates are the same bits, the recorded residuals and objectives agree to
    rounding, and the loop stops at the same iteration."""
    from importlib import import_module
    L = import_module("adaptive-proximal-algorithms_amd._lib")
    rng = np.random.default_rng(m + n + 17)
    A = np.asfortranarray(np.hstack([rng.uniform(-1, 1, (m, n - 1)), np.ones((m, 1))]))
    yv = A @ (rng.standard_normal(n) / np.sqrt(n)) + rng.laplace(size=m)
    nA = float(np.linalg.norm(A))
    h = {"tl1": apx.Translate(apx.NormL1(1.0), -yv), "tbox": apx.Translate(apx.IndBox(-0.5, 0.8), -yv),
         "tl2": apx.Translate(apx.NormL2(1.0), -yv)}[hkind]
    runs = []
    for defer in (1, 0):
        ctx = apx.Context(0)
        ctx.one_pass_threshold(16 << 20)
        ctx.set_option(L.APX_OPT_TEAM_PD_DEFER, defer)
        dA = apx.DeviceMatrix.from_numpy(A, ctx)
        t = dict(f=apx.Zero(), g=apx.NormL1(0.05 if hkind == "tl2" else 5.0), h=h, A=dA, rule=apx.OurRule(t=1.0, norm_A=nA), ctx=ctx)
        recs = []
        x, y, it = apx.adaptive_primal_dual(np.zeros(n), np.zeros(m), tol=0.0, maxit=37, record=recs.append, **t)
        tol = recs[20]["norm_res"] * (1 + 1e-9)
        xt, yt, itt = apx.adaptive_primal_dual(np.zeros(n), np.zeros(m), tol=tol, maxit=200, poll_every=5, **t)
        runs.append((x, y, recs, xt, yt, itt))
        dA.free()
    a, b = runs
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert [r["norm_res"] for r in a[2]] == pytest.approx([r["norm_res"] for r in b[2]], rel=1e-12)
    assert [r["objective"] for r in a[2]] == pytest.approx([r["objective"] for r in b[2]], rel=1e-12)
    assert a[5] == b[5] <= 37 and np.array_equal(a[3], b[3]) and np.array_equal(a[4], b[4])


def test_primal_dual_loops_with_a_smooth_f_on_its_own_one_pass_kernels(apx):
    """The primal-dual loops with a smooth f whose matrix is large enough for ITS one-pass kernels (f and grad f from one read of
    Q, csrc/slab.hip — round 4: a linear operator in the problem no longer keeps f on the two GEMV passes) next to the operator's
    (A x, the dual step and A' y from one read of A, csrc/slab_pd.hip): adaptive_primal_dual, AdaPDM+ and Malitsky-Pock against
    the oracle — all branches of the step-size formulas are live (C, L != 0)."""
    rng = np.random.default_rng(77)
    m, n, k = 9000, 1000, 8300
    A = np.asfortranarray(np.hstack([rng.uniform(-1, 1, (m, n - 1)), np.ones((m, 1))]))
    Q = np.asfortranarray(rng.uniform(-1, 1, (k, n)) / np.sqrt(n))
    q = rng.standard_normal(k)
    yv = A @ (rng.standard_normal(n) / np.sqrt(n)) + rng.laplace(size=m)
    nA = float(np.linalg.norm(A))
    K = 12
    ctx = apx.Context(0)
    ctx.one_pass_threshold(16 << 20)
    dA = apx.DeviceMatrix.from_numpy(A, ctx)
    dQ = apx.DeviceMatrix.from_numpy(Q, ctx).build_transpose()
    for name, run, kw in (("adapdm", "adaptive_linesearch_primal_dual", dict(eta=nA / 500.0, t=0.5)),
                          ("mp", "malitsky_pock", dict(sigma=100.0 / nA, t=1.0)),
                          ("pd", "adaptive_primal_dual", dict(rule_args=dict(t=1.0, norm_A=nA)))):
        ro, rd = [], []
        kwo, kwd = dict(kw), dict(kw)
        if "rule_args" in kw:
            kwo = dict(rule=O.OurRule(**kw["rule_args"]))
            kwd = dict(rule=apx.OurRule(**kw["rule_args"]))
        xo, yo, _ = getattr(O, run)(np.zeros(n), np.zeros(m), f=O.LinearLeastSquares(Q, q), g=O.NormL1(0.3), h=O.Translate(O.NormL1(1.0), -yv),
                                    A=O.Counting(A), tol=0.0, maxit=K, record=ro.append, **kwo)
        ctx.kernel_stats(reset=True)
        ctx.profile(True)
        x, y, it = getattr(apx, run)(np.zeros(n), np.zeros(m), f=apx.LinearLeastSquares(dQ, q, ctx=ctx), g=apx.NormL1(0.3),
                                     h=apx.Translate(apx.NormL1(1.0), -yv), A=apx.Counting(dA), tol=0.0, maxit=K, ctx=ctx, record=rd.append, **kwd)
        st = ctx.kernel_stats(reset=True)
        ctx.profile(False)
        # both kinds of one-pass launches ran: f's (one per evaluation) and the operator's (one per iteration / per trial) — the
        # only GEMVs left are the two in front of the loop and the rejected trials'
        assert it == K and st["one_pass"]["launches"] >= 2 * K and st["gemv_n"]["launches"] <= 1, (name, st)
        assert _rel(x, xo) <= 1e-10 and _rel(y, yo) <= 1e-9, name
        for a, b_ in zip(rd, ro):
            assert a["gamma"] == pytest.approx(b_["gamma"], rel=1e-9) and a["objective"] == pytest.approx(b_["objective"], rel=1e-9), name
            assert (a["A_evals"], a["At_evals"]) == (b_["A_evals"], b_["At_evals"]), name
    dA.free()
    dQ.free()


def test_team_wait_slots_after_a_team_run(apx):
    """apx_problem_team_wait_slots (ABI 4) behind a run on the team kernels: every reported slot waited at least once, its longest
    wait lies inside the run (a launch of the run, a row of the team's deal, a member of its team, an XCD of the part), and the
    per-slot maxima add up to apx_problem_team_waits' summary."""
    rng = np.random.default_rng(5)
    m, n = 4100, 30000
    A = np.asfortranarray(rng.uniform(-1, 1, (m, n)))
    b = rng.standard_normal(m)
    ctx = apx.Context(0)
    ctx.one_pass_threshold(16 << 20)
    dA = apx.DeviceMatrix.from_numpy(A, ctx).build_transpose()
    s = apx.Solver(np.zeros(n), f=apx.LinearLeastSquares(dA, b, ctx=ctx), g=apx.NormL1(1.0), rule=apx.OurRule(gamma=1e-8), tol=0.0, ctx=ctx)
    s.step(40, sync=True)
    summary, slots = s.team_waits(), s.team_wait_slots()
    assert summary["team_launches"] == 41
    assert 0 < len(slots) <= 256 and all(q["waits"] > 0 and q["kind"] == 0 for q in slots)
    assert sum(q["waits"] for q in slots) == summary["waits"]
    assert max(q["max_wait_ns"] for q in slots) == summary["max_wait_ns"] and max(q["max_polls"] for q in slots) == summary["max_polls"]
    for q in slots:
        assert 0 <= q["at_launch"] < 41 and 0 <= q["at_step"] < m and 0 <= q["team"] < 64 and 0 <= q["member"] < 4 and 0 <= q["xcc"] < 8
        assert q["long_waits"] <= q["waits"] and q["at_clock_ns"] > 0 and 0 <= q["block"] < 256
    s.close()
    dA.free()
