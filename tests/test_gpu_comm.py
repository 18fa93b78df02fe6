"""RCCL inside the library (apx_comm_*, csrc/comm.hip) with the one rank a one-GPU box offers: an all-reduce over one
rank is the identity, so every row-sharded run through the communicator must equal the un-sharded run of the same
(generic) kernels BIT FOR BIT — the gradient of src/AdaProx.jl:336 is what the collective completes.  N > 1 ranks over
xGMI are only ever run by the driver (bench.py --gpus N); the sharding algebra with two ranks is covered on the CPU
(tests/test_sharded_gloo.py) and with two real ranks sharing the GPU over the callback transport
(tests/test_gpu_two_ranks.py)."""
import ctypes as C
import importlib
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import adaprox_oracle as O
from oracle import problems as P

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def apx():
    import adaprox_amd as apx
    apx.load()
    return apx


@pytest.fixture(scope="module")
def sharded():
    return importlib.import_module("adaptive-proximal-algorithms_amd.sharded")


@pytest.fixture()
def generic_ctx(apx):
    """A context on the generic kernels: a problem with a communicator never takes the small-size paths, so the
    un-sharded run it is compared with must not either."""
    ctx = apx.Context(device=0)
    ctx.fused_small(False)
    yield ctx
    ctx.close()


def test_init_rank_one_rank_is_bit_equal_to_unsharded(apx, sharded, generic_ctx):
    ctx = generic_ctx
    comm = sharded.Comm.init_rank(ctx, 1, 0, sharded.Comm.unique_id())
    assert comm.info() == (1, 0)                      # ncclCommCount / ncclCommUserRank
    v = np.arange(5, dtype=np.float64) - 1.5
    assert np.array_equal(comm(v), v)                 # host all-reduce (set-up code)
    prob = P.random_lasso(300, 700, pfactor=10, seed=6)
    gam = 1.0 / P.opnorm_sq_power(prob.A, iters=30)
    x0 = np.zeros(700)
    A = apx.DeviceMatrix.from_numpy(prob.A, ctx=ctx)
    f_plain = apx.LinearLeastSquares(A, prob.b)
    f_comm = apx.LinearLeastSquares(A, prob.b, allreduce=comm)
    ra, rb = [], []
    xa, ita = apx.adaptive_proxgrad(x0, f=f_plain, g=apx.NormL1(1.0), rule=apx.OurRule(gamma=gam), tol=0.0, maxit=60,
                                    record=ra.append, ctx=ctx)
    xb, itb = apx.adaptive_proxgrad(x0, f=f_comm, g=apx.NormL1(1.0), rule=apx.OurRule(gamma=gam), tol=0.0, maxit=60,
                                    record=rb.append, ctx=ctx)
    assert ita == itb == 60 and np.array_equal(xa, xb)
    assert [r["gamma"] for r in ra] == [r["gamma"] for r in rb]
    assert [r["objective"] for r in ra] == [r["objective"] for r in rb]
    # ... and both are the oracle's iterates
    xo, _ = O.adaptive_proxgrad(x0, f=O.LinearLeastSquares(prob.A, prob.b), g=O.NormL1(1.0), rule=O.OurRule(gamma=gam),
                                tol=0.0, maxit=60)
    assert np.max(np.abs(xb - xo)) <= 1e-10 * np.max(np.abs(xo))
    # the host-driven loop (a scalar all-reduce per trial)
    za, _ = apx.backtracking_proxgrad(x0, f=f_plain, g=apx.NormL1(1.0), gamma0=gam, xi=1.5, tol=0.0, maxit=15, ctx=ctx)
    zb, _ = apx.backtracking_proxgrad(x0, f=f_comm, g=apx.NormL1(1.0), gamma0=gam, xi=1.5, tol=0.0, maxit=15, ctx=ctx)
    assert np.array_equal(za, zb)
    # logistic fp32: the gradient and the two loss scalars travel separately (fp32 buffer + fp64 scalars)
    rng = np.random.default_rng(3)
    X = rng.uniform(-1, 1, (500, 64)).astype(np.float32)
    y = (rng.uniform(size=500) < 0.5).astype(np.float32)
    Xd = apx.DeviceMatrix.from_numpy(X, ctx=ctx)
    w0 = np.zeros(65, dtype=np.float32)
    wa, _ = apx.adaptive_proxgrad(w0, f=apx.LogisticLoss(Xd, y), g=apx.NormL1(0.01), rule=apx.OurRule(gamma=1.0), tol=0.0,
                                  maxit=25, ctx=ctx)
    wb, _ = apx.adaptive_proxgrad(w0, f=apx.LogisticLoss(Xd, y, allreduce=comm), g=apx.NormL1(0.01),
                                  rule=apx.OurRule(gamma=1.0), tol=0.0, maxit=25, ctx=ctx)
    assert np.array_equal(wa, wb)
    comm.close()


def test_one_pass_kernels_on_a_row_shard(apx, sharded):
    """A row shard of f is as good as a whole matrix for the one-pass kernels (the rows are local; only the block sums
    travel): with a communicator the run takes the SAME kernels — first- and second-generation row kernels, the teams of
    workgroups for rows of C2's kind — and, with one rank, gives the same bits.  This is the path `bench.py --gpus N` times."""
    ctx = apx.Context(device=0)
    ctx.one_pass_threshold(16 << 20)
    comm = sharded.Comm.init_rank(ctx, 1, 0, sharded.Comm.unique_id())
    rng = np.random.default_rng(17)
    for m, n, dtype, logistic in ((9000, 1000, np.float64, False), (6000, 6000, np.float32, True), (4200, 12000, np.float64, False),
                                  (4100, 30000, np.float64, True)):
        A = np.asfortranarray(rng.uniform(-1, 1, (m, n)).astype(dtype))
        dA = apx.DeviceMatrix.from_numpy(A, ctx).build_transpose()
        if logistic:
            b = (rng.uniform(size=m) < 0.5).astype(dtype)
            mk = lambda **kw: apx.LogisticLoss(dA, b, ctx=ctx, **kw)      # noqa: E731
            nx, gam, lam = n + 1, 4.0 / (n / 3.0 + 1.0), 0.01
        else:
            b = rng.standard_normal(m).astype(dtype)
            mk = lambda **kw: apx.LinearLeastSquares(dA, b, ctx=ctx, **kw)      # noqa: E731
            nx, gam, lam = n, 1.0 / (m * n / 3.0), 1.0
        x0 = (0.01 * rng.standard_normal(nx)).astype(dtype)
        out = []
        for f in (mk(), mk(allreduce=comm)):
            ctx.kernel_stats(reset=True)
            ctx.profile(True)
            recs = []
            x, it = apx.adaptive_proxgrad(x0, f=f, g=apx.NormL1(lam), rule=apx.OurRule(gamma=gam), tol=0.0, maxit=15, ctx=ctx,
                                          record=recs.append)
            st = ctx.kernel_stats(reset=True)
            ctx.profile(False)
            assert st["gemv_t"]["launches"] == 0 and st["gemv_n"]["launches"] == 0 and st["one_pass"]["launches"] == 16, (m, n, st)      # one pass over A per evaluation
            out.append((x, [r["objective"] for r in recs], [r["gamma"] for r in recs]))
        assert np.array_equal(out[0][0], out[1][0]) and out[0][1] == out[1][1] and out[0][2] == out[1][2], (m, n)
        dA.free()
    comm.close()
    ctx.close()


def test_row_sharded_linear_operator_through_the_communicator(apx, sharded, generic_ctx):
    """C5's shape of problem: A_lin sharded (shard_mask bit 1) — A'y, the dual residual and h(Ax) are all-reduced."""
    ctx = generic_ctx
    comm = sharded.Comm.init_rank(ctx, 1, 0, sharded.Comm.unique_id())
    rng = np.random.default_rng(5)
    m, n = 240, 90
    A = rng.uniform(-1, 1, (m, n))
    b = A @ (rng.standard_normal(n) * (rng.uniform(size=n) < 0.2)) + rng.laplace(size=m)
    Ad = apx.DeviceMatrix.from_numpy(A, ctx=ctx)
    norm_A = float(np.linalg.norm(A))
    for h in (apx.Translate(apx.NormL1(1.0), -b), apx.Translate(apx.NormL2(1.0), -b)):
        kw = dict(f=apx.Zero(), g=apx.NormL1(2.0), h=h, A=Ad, tol=0.0, maxit=80, ctx=ctx)
        xa, ya, _ = apx.condat_vu(np.zeros(n), np.zeros(m), Lf=0.0, norm_A=norm_A, **kw)
        xb, yb, _ = apx.condat_vu(np.zeros(n), np.zeros(m), Lf=0.0, norm_A=norm_A, A_allreduce=comm, **kw)
        # (the sharded loop closes the iteration from all-reduced sums in a kernel of its own: same arithmetic,
        # the residual norm is summed in another order — iterates agree to rounding, not bit for bit)
        assert np.max(np.abs(xa - xb)) <= 1e-13 * max(np.max(np.abs(xa)), 1.0)
        assert np.max(np.abs(ya - yb)) <= 1e-13 * max(np.max(np.abs(ya)), 1.0)
        rule = apx.OurRule(norm_A=norm_A, t=1.0)
        xa, ya, _ = apx.adaptive_primal_dual(np.zeros(n), np.zeros(m), rule=rule, **kw)
        xb, yb, _ = apx.adaptive_primal_dual(np.zeros(n), np.zeros(m), rule=rule, A_allreduce=comm, **kw)
        assert np.max(np.abs(xa - xb)) <= 1e-12 * max(np.max(np.abs(xa)), 1.0)
    # AdaPDM+ on the sharded operator (round 4: A' y_next and the trial's three sums are all-reduced per trial) — with one rank the
    # unsharded iterates to rounding, rejected trials included ; the l2 ball (a norm over all ranks' rows in front of every trial's
    # prox) is refused, not silently wrong ; Malitsky-Pock the same way
    kw = dict(f=apx.Zero(), g=apx.NormL1(2.0), h=apx.Translate(apx.NormL1(1.0), -b), eta=norm_A / 200.0, t=1.0, tol=0.0, maxit=30, ctx=ctx)
    ra, rb = [], []
    xa, ya, _ = apx.adaptive_linesearch_primal_dual(np.zeros(n), np.zeros(m), A=apx.Counting(Ad), record=ra.append, **kw)
    xb, yb, _ = apx.adaptive_linesearch_primal_dual(np.zeros(n), np.zeros(m), A=apx.Counting(Ad), record=rb.append, A_allreduce=comm, **kw)
    assert ra[-1]["At_evals"] > 31 and [r["At_evals"] for r in ra] == [r["At_evals"] for r in rb]
    assert np.max(np.abs(xa - xb)) <= 1e-12 * max(np.max(np.abs(xa)), 1.0) and np.max(np.abs(ya - yb)) <= 1e-12
    kwm = dict(f=apx.Zero(), g=apx.NormL1(2.0), h=apx.Translate(apx.NormL1(1.0), -b), sigma=50.0 / norm_A, t=1.0, tol=0.0, maxit=30, ctx=ctx)
    ra, rb = [], []
    xa, ya, _ = apx.malitsky_pock(np.zeros(n), np.zeros(m), A=apx.Counting(Ad), record=ra.append, **kwm)
    xb, yb, _ = apx.malitsky_pock(np.zeros(n), np.zeros(m), A=apx.Counting(Ad), record=rb.append, A_allreduce=comm, **kwm)
    assert ra[-1]["A_evals"] > 31 and [r["A_evals"] for r in ra] == [r["A_evals"] for r in rb]
    assert np.max(np.abs(xa - xb)) <= 1e-12 * max(np.max(np.abs(xa)), 1.0) and np.max(np.abs(ya - yb)) <= 1e-12
    with pytest.raises(apx.ApxError, match="separable h only"):
        apx.malitsky_pock(np.zeros(n), np.zeros(m), f=apx.Zero(), g=apx.NormL1(2.0), h=apx.Translate(apx.NormL2(1.0), -b), A=Ad,
                          sigma=1.0 / norm_A, tol=0.0, maxit=3, ctx=ctx, A_allreduce=comm)
    with pytest.raises(apx.ApxError, match="separable h only"):
        apx.adaptive_linesearch_primal_dual(np.zeros(n), np.zeros(m), f=apx.Zero(), g=apx.NormL1(2.0),
                                            h=apx.Translate(apx.NormL2(1.0), -b), A=Ad, eta=norm_A, tol=0.0, maxit=3,
                                            ctx=ctx, A_allreduce=comm)

    comm.close()


def test_row_sharded_operator_on_the_one_pass_kernels(apx, sharded):
    """A row shard of the OPERATOR is as good as a whole one for the one-pass primal-dual kernels (round 4): the rows are local,
    A' y and the two residual sums are all-reduced behind the closing kernel, the iteration is closed from the global sums.  With
    one rank through the communicator: the same kernels as the unsharded run (k_multi_rows, k_team_rows alone and in teams),
    iterates equal to rounding (the residual norm is closed in a kernel of its own).  The l2 ball too: A' z and sum z^2 are
    all-reduced first, the scaling and y+ come behind that."""
    ctx = apx.Context(device=0)
    ctx.one_pass_threshold(16 << 20)
    comm = sharded.Comm.init_rank(ctx, 1, 0, sharded.Comm.unique_id())
    rng = np.random.default_rng(23)
    for m, n, hk in ((9000, 1000, "l1"), (4100, 6000, "box"), (4100, 20000, "l1"), (9000, 1000, "l2"), (4100, 20000, "l2")):
        A = np.asfortranarray(np.hstack([rng.uniform(-1, 1, (m, n - 1)), np.ones((m, 1))]))
        yv = A @ (rng.standard_normal(n) / np.sqrt(n)) + rng.laplace(size=m)
        nA = float(np.linalg.norm(A))
        dA = apx.DeviceMatrix.from_numpy(A, ctx).build_transpose()
        h = {"l1": apx.Translate(apx.NormL1(1.0), -yv), "box": apx.Translate(apx.IndBox(-0.5, 0.8), -yv),
             "l2": apx.Translate(apx.NormL2(1.0), -yv)}[hk]
        kw = dict(f=apx.Zero(), g=apx.NormL1(0.3 if hk == "l2" else 5.0), h=h, A=dA, rule=apx.OurRule(t=1.0, norm_A=nA), tol=0.0, maxit=12, ctx=ctx)
        out = []
        for extra in ({}, {"A_allreduce": comm}):
            recs = []
            ctx.kernel_stats(reset=True)
            ctx.profile(True)
            x, y, it = apx.adaptive_primal_dual(np.zeros(n), np.zeros(m), record=recs.append, **kw, **extra)
            st = ctx.kernel_stats(reset=True)
            ctx.profile(False)
            out.append((x, y, [r["norm_res"] for r in recs], st["one_pass"]["launches"]))
        assert out[0][3] == 12 and out[1][3] == 12, (m, n, hk, out[0][3], out[1][3])
        assert np.max(np.abs(out[0][0] - out[1][0])) <= 1e-12 * max(np.max(np.abs(out[0][0])), 1.0), (m, n, hk)
        assert np.max(np.abs(out[0][1] - out[1][1])) <= 1e-12 * max(np.max(np.abs(out[0][1])), 1.0), (m, n, hk)
        assert out[0][2] == pytest.approx(out[1][2], rel=1e-10)
        dA.free()
    comm.close()
    ctx.close()


def test_init_all_and_the_group_entry_point(apx, sharded, generic_ctx):
    """One process, one context per device (here: the one device): ncclCommInitAll + apx_adaptive_primal_dual_group, which
    runs every device's loop on its own host thread."""
    L = importlib.import_module("adaptive-proximal-algorithms_amd._lib")
    api = importlib.import_module("adaptive-proximal-algorithms_amd.api")
    ctx = generic_ctx
    comms = sharded.Comm.init_all([ctx])
    assert len(comms) == 1 and comms[0].info() == (1, 0)
    prob = P.random_lasso(260, 500, pfactor=10, seed=2)
    gam = 1.0 / P.opnorm_sq_power(prob.A, iters=30)
    A = apx.DeviceMatrix.from_numpy(prob.A, ctx=ctx)
    x0 = np.zeros(500)
    xa, _ = apx.adaptive_proxgrad(x0, f=apx.LinearLeastSquares(A, prob.b), g=apx.NormL1(1.0), rule=apx.OurRule(gamma=gam),
                                  tol=0.0, maxit=40, ctx=ctx)
    f = apx.LinearLeastSquares(A, prob.b, allreduce=comms[0])
    pr = api._Problem(f, apx.NormL1(1.0), apx.Zero(), None, np.float64, ctx).finalize(500)
    lib = ctx._lib
    ps = (C.c_void_p * 1)(pr.h)
    rule = apx.OurRule(gamma=gam)._rule()
    x_out = np.empty(500)
    res = L.Result()
    opts = L.SolveOpts()
    L.check(lib.apx_adaptive_primal_dual_group(1, ps, C.byref(rule), x0.ctypes.data_as(C.c_void_p), None, 0.0, 40,
                                               C.byref(opts), x_out.ctypes.data_as(C.c_void_p), None, C.byref(res)))
    assert res.numit == 40 and np.array_equal(x_out, xa)
    # two contexts on ONE device: refused before RCCL is asked (it would refuse too)
    ctx2 = apx.Context(device=0)
    with pytest.raises(apx.ApxError, match="one context per DEVICE"):
        sharded.Comm.init_all([ctx, ctx2])
    pr.close()
    comms[0].close()
    ctx2.close()


def test_comm_argument_errors(apx, sharded, generic_ctx):
    L = importlib.import_module("adaptive-proximal-algorithms_amd._lib")
    ctx = generic_ctx
    lib = ctx._lib
    h = C.c_void_p()
    uid = sharded.Comm.unique_id()
    assert lib.apx_comm_init_rank(ctx._h, 2, 2, C.c_char_p(uid), C.byref(h)) == L.APX_ERR_INVALID      # rank >= nranks
    assert lib.apx_comm_init_rank(None, 1, 0, C.c_char_p(uid), C.byref(h)) == L.APX_ERR_INVALID
    assert lib.apx_comm_unique_id(None) == L.APX_ERR_INVALID
    assert lib.apx_comm_destroy(None) == L.APX_OK
    # a communicator AND a callback on one problem
    comm = sharded.Comm.init_rank(ctx, 1, 0, uid)
    A = apx.DeviceMatrix.from_numpy(np.eye(40), ctx=ctx)
    b = np.zeros(40)
    d = L.ProblemDesc()
    d.f_kind, d.dtype, d.n = L.APX_F_LEAST_SQUARES, L.APX_F64, 40
    d.A_f, d.b = A._h, b.ctypes.data
    d.comm = comm._h
    cb = L.ALLREDUCE_FN(lambda *a: 0)
    d.allreduce = cb
    assert lib.apx_problem_create(ctx._h, C.byref(d), C.byref(h)) == L.APX_ERR_INVALID
    assert b"not both" in lib.apx_last_error()
    # a communicator of another context
    ctx2 = apx.Context(device=0)
    d.allreduce = L.ALLREDUCE_FN()
    d.A_f = apx.DeviceMatrix.from_numpy(np.eye(40), ctx=ctx2)._h
    assert lib.apx_problem_create(ctx2._h, C.byref(d), C.byref(h)) == L.APX_ERR_INVALID
    assert b"another context" in lib.apx_last_error()
    comm.close()
    ctx2.close()


def test_bench_takes_the_in_library_path(apx):
    """bench.py's N-rank code path with the single rank of this box (APX_BENCH_FORCE_DIST=1): the communicator comes from
    apx_comm_init_rank, `rccl_ranks` is what ncclCommCount says, and the run equals the plain one."""
    env = dict(os.environ, APX_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29541")
    args = ["--rows", "3000", "--cols", "12000", "--steps", "20", "--warmup", "3", "--no-cpu-baseline"]
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert len(out.stdout.strip().splitlines()) == 1, "rank 0 prints ONE line on stdout:\n" + out.stdout[:600]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["rccl_ranks"] == 1 and line["collective"]["transport"] == "rccl"
    env.pop("APX_BENCH_FORCE_DIST")
    out2 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True,
                          timeout=600)
    assert out2.returncode == 0, out2.stderr[-2000:]
    plain = json.loads(out2.stdout.strip().splitlines()[-1])
    assert "rccl_ranks" not in plain
    # 3000 x 12000 is on the generic kernels either way (rows too long for the one-pass kernel, too many for the small paths)
    assert line["final_objective"] == plain["final_objective"]


def test_bench_falls_back_when_the_communicator_cannot_be_set_up(apx):
    """If the in-library communicator cannot be set up on any rank, bench.py's N-rank path agrees on the callback transport
    over the launcher's gloo group and says so in its line, instead of dying without a number (APX_BENCH_SIMULATE_RCCL_FAILURE
    is the test hook; the sums are the same, so the iterates are)."""
    env = dict(os.environ, APX_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29543")
    args = ["--rows", "3000", "--cols", "12000", "--steps", "20", "--warmup", "3", "--no-cpu-baseline"]
    ok = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=600)
    assert ok.returncode == 0, ok.stderr[-2000:]
    env["APX_BENCH_SIMULATE_RCCL_FAILURE"] = "1"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "RCCL inside the library is not available" in out.stderr
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["collective"]["transport"] == "gloo-fallback" and line["rccl_ranks"] is None
    assert "FALLBACK" in line["collective"]["what"]
    assert line["final_objective"] == json.loads(ok.stdout.strip().splitlines()[-1])["final_objective"]
