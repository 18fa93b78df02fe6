import os, sys, numpy as np, scipy.sparse as sp
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import adaprox_amd as apx
from importlib import import_module
L = import_module("adaptive-proximal-algorithms_amd._lib")
rng = np.random.default_rng(0)
bad = 0
cases = [(300, 200, 0.05), (9000, 20000, 0.002), (20000, 9000, 0.004), (8192, 8192, 0.001), (8193, 16385, 0.0007), (1, 50000, 0.3), (50000, 1, 0.5),
         (17000, 40000, 0.0005), (40000, 17000, 0.00002), (5, 5, 1.0)]
for dtype in (np.float64, np.float32):
    for (m, n, dens) in cases:
        A = sp.random(m, n, density=dens, format="csr", random_state=rng, dtype=np.float64)
        A.data = rng.standard_normal(A.nnz)
        if m > 100 and n > 100:
            # a heavy row, a heavy column, empty rows
            A = A.tolil(); A[3, :] = rng.standard_normal(n); A[:, 7] = rng.standard_normal((m, 1)); A[10:20, :] = 0; A = A.tocsr(); A.eliminate_zeros()
        A = A.astype(dtype)
        x = rng.standard_normal(n).astype(dtype); r = rng.standard_normal(m).astype(dtype)
        ctx = apx.Context(0); ctx.set_option(L.APX_OPT_SPARSE_WINDOWS, 3)
        dA = apx.DeviceMatrix.from_sparse(A, ctx)
        ctx.profile(True)
        y = dA @ x; g = dA.T @ r
        kn, kt = ctx.kernel_name("gemv_n"), ctx.kernel_name("gemv_t")
        y2 = dA @ x; g2 = dA.T @ r
        A64 = A.astype(np.float64)
        yr = A64 @ x.astype(np.float64); gr = A64.T @ r.astype(np.float64)
        sy = np.abs(A64) @ np.abs(x.astype(np.float64)) + 1e-300; sg = np.abs(A64).T @ np.abs(r.astype(np.float64)) + 1e-300
        eps = np.finfo(dtype).eps
        ey = np.max(np.abs(y - yr) / sy) / eps; eg = np.max(np.abs(g - gr) / sg) / eps
        ok = ey < 4 and eg < 4 and np.array_equal(y, y2) and np.array_equal(g, g2) and "tiles" in kn and "tiles" in kt
        bad += not ok
        print(np.dtype(dtype).name, (m, n), A.nnz, "err/eps", round(float(ey), 3), round(float(eg), 3), "OK" if ok else "FAIL", "|", kn[:60], "|", kt[:60], flush=True)
        dA.free()
print("bad", bad)
sys.exit(1 if bad else 0)
