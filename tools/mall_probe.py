"""Does the 256 MiB Infinity Cache serve the second pass over a row panel?
Round-robin over K matrices of `mb` MB each: gemv_n(A_k) then gemv_t(A_k).  With K*mb > 256 MiB the
N pass streams from HBM (evicted since its last use) and the T pass re-reads what N just read."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import adaprox_amd as apx

ctx = apx.Context(0)
apx.set_default_context(ctx)
lib = ctx._lib
import ctypes as C
n = 100000
for rows, K in [(128, 1), (128, 4), (128, 8), (256, 4), (64, 8), (192, 6)]:
    mats = [apx.DeviceMatrix.uniform(rows, n, seed=k, ctx=ctx) for k in range(K)]
    x = np.ones(n); r = np.ones(rows)
    # device-resident vectors are not exposed at op level; use the solver-free path: time raw launches through
    # apx_gemv_* (includes small H2D/D2H, excluded by the HIP-event kernel stats)
    for rep in range(2):
        for A in mats:
            A.matvec(x); A.rmatvec(r)
    ctx.kernel_stats(reset=True); ctx.profile(True)
    for rep in range(6):
        for A in mats:
            A.matvec(x); A.rmatvec(r)
    ctx.profile(False)
    st = ctx.kernel_stats(reset=True)
    mb = rows * n * 8 / 1e6
    out = {k: (v['bytes'] / v['launches'] / (v['total_ms'] / v['launches'] * 1e-3) / 1e9, v['total_ms'] / v['launches'] * 1e3) for k, v in st.items()}
    print(f"rows={rows} panel={mb:.0f}MB K={K} total={K*mb:.0f}MB  gemv_n {out['gemv_n'][0]:.0f} GB/s ({out['gemv_n'][1]:.1f} us)  gemv_t {out['gemv_t'][0]:.0f} GB/s ({out['gemv_t'][1]:.1f} us)")
    for A in mats: A.free()
