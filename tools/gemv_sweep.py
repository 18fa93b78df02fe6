"""Sweeps the compiled GEMV variants x grid sizes on the bench matrix (one gpurun call).
Kernel time is measured with the library's HIP events (kernel only, no copies)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import adaprox_amd as apx

m = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
dtype = np.float32 if (len(sys.argv) > 3 and sys.argv[3] == "f32") else np.float64
reps = 6
ctx = apx.Context(0)
apx.set_default_context(ctx)
A = apx.DeviceMatrix.uniform(m, n, seed=0, dtype=dtype, ctx=ctx)
rng = np.random.default_rng(0)
x = rng.standard_normal(n).astype(dtype); r = rng.standard_normal(m).astype(dtype)
nbytes = m * n * A.dtype.itemsize
res = []
def run(kind, fn):
    fn(); ctx.kernel_stats(reset=True); ctx.profile(True)
    for _ in range(reps): fn()
    ctx.profile(False)
    st = ctx.kernel_stats(reset=True)[kind]
    return st["total_ms"] / st["launches"]
for vi, name in enumerate(ctx.gemv_variants(False)):
    for blocks in (2048, 4096, 8192, 16384):
        ctx.set_gemv_config(n_variant=vi, n_blocks=blocks)
        ms = run("gemv_n", lambda: A.matvec(x))
        res.append(("N", name, blocks, ms, nbytes / ms / 1e6))
        print(f"N {name:28s} blocks={blocks:6d} {ms:8.3f} ms {nbytes/ms/1e6:7.0f} GB/s", flush=True)
ctx.set_gemv_config(n_variant=0, n_blocks=4096)
for vi, name in enumerate(ctx.gemv_variants(True)):
    for blocks in (1, 16384, 65536):
        ctx.set_gemv_config(t_variant=vi, t_blocks=blocks)
        ms = run("gemv_t", lambda: A.rmatvec(r))
        res.append(("T", name, blocks, ms, nbytes / ms / 1e6))
        print(f"T {name:28s} blocks={blocks:6d} {ms:8.3f} ms {nbytes/ms/1e6:7.0f} GB/s", flush=True)
json.dump(res, open(f"gpurun_out/gemv_sweep_{m}x{n}_{A.dtype.name}.json", "w"))
best = {k: max((r for r in res if r[0] == k), key=lambda r: r[4]) for k in ("N", "T")}
print("BEST", best)
