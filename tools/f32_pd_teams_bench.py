"""Float32 teams in the primal-dual form (round 5): a 16 GB Float32 LAD-type problem (Condat-Vu steps) on the one-pass team kernels against
the two GEMV passes, three row lengths.  One JSON line each ; profiles/r05_f32_pd_teams.jsonl.  `python tools/f32_pd_teams_bench.py`."""
import os, sys, json, time, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import adaprox_amd as apx
for (m,n) in ((200000,20000),(60000,60000),(40000,100000)):
    out={}
    for mode in (1,0):
        ctx=apx.Context(0); ctx.one_pass(mode)
        A=apx.DeviceMatrix.uniform(m,n,seed=1,dtype=np.float32,ctx=ctx,layout="rows")
        A.fill_column(n-1,1.0)
        rng=np.random.default_rng(0); yv=rng.laplace(size=m).astype(np.float32)
        nA=float(np.sqrt(m*n/3.0))
        s=apx.Solver(np.zeros(n,dtype=np.float32),np.zeros(m,dtype=np.float32),f=apx.Zero(),g=apx.NormL1(5.0),h=apx.Translate(apx.NormL1(1.0),-yv),A=A,rule=apx.FixedStepsize(1/nA,0.99),tol=0.0,ctx=ctx,poll_every=1<<30)
        s.step(10,sync=True); ctx.profile(True); ctx.kernel_stats(reset=True)
        t0=time.perf_counter(); s.step(100,sync=True); dt=time.perf_counter()-t0
        st=ctx.kernel_stats(reset=True)
        out[mode]=dict(ms_per_iter=dt*10, kernel=ctx.kernel_name("one_pass") if mode else "two GEMV passes", one_pass=st["one_pass"]["launches"], GBps=(st["one_pass"]["bytes"]/st["one_pass"]["total_ms"]/1e6 if st["one_pass"]["launches"] else None))
        s.close(); A.free()
    print(json.dumps({"shape":[m,n],"one_pass":out[1],"two_pass":out[0]}),flush=True)
