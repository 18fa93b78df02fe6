"""The two products of the large sparse case (1 000 000 x 100 000, ~60 entries per row, Float64 and Float32) per KERNEL — HIP-event
times of `X w` and `X' r` from the library's own statistics — with the round-5 forms on and off in one process
(APX_OPT_SPARSE_WINDOWS: L2 windows of the gathered operand, three strides per group).  One JSON line per dtype and setting ;
profiles/r05_sparse_kernels_windows_ab.jsonl.  `python tools/sparse_kernels.py` on an MI355X.  SKEW=3: the column of an entry is
n * u^3 instead of n * u (u uniform): 43 % of the entries in the first 8 % of the columns — the head of a power-law X'."""
import os, sys, json, time, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import adaprox_amd as apx
from importlib import import_module
L=import_module("adaptive-proximal-algorithms_amd._lib")
from sparse_bench import rand_csr
for dtype in (np.float64, np.float32):
    A=rand_csr(1_000_000,100_000,60,0,dtype)
    skew=float(os.environ.get('SKEW','0'))
    if skew:
        import scipy.sparse as sp
        A=sp.csr_matrix((A.data,np.minimum((100_000*np.random.default_rng(5).uniform(size=A.nnz)**skew).astype(np.int32),99_999),A.indptr),shape=A.shape); A.sum_duplicates()
    x=np.random.default_rng(1).standard_normal(100_000).astype(dtype); r=np.random.default_rng(2).standard_normal(1_000_000).astype(dtype)
    for win in [int(v) for v in os.environ.get('WINS', '1,0').split(',')]:
        ctx=apx.Context(0); ctx.set_option(L.APX_OPT_SPARSE_WINDOWS,win)
        dA=apx.DeviceMatrix.from_sparse(A,ctx)
        ctx.profile(True)
        for _ in range(3): dA@x; dA.T@r
        ctx.kernel_stats(reset=True)
        for _ in range(10): dA@x; dA.T@r
        st=ctx.kernel_stats(reset=True)
        print(json.dumps({"dtype":np.dtype(dtype).name,"windows":win,"skew":skew,"nnz":int(A.nnz),"A@x":{"kernel":ctx.kernel_name("gemv_n"),"us":st["gemv_n"]["total_ms"]/st["gemv_n"]["launches"]*1e3,"GBps":st["gemv_n"]["bytes"]/st["gemv_n"]["total_ms"]/1e6},
                          "A.T@r":{"kernel":ctx.kernel_name("gemv_t"),"us":st["gemv_t"]["total_ms"]/st["gemv_t"]["launches"]*1e3,"GBps":st["gemv_t"]["bytes"]/st["gemv_t"]["total_ms"]/1e6}}),flush=True)
        dA.free()
