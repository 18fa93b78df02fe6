#!/bin/bash
set -o pipefail
O=gpurun_out/r03o
mkdir -p $O
N=12000,16000,20000,22528,23000,28000,40000,56000,80000,112000,160000,225280
for k in ls log; do
  timeout -k 10 300 python tools/shape_sweep.py f32 $k 6 one $N > $O/sweep_f32_long_${k}_one.jsonl 2>> $O/sweep.err; echo "$k one rc $?"
  timeout -k 10 300 python tools/shape_sweep.py f32 $k 6 two $N > $O/sweep_f32_long_${k}_two.jsonl 2>> $O/sweep.err; echo "$k two rc $?"
done
python3 - <<'PY'
import json
O="gpurun_out/r03o"
for k in ("ls","log"):
    a={json.loads(l)["n"]:json.loads(l) for l in open(f"{O}/sweep_f32_long_{k}_one.jsonl") if l.startswith("{")}
    b={json.loads(l)["n"]:json.loads(l) for l in open(f"{O}/sweep_f32_long_{k}_two.jsonl") if l.startswith("{")}
    for n in sorted(a):
        if n in b: print(k, n, "one pass:", a[n]["passes_per_iter"], a[n]["ms_per_iter"], "ms", a[n]["GBps_of_bytes_read"], "GB/s   two passes:", b[n]["ms_per_iter"], "ms", b[n]["GBps_of_bytes_read"], "GB/s each")
PY
