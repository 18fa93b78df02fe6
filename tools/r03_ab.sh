#!/bin/bash
set -o pipefail
O=gpurun_out/r03e
mkdir -p $O
N=130,200,300,450,600,1000,1500,2000,2100,2600,3100,4096,5000,5200,6000,7000,7700,8500,9300,10000,10752,11264
for k in ls log; do
  for mask in 15 7; do
    timeout -k 10 240 python tools/shape_sweep.py f32 $k 4 one $N $mask > $O/sweep_f32_${k}_mask$mask.jsonl 2>> $O/sweep.err; echo "$k $mask rc $?"
  done
done
python3 - <<'PY'
import json
O="gpurun_out/r03e"
for k in ("ls","log"):
    a={json.loads(l)["n"]:json.loads(l) for l in open(f"{O}/sweep_f32_{k}_mask15.jsonl") if l.startswith("{")}
    b={json.loads(l)["n"]:json.loads(l) for l in open(f"{O}/sweep_f32_{k}_mask7.jsonl") if l.startswith("{")}
    for n in sorted(a):
        if n in b: print(k, n, "multi", a[n]["GBps_of_bytes_read"], a[n]["passes_per_iter"], "gen1", b[n]["GBps_of_bytes_read"], b[n]["passes_per_iter"])
PY
