#!/bin/bash
set -o pipefail
O=gpurun_out/r03l
mkdir -p $O
for i in 1 2 3; do
timeout -k 10 300 python tools/ab.py c3 base "ROW_DEAL=0" --rounds 6 --steps 30 > $O/ab_c3_$i.jsonl 2> $O/ab_c3.err; echo "c3 rc $?"
timeout -k 10 300 python tools/ab.py c5 base "ROW_DEAL=0" --rounds 4 --steps 40 > $O/ab_c5_$i.jsonl 2> $O/ab_c5.err; echo "c5 rc $?"
done
timeout -k 10 300 python tools/ab.py c3f64 base "ROW_DEAL=0" --rounds 4 --steps 30 > $O/ab_c3f64.jsonl 2> $O/ab_c3f64.err; echo "c3f64 rc $?"
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03l/ab_*.jsonl")):
    print(f)
    for l in open(f):
        d=json.loads(l); print("   ", d["variant"], d["pass_ms_median"], d["pass_ms_min"], d["pass_ms_max"], d["TBps_of_bytes_read"], d["kernel"][:40])
PY
