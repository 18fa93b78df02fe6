#!/bin/bash
# bench.py under different column-streaming (T) kernel variants of the two-pass iteration; prints value + per-kernel GB/s
# (the variant is an option of the context now — the library reads no environment: bench.py --gemv-t-variant)
for v in "$@"; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-other-configs --two-pass --gemv-t-variant $v --steps 20 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('t_variant', sys.argv[1], round(d['value'],2), {k:(round(x['avg_ms'],3), round(x['GBps'])) for k,x in d['roofline']['per_kernel'].items()})" "$v"
done
