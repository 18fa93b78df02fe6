"""JSONL record sink with the wire format of experiments/logging.jl.

The reference installs `get_logger(path)` (experiments/logging.jl:19-42): every
`@logmsg Record` becomes one JSON line holding the record's keyword arguments, and the
console shows only log-spaced iterations.  The solvers of this package take a
``record=`` callable instead of a Julia logger; ``get_logger(path)`` returns such a
callable writing the same lines, so `plot_convergence` / `plot_residual`
(experiments/lasso/runme.jl:163-189) read the files unchanged:

    {"method": "AdaPGM (Ours)", "it": 3, "gamma": ..., "sigma": ..., "norm_res": ...,
     "objective": ..., "grad_f_evals": 4, "prox_g_evals": null, "prox_h_evals": null,
     "A_evals": null, "At_evals": null, "f_evals": 4}
"""
from __future__ import annotations

import json
import math
import sys
from datetime import datetime
from typing import Callable, Iterable, Optional


def is_logstep(base: int, it: int) -> bool:
    """experiments/logging.jl:13-17."""
    scale = int(math.floor(math.log(it, base) + 1e-12))
    step = base ** scale
    return it % step == 0


def _jsonable(v):
    if isinstance(v, float) and not math.isfinite(v):
        return None                       # JSON.jl writes null for NaN / Inf
    return v


def get_logger(path: str, keys: Optional[Iterable[str]] = None, base: int = 10, console=None) -> Callable[[dict], None]:
    """experiments/logging.jl:19-42 — returns the `record=` callable."""
    keys = None if keys is None else list(keys)
    fh = open(path, "a", buffering=1)
    out = sys.stderr if console is None else console

    def record(rec: dict) -> None:
        row = rec if keys is None else {k: rec[k] for k in keys}
        fh.write(json.dumps({k: _jsonable(v) for k, v in row.items()}) + "\n")      # :21-27
        it = rec.get("it")
        if console is not False and isinstance(it, int) and it >= 1 and is_logstep(base, it):   # :35-38
            print(f"[{datetime.now():%Y-%m-%d %H:%M:%S}] " +
                  " ".join(f"{k}={v}" for k, v in rec.items()), file=out)

    record.close = fh.close
    return record
