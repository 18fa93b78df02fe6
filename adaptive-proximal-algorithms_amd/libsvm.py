"""LIBSVM text -> (X, y): host mirror of experiments/libsvm.jl:3-61 (ingest only — parsing
text is host work in the reference too; X goes to the device with DeviceMatrix.from_sparse).

Same behaviour: 1-based feature indices, the matrix is as wide as the largest index seen,
`labels=[l0, l1]` remaps a two-valued label column to (l0, l1) unless it already uses them."""
from __future__ import annotations

import warnings
from typing import Optional, Sequence, Tuple

import numpy as np


def load_libsvm_dataset(file_path: str, dtype=np.float64, labels: Optional[Sequence[float]] = None):
    import scipy.sparse as sp

    if labels is not None:
        assert len(labels) == 2                                   # :15
        assert labels[0] != labels[1]                             # :16
        if labels[0] > labels[1]:                                 # :17-20
            warnings.warn("Labels provided are not in ascending order")
    y, rows, cols, vals = [], [], [], []
    with open(file_path) as fh:
        for row, line in enumerate(fh.read().splitlines()):       # :23
            tokens = line.strip().split(" ")                      # :24
            y.append(float(tokens[0]))                            # :25
            for token in tokens[1:]:                              # :26-31
                if not token:
                    continue
                col_str, val_str = token.split(":")
                rows.append(row)
                cols.append(int(col_str) - 1)
                vals.append(float(val_str))
    y = np.asarray(y, dtype=dtype)
    n = (max(cols) + 1) if cols else 0
    X = sp.csr_matrix((np.asarray(vals, dtype=dtype), (rows, cols)), shape=(len(y), n))   # :34 sparse(rows, cols, vals)
    if labels is not None:                                        # :41-58
        assert len(np.unique(y)) == 2
        y0, y1 = y.min(), y.max()
        l0, l1 = labels
        if not (y0 in labels and y1 in labels):
            lo, hi = y == y0, y == y1
            y[lo] = l0
            y[hi] = l1
        assert all(v in labels for v in np.unique(y))
    return X, y
