"""The HIP mirror emits the reference's record schema (src/AdaProx.jl:56, :351) and the JSONL
sink writes what experiments/logging.jl writes — key for key, null for uncounted terms."""
import importlib
import json

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import problems as P
from test_record_schema import BT_KEYS, PD_KEYS


def test_records_and_jsonl_from_the_hip_path(tmp_path):
    import adaprox_amd as apx
    lg = importlib.import_module("adaptive-proximal-algorithms_amd.logging")
    prob = P.random_lasso(40, 80, seed=1)
    gam = 1.0 / P.opnorm_sq_power(prob.A, iters=30)
    path = tmp_path / "run.jsonl"
    sink = lg.get_logger(str(path), console=False)
    f = apx.Counting(apx.LinearLeastSquares(prob.A, prob.b))
    apx.adaptive_proxgrad(np.zeros(80), f=f, g=apx.NormL1(1.0), rule=apx.OurRule(gamma=gam), tol=0.0, maxit=12,
                          name="AdaPGM (Ours)", record=sink)
    apx.backtracking_proxgrad(np.zeros(80), f=f, g=apx.NormL1(1.0), gamma0=gam, tol=0.0, maxit=5,
                              name="PGM (backtracking)", record=sink)
    lad = P.random_lad(50, 20, seed=3, lam=1.0)
    apx.condat_vu(np.zeros(20), np.zeros(50), f=apx.Zero(), g=apx.NormL1(1.0),
                  h=apx.Translate(apx.NormL1(1.0), -lad.y), A=apx.Counting(apx.DeviceMatrix.from_numpy(lad.A)),
                  Lf=0.0, norm_A=lad.norm_A, tol=0.0, maxit=4, record=sink)
    sink.close()
    rows = [json.loads(l) for l in open(path)]
    assert len(rows) == 21
    assert list(rows[0]) == PD_KEYS and list(rows[12]) == BT_KEYS and list(rows[-1]) == PD_KEYS
    assert rows[0]["method"] == "AdaPGM (Ours)" and rows[0]["A_evals"] is None and rows[0]["prox_g_evals"] is None
    assert [r["f_evals"] for r in rows[:12]] == list(range(2, 14))           # it + 1
    assert rows[-1]["method"] == "Condat-Vu" and rows[-1]["A_evals"] == 5 and rows[-1]["At_evals"] == 4
