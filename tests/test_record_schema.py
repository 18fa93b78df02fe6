"""The JSONL wire format of experiments/logging.jl:19-42 and the record keys of
src/AdaProx.jl:56 / :351."""
import importlib
import io
import json

import numpy as np

from oracle import adaprox_oracle as O
from oracle import problems as P

PD_KEYS = ["method", "it", "gamma", "sigma", "norm_res", "objective", "grad_f_evals", "prox_g_evals",
           "prox_h_evals", "A_evals", "At_evals", "f_evals"]                 # src/AdaProx.jl:351
BT_KEYS = ["method", "it", "gamma", "norm_res", "objective", "grad_f_evals", "prox_g_evals", "f_evals"]   # :56


def test_jsonl_sink_matches_reference_wire_format(tmp_path):
    lg = importlib.import_module("adaptive-proximal-algorithms_amd.logging")
    prob = P.random_lasso(40, 80, seed=1)
    gam = 1.0 / P.opnorm_sq_power(prob.A, iters=30)
    path = tmp_path / "lasso.jsonl"
    sink = lg.get_logger(str(path), console=io.StringIO())
    O.adaptive_proxgrad(np.zeros(80), f=O.Counting(O.LinearLeastSquares(prob.A, prob.b)), g=O.NormL1(1.0),
                        rule=O.OurRule(gamma=gam), tol=0.0, maxit=12, name="AdaPGM (Ours)", record=sink)
    O.backtracking_proxgrad(np.zeros(80), f=O.Counting(O.LinearLeastSquares(prob.A, prob.b)), g=O.NormL1(1.0),
                            gamma0=gam, tol=0.0, maxit=5, name="PGM (backtracking)", record=sink)
    sink.close()
    rows = [json.loads(l) for l in open(path)]
    assert len(rows) == 17
    assert list(rows[0]) == PD_KEYS and list(rows[-1]) == BT_KEYS
    # A = 0 and an unwrapped g: the counts are `nothing` -> JSON null (Appendix A.7)
    assert rows[0]["A_evals"] is None and rows[0]["prox_g_evals"] is None and rows[0]["f_evals"] == 2
    assert [r["it"] for r in rows[:12]] == list(range(1, 13))
    # keys subset (experiments/logging.jl:24-26)
    sub = lg.get_logger(str(tmp_path / "sub.jsonl"), keys=["it", "objective"], console=False)
    sub(dict(rows[3]))
    sub.close()
    assert list(json.loads(open(tmp_path / "sub.jsonl").read())) == ["it", "objective"]


def test_is_logstep():
    lg = importlib.import_module("adaptive-proximal-algorithms_amd.logging")
    shown = [it for it in range(1, 1001) if lg.is_logstep(10, it)]
    assert shown == list(range(1, 10)) + list(range(10, 100, 10)) + list(range(100, 1000, 100)) + [1000]
