"""Soak of the team kernels: tens of thousands of launches of k_team_rows (C2's shape) and k_team_rows in its primal-dual form (C5's) in one
process — sequence numbers, ring re-use and the per-XCD tickets run far past their first wrap of the 8-slot rings; any
timed-out hand-off raises (the error word is read with every poll and with the result).  Prints one JSON line per leg."""
import json
import os
import sys
import time

import numpy as np

_ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")      # (run from anywhere: tools/prof_round.sh does `cd /tmp` first)
sys.path.insert(0, _ROOT)
import adaprox_amd as apx  # noqa: E402

apx.load()
from importlib import import_module  # noqa: E402

problems = import_module("adaptive-proximal-algorithms_amd.problems")


THRESHOLDS = json.load(open(os.path.join(_ROOT, "profiles", "incidents", "wait_thresholds.json")))


def where_the_long_waits_are(slots_by_solve, nsteps_hint=None) -> dict:
    """Every workgroup slot remembers where and when it saw its longest hand-off wait (apx_problem_team_wait_slots).  Two pictures
    are possible behind a millisecond-long wait with a dozen polls: the whole QUEUE was not running (preempted / the chip
    stalled) — then every wave that was polling at that moment has its maximum there: many slots, many teams, all XCDs, one
    clock value ; or ONE workgroup was late — then only its partners (P - 1 slots of one team) do.  Also: at a launch's first
    rows (workgroups starting at different times) or in mid-stream."""
    allrec = [(k, r) for k, recs in enumerate(slots_by_solve) for r in recs]
    if not allrec:
        return {"slots_that_waited": 0}
    top = sorted(allrec, key=lambda kr: -kr[1]["max_wait_ns"])[:6]
    out = {"slots_that_waited": len(allrec), "waits_over_100us": sum(r["long_waits"] for _, r in allrec),
           "slot_maxima_at_first_4_rows": sum(1 for _, r in allrec if r["at_step"] < 4),
           "slot_maxima_over_100us": sum(1 for _, r in allrec if r["max_wait_ns"] > 100_000), "longest": []}
    for k, r in top:
        # the other slots of the same solve whose own maximum ended within the span of this wait (same launch)
        span = max(r["max_wait_ns"], 20_000)
        near = [q for q in slots_by_solve[k] if q["at_launch"] == r["at_launch"] and abs(q["at_clock_ns"] - r["at_clock_ns"]) <= span
                and q["max_wait_ns"] >= 0.25 * r["max_wait_ns"]]
        out["longest"].append({"solve": k, "wait_us": r["max_wait_ns"] / 1e3, "polls": r["max_polls"], "launch": r["at_launch"],
                               "row": r["at_step"], "team": r["team"], "member": r["member"], "xcc": r["xcc"], "block": r["block"],
                               "slots_with_their_maximum_in_the_same_gap": len(near),
                               "teams_among_them": len({q["team"] for q in near}), "xcds_among_them": len({q["xcc"] for q in near})})
    return out


def flag(leg: str, waits: dict) -> dict:
    """the regression threshold: the longest waits the soaks of round 3 / 4 saw are on record in
    profiles/incidents/wait_thresholds.json ; a run that comes near the OLD 20 ms bound says so."""
    t = THRESHOLDS
    ms = waits["max_wait_ns"] / 1e6
    verdict = "ok" if ms <= t["flag_at_ms"] else ("FLAG: beyond %.1f ms" % t["flag_at_ms"])
    return {"leg": leg, "longest_wait_ms": ms, "most_polls": waits["max_polls"], "on_record_ms": t["on_record"][leg]["max_wait_ms"],
            "on_record_polls": t["on_record"][leg]["max_polls"], "flag_at_ms": t["flag_at_ms"], "verdict": verdict}


def leg_c2(iters):
    ctx = apx.Context(0)
    p = problems.device_lasso(50_000, 100_000, pfactor=10, seed=0, power_iters=30, ctx=ctx, layout="rows")      # (bench.py's default: one resident copy)
    f = apx.LinearLeastSquares(p.A, p.b)
    t0 = time.perf_counter()
    obj, slots = [], []
    # (with tol = 0 the loop runs past its fixed point, where the reference's rule lets gamma grow without bound — NaN in the
    #  oracle too: so ten solves of 2000 iterations from x0 = 0 instead of one of 20 000 ; each must end at the same objective
    #  bit for bit, the team kernels being deterministic)
    for rep in range(iters // 2000):
        s = apx.Solver(np.zeros(100_000), f=f, g=apx.NormL1(p.lam), rule=apx.OurRule(gamma=1.0 / p.Lf), tol=0.0, ctx=ctx, poll_every=64)
        s.step(2000)
        x, _, res = s.result()
        assert np.all(np.isfinite(x)) and int(res.numit) == 2000
        obj.append(res.objective)
        w = s.team_waits()            # (every Solver is a problem of its own: summed / maximised over the solves below)
        waits = w if rep == 0 else dict(waits=waits["waits"] + w["waits"], max_polls=max(waits["max_polls"], w["max_polls"]),
                                        max_wait_ns=max(waits["max_wait_ns"], w["max_wait_ns"]),
                                        team_launches=waits["team_launches"] + w["team_launches"])
        slots.append(s.team_wait_slots())
        s.close()
    dt = time.perf_counter() - t0
    assert len(set(obj)) == 1, obj
    print(json.dumps({"leg": "C2 k_team_rows", "solves": len(obj), "iterations_each": 2000, "seconds": dt, "it_per_s": iters / dt,
                      "objective_after_2000": obj[0], "optimum": p.optimum, "identical_objectives": True,
                      # hand-offs that were not complete at the first look, the most polls / the longest any of them took: how
                      # close normal operation comes to the bound of a wait (APX_OPT_TEAM_WAIT_MS: 1000 ms AND 4096 polls)
                      "team_hand_off_waits": waits, "threshold": flag("c2", waits),
                      "where_the_long_waits_are": where_the_long_waits_are(slots)}), flush=True)
    p.A.free()


def leg_c5(iters):
    ctx = apx.Context(0)
    m, n = 50_000, 20_000
    p = problems.device_lad(m, n, seed=0, lam=10.0, ctx=ctx)
    p.A.drop_primary()
    gamma = 1.0 / p.norm_A
    sigma = 0.99 / p.norm_A
    s = apx.Solver(np.zeros(n), np.zeros(m), f=apx.Zero(), g=apx.NormL1(p.lam), h=apx.Translate(apx.NormL1(1.0), -p.y), A=p.A,
                   rule=apx.FixedStepsize(gamma, np.sqrt(sigma / gamma)), tol=0.0, ctx=ctx, poll_every=64)
    t0 = time.perf_counter()
    res_hist = []
    done = 0
    while done < iters:
        s.step(min(5000, iters - done))
        done += min(5000, iters - done)
        x, y, res = s.result()
        res_hist.append((res.objective, res.norm_res))
    dt = time.perf_counter() - t0
    assert np.all(np.isfinite(x)) and np.all(np.isfinite(y))
    print(json.dumps({"leg": "C5 k_team_rows (primal-dual)", "iterations": int(res.numit), "seconds": dt, "it_per_s": iters / dt,
                      "objective_norm_res_every_5000": res_hist, "team_hand_off_waits": s.team_waits(),
                      "threshold": flag("c5", s.team_waits()), "where_the_long_waits_are": where_the_long_waits_are([s.team_wait_slots()])}),
          flush=True)
    s.close()
    p.A.free()


if __name__ == "__main__":
    which = sys.argv[1:] or ["c2", "c5"]
    if "c2" in which:
        leg_c2(int(20000))
    if "c5" in which:
        leg_c5(int(60000))
