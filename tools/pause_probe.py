"""What pauses the dispatch ?  (VERDICT r4 #8.)  The soaks of rounds 3 - 4 found hand-off waits of 0.9 - 1.6 ms with a dozen polls
that MANY workgroup slots on all XCDs see at the same moment, in mid-stream: the whole dispatch stops two to three times per
100 s, for an almost constant time.  This is one instrumented run of the same loop (C5's team kernel, 1.16 ms per launch) that lines
those moments up against what the HOST was doing:

  * the device's wall clock is tied to the host's CLOCK_REALTIME (apx_ctx_device_clock, before and after every segment) ;
  * the run is cut into segments with a Solver of their own (a slot keeps only its LONGEST wait: a fresh Solver per ~6 s makes
    every pause of the run the maximum of some segment) ;
  * three phases: (A) nothing beside the loop ; (B) a thread calling `rocm-smi` every second (what tools/power_watch.py and a
    monitoring agent do) ; (C) a thread reading the amdgpu sysfs files directly every 0.25 s (gpu_metrics, power, clocks) ;
    every such call is logged with its host start / end time ;
  * the main thread's own calls (step, result) are logged the same way.

Output: one JSON line per phase — the pauses found (host time, length, slots / XCDs that saw them), the host activity that
overlaps each (within the clock tie's uncertainty + 2 ms), and the rate of pauses per 100 s in each phase."""
import glob
import json
import os
import subprocess
import sys
import threading
import time

import numpy as np

_ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, _ROOT)
import adaprox_amd as apx  # noqa: E402
from importlib import import_module  # noqa: E402

problems = import_module("adaptive-proximal-algorithms_amd.problems")

SEG_ITERS = int(os.environ.get("SEG_ITERS", "5000"))
SEGS = int(os.environ.get("SEGS", "8"))          # per phase: 8 x 5000 x 1.17 ms = 47 s


def sysfs_files():
    out = []
    for card in sorted(glob.glob("/sys/class/drm/card*/device")):
        for name in ("gpu_metrics", "pp_dpm_sclk", "pp_dpm_mclk", "pp_dpm_fclk", "gpu_busy_percent", "mem_busy_percent"):
            p = os.path.join(card, name)
            if os.path.exists(p):
                out.append(p)
        out += glob.glob(os.path.join(card, "hwmon", "hwmon*", "power1_average")) + glob.glob(os.path.join(card, "hwmon", "hwmon*", "power1_input"))
    return out


class Activity:
    def __init__(self):
        self.log = []                     # (what, t_start_ns, t_end_ns)
        self.lock = threading.Lock()

    def note(self, what, t0, t1):
        with self.lock:
            self.log.append((what, t0, t1))


def sampler(kind, every, act, stop, files):
    while not stop.is_set():
        t0 = time.time_ns()
        if kind == "rocm-smi":
            try:
                subprocess.run(["rocm-smi", "-P", "-c", "-t", "-u", "--json"], capture_output=True, text=True, timeout=10)
            except Exception:      # noqa: BLE001
                pass
            act.note("rocm-smi", t0, time.time_ns())
        else:
            for f in files:
                t0 = time.time_ns()
                try:
                    with open(f, "rb") as fh:
                        fh.read()
                except OSError:
                    pass
                act.note("sysfs:" + os.path.basename(f), t0, time.time_ns())
        stop.wait(every)


def main():
    ctx = apx.Context(0)
    m, n = 50_000, 20_000
    p = problems.device_lad(m, n, seed=0, lam=10.0, ctx=ctx)
    p.A.drop_primary()
    gamma, sigma = 1.0 / p.norm_A, 0.99 / p.norm_A
    files = sysfs_files()
    print(json.dumps({"sysfs_files": files}), flush=True)
    for phase, kind, every in (("A: nothing beside the loop", None, 0), ("B: rocm-smi every 1 s", "rocm-smi", 1.0),
                               ("C: amdgpu sysfs files every 0.25 s", "sysfs", 0.25), ("A again", None, 0)):
        act, stop = Activity(), threading.Event()
        th = None
        if kind is not None:
            th = threading.Thread(target=sampler, args=(kind, every, act, stop, files), daemon=True)
            th.start()
        pauses, ties, seconds = [], [], 0.0
        for seg in range(SEGS):
            s = apx.Solver(np.zeros(n), np.zeros(m), f=apx.Zero(), g=apx.NormL1(p.lam), h=apx.Translate(apx.NormL1(1.0), -p.y), A=p.A,
                           rule=apx.FixedStepsize(gamma, np.sqrt(sigma / gamma)), tol=0.0, ctx=ctx, poll_every=64)
            d0, h0, u0 = ctx.device_clock()
            t0 = time.time_ns()
            s.step(SEG_ITERS)
            s.result()
            t1 = time.time_ns()
            act.note("main: step(%d) + result()" % SEG_ITERS, t0, t1)
            d1, h1, u1 = ctx.device_clock()
            seconds += (t1 - t0) / 1e9
            ties.append((u0, u1, (d1 - d0) - (h1 - h0)))         # the two clocks' drift over the segment
            slots = s.team_wait_slots()
            s.close()
            # pauses: waits of > 300 us — cluster the slots' maxima by device clock
            big = sorted((q for q in slots if q["max_wait_ns"] > 300_000), key=lambda q: q["at_clock_ns"])
            clusters = []
            for q in big:
                if clusters and q["at_clock_ns"] - clusters[-1][-1]["at_clock_ns"] < 2_000_000:
                    clusters[-1].append(q)
                else:
                    clusters.append([q])
            for c in clusters:
                end_dev = max(q["at_clock_ns"] for q in c)
                length = max(q["max_wait_ns"] for q in c)
                # device clock -> host clock, tied at both ends of the segment (linear in between)
                frac = (end_dev - d0) / max(d1 - d0, 1)
                end_host = h0 + frac * (h1 - h0) + (1 - frac) * 0 + 0
                pauses.append({"segment": seg, "end_host_ns": int(end_host), "length_us": length / 1e3, "polls": max(q["max_polls"] for q in c),
                               "slots": len(c), "xcds": len({q["xcc"] for q in c}), "launch": c[0]["at_launch"],
                               "tie_uncertainty_us": (u0 + u1) / 2e3})
        stop.set()
        if th is not None:
            th.join(timeout=15)
        # what the host was doing when each pause began .. ended (+- the tie's uncertainty and 2 ms)
        for q in pauses:
            lo = q["end_host_ns"] - q["length_us"] * 1e3 - 2e6 - q["tie_uncertainty_us"] * 1e3
            hi = q["end_host_ns"] + 2e6 + q["tie_uncertainty_us"] * 1e3
            q["host_activity_overlapping"] = sorted({w for (w, a, b) in act.log if a <= hi and b >= lo and not w.startswith("main:")})
            q["end_host_iso"] = time.strftime("%H:%M:%S", time.gmtime(q["end_host_ns"] / 1e9)) + ".%03d" % ((q["end_host_ns"] // 1_000_000) % 1000)
        busy = {}
        for (w, a, b) in act.log:
            if not w.startswith("main:"):
                busy[w] = busy.get(w, 0) + (b - a) / 1e9
        calls = {}
        for (w, a, b) in act.log:
            calls[w] = calls.get(w, 0) + 1
        print(json.dumps({"phase": phase, "seconds_of_loop": seconds, "segments": SEGS, "iterations_per_segment": SEG_ITERS,
                          "pauses_over_300us": len(pauses), "pauses_per_100s": 100.0 * len(pauses) / max(seconds, 1e-9),
                          "pause_lengths_us": [round(q["length_us"], 1) for q in pauses],
                          "pauses": pauses, "host_calls": calls, "host_seconds_inside_calls": busy,
                          "fraction_of_loop_time_covered_by_sampler_calls": sum(busy.values()) / max(seconds, 1e-9),
                          "clock_tie": {"uncertainty_us_max": max(max(a, b) for a, b, _ in ties) / 1e3,
                                        "drift_us_per_segment_max": max(abs(c) for _, _, c in ties) / 1e3}}), flush=True)
    p.A.free()


if __name__ == "__main__":
    main()
