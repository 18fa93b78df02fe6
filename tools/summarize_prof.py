#!/usr/bin/env python3
"""Condenses rocprofv3 CSV output (kernel-trace stats + PMC passes) into the small
summaries committed under profiles/.  Usage:
    python tools/summarize_prof.py <prof_dir> <out_prefix>
<prof_dir> holds sub-directories trace/, pmc_fetch/, pmc_write/ written by rocprofv3
(--output-format csv).  FETCH_SIZE on gfx950 reports exactly half of the bytes of a wide
coalesced streaming read (MI355X_MICROARCH.md §HBM) and is doubled here, as that guide
prescribes; WRITE_SIZE is exact.  Both counters are in KiB."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def find(d, pat):
    return sorted(glob.glob(os.path.join(d, "**", pat), recursive=True))


def short(name):
    # k_team_rows serves f's gradient (FK 1 / 2) and the primal-dual loop (FK 100, csrc/team_rows.h kTeamFkPd): two entries
    if "k_team_rows" in name:
        return "k_team_rows (primal-dual)" if "k_team_rows<100" in name.replace(" ", "") else "k_team_rows"
    for key in ("k_multi_rows", "k_teampd_rows", "k_slab2_rows", "k_slabpd_rows", "k_slabpd_close", "k_slab_rows", "k_slab_close",
                "gemv_n_kernel", "gemv_t_kernel", "gemv_t_finish", "k_finish_rows", "k_f_reduce", "k_reduce_n",
                "k_rule", "k_step", "k_dual", "k_final", "k_fill_uniform", "k_scale_columns", "sum_partials_rows", "spmv_tiles", "spmv_stream", "spmv_rows", "spmv_long", "k_bt_trial"):
        if key in name:
            return key
    return name[:60]


def main():
    prof, out = sys.argv[1], sys.argv[2]
    summary = {}
    # ---- kernel trace: per-kernel durations
    rows = defaultdict(list)
    for f in find(os.path.join(prof, "trace"), "*kernel_trace.csv"):
        for r in csv.DictReader(open(f)):
            rows[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    kt = {}
    total = sum(sum(v) for v in rows.values()) or 1
    for k, v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
        v_sorted = sorted(v)
        kt[k] = {"calls": len(v), "total_ms": sum(v) / 1e6, "avg_us": sum(v) / len(v) / 1e3,
                 "median_us": v_sorted[len(v) // 2] / 1e3, "min_us": v_sorted[0] / 1e3, "max_us": v_sorted[-1] / 1e3,
                 "pct": 100.0 * sum(v) / total}
    summary["kernel_trace"] = kt
    # ---- what lies between two consecutive passes over A: from the end of a row kernel to the start of the next launch of the
    # same kernel (the closing kernel, the reductions, the rule, the step, the launch gaps) — pauses of more than 1 ms (set-up
    # between legs, host read-backs) are not iterations and are left out
    ROWK = ("k_team_rows", "k_team_rows (primal-dual)", "k_multi_rows", "k_slab_rows", "k_slab2_rows", "k_slabpd_rows")
    ev = []
    for f in find(os.path.join(prof, "trace"), "*kernel_trace.csv"):
        for r in csv.DictReader(open(f)):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    ev.sort()
    gaps, between = defaultdict(list), defaultdict(lambda: defaultdict(list))
    last = {}
    for i, (t0, t1, name) in enumerate(ev):
        k = short(name)
        if k not in ROWK:
            continue
        if name in last:
            j, e_prev = last[name]
            g = t0 - e_prev
            if 0 <= g < 1_000_000:
                gaps[name].append(g)
                for (a0, a1, an) in ev[j + 1:i]:
                    between[name][short(an)].append(a1 - a0)
        last[name] = (i, t1)
    gp = {}
    for name, v in gaps.items():
        v.sort()
        gp[name] = {"launches": len(v) + 1, "median_gap_us": v[len(v) // 2] / 1e3, "mean_gap_us": sum(v) / len(v) / 1e3,
                    "kernels_in_the_gap_avg_us": {k: sum(d) / len(d) / 1e3 for k, d in between[name].items()},
                    # (how many of the gaps a kernel appears in: a one-off — an allocation's fill on the first iteration — is not
                    #  part of the iteration)
                    "kernels_in_the_gap_count": {k: len(d) for k, d in between[name].items()}}
    summary["between_consecutive_row_kernels"] = gp
    # ---- PMC passes
    pmc = defaultdict(lambda: defaultdict(list))
    for sub in ("pmc_fetch", "pmc_write"):
        for f in find(os.path.join(prof, sub), "*counter_collection.csv"):
            for r in csv.DictReader(open(f)):
                pmc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    traffic = {}
    for k, cs in pmc.items():
        ent = {}
        if "FETCH_SIZE" in cs:
            v = sorted(cs["FETCH_SIZE"])
            ent["FETCH_SIZE_KiB_median"] = v[len(v) // 2]
            ent["fetch_bytes_corrected"] = 2.0 * 1024.0 * v[len(v) // 2]
        if "WRITE_SIZE" in cs:
            v = sorted(cs["WRITE_SIZE"])
            ent["WRITE_SIZE_KiB_median"] = v[len(v) // 2]
            ent["write_bytes"] = 1024.0 * v[len(v) // 2]
        ent["hbm_bytes_per_launch"] = ent.get("fetch_bytes_corrected", 0.0) + ent.get("write_bytes", 0.0)
        traffic[k] = ent
    summary["pmc"] = traffic
    json.dump(summary, open(out + ".json", "w"), indent=1)
    with open(out + ".md", "w") as fh:
        fh.write("| kernel | calls | avg us | median us | min us | max us | % of GPU time |\n|---|---|---|---|---|---|---|\n")
        for k, e in kt.items():
            fh.write(f"| {k} | {e['calls']} | {e['avg_us']:.1f} | {e['median_us']:.1f} | {e['min_us']:.1f} | "
                     f"{e['max_us']:.1f} | {e['pct']:.2f} |\n")
        fh.write("\n| kernel | FETCH_SIZE KiB (raw, median) | fetch bytes (x2 gfx950 correction) | WRITE_SIZE KiB | "
                 "HBM bytes / launch |\n|---|---|---|---|---|\n")
        for k, e in traffic.items():
            fh.write(f"| {k} | {e.get('FETCH_SIZE_KiB_median', float('nan')):.0f} | "
                     f"{e.get('fetch_bytes_corrected', float('nan')):.4g} | "
                     f"{e.get('WRITE_SIZE_KiB_median', float('nan')):.0f} | {e['hbm_bytes_per_launch']:.4g} |\n")
        fh.write("\n| row kernel | launches | median time to the NEXT launch of it (us) | kernels in between, average us |\n|---|---|---|---|\n")
        for name, e in gp.items():
            fh.write(f"| {name[:150]} | {e['launches']} | {e['median_gap_us']:.1f} | " +
                     ", ".join(f"{k} {u:.1f}" + ("" if e["kernels_in_the_gap_count"][k] >= e["launches"] - 1 else f" (in {e['kernels_in_the_gap_count'][k]} of the gaps)")
                               for k, u in e["kernels_in_the_gap_avg_us"].items()) + " |\n")
    # what bench.py reads for roofline.traffic
    # with the dual layout (bench default) A*x runs the column-streaming kernel on the transposed copy,
    # so both passes appear as gemv_t_kernel in the trace
    pt = {"gemv_n": traffic.get("gemv_n_kernel") or traffic.get("gemv_t_kernel", {}),
          "gemv_t": traffic.get("gemv_t_kernel", {})}
    for k in ("k_team_rows", "k_team_rows (primal-dual)", "k_multi_rows", "k_teampd_rows", "k_slab_rows", "k_slab2_rows",      # the one-pass kernels (A read once per launch)
              "spmv_tiles"):                                                                                                   # the sparse leg's products
        if k in traffic:
            pt[k] = traffic[k]
    if len(sys.argv) > 3:
        pt["source"] = sys.argv[3]
    json.dump(pt, open(os.path.join(os.path.dirname(out), "pmc_traffic.json"), "w"), indent=1)
    print(open(out + ".md").read())


if __name__ == "__main__":
    main()
