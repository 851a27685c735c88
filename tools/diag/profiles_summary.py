"""Diagnostic (not a test): condenses the rocprofv3 output of tools/diag/profiles.sh into one JSON: per config, per kernel,
average dispatch duration (kernel trace) and per-dispatch counter means (PMC passes).  HBM bytes as
MI355X_MICROARCH.md §HBM prescribes: FETCH_SIZE / WRITE_SIZE are in KB, and on gfx950 FETCH_SIZE counts half of the bytes
of a wide coalesced stream (doubled here; the raw value is kept beside it)."""
import collections
import csv
import glob
import json
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
out = {}


def short(name):
    name = name.replace("void ", "")
    return name.split("(")[0]


for cfg in ("A", "B", "C", "Ax64"):
    entry = {}
    fs = glob.glob("gpurun_out/%s_%s_kernel_stats.csv" % (tag, cfg))
    if fs:
        st = {}
        for r in csv.DictReader(open(fs[0])):
            st[short(r["Name"])] = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3, "total_ms": float(r["TotalDurationNs"]) / 1e6,
                                   "pct": float(r["Percentage"])}
        entry["kernel_stats"] = st
    ctr = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in glob.glob("gpurun_out/%s_pmc_%s_*" % (tag, cfg)):
        for f in glob.glob(d + "/*/*counter_collection.csv"):
            for r in csv.DictReader(open(f)):
                ctr[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    if ctr:
        cs = {}
        for kn, dd in ctr.items():
            if not ("k_search" in kn or "finalize" in kn or "k_chain" in kn or "k_run" in kn):
                continue
            c = {k: sum(v) / len(v) for k, v in dd.items()}
            c["dispatches"] = max(len(v) for v in dd.values())
            if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
                c["hbm_bytes_per_launch_uncorrected"] = (c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024
                c["hbm_bytes_per_launch"] = (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024
            if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c and c["TCC_HIT_sum"] + c["TCC_MISS_sum"] > 0:
                c["l2_hit_rate"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
            if "SQ_WAIT_ANY" in c and c.get("SQ_WAVE_CYCLES"):
                c["wait_fraction_of_wave_cycles"] = c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]
            cs[kn] = c
        entry["counters_per_dispatch"] = cs
    if entry:
        out[cfg] = entry
json.dump(out, open("gpurun_out/%s_profile_summary.json" % tag, "w"), indent=1)
for cfg, e in out.items():
    print("==", cfg)
    for kn, s in sorted(e.get("kernel_stats", {}).items(), key=lambda kv: -kv[1]["total_ms"])[:6]:
        print("   %-60s calls %6d avg %10.2f us  %5.1f %%" % (kn[:60], s["calls"], s["avg_us"], s["pct"]))
    for kn, c in e.get("counters_per_dispatch", {}).items():
        print("   ", kn[:60], {k: round(v, 1) for k, v in c.items()})
