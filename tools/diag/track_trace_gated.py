"""Reads a rocprofv3 kernel-trace CSV of tools/diag/track_trace.py (frames gated on the device, two streams): the device's time per tracked
frame split into search launches (live / left at once), the RBC construction, the gate, other kernels and idle time."""
import csv, sys
import numpy as np
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda x: x[0])
def kind(n):
    if "k_gate" in n: return "gate"
    if "k_place_lists" in n or "k_list_boxes" in n: return "build"
    if "k_search" in n and "2, 16, true" in n: return "build"          # the owner search
    if "k_search" in n and "true, true, 2, 16" in n: return "search"
    if "k_get_lms" in n: return "lms"
    if "k_set_T" in n: return "setT"
    if "k_chain_end" in n: return "end"
    return "other"
# frames: from one gate kernel's END to the next gate kernel's end (the gate opens = the previous registration is over)
gates = [e for e in ev if kind(e[2]) == "gate"]
print("kernels", len(ev), "gates", len(gates))
per = []
for a, b in zip(gates[8:-1], gates[9:]):
    t0, t1 = a[1], b[1]
    inside = [e for e in ev if e[0] >= t0 and e[0] < t1]
    acc = {}
    for s, e, n in inside:
        k = kind(n)
        d = (min(e, t1) - s) / 1e3
        if k == "search": k = "search_live" if (e - s) > 4500 else "search_dead"
        acc[k] = acc.get(k, 0.0) + d
        acc["n_" + k] = acc.get("n_" + k, 0) + 1
    # union busy time
    busy, end = 0.0, t0
    for s, e, n in inside:
        s2, e2 = max(s, end), min(e, t1)
        if e2 > s2: busy += e2 - s2; end = e2
    acc["period"] = (t1 - t0) / 1e3; acc["idle"] = acc["period"] - busy / 1e3
    per.append(acc)
keys = ["period", "search_live", "n_search_live", "search_dead", "n_search_dead", "build", "gate", "lms", "setT", "end", "other", "idle"]
print("per frame, mean over %d frames [us or count]:" % len(per))
for k in keys:
    v = [p.get(k, 0.0) for p in per]
    print("  %-14s mean %8.2f  p50 %8.2f  p99 %8.2f" % (k, np.mean(v), np.percentile(v, 50), np.percentile(v, 99)))
