"""Reads a rocprofv3 kernel-trace CSV: every idle gap of the device longer than `thr` ms with the kernels on both sides (where does a run
stall?).  usage: python tools/diag/gaps.py trace.csv [thr_ms]"""
import csv, sys
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 5.0
ev = []
for r in csv.DictReader(open(sys.argv[1])):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:70]))
ev.sort()
t0 = ev[0][0]
print("kernels:", len(ev), "span %.1f ms" % ((ev[-1][1] - t0) / 1e6))
end = ev[0][1]
for i in range(1, len(ev)):
    gap = (ev[i][0] - end) / 1e6
    if gap > thr:
        print("gap %8.2f ms at t = %9.1f ms: after [%s] before [%s]; the 3 next: %s" % (gap, (end - t0) / 1e6, ev[i - 1][2], ev[i][2], [e[2][:30] for e in ev[i + 1:i + 4]]))
    end = max(end, ev[i][1])
