"""Reads a rocprofv3 kernel-trace CSV of tools/diag/track_trace.py: per frame (from one k_get_lms_band to the next on the main timeline)
the device's busy time, idle time between kernels and the largest idle gap."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda x: x[0])
# frames are delimited by the RBC construction's first kernel (the owner search: k_search<..., true (OWNER), ...> with 2, 16)
def short(n):
    return n.split("(")[0][:60]
frames, cur = [], []
for s, e, n in ev:
    if "k_place_lists" in n and cur:
        cur.append((s, e, n)); continue
    if "k_get_lms_band" in n: continue
    if "true, false, 2, 16, true" in n.replace(" ", "").replace(",", ", ") or ("k_search" in n and ", true, 1, 1024" in n):   # owner search starts a frame
        if cur: frames.append(cur)
        cur = []
    cur.append((s, e, n))
if cur: frames.append(cur)
print("frames:", len(frames))
out = []
for f in frames[8:]:
    t0, t1 = f[0][0], f[-1][1]
    busy = sum(e - s for s, e, _ in f)
    gaps = [f[i + 1][0] - f[i][1] for i in range(len(f) - 1)]
    out.append(((t1 - t0) / 1e3, busy / 1e3, len(f), max(gaps) / 1e3 if gaps else 0, max(e - s for s, e, _ in f) / 1e3))
import numpy as np
a = np.array(out)
print("per frame [us]: span mean %.1f p50 %.1f p99 %.1f max %.1f; busy mean %.1f; kernels mean %.1f; largest in-frame gap p50 %.2f p99 %.2f max %.2f; longest kernel p50 %.2f p99 %.2f" %
      (a[:, 0].mean(), np.percentile(a[:, 0], 50), np.percentile(a[:, 0], 99), a[:, 0].max(), a[:, 1].mean(), a[:, 2].mean(),
       np.percentile(a[:, 3], 50), np.percentile(a[:, 3], 99), a[:, 3].max(), np.percentile(a[:, 4], 50), np.percentile(a[:, 4], 99)))
# frame-to-frame period on the device
starts = np.array([f[0][0] for f in frames[8:]]) / 1e3
per = np.diff(starts)
print("frame period on the device [us]: mean %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f" % (per.mean(), np.percentile(per, 50), np.percentile(per, 90), np.percentile(per, 99), per.max()))
idle = per - a[:-1, 0]
print("idle between a frame's last kernel and the next frame's first [us]: mean %.1f p50 %.1f p99 %.1f max %.1f" % (idle.mean(), np.percentile(idle, 50), np.percentile(idle, 99), idle.max()))
worst = np.argsort(per)[-6:]
for w in worst:
    f = frames[8 + w]
    gaps = [(f[i + 1][0] - f[i][1]) / 1e3 for i in range(len(f) - 1)]
    durs = [(e - s) / 1e3 for s, e, _ in f]
    print("frame %d: period %.1f span %.1f kernels %d idle-after %.1f; gaps > 3 us: %s; kernels > 12 us: %s" %
          (w, per[w], a[w, 0], len(f), idle[w], [(i, round(x, 1)) for i, x in enumerate(gaps) if x > 3], [(i, round(x, 1), short(f[i][2])) for i, x in enumerate(durs) if x > 12]))
