"""Diagnostic (not a test): per-frame completion times of the pipelined tracking passes bench.py times — where the slow first pass
after icp_track_reset comes from."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import icp_amd

hops = 64
frames = [icp_amd.synth_cloud_vga(moved=f) for f in range(5)]
order = [0, 1, 2, 3, 4, 3, 2, 1]
seq = [frames[order[i % len(order)]] for i in range(hops + 8)]


def timed_pipelined(g, fr, warm, pinned=False, depth=2):
    stamps, inflight, sub = [], 0, []
    t0 = time.perf_counter()
    for i, f in enumerate(fr):
        if inflight >= depth:
            g.track_collect(); stamps.append(time.perf_counter() - t0); inflight -= 1
        ts = time.perf_counter()
        if pinned:
            g.track_staging(i & 1)[...] = f
            g.track_submit(i & 1, warm)
        else:
            g.track_submit(f, warm)
        sub.append(time.perf_counter() - ts)
        inflight += 1
    while inflight:
        g.track_collect(); stamps.append(time.perf_counter() - t0); inflight -= 1
    d = np.diff(np.array([0.0] + stamps)) * 1e6
    return d, np.array(sub) * 1e6


for warm in (False, True):
    g = icp_amd.ICP(0); g.init(16384, 256, 2e2, 1e-6)
    for f in seq[:8]: g.track_next(f, warm_start=warm)
    for f in seq[8:24]: g.track_next(f, warm_start=warm)
    g.track_reset()
    g.track_pipelined(seq[:8], warm_start=warm)
    for p in range(3):
        g.sync()
        d, s = timed_pipelined(g, seq[8:], warm)
        print("warm=%d pageable pass %d: mean %.0f us/frame; p50 %.0f p90 %.0f p99 %.0f max %.0f; submit mean %.0f max %.0f" %
              (warm, p, d.mean(), np.percentile(d, 50), np.percentile(d, 90), np.percentile(d, 99), d.max(), s.mean(), s.max()))
        print("   first 24 frame gaps:", " ".join("%.0f" % x for x in d[:24]))
        print("   first 24 submits   :", " ".join("%.0f" % x for x in s[:24]))
    g.close()
