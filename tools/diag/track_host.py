"""Diagnostic (not a test): the host's own time per tracked frame — duration of every icp_track_submit / icp_track_collect call of a warm-started
pass with two frames in flight, frames from registered buffers, by iteration count."""
import os, sys, time, gc
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import icp_amd
frames = [icp_amd.synth_cloud_vga(moved=f) for f in range(5)]
order = [0, 1, 2, 3, 4, 3, 2, 1]
hops = 256
seq = [frames[order[i % len(order)]] for i in range(hops + 8)]
g = icp_amd.ICP(0); g.init(16384, 256, 2e2, 1e-6)
if os.environ.get("REG", "1") == "1":
    for fr in frames: g.track_register(fr)
warm = os.environ.get("WARM", "1") == "1"
g.track_pipelined(seq[:8], warm_start=warm); g.sync()
g.track_reset()                                   # (ICP_AMD_TRACK_PROF: the set-up calls — allocations, the stream probe — are printed and dropped here)
g.track_pipelined(seq[:8], warm_start=warm); g.sync()
pc = time.perf_counter
sub, col, ks, stamps = [], [], [], []
gc.collect(); gc.disable()
t0 = pc()
for i, f in enumerate(seq[8:]):
    a = pc(); g.track_submit(f, warm); b = pc(); sub.append((b - a) * 1e6)
    if i >= 1:
        a = pc(); ks.append(g.track_collect()[0]); b = pc(); col.append((b - a) * 1e6); stamps.append(b)
a = pc(); ks.append(g.track_collect()[0]); col.append((pc() - a) * 1e6); stamps.append(pc())
el = pc() - t0
gc.enable()
sub, col, ks = np.array(sub), np.array(col), np.array(ks)
print("frames/s %.0f; submit us: mean %.1f p50 %.1f p90 %.1f max %.1f; collect us: mean %.1f p50 %.1f" % (hops / el, sub.mean(), np.median(sub), np.percentile(sub, 90), sub.max(), col.mean(), np.median(col)))
# submit(i) enqueues frame i; its k is ks[i]
for k in sorted(set(ks.tolist())):
    m = ks == k
    gaps = np.diff(np.array([t0] + stamps)) * 1e6
    print("k=%2d n=%3d  submit %.1f us  collect-wait %.1f us  completion gap p50 %.0f max %.0f" % (k, m.sum(), sub[m].mean(), col[m].mean(), np.median(gaps[m]), gaps[m].max()))
print("launch stats", g.launch_stats())
g.track_reset()
