"""Diagnostic (not a test): one pipelined tracking pass (cold start, pageable source) for rocprofv3 --kernel-trace; prints the host-side
gaps so that outlier frames can be matched with the device timeline."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import icp_amd
warm = bool(int(os.environ.get("WARM", "0")))
pinned = bool(int(os.environ.get("PINNED", "0")))     # two frames one step apart left in the engine's pinned frame buffers (as bench.py's pinned line)
hops = 256
frames = [icp_amd.synth_cloud_vga(moved=f) for f in range(5)]
order = [0, 1, 2, 3, 4, 3, 2, 1]
seq = [frames[order[i % len(order)]] for i in range(hops + 8)]
g = icp_amd.ICP(0); g.init(16384, 256, 2e2, 1e-6)
if int(os.environ.get("REG", "0")):                   # the frame buffers registered as DMA sources (round 5)
    for fr in frames: g.track_register(fr)
g.track_pipelined(seq[:8], warm_start=warm); g.sync()
if pinned:
    g.track_reset()
    g.track_staging(0)[...] = frames[1]; g.track_staging(1)[...] = frames[2]
    for i in range(8):
        g.track_submit(i & 1, warm)
        if i >= 1: g.track_collect()
    g.track_collect()
pc = time.perf_counter
stamps, sub = [], []
t0 = pc()
for i in range(hops):
    ts = pc(); g.track_submit((i & 1) if pinned else seq[8 + i], warm); sub.append(pc() - ts)
    if i >= 1: g.track_collect(); stamps.append(pc())
g.track_collect(); stamps.append(pc())
gaps = np.diff(np.array([t0] + stamps)) * 1e6
print("mean %.1f p50 %.1f p99 %.1f max %.1f" % (gaps.mean(), np.percentile(gaps, 50), np.percentile(gaps, 99), gaps.max()))
print("gaps:", " ".join("%.0f" % x for x in gaps))
print("submit:", " ".join("%.0f" % (x * 1e6) for x in sub))
