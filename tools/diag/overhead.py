"""Diagnostic (not a test): wall-clock time around K fresh 40-iteration passes against the HIP-event time of the same passes (what the
bench's bracket costs beyond the GPU time), three ways: events around all K passes, events behind the first pass, no events."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import icp_amd
F, M = icp_amd.synth_pair(128)
g = icp_amd.ICP(0); g.init(16384, 256, 2e2, 1e-6)
g.write(icp_amd.Memory.F, F); g.write(icp_amd.Memory.M, M); g.buildRBC()
for _ in range(5): g.run_fixed_fresh(40)
g.sync()
def one(kind, K):
    g.sync(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    if kind == "all":
        ev, n = g.time_run_fixed(40, K, from_identity=True), K
    elif kind == "tail":
        ev, n = g.time_run_fixed_tail(40, K, from_identity=True)
    else:
        for _ in range(K): g.run_fixed_fresh(40)
        ev, n = 0.0, 1
    g.sync(); torch.cuda.synchronize()
    w = time.perf_counter() - t0
    return w * 1e6 / (K * 40), ev * 1e3 / (n * 40)
for rep in range(3):
    for K in (20, 200):
        print("K=%3d  " % K + "   ".join("%s: wall %.3f events %.3f us/it" % ((kind,) + one(kind, K)) for kind in ("all", "tail", "none")))
