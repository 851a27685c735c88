"""Diagnostic (not a test): per-iteration rate at A of a checked run that never converges (thresholds 0, max_iterations 200) against fixed
runs of the same length — as a graph and as plain launches —, with the per-query outputs stored every iteration or not."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import icp_amd

N = 200
g = icp_amd.ICP(0); g.init(16384, 256, 2e2, 1e-6, max_iterations=N, angle_threshold=0.0, translation_threshold=0.0)
F, M = icp_amd.synth_pair(128)
g.write(icp_amd.Memory.F, F); g.write(icp_amd.Memory.M, M); g.buildRBC(); g.sync()
def timed(fn, reps=20):
    ts = []
    for _ in range(reps):
        g.reset_transform(); g.sync()
        t0 = time.perf_counter(); fn(); g.sync(); ts.append(time.perf_counter() - t0)
    return np.median(ts[3:]) * 1e6 / N
print("plain=%s" % os.environ.get("ICP_AMD_RUN_GRAPH", "1"))
print("fixed run of %d                 : %.3f us per iteration" % (N, timed(lambda: g.run_fixed(N))))
for every in (False, True):
    g.set_output_mode(every)
    for depth in (2, 3, 6, 24):
        g.set_run_depth(depth, True)
        print("checked, host-driven, depth %d, outputs %s: %.3f us per iteration" % (depth, "every iteration" if every else "lazy", timed(lambda: g.run())))
    g.set_run_depth(3, False)
    print("checked, one graph of %d, outputs every iteration: %.3f us per iteration" % (N, timed(lambda: g.run())))
