"""Diagnostic (not a test): 40-iteration fresh runs as a cached graph against the same launches enqueued one by one (ICP_AMD_RUN_GRAPH=0)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import icp_amd
for side, nr, batch in ((128, 256, 1), (256, 1024, 1), (128, 256, 64)):
    g = icp_amd.ICP(0); g.init(side * side, nr, 2e2, 1e-6, batch=batch)
    for b in range(batch):
        F, M = icp_amd.synth_pair(side, seed=0x1C9D5EED + b)
        g.write(icp_amd.Memory.F, F, batch_index=b); g.write(icp_amd.Memory.M, M, batch_index=b)
    g.buildRBC()
    for _ in range(5): g.run_fixed_fresh(40)
    g.sync()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(50): g.run_fixed_fresh(40)
        g.sync()
        best = min(best, (time.perf_counter() - t0) / 2000 * 1e6)
    print("RUN_GRAPH=%s  (%d, %d) x %d: %.2f us per iteration" % (os.environ.get("ICP_AMD_RUN_GRAPH", "1"), side * side, nr, batch, best))
    g.close()
