"""Diagnostic (not a test): steady-state RBC construction time (cached graph), config A / B / C."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import icp_amd
for name, side, nr in (("A", 128, 256), ("B", 256, 1024), ("C", 1024, 4096)):
    F, M = icp_amd.synth_pair(side)
    g = icp_amd.ICP(0); g.init(side * side, nr, 2e2, 1e-6)
    g.write(icp_amd.Memory.F, F); g.write(icp_amd.Memory.M, M)
    g.buildRBC(); g.sync()
    t0 = time.perf_counter()
    for _ in range(50): g.buildRBC()
    g.sync()
    print("config %s: buildRBC %.1f us" % (name, (time.perf_counter() - t0) / 50 * 1e6))
    g.close()
