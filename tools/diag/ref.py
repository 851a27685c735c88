"""Diagnostic (not a test): reference-order mode timing at config A (us per iteration, fixed 40-iteration graph)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import icp_amd
F, M = icp_amd.synth_pair(128)
g = icp_amd.ICP(0); g.init(16384, 256, 2e2, 1e-6); g.setPowerMode(1); g.setReduceMode(0)
g.write(icp_amd.Memory.F, F); g.write(icp_amd.Memory.M, M); g.buildRBC(); g.run_fixed(2); g.sync()
for _ in range(3):
    print("reference-order, squared start: %.2f us/iter" % (g.time_run_fixed(40, 20, True) * 1e3 / 800))
print({n: round(g.time_masked(mk, 40, 20), 2) for n, mk in (("search", 1), ("means", 2), ("sij", 4), ("finalize", 8))})
