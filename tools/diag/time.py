"""Diagnostic (not a test): marginal cost of each kernel inside a hipGraph chain."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import icp_amd
side, nr = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (128, 256)
F, M = icp_amd.synth_pair(side)
g = icp_amd.ICP(0)
g.init(side * side, nr, 2e2, 1e-6)
g.setPowerMode(1)
fused = int(os.environ.get("FUSED", "0"))
g.setReduceMode(fused)
g.write(icp_amd.Memory.F, F); g.write(icp_amd.Memory.M, M); g.buildRBC(); g.run_fixed(3); g.sync()
names = {1: "search", 2: "means", 4: "sij", 8: "finalize", 16: "nop", 15: "all", 3: "search+means", 6: "means+sij", 12: "sij+fin", 17: "search+nop", 14: "means+sij+fin"}
for mask, n in names.items():
    print("%-14s %7.2f us/iter" % (n, g.time_masked(mask)))
