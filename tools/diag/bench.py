"""Diagnostic (not a test): steady-state per-iteration time through the same path bench.py uses."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import icp_amd
F, M = icp_amd.synth_pair(128)
g = icp_amd.ICP(0); g.init(16384, 256, 2e2, 1e-6); g.setPowerMode(1); g.setReduceMode(int(os.environ.get("FUSED", "1")))
g.write(icp_amd.Memory.F, F); g.write(icp_amd.Memory.M, M); g.buildRBC()
for rep in range(4):
    ms = g.time_run_fixed(40, 50, True)
    st = g.state()
    print("run_fixed graph: %.2f us/iter   k=%d power_iters=%d" % (ms * 1e3 / 2000, st.k, st.power_iterations))
print("masked all    : %.2f us/iter" % g.time_masked(15, 40, 50))
print("masked search : %.2f us/iter" % g.time_masked(1, 40, 50))
print("masked final  : %.2f us/iter" % g.time_masked(8, 40, 50))
