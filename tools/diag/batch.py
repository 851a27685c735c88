"""Diagnostic (not a test): batched independent registrations on one GPU (BASELINE config 4, per-GPU share)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import icp_amd
side, nr = 128, 256
pairs = [icp_amd.synth_pair(side, seed=0x1C9D5EED + i) for i in range(8)]
for B in (1, 2, 8, 64):
    g = icp_amd.ICP(0); g.init(side * side, nr, 2e2, 1e-6, batch=B); g.setPowerMode(1); g.setReduceMode(1)
    for b in range(B):
        F, M = pairs[b % 8]
        g.write(icp_amd.Memory.F, F, batch_index=b); g.write(icp_amd.Memory.M, M, batch_index=b)
    g.buildRBC(); g.run_fixed(2); g.sync()
    ms = g.time_run_fixed(40, 10, True)
    per_iter = ms * 1e3 / (40 * 10)
    print("batch %3d: %8.2f us per batched iteration = %6.2f us per registration-iteration  -> %9.0f iterations/s aggregate"
          % (B, per_iter, per_iter / B, B / per_iter * 1e6))
    g.close()
