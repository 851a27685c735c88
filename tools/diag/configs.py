"""Diagnostic (not a test): per-iteration time and kernel split at the BASELINE configs A, B, C."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import icp_amd
for name, side, nr, iters, reps in (("A", 128, 256, 40, 20), ("B", 256, 1024, 40, 10), ("C", 1024, 4096, 10, 2)):
    m = side * side
    F, M = icp_amd.synth_pair(side)
    for fused in (1, 0):
        g = icp_amd.ICP(0); g.init(m, nr, 2e2, 1e-6); g.setPowerMode(1); g.setReduceMode(fused)
        g.write(icp_amd.Memory.F, F); g.write(icp_amd.Memory.M, M)
        g.buildRBC(); g.sync()                      # first call: graph capture + instantiation
        t0 = time.perf_counter()
        for _ in range(10): g.buildRBC()
        g.sync(); tb = (time.perf_counter() - t0) / 10
        g.run_fixed(2); g.sync()
        us = g.time_run_fixed(iters, reps, True) * 1e3 / (iters * reps)
        ks = g.time_masked(1, iters, reps)
        algo = 72 * m + 32 * nr + 64
        print("config %s m=%d nr=%d %-9s: %9.2f us/iter  search %9.2f us  (algorithmic %.2f MB -> %.1f GB/s; 18*m*(nr+m/nr) flop -> %.2f TFLOP/s)  buildRBC %.2f ms"
              % (name, m, nr, "fused" if fused else "reference", us, ks, algo / 1e6, algo / ks / 1e3, 18.0 * m * (nr + m / nr) / ks / 1e6, tb * 1e3))
        g.close()
