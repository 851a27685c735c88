"""Diagnostic (not a test): time of the rotation solvers as the iteration's finalize runs them — the reference-order pipeline with the
literal power loop (ICP_AMD_MODE=reference: k_finalize<1> carries the loop) against the default modes, per kernel (icp_time_kernels) and per
iteration."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import icp_amd
F, M = icp_amd.synth_pair(128)
for name, pm, rm in (("reference order + literal", icp_amd.PowerMode.LITERAL, icp_amd.ReduceMode.REFERENCE_ORDER), ("fused + squared", icp_amd.PowerMode.SQUARED, icp_amd.ReduceMode.FUSED),
                     ("fused + literal", icp_amd.PowerMode.LITERAL, icp_amd.ReduceMode.FUSED)):
    g = icp_amd.ICP(0); g.init(16384, 256, 2e2, 1e-6); g.setPowerMode(pm); g.setReduceMode(rm)
    g.write(icp_amd.Memory.F, F); g.write(icp_amd.Memory.M, M); g.buildRBC()
    g.time_run_fixed(40, 3, True)
    ms = min(g.time_run_fixed(40, 20, True) for _ in range(3))
    trips = []
    g.reset_transform()
    for it in range(40):
        g.step(); trips.append(g.state().power_iterations)
    print("%-28s %7.2f us per iteration; power-method trips per iteration: mean %.1f (min %d, max %d)" % (name, ms * 1e3 / 800, sum(trips) / 40.0, min(trips), max(trips)))
    g.close()
