"""Diagnostic (not a test): the fused finalize alone and the whole iteration, per engine build.  usage: ICP_AMD_LIB=... python tools/diag/finalize.py [SIDE NR]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import icp_amd
side, nr = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (256, 1024)
F, M = icp_amd.synth_pair(side)
g = icp_amd.ICP(0); g.init(side * side, nr, 2e2, 1e-6)
g.write(icp_amd.Memory.F, F); g.write(icp_amd.Memory.M, M); g.buildRBC(); g.run_fixed(3); g.sync()
fin = min(g.time_masked(8, 40, 20) for _ in range(3))
sea = min(g.time_masked(1, 40, 20) for _ in range(3))
it = min(g.time_run_fixed(40, 20, True) for _ in range(3)) * 1e3 / 800
print("%-28s finalize %.2f us  search %.2f us  iteration %.2f us  (launches per iteration %d)" % (os.environ.get("ICP_AMD_LIB", "default"), fin, sea, it, g.launches_per_iteration()))
