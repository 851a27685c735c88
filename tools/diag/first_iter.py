"""Diagnostic (not a test): what a registration's FIRST search costs (seeded from the queries' own grid cells) against the later ones, with and
without invalid points: fresh passes of 1, 2 and 10 iterations."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import icp_amd
from icp_amd import workloads as W
side, nr = W.CONFIGS[os.environ.get("CFG", "C")]
m = side * side
for name in ["clean", "scattered10", "blobs30"]:
    F, M = icp_amd.synth_pair(side) if name == "clean" else W.holes_pair(icp_amd, name, side)
    g = icp_amd.ICP(0); g.init(m, nr, 2e2, 1e-6)
    g.write(icp_amd.Memory.F, F); g.write(icp_amd.Memory.M, M); g.buildRBC()
    out = []
    for it in (1, 2, 10):
        g.time_run_fixed(it, 2, True)
        out.append(min(g.time_run_fixed(it, 5, True) for _ in range(3)) * 1e3 / 5)
    print("%-12s pass of 1: %8.1f us   of 2: %8.1f us   of 10: %8.1f us   -> first %.1f, second %.1f, later %.1f us per iteration" % (
        name, out[0], out[1], out[2], out[0], out[1] - out[0], (out[2] - out[1]) / 8))
    g.close()
