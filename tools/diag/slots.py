"""Diagnostic (not a test): 64 registrations of config 4 on ONE GPU as 1, 2 or 4 device slots of the in-library batch API
(slots on the same device = independent handles / streams / host threads: the finalize of one overlaps the search of another)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import icp_amd
from icp_amd import workloads as W
pairs = [W.pair(icp_amd, i) for i in range(64)]
for nslots in (1, 2, 4, 1, 2, 4):
    g = icp_amd.ICPBatch([0] * nslots)
    g.init(64, W.M_POINTS, W.NR, W.A, W.C_)
    for i, (F, M) in enumerate(pairs):
        g.write(i, icp_amd.Memory.F, F); g.write(i, icp_amd.Memory.M, M)
    g.buildRBC()
    g.run_fixed(40); g.run_fixed(40)
    reps = 10
    s = g.time_run_fixed(40, reps)
    print("%d slot(s): %.3f us per registration-iteration (%.1f us per batched iteration)" % (nslots, s / (reps * 40 * 64) * 1e6, s / (reps * 40) * 1e6), flush=True)
    g.close()
