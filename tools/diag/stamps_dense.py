"""Diagnostic (not a test): per-phase s_memtime stamps of the dense k_search (ICP_DBG_STAMPS build: ICP_AMD_LIB=build/libicp_stamps.so).
usage: python tools/diag/stamps_dense.py SIDE NR [BATCH]"""
import sys, os, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import icp_amd
side, nr = int(sys.argv[1]), int(sys.argv[2])
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 1
g = icp_amd.ICP(0); g.init(side * side, nr, 2e2, 1e-6, batch=batch)
for b in range(batch):
    F, M = icp_amd.synth_pair(side, seed=0x1C9D5EED + b) if not os.environ.get("CASE") else __import__("icp_amd.workloads", fromlist=["x"]).holes_pair(icp_amd, os.environ["CASE"], side, seed=0x1C9D5EED + b)      # CASE: a hole case of workloads.HOLES
    g.write(icp_amd.Memory.F, F, batch_index=b); g.write(icp_amd.Memory.M, M, batch_index=b)
g.buildRBC(); g.run_fixed(int(os.environ.get("ITERS", "5"))); g.sync()
if os.environ.get("FUSED", "1") == "0":
    g.setReduceMode(0)
L = icp_amd.lib(); nb = side * side // 64 * batch
out = np.zeros((nb, 16), np.uint64)
for rep in range(3):
    rc = L.icp_debug_stamps(g._h, out.ctypes.data_as(C.c_void_p), nb); assert rc == 0
t = out.astype(np.int64)[2:]                         # (rows 0 and 1 also hold the finalize kernel's stamps)
GHZ = float(os.environ.get("GHZ", "2.3"))
print("layout", g.search_layout(), "blocks", nb, "(s_memtime = shader clock of the block's own XCD: differences inside a block only; ns at %.2f GHz)" % GHZ)
chained = g.launches_per_iteration() == 1
seq = ([(13, "chain: prologue loads arrived (state, moments, reps, query)"), (10, "chain: moment trees + barrier"), (11, "chain: finish (means / S from the moments)"),
        (12, "chain: power method"), (9, "chain: compose, hand-over of the queries, barrier"), (0, "to the search proper"), (1, "first barrier (reps + queries in LDS)"),
        (2, "stage 1")] if chained else
       [(0, "prologue: loads, transform, hand-over"), (10, "seed bound + tile masks (2 barriers)"), (14, "stage 1 (tiles staged + scanned)"), (2, "the representatives at the origin (vote, staged list, scan)")]) + [(-1, "")]
seq = [x for x in seq if x[0] >= 0] + [
       (3, "nearest representative"), (4, "stage 2 list scan"), (5, "stage 2 reduce"), (6, "hand-off barrier + epilogue wave"), (7, "moment tree + store")]
prev, total = 8, 0.0
for k, name in seq:
    okk = (t[:, k] > 0) & (t[:, prev] > 0) & (t[:, k] >= t[:, prev]) & (t[:, k] - t[:, prev] < 10**7)
    if okk.sum() < len(t) // 2:
        continue
    d = (t[okk, k] - t[okk, prev]) / GHZ
    total += np.median(d)
    print("%-42s median %7.0f ns  p10 %7.0f  p90 %7.0f   (%d blocks)   cumulative %7.0f" % (name, np.median(d), np.percentile(d, 10), np.percentile(d, 90), okk.sum(), total))
    prev = k
okk = (t[:, 7] > t[:, 8]) & (t[:, 8] > 0) & (t[:, 7] - t[:, 8] < 10**7)
d = (t[okk, 7] - t[okk, 8]) / GHZ
print("block lifetime: median %.0f ns  p10 %.0f  p90 %.0f" % (np.median(d), np.percentile(d, 10), np.percentile(d, 90)))
print("masked search %.2f us per launch in this build (boundary included)" % g.time_masked(1, 40, 10))
