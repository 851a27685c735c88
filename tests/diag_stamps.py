"""Diagnostic (not a test): per-phase s_memtime stamps of k_search (needs the ICP_DBG_STAMPS build)."""
import sys, os, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import icp_amd
side, nr = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (128, 256)
F, M = icp_amd.synth_pair(side)
g = icp_amd.ICP(0); g.init(side * side, nr, 2e2, 1e-6); g.setPowerMode(1); g.setReduceMode(int(os.environ.get('FUSED','0')))
g.write(icp_amd.Memory.F, F); g.write(icp_amd.Memory.M, M); g.buildRBC(); g.run_fixed(3); g.sync()
L = icp_amd.lib(); nb = side * side // 64
out = np.zeros((nb, 16), np.uint64)
for rep in range(3):
    rc = L.icp_debug_stamps(g._h, out.ctypes.data_as(C.c_void_p), nb); assert rc == 0
t = out.astype(np.int64)
t0 = t[:, 8].min()
names = ["loads+transform", "stage reps->LDS", "S1 loop", "S1 sync", "S2 loop", "S2 sync", "epilogue", "tree"]
prev = t[:, 8]
print("block start spread (ticks): %d" % (t[:, 8].max() - t0))
for k in range(8):
    d = t[:, k] - prev
    print("%-18s mean %8.0f  min %8d  max %8d" % (names[k], d.mean(), d.min(), d.max()))
    prev = t[:, k]
if t[:, 13].max() > 0:
    d = t[:, 13] - t[:, 3]; print("  %-18s mean %8.0f  min %8d  max %8d" % ("S1 combine+stage lists", d.mean(), d.min(), d.max()))
    d = t[:, 4] - t[:, 13]; print("  %-18s mean %8.0f  min %8d  max %8d" % ("list scan", d.mean(), d.min(), d.max()))
if t[:, 9].max() > 0:
    d = t[:, 9] - t[:, 8]
    print("%-18s mean %8.0f  min %8d  max %8d   (start -> end of the chained prologue)" % ("prologue", d.mean(), d.min(), d.max()))
if t[:, 12].max() > 0:
    for a, b_, n in ((8, 10, "moments->trees"), (10, 11, "finish (f64 divs)"), (11, 12, "power method"), (12, 9, "compose+publish")):
        d = t[:, b_] - t[:, a]
        print("  %-18s mean %8.0f  min %8d  max %8d" % (n, d.mean(), d.min(), d.max()))
if t[0, 15] > 0 and t[0, 14] > 0:
    print("finalize kernel (block 0): fetch+trees %d  finish %d  power method %d  compose+publish %d   total %d ticks"
          % (t[0, 10] - t[0, 14], t[0, 11] - t[0, 10], t[0, 12] - t[0, 11], t[0, 15] - t[0, 12], t[0, 15] - t[0, 14]))
print("kernel span (first start -> last end): %d ticks; per-block mean %0.f" % (t[:, 7].max() - t0, (t[:, 7] - t[:, 8]).mean()))
print("row0 raw:", [int(x) for x in t[0]])
pm = t[1, :7]
if pm[6] > pm[0] > 0:
    print("power method (ticks): squarings %d  x0 %d  loop %d  lambda %d  final normalize %d  Tk %d" % tuple(int(pm[k + 1] - pm[k]) for k in range(6)))
try:
    us = g.time_masked(1, 40, 20)
    good = t[(t[:, 8] > 0) & (t[:, 7] > t[:, 8])]
    span = int(good[:, 7].max() - good[:, 8].min())
    print("calibration: masked search %.2f us per launch in this build (incl. ~1.74 us boundary); k_search span %d ticks -> %.3f ns/tick if boundary excluded"
          % (us, span, (us - 1.74) * 1e3 / span))
except Exception as e:
    print("calibration failed:", e)
