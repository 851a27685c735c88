"""Diagnostic (not a test): where an iteration of the persistent run spends its cycles (needs an -DICP_DBG_STAMPS build:
    hipcc ... -DICP_DBG_STAMPS -o /tmp/libicp_dbg.so ;  ICP_AMD_LIB=/tmp/libicp_dbg.so python tests/diag_persist_stamps.py)
Per-phase s_memtime totals of wave 0 of every block over a 40-iteration run, printed per iteration (100 MHz ticks)."""
import sys, os, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["ICP_AMD_PERSISTENT"] = "1"
import icp_amd
side, nr = 128, 256
F, M = icp_amd.synth_pair(side)
g = icp_amd.ICP(0); g.init(side * side, nr, 2e2, 1e-6)
g.write(icp_amd.Memory.F, F); g.write(icp_amd.Memory.M, M); g.buildRBC(); g.run_fixed(3); g.sync()
assert g.run_form() == 2
L = icp_amd.lib(); nb = 256
out = np.zeros((nb, 16), np.uint64)
for rep in range(3):
    rc = L.icp_debug_stamps(g._h, out.ctypes.data_as(C.c_void_p), nb); assert rc == 0
t = out.astype(np.float64) / 40.0
names = ["transform + barrier 1", "stage 1", "stage 2", "barrier 2 (hand-off)", "epilogue (wave 0)", "barrier 3", "block tree + publish",
         "gather (polls)", "barrier 4", "finalize (wave 0)", "barrier 5"]
tot = 0.0
for k, n in enumerate(names):
    print("%-24s mean %8.1f  min %8.1f  max %8.1f ticks per iteration" % (n, t[:, k].mean(), t[:, k].min(), t[:, k].max()))
    tot += t[:, k].mean()
print("sum %.1f ticks per iteration; polls of wave 0 per iteration: mean %.2f max %.2f" % (tot, t[:, 11].mean(), t[:, 11].max()))
us = g.time_run_fixed(40, 20, True) * 1e3 / 800
print("this build: %.2f us per iteration -> %.2f ns per tick" % (us, us * 1e3 / tot))
