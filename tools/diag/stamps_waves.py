"""Diagnostic (not a test): per-WAVE s_memtime stamps of the dense k_search (a -DICP_DBG_STAMPS -DICP_DBG_STAMPS_WAVES build:
ICP_AMD_LIB=build/libicp_stampsw.so): for every block the timeline of each of its waves — which wave the block's barriers wait for, and
in which phase that wave was.  usage: [CASE=scattered10] python tools/diag/stamps_waves.py SIDE NR [BATCH]"""
import sys, os, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import icp_amd
side, nr = int(sys.argv[1]), int(sys.argv[2])
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 1
W = 8                                               # waves per block of the dense variants
g = icp_amd.ICP(0); g.init(side * side, nr, 2e2, 1e-6, batch=batch)
inval = []
for b in range(batch):
    F, M = icp_amd.synth_pair(side, seed=0x1C9D5EED + b) if not os.environ.get("CASE") else __import__("icp_amd.workloads", fromlist=["x"]).holes_pair(icp_amd, os.environ["CASE"], side, seed=0x1C9D5EED + b)
    g.write(icp_amd.Memory.F, F, batch_index=b); g.write(icp_amd.Memory.M, M, batch_index=b)
g.buildRBC(); g.run_fixed(int(os.environ.get("ITERS", "5"))); g.sync()
L = icp_amd.lib(); nb = side * side // 64 * batch
out = np.zeros((nb * W, 16), np.uint64)
for rep in range(3):
    rc = L.icp_debug_stamps(g._h, out.ctypes.data_as(C.c_void_p), nb * W); assert rc == 0
t = out.astype(np.int64).reshape(nb, W, 16)[4:]
GHZ = float(os.environ.get("GHZ", "2.3"))
names = {8: "start", 0: "prologue end", 10: "seed+masks end", 14: "stage 1 tiles end", 2: "origin section end", 3: "nearest rep", 5: "stage 2 end", 6: "epilogue wave end", 7: "end"}
order = [k for k in (8, 0, 10, 14, 2, 3, 5, 6, 7) if (t[:, :, k] > 0).mean() > 0.5]
print("layout", g.search_layout(), "blocks", nb, "stamps present:", [names[k] for k in order])
t0 = t[:, :, 8].min(axis=1, keepdims=True)          # the block's first wave start
for k in order[1:]:
    v = (t[:, :, k] - t0) / GHZ                      # ns since the block began, per wave
    ok = (t[:, :, k] > 0).all(axis=1)
    v = v[ok]
    last = v.argmax(axis=1)                          # the wave that reaches this stamp last
    spread = v.max(axis=1) - v.min(axis=1)
    print("%-20s at (median over blocks) first wave %6.0f  median wave %6.0f  last wave %6.0f ns   spread %5.0f   last wave is wave 7 in %4.1f %% / wave 0 in %4.1f %% of the blocks" % (
        names[k], np.median(v.min(axis=1)), np.median(np.median(v, axis=1)), np.median(v.max(axis=1)), np.median(spread), 100.0 * (last == W - 1).mean(), 100.0 * (last == 0).mean()))
# per phase: the time the LAST wave (of the phase's end stamp) spent in it against the median wave
for a, b in zip(order[:-1], order[1:]):
    ok = (t[:, :, a] > 0).all(axis=1) & (t[:, :, b] > 0).all(axis=1)
    d = (t[ok][:, :, b] - t[ok][:, :, a]) / GHZ
    print("phase %-18s -> %-18s  median wave %6.0f ns   slowest wave of the block %6.0f   wave 7: %6.0f   wave 0: %6.0f" % (names[a], names[b], np.median(d), np.median(d.max(axis=1)), np.median(d[:, W - 1]), np.median(d[:, 0])))
