"""Invalid points (holes) at config A: list statistics (oracle-free: read from the engine), us per iteration of fixed 40-iteration
fresh runs, blocking run; optional parity against the oracle (ICP_DIAG_PARITY=1)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import icp_amd
from icp_amd import workloads as W

side, nr = W.CONFIGS[os.environ.get("CFG", "A")]
if os.environ.get("SIDE"):                        # any other shape: SIDE=256 NR=256 (long lists: the lanes = candidates form of stage 2)
    side, nr = int(os.environ["SIDE"]), int(os.environ["NR"])
m = side * side
batch = int(os.environ.get("BATCH", "1"))
names = ["clean"] + list(W.HOLES)
if os.environ.get("CASE"):
    names = os.environ["CASE"].split(",")
for name in names:
    g = icp_amd.ICP(0)
    g.init(m, nr, 2e2, 1e-6, batch=batch)
    for b in range(batch):
        F, M = icp_amd.synth_pair(side, seed=W.BASE_SEED + b) if name == "clean" else W.holes_pair(icp_amd, name, side, seed=W.BASE_SEED + b)
        g.write(icp_amd.Memory.F, F, batch_index=b); g.write(icp_amd.Memory.M, M, batch_index=b)
    g.buildRBC()
    N = g.read(icp_amd.Memory.RBC_N)
    it = 40 if m <= 65536 else 10
    g.time_run_fixed(it, 3, from_identity=True)
    ms = min(g.time_run_fixed(it, 20, from_identity=True) for _ in range(3))
    rid = g.read(icp_amd.Memory.RID)
    cand = int(N[rid].astype(np.int64).sum())
    line = "%-18s N.max %6d  cand/iter %7.2fM  %8.2f us/iter" % (name, N.max(), cand / 1e6, ms * 1e3 / 20 / it / batch)
    if batch == 1:
        g.reset_transform(); g.buildRBC(); k = g.run(); line += "  run k=%d" % k
    if os.environ.get("ICP_DIAG_PARITY") and batch == 1:
        from oracle import oracle as O
        o = O.OracleICP(m, nr, 2e2, 1e-6, threads=16, power_fast=True, fused=True)
        o.write_f(F); o.write_m(M); o.build_rbc(); ko = o.run()
        ok = (k == ko and np.array_equal(g.read(icp_amd.Memory.NN_ID)["id"], o.nn_id["id"]) and
              np.array_equal(g.read(icp_amd.Memory.T).view(np.uint32), o.T.view(np.uint32)))
        line += "  parity %s" % ("ok" if ok else "DIFF")
    print(line, flush=True)
    g.close()
