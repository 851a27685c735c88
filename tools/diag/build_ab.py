"""Diagnostic (not a test): buildRBC time (cached graph, back to back) at the BASELINE sizes.  usage: ICP_AMD_LIB=... [CASE=blobs30] python tools/diag/build_ab.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import icp_amd
out = []
for side, nr, batch in ((128, 256, 1), (256, 1024, 1), (1024, 4096, 1), (128, 256, 64)):
    m = side * side
    g = icp_amd.ICP(0); g.init(m, nr, 2e2, 1e-6, batch=batch)
    for b in range(batch):
        F, M = icp_amd.synth_pair(side, seed=0x1C9D5EED + b)
        if os.environ.get("CASE"):                       # a hole case of workloads.HOLES (invalid points in both frames)
            from icp_amd import workloads as W
            F, M = W.holes_pair(icp_amd, os.environ["CASE"], side, seed=0x1C9D5EED + b)
        g.write(icp_amd.Memory.F, F, batch_index=b); g.write(icp_amd.Memory.M, M, batch_index=b)
    g.buildRBC(); g.sync()
    n = 5 if m * batch >= 1 << 20 else 50
    t0 = time.perf_counter()
    for _ in range(n): g.buildRBC()
    g.sync()
    out.append("%8.1f" % ((time.perf_counter() - t0) / n * 1e6))
    g.close()
print("buildRBC us (A, B, C, A x 64):", " ".join(out))
