"""Diagnostic (not a test): the blocking ICP::run () at A, B and A x 64 — host-driven checked run (depth sweep) against rounds 1 - 3's
one graph of max_iterations launches; all-in wall-clock per run (reset + buildRBC excluded), k, launches, dead launches."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import icp_amd

for side, nr, batch in ((128, 256, 1), (256, 1024, 1), (128, 256, 64)):
    g = icp_amd.ICP(0); g.init(side * side, nr, 2e2, 1e-6, batch=batch)
    for b in range(batch):
        F, M = icp_amd.synth_pair(side, seed=0x1C9D5EED + b)
        g.write(icp_amd.Memory.F, F, batch_index=b); g.write(icp_amd.Memory.M, M, batch_index=b)
    g.buildRBC(); g.sync()
    ref = None
    for depth, adaptive in ((3, False), (1, True), (2, True), (3, True), (4, True), (6, True), (3, False)):
        g.set_run_depth(depth, adaptive)
        times = []
        for rep in range(30):
            g.reset_transform(); g.buildRBC(); g.sync()
            t0 = time.perf_counter()
            k = g.run()
            times.append(time.perf_counter() - t0)
        T = g.read(icp_amd.Memory.T)
        if ref is None: ref = (k, T.copy())
        same = k == ref[0] and np.array_equal(T.view(np.uint32), ref[1].view(np.uint32))
        n, kk, dead = g.run_stats()
        tl = g.run_timeline() if adaptive else None
        t = np.array(times[5:]) * 1e6
        print("(%d, %d) x %d  %s depth %d: k = %d, run () median %.1f us (min %.1f, p90 %.1f) = %.2f us x k; launches %d, dead %d; same bits as the first form: %s" %
              (side * side, nr, batch, "adaptive" if adaptive else "graph-40", depth, k, np.median(t), t.min(), np.percentile(t, 90), np.median(t) / max(k, 1), n, dead, same), flush=True)
        if tl: print("      host timeline of the last run [us]: enqueued %.1f, first word %.1f, decided %.1f, end enqueued %.1f, FINAL %.1f" % tuple(tl[1:]), flush=True)
    g.close()
