"""Diagnostic (not a test): the two forms of stage 2 (a query's lanes scan its list / lanes = candidates) over list lengths."""
import sys, os, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if len(sys.argv) > 1:
    import icp_amd
    side, nr, batch = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    m = side * side
    g = icp_amd.ICP(0); g.init(m, nr, 2e2, 1e-6, batch=batch)
    for b in range(batch):
        F, M = icp_amd.synth_pair(side, seed=0x1C9D5EED + b)
        g.write(icp_amd.Memory.F, F, batch_index=b); g.write(icp_amd.Memory.M, M, batch_index=b)
    g.buildRBC(); g.run_fixed(2); g.sync()
    iters, reps = (10, 3) if m * batch >= 1 << 20 else (40, 10)
    us = g.time_run_fixed(iters, reps, True) * 1e3 / (iters * reps)
    print("%s %9.2f" % (g.search_layout(), us))
else:
    for side, nr, batch in ((256, 1024, 1), (256, 512, 1), (256, 256, 1), (512, 2048, 1), (512, 1024, 1), (512, 512, 1), (128, 128, 64), (128, 64, 64), (1024, 8192, 1), (1024, 4096, 1)):
        row = []
        for s2 in ("0", "1"):
            out = subprocess.run([sys.executable, __file__, str(side), str(nr), str(batch)], env=dict(os.environ, ICP_AMD_S2WAVE=s2), capture_output=True, text=True).stdout.strip()
            row.append(out)
        print("side %4d nr %5d batch %2d list %4d : query lanes %s | lanes = candidates %s" % (side, nr, batch, side * side // nr, row[0], row[1]), flush=True)
