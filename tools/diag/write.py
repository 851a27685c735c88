import sys, time, os
sys.path.insert(0, '/root/repo')
import icp_amd
F, M = icp_amd.synth_pair(128)
g = icp_amd.ICP(0); g.init(16384, 256, 2e2, 1e-6)
g.write(icp_amd.Memory.F, F); g.write(icp_amd.Memory.M, M); g.buildRBC(); g.run_fixed_fresh(40); g.sync()
for name, fn in (("write F+M only", lambda: (g.write(icp_amd.Memory.F, F), g.write(icp_amd.Memory.M, M))),
                 ("write F+M + build + run", lambda: (g.write(icp_amd.Memory.F, F), g.write(icp_amd.Memory.M, M), g.buildRBC(), g.run_fixed_fresh(40))),
                 ("build + run", lambda: (g.buildRBC(), g.run_fixed_fresh(40)))):
    for rep in range(2):
        g.sync(); t0 = time.perf_counter()
        for _ in range(20): fn()
        g.sync()
        print("%-28s %.1f us per round" % (name, (time.perf_counter() - t0) / 20 * 1e6))
