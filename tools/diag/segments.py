"""Diagnostic (not a test): what a checked run could be made of at A — one graph of 40 chained launches, the same 40 launches
enqueued one by one (ICP_AMD_RUN_GRAPH=0), and graphs of S iterations back to back; GPU time per iteration and the host's
enqueue time.  Each form in a process of its own (the switch is read once)."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def child(seg):
    import icp_amd
    g = icp_amd.ICP(0); g.init(16384, 256, 2e2, 1e-6)
    F, M = icp_amd.synth_pair(128)
    g.write(icp_amd.Memory.F, F); g.write(icp_amd.Memory.M, M); g.buildRBC()
    nseg = 40 // seg
    def one_pass():
        g.run_fixed_fresh(seg)
        for _ in range(nseg - 1): g.run_fixed(seg)
    for _ in range(50): one_pass()
    g.sync()
    best, host = 1e9, 1e9
    for rep in range(5):
        t0 = time.perf_counter()
        for _ in range(50): one_pass()
        t1 = time.perf_counter()
        g.sync()
        t2 = time.perf_counter()
        best = min(best, (t2 - t0) / 2000 * 1e6); host = min(host, (t1 - t0) / 2000 * 1e6)
    print("RUN_GRAPH=%s segments of %2d: %.2f us per iteration on the GPU, host enqueue %.2f us per iteration" %
          (os.environ.get("ICP_AMD_RUN_GRAPH", "1"), seg, best, host), flush=True)
    g.close()


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(int(sys.argv[1]))
    else:
        for env, seg in (("1", 40), ("0", 40), ("1", 20), ("1", 8), ("1", 4), ("0", 4), ("1", 2)):
            e = dict(os.environ); e["ICP_AMD_RUN_GRAPH"] = env
            subprocess.run([sys.executable, os.path.abspath(__file__), str(seg)], env=e, check=False)
