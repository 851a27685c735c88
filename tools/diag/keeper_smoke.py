import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import icp_amd as engine
clouds = [engine.synth_cloud_vga(moved=f) for f in range(4)]
order = [0, 1, 2, 3, 2, 1, 0]
g = engine.ICP(0)
g.init(16384, 256, 2e2, 1e-6)
print("form", g.track_form(), flush=True)
res = g.track_pipelined([clouds[i] for i in order], warm_start=False, depth=2)
print([r if r is None else r[0] for r in res], flush=True)
g.close()
