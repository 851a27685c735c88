import json, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench, icp_amd
g = icp_amd.ICP(0); g.init(16384, 256, 2e2, 1e-6); print("track_form:", g.track_form()); g.close()
o = bench.measure_tracking(icp_amd, 0)
for n in ("cold_start","warm_start"):
    for v,r in o[n].items():
        print(n,v,"%.0f f/s"%r["frames_per_s"],"gap",{k:round(x) for k,x in r["completion_gap_us"].items()},"lat p50 %d"%r["latency_us"]["p50"],"same-k p99",round(r["gap_over_same_k"]["p99"],2),"hop p99",round(r["gap_over_same_hop"]["p99"],2) if r.get("gap_over_same_hop") else None,"submit call us",{k:round(x) for k,x in r["submit_call_us"].items()} if "submit_call_us" in r else None,"above",r["frames_above_1.25x_median_gap"],{k:round(x) for k,x in r["host_launch_calls"].items()})
