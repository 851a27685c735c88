import json, sys
for f in sys.argv[1:]:
    d=json.load(open(f)); print(f, round(d["value"]), round(d["us_per_iteration"],3), round(d["roofline"]["frac"],5), d["cpu_baseline"]["value"] if "cpu_baseline" in d else None)
    oc=d["other_configs"]
    print({k:(round(v["us_per_iteration"],3) if isinstance(v,dict) and "us_per_iteration" in v else None) for k,v in oc.items()})
    t=oc["track"]
    for n in ("cold_start","warm_start"):
        for v,r in t[n].items(): print(" ", n, v, round(r["frames_per_s"]), {k:round(x,2) for k,x in r.get("gap_over_same_hop",{}).items() if k in ("p99","max")}, r.get("frames_above_1.25x_median_gap"))
    h=oc["holes"]
    for key in ("A_holes","A_x64_holes","B_holes"):
        print(" ", key, {k:round(v.get("us_per_iteration",0),2) for k,v in h[key].items()})
    print(" ", {k:(round(v["us_per_iteration"],2), v["run_k"]) for k,v in h["A_wall"].items() if isinstance(v,dict)}, round(h["track_blobs10"]["frames_per_s"]))
    print(" ref order", d.get("reference_order_us_per_iteration"))
