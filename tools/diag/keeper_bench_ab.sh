#!/bin/bash
# Diagnostic (not a test): the tracking lines of bench.py itself (a process that has imported torch) with / without the keeper thread, alternating.
for i in 1 2 3; do
    for k in 1 0; do
        ICP_AMD_TRACK_KEEPER=$k python3 bench.py --no-cpu-baseline --steps 20 --warmup 5 > /tmp/kb.txt 2>/dev/null
        python3 - "$k" <<'P'
import json, sys
d = json.load(open("bench_extra.json")); t = d["other_configs"]["track"]
print("keeper", sys.argv[1], "value %.0f" % d["value"], " ".join("%s/%s %d p99 %.2f call %s" % (w[:4], v.replace("pipelined_", ""), x["frames_per_s"], (x.get("gap_over_same_hop") or {}).get("p99", 0), round(x["submit_call_us"]["mean"]) if "submit_call_us" in x else "-") for w in ("cold_start", "warm_start") for v, x in t[w].items()))
P
    done
done
