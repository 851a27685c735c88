#!/bin/bash
# Diagnostic (not a test): the search kernel's duration by iteration number within the timed fresh registrations (rocprofv3 --kernel-trace of
# tools/diag/holes.py), hole cases against the clean pair.   usage: CFG=B tools/diag/holes_iter.sh case...
export TMPDIR=/tmp
for c in "$@"; do
    export CASE=$c
    d=gpurun_out/hi_$c; rm -rf $d
    timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $d -- python3 tools/diag/holes.py > $d.log 2>&1 || { echo failed $c; tail -3 $d.log; exit 1; }
    grep "us/iter" $d.log
    python3 - $d <<'P'
import sys, csv, glob, collections
rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the dominant search instantiation (not the owner search)
cnt = collections.Counter(r["Kernel_Name"] for r in rows if r["Kernel_Name"].startswith("void k_search"))
name = cnt.most_common(1)[0][0]
import os
P = int(os.environ.get('PERIOD', '40'))
per, i, others = collections.defaultdict(list), 0, collections.defaultdict(list)
for r in rows:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if r["Kernel_Name"] == name: per[i % P].append(d); i += 1
    else: others[r["Kernel_Name"].split("(")[0][:50]].append(d)
med = lambda v: sorted(v)[len(v) // 2]
import os

print("   search by iteration (median us):", " ".join("%d:%.1f" % (k, med(per[k])) for k in (0, 1, 2, 3, 5, 9, 10, 20, 39) if k < P), " mean over all %.2f" % (sum(sum(v) for v in per.values()) / sum(len(v) for v in per.values())))
for k, v in sorted(others.items(), key=lambda kv: -sum(kv[1]))[:4]: print("   %-50s n %5d median %.2f" % (k, len(v), med(v)))
P
    rm -rf $d $d.log
done
