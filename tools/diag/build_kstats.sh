#!/bin/bash
# Diagnostic (not a test): buildRBC wall time (tools/diag/build_ab.py) for several engine builds, alternating, and the per-kernel
# durations of the construction under rocprofv3 for the last one.   usage: tools/diag/build_kstats.sh LIB...
export TMPDIR=/tmp
for rep in 1 2; do for l in "$@"; do echo "== $l"; ICP_AMD_LIB=$l timeout -k 10 200 python3 tools/diag/build_ab.py || exit 1; done; done
for l in "$@"; do last=$l; done
export ICP_AMD_LIB=$last
d=gpurun_out/bk; rm -rf $d
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 tools/diag/build_ab.py > $d.log 2>&1 || { tail -3 $d.log; exit 1; }
python3 - $d <<'P'
import sys, csv, glob
for f in glob.glob(sys.argv[1] + "/*/*kernel_stats.csv"):
    for r in list(csv.DictReader(open(f)))[:14]:
        print("  %-64s calls %6s avg %9.3f us  min %9.3f max %9.3f" % (r["Name"][:64], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
P
