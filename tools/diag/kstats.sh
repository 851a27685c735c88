#!/bin/bash
# Diagnostic (not a test): per-kernel average durations (rocprofv3 --kernel-trace --stats) of bench.py at one configuration for several engine builds.
# usage: tools/diag/kstats.sh "<bench args>" LIB...
export TMPDIR=/tmp
args=$1; shift
for l in "$@"; do
    export ICP_AMD_LIB=$l
    d=gpurun_out/ks_$(basename $l .so)
    rm -rf $d
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 bench.py $args --no-cpu-baseline --no-other-configs > $d.log 2>&1 || { echo "failed $l"; tail -3 $d.log; exit 1; }
    echo "== $l"
    python3 - $d <<'P'
import sys, csv, glob
for f in glob.glob(sys.argv[1] + "/*/*kernel_stats.csv"):
    for r in list(csv.DictReader(open(f)))[:5]:
        print("  %-64s calls %6s avg %9.3f us  %5.1f %%" % (r["Name"][:64], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
P
    rm -rf $d $d.log
done
