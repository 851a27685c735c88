#!/bin/bash
# Diagnostic (not a test): rocprofv3 --kernel-trace --stats of the hole cases at config A, one run per case, the top kernels of each into
# gpurun_out/<TAG>_A_holes_kernel_stats.csv (case, kernel, calls, average ns, percentage).  usage: tools/diag/holes_stats.sh TAG
export TMPDIR=/tmp
tag=${1:-r05}
mkdir -p gpurun_out
out=gpurun_out/${tag}_A_holes_kernel_stats.csv
echo '"Case","Name","Calls","TotalDurationNs","AverageNs","Percentage","us_per_iteration_of_the_fixed_passes"' > $out
for c in clean scattered10 blobs10 blobs30 scattered10_rgb0 blobs10_rgb0 blobs30_rgb0; do
    export CASE=$c
    d=gpurun_out/${tag}_hs_$c
    rm -rf $d
    timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 tools/diag/holes.py > $d.log 2>&1 || { echo failed $c; tail -3 $d.log; exit 1; }
    us=$(grep "us/iter" $d.log | sed 's/.* \([0-9.]*\) us\/iter.*/\1/')
    python3 - $d $c "$us" >> $out <<'P'
import sys, csv, glob
for f in glob.glob(sys.argv[1] + "/*/*kernel_stats.csv"):
    for r in list(csv.DictReader(open(f)))[:5]:
        print('"%s","%s",%s,%s,%s,%s,%s' % (sys.argv[2], r["Name"].split("(")[0], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], sys.argv[3]))
P
    rm -rf $d $d.log
done
cat $out | cut -c1-170
