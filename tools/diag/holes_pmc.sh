#!/bin/bash
# Diagnostic (not a test): per-kernel time and SQ counters of the search with and without invalid points.  usage: tools/diag/holes_pmc.sh TAG [BATCH] [CFG]
export TMPDIR=/tmp
tag=${1:-r05}; export BATCH=${2:-64}; export CFG=${3:-A}
mkdir -p gpurun_out
out=gpurun_out/${tag}_holes_pmc_${CFG}x${BATCH}.txt
: > $out
for c in clean scattered10 blobs30; do
    export CASE=$c
    d=gpurun_out/${tag}_hp_$c
    timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 tools/diag/holes.py > $d.log 2>&1 || { echo failed $c; tail -3 $d.log; exit 1; }
    echo "== $c: kernel stats" >> $out; cat $d.log >> $out
    python3 - $d >> $out <<'P'
import sys, csv, glob
for f in glob.glob(sys.argv[1] + "/*/*kernel_stats.csv"):
    for r in list(csv.DictReader(open(f)))[:6]:
        print("  %-70s calls %6s avg %10.2f us  total %5.1f %%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
P
    rm -rf $d $d.log
    for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU"; do
        timeout -k 10 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $d -- python3 tools/diag/holes.py > $d.log 2>&1 || { echo failed pmc $c; tail -3 $d.log; exit 1; }
        python3 - $d >> $out <<'P'
import sys, csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for kn, d in acc.items():
    if "k_search" in kn and len(next(iter(d.values()))) > 100:
        print("  %-60s " % kn + "  ".join("%s %.3gM" % (k, sum(v) / len(v) / 1e6) for k, v in d.items()) + "  (n=%d)" % len(next(iter(d.values()))))
P
        rm -rf $d $d.log
    done
done
cat $out
