#!/bin/bash
# Diagnostic (not a test): VALU lane utilisation of the search kernels (SQ_THREAD_CYCLES_VALU / (SQ_ACTIVE_INST_VALU x 64)) and the
# scalar / LDS instruction mix beside it, per config.  usage: tools/diag/lanes.sh TAG.  Counters in a run of their own (--kernel-trace only).
export TMPDIR=/tmp
tag=${1:-r03}
mkdir -p gpurun_out
for cfg in "Ax64 --batch 64 --steps 2 --warmup 1" "C --config C --steps 2 --warmup 1" "B --config B --steps 3 --warmup 1" "A --steps 5 --warmup 1"; do
    set -- $cfg; n=$1; shift
    timeout -k 10 240 rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU --kernel-trace --output-format csv -d gpurun_out/${tag}_lanes_$n -- python3 bench.py --no-cpu-baseline --no-other-configs "$@" > gpurun_out/${tag}_lanes_$n.log 2>&1 || { echo "failed $n"; tail -5 gpurun_out/${tag}_lanes_$n.log; break; }
    python3 - "$n" gpurun_out/${tag}_lanes_$n <<'P'
import sys, csv, glob, collections
n, d = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(d + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_INSTS_VALU": cnt[k] += 1
for k, c in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_INSTS_VALU", 0))[:4]:
    d_ = max(cnt[k], 1)
    if c.get("SQ_ACTIVE_INST_VALU"):
        print("%-5s %-60s dispatches %5d  VALU %.3e  SALU %.3e  lane utilisation %.3f" % (n, k[:60], d_, c["SQ_INSTS_VALU"] / d_, c.get("SQ_INSTS_SALU", 0) / d_, c["SQ_THREAD_CYCLES_VALU"] / (c["SQ_ACTIVE_INST_VALU"] * 64)))
P
    rm -rf gpurun_out/${tag}_lanes_$n
done
