#!/bin/bash
# Diagnostic (not a test): one PMC pass around a short bench run; prints per-dispatch means for k_search.
# usage: tools/diag/pmc.sh NAME COUNTER...      (TA_* counters abort rocprofv3 on this pool: do not use them)
export TMPDIR=/tmp
name=$1; shift
timeout 120 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d gpurun_out/pmc_$name -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_$name.log 2>&1
python3 - "$name" <<'PY'
import csv, glob, collections, sys
for f in sorted(glob.glob('gpurun_out/pmc_%s/*/*counter_collection.csv' % sys.argv[1])):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        acc[r['Kernel_Name'][:40]][r['Counter_Name']].append(float(r['Counter_Value']))
    for kn, d in acc.items():
        if 'k_search' in kn or 'finalize' in kn:
            for k, v in d.items():
                print("%-42s %-28s mean %14.1f  (n=%d)" % (kn, k, sum(v) / len(v), len(v)))
PY
