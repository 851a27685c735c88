#!/bin/bash
# Diagnostic (not a test): SQ counters of the dense k_search variant (bench.py --batch 64), one pass per counter group.
export TMPDIR=/tmp
mkdir -p gpurun_out
run () {
    name=$1; shift
    timeout 150 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d gpurun_out/pmcb_$name -- python3 bench.py --batch 64 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/pmcb_$name.log 2>&1
    python3 - "$name" <<'PY'
import csv, glob, collections, sys, os
fs = sorted(glob.glob('gpurun_out/pmcb_%s/*/*counter_collection.csv' % sys.argv[1]), key=os.path.getmtime)
if fs:
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[-1])):
        acc[r['Kernel_Name'][:44]][r['Counter_Name']].append(float(r['Counter_Value']))
    for kn, d in acc.items():
        if 'k_search' in kn or 'finalize' in kn:
            for k, v in d.items():
                print("%-46s %-28s mean %16.1f  (n=%d)" % (kn, k, sum(v) / len(v), len(v)))
PY
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY
run sq2 SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS
run sq3 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM
run grbm GRBM_GUI_ACTIVE SQ_WAVES
