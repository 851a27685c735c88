#!/bin/bash
# Diagnostic (not a test): instruction-mix counters of one bench configuration.  usage: tools/diag/pmc_cfg.sh NAME bench-args...
export TMPDIR=/tmp
name=$1; shift
mkdir -p gpurun_out
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVES"; do
    tag=$(echo $set | cut -d' ' -f1)
    timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/pmcx_${name}_$tag -- python3 bench.py --no-cpu-baseline --no-other-configs "$@" > gpurun_out/pmcx_${name}_$tag.log 2>&1 || break
done
python3 - "$name" <<'PY'
import csv, glob, collections, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/pmcx_%s_*/*/*counter_collection.csv' % sys.argv[1]):
    for r in csv.DictReader(open(f)):
        acc[r['Kernel_Name'].replace('void ', '').split('(')[0]][r['Counter_Name']].append(float(r['Counter_Value']))
for kn, d in acc.items():
    if 'k_search' in kn or 'finalize' in kn:
        print(kn[:56], {k: round(sum(v) / len(v)) for k, v in sorted(d.items())}, 'n=%d' % max(len(v) for v in d.values()))
PY
