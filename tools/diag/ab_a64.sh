#!/bin/bash
# Diagnostic (not a test): same-box A/B of engine builds at A and A x 64 (4 alternating rounds).  usage: tools/diag/ab_a64.sh LIB...
for i in 1 2 3 4; do
    for l in "$@"; do
        printf "%-34s" "$l"
        for cfg in "" "--batch 64 --steps 8 --warmup 2"; do
            ICP_AMD_LIB=$l python3 bench.py $cfg --no-cpu-baseline --no-other-configs 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(' %9.4f' % d['us_per_iteration'], end='')"
        done
        echo
    done
done
