#!/bin/bash
# Diagnostic (not a test): same-box A/B of engine builds at config A only (5 alternating rounds).  usage: tools/diag/ab_a.sh LIB...
for i in 1 2 3 4 5; do
    for l in "$@"; do
        printf "%-34s" "$l"
        ICP_AMD_LIB=$l python3 bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(' %9.4f' % d['us_per_iteration'])"
    done
done
