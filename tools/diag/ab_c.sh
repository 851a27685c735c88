#!/bin/bash
# Diagnostic (not a test): same-box A/B of engine builds at config C only (3 alternating rounds).  usage: tools/diag/ab_c.sh LIB...
for i in 1 2 3; do
    for l in "$@"; do
        printf "%-34s" "$l"
        ICP_AMD_LIB=$l python3 bench.py --config C --steps 3 --no-cpu-baseline --no-other-configs 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(' %9.3f' % d['us_per_iteration'])"
    done
done
