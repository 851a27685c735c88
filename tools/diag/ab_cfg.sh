#!/bin/bash
# Diagnostic (not a test): same-box A/B of engine builds over the bench configurations (3 alternating rounds).
# usage: tools/diag/ab_cfg.sh LIB...      prints us per iteration of: A, B, C, A x 64
for i in 1 2 3; do
    for l in "$@"; do
        printf "%-34s" "$l"
        for cfg in "" "--config B --steps 20" "--config C --steps 3" "--batch 64 --steps 8 --warmup 2"; do
            ICP_AMD_LIB=$l python3 bench.py $cfg --no-cpu-baseline --no-other-configs 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(' %9.3f' % d['us_per_iteration'], end='')"
        done
        echo
    done
done
