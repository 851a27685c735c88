#!/bin/bash
# Diagnostic (not a test): same-box A/B of an environment switch over the bench configurations (3 alternating rounds).
# usage: tools/diag/env_cfg.sh VAR=VALUE    prints us per iteration of: A, B, C, A x 64 without and with the setting
for i in 1 2 3; do
    for e in "_ICP_NONE=1" "$1"; do
        printf "%-28s" "$e"
        for cfg in "" "--config B --steps 20" "--config C --steps 3" "--batch 64 --steps 8 --warmup 2"; do
            env "$e" python3 bench.py $cfg --no-cpu-baseline --no-other-configs 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(' %9.3f' % d['us_per_iteration'], end='')"
        done
        echo
    done
done
