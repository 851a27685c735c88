#!/bin/bash
# Diagnostic (not a test): same-box A/B of engine builds at config A.  usage: tools/diag/ab.sh LIB... (3 alternating rounds)
for i in 1 2 3; do
    for l in "$@"; do
        printf "%-32s " "$l"; ICP_AMD_LIB=$l python tools/diag/bench.py 2>&1 | grep run_fixed | tail -1
    done
done
