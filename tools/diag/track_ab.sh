#!/bin/bash
# Diagnostic (not a test): tracking frames/s, same box, alternating engine builds.  usage: tools/diag/track_ab.sh LIB...
for i in 1 2; do
    for l in "$@"; do
        echo "== $l"; ICP_AMD_LIB=$l python3 tools/diag/track_bench.py 2>&1 | grep -v "^track_form"
    done
done
