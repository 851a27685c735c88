#!/bin/bash
# Diagnostic (not a test): tracking frames/s, same box, alternating engine builds.  usage: tests/diag_track_ab.sh LIB...
for i in 1 2; do
    for l in "$@"; do
        echo "== $l"; ICP_AMD_LIB=$l python3 tests/diag_track_bench.py 2>&1 | grep -v "^track_form"
    done
done
