#!/bin/bash
# Diagnostic (not a test): same-box A/B of engine builds on the hole cases of the dense shapes (B; A x 64; a long-list shape), alternating.
# usage: tools/diag/holes_ab.sh LIB...
for i in 1 2; do
    for l in "$@"; do
        echo "== $l"
        ICP_AMD_LIB=$l CFG=B CASE=clean,scattered10,blobs10,blobs30 python3 tools/diag/holes.py | cut -c1-75
        ICP_AMD_LIB=$l CFG=A BATCH=64 CASE=clean,scattered10,blobs10,blobs30 python3 tools/diag/holes.py | cut -c1-75
    done
done
