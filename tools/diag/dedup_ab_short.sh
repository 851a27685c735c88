#!/bin/bash
# Diagnostic: the B / (65536, 256) / A x 64 part of tools/diag/dedup_ab.sh.  usage: tools/diag/dedup_ab_short.sh LIB...
for i in 1 2; do
    for l in "$@"; do
        echo "== $l"
        ICP_AMD_LIB=$l CFG=B CASE=clean,blobs30,scattered10_rgb0,blobs30_rgb0 python3 tools/diag/holes.py | cut -c1-90
        ICP_AMD_LIB=$l CFG=A BATCH=64 CASE=clean,blobs30,blobs30_rgb0 python3 tools/diag/holes.py | cut -c1-90
        ICP_AMD_LIB=$l CFG=C CASE=blobs30_rgb0 python3 tools/diag/holes.py | cut -c1-90
    done
done
