#!/bin/bash
# Diagnostic (not a test): same-box A/B of engine builds on the degenerate-list cases (invalid points with the colour zeroed: one list of
# identical points) and the clean frames beside them, alternating.  usage: tools/diag/dedup_ab.sh LIB...
for i in 1 2; do
    for l in "$@"; do
        echo "== $l"
        ICP_AMD_LIB=$l CFG=A CASE=clean,blobs30,scattered10_rgb0,blobs10_rgb0,blobs30_rgb0 python3 tools/diag/holes.py | cut -c1-90
        ICP_AMD_LIB=$l CFG=B CASE=clean,blobs30,blobs30_rgb0 python3 tools/diag/holes.py | cut -c1-90
        ICP_AMD_LIB=$l SIDE=256 NR=256 CASE=clean,blobs10_rgb0,blobs30_rgb0 python3 tools/diag/holes.py | cut -c1-90
        ICP_AMD_LIB=$l CFG=C CASE=clean,blobs30,blobs30_rgb0 python3 tools/diag/holes.py | cut -c1-90
        ICP_AMD_LIB=$l python3 tools/diag/build_ab.py | cut -c1-120
        ICP_AMD_LIB=$l CASE=blobs30_rgb0 python3 tools/diag/build_ab.py | cut -c1-120
    done
done
