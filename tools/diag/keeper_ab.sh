#!/bin/bash
# Diagnostic (not a test): tracking with / without the keeper thread (ICP_AMD_TRACK_KEEPER=0: the calling thread looks after the runs), same box, alternating.
for i in 1 2 3; do
    for k in 1 0; do
        echo "== ICP_AMD_TRACK_KEEPER=$k"; ICP_AMD_TRACK_KEEPER=$k python3 tools/diag/track_bench.py 2>&1 | grep -v "^track_form" | cut -c1-330
    done
done
