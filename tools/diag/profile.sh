#!/bin/bash
# Diagnostic (not a test): the round's profile set.  Kernel-trace statistics of the bench command in both
# reduction modes, then separate FETCH_SIZE / WRITE_SIZE passes (counters never share a run with --stats).
# Output under gpurun_out/; the summaries that are judged get copied to profiles/ by hand.
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_fused -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/prof_fused.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ref -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --reduce-mode reference > gpurun_out/prof_ref.log 2>&1
for d in prof_fused prof_ref; do
    f=$(ls gpurun_out/$d/*/*kernel_stats.csv 2>/dev/null | head -1)
    echo "== $d: $f"; [ -n "$f" ] && head -12 "$f"
    tail -1 gpurun_out/$d.log | cut -c1-400
done
bash tools/diag/pmc.sh fetch FETCH_SIZE
bash tools/diag/pmc.sh write WRITE_SIZE
bash tools/diag/pmc.sh l2 TCC_HIT_sum TCC_MISS_sum
bash tools/diag/pmc.sh sq SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU
