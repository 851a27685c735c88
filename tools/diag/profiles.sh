#!/bin/bash
# Diagnostic (not a test): the profile set of a round, per BASELINE config.  usage: tools/diag/profiles.sh TAG  (e.g. r03)
#   kernel-trace statistics of bench.py at the default line (A, chained) and with --config B / --config C / --batch 64,
#   then separate PMC passes (counters never share a run with --stats; the program comes directly after `--`):
#   FETCH_SIZE, WRITE_SIZE, SQ instruction / wait counters, for C, A x 64 and the default line.
# Output under gpurun_out/<TAG>_*; tools/diag/profiles_summary.py condenses it into gpurun_out/<TAG>_profile_summary.json,
# and the summaries that are judged get copied to profiles/ by hand.
export TMPDIR=/tmp
tag=${1:-r03}
mkdir -p gpurun_out
stats () {   # name, bench args...
    name=$1; shift
    timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_stats_$name -- python3 bench.py --no-cpu-baseline --no-other-configs "$@" > gpurun_out/${tag}_stats_$name.log 2>&1 || return 1
    f=$(ls gpurun_out/${tag}_stats_$name/*/*kernel_stats.csv 2>/dev/null | head -1)
    [ -n "$f" ] && cp "$f" gpurun_out/${tag}_${name}_kernel_stats.csv
    echo "== stats $name"; [ -n "$f" ] && head -6 "$f" | cut -c1-160
}
pmc () {     # name, counters..., then "--" and bench args
    name=$1; shift
    ctrs=(); while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done; shift
    timeout -k 10 240 rocprofv3 --pmc "${ctrs[@]}" --kernel-trace --output-format csv -d gpurun_out/${tag}_pmc_$name -- python3 bench.py --no-cpu-baseline --no-other-configs "$@" > gpurun_out/${tag}_pmc_$name.log 2>&1 || return 1
    echo "== pmc $name done"
}
stats A --steps 50 --warmup 5 &&
stats B --config B --steps 20 --warmup 3 &&
stats C --config C --steps 3 --warmup 1 &&
stats Ax64 --batch 64 --steps 5 --warmup 1 &&
stats REF --reduce-mode reference --power-mode literal --steps 50 --warmup 5 &&
for cfg in "C --config C --steps 2 --warmup 1" "Ax64 --batch 64 --steps 2 --warmup 1" "B --config B --steps 3 --warmup 1" "A --steps 5 --warmup 1"; do
    set -- $cfg; n=$1; shift
    pmc ${n}_fetch FETCH_SIZE -- "$@" &&
    pmc ${n}_write WRITE_SIZE -- "$@" &&
    pmc ${n}_sq1 SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM_RD -- "$@" &&
    pmc ${n}_sq2 SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY -- "$@" &&
    pmc ${n}_l2 TCC_HIT_sum TCC_MISS_sum -- "$@" || break
done
python3 tools/diag/profiles_summary.py $tag
# (the raw traces stay on the box: gpurun_out/ is merged back up to 64 MiB)
rm -rf gpurun_out/${tag}_stats_*/ gpurun_out/${tag}_pmc_*/
