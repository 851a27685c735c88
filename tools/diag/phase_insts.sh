#!/bin/bash
# Diagnostic (not a test): executed VALU / SALU / LDS / vector-memory instructions of the dense search kernel PER PHASE — builds of the
# engine whose k_search returns behind phase k (-DICP_DBG_EXIT_AFTER=k, KS_STAMP points), one PMC run each; the differences between
# consecutive builds are the phases.  usage: tools/diag/phase_insts.sh build|run [TAG]   (build here, run on the GPU box)
SRC="$(echo icp_amd/csrc/*.hip icp_amd/csrc/*.cpp)"
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wno-unused-result -pthread -mllvm -amdgpu-kernarg-preload-count=14 -Iinclude -shared"
PH="0 10 2 3 5 6 99"
if [ "$1" = build ]; then
    mkdir -p build
    for k in $PH; do /opt/rocm/bin/hipcc $FLAGS -DICP_DBG_EXIT_AFTER=$k -o build/libicp_exit$k.so $SRC 2>&1 | grep -E "error" ; done
    ls -la build/libicp_exit*.so
    exit 0
fi
export TMPDIR=/tmp
tag=${2:-r03}
mkdir -p gpurun_out
out=gpurun_out/${tag}_phase_insts.txt
echo "# tools/diag/phase_insts.sh: instructions per dispatch of the dense k_search with the kernel cut behind phase k (0 prologue + hand-over, 10 seed bound + tile masks, 2 stage 1, 3 nearest representative + list header, 5 stage 2 + its reduction, 6 epilogue wave, 99 = the whole kernel; the finalize is skipped in all of them, so every iteration searches with the identity transform)" > $out
for cfg in "C --config C --steps 2 --warmup 1" "Ax64 --batch 64 --steps 2 --warmup 1" "B --config B --steps 3 --warmup 1"; do
    set -- $cfg; n=$1; shift
    for k in $PH; do                                  # (99: no phase matches — the whole kernel, like the others with the transform held)
        lib=build/libicp_exit$k.so
        export ICP_AMD_LIB=$lib
        timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d gpurun_out/${tag}_ph_${n}_$k -- python3 bench.py --no-cpu-baseline --no-other-configs "$@" > gpurun_out/${tag}_ph_${n}_$k.log 2>&1 || { echo "failed $n $k" | tee -a $out; tail -3 gpurun_out/${tag}_ph_${n}_$k.log; exit 1; }
        python3 - "$n" "$k" gpurun_out/${tag}_ph_${n}_$k >> $out <<'P'
import sys, csv, glob, collections
n, k, d = sys.argv[1:4]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(d + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        kn = r["Kernel_Name"].split("(")[0]
        if "k_search" not in kn or ", true, 1," in kn.replace("true, 1, 2", ""): pass
        acc[kn][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_INSTS_VALU": cnt[kn] += 1
best = max((kn for kn in acc if "k_search" in kn), key=lambda kn: cnt[kn])
c, dsp = acc[best], cnt[best]
print("%-5s exit %-4s %-62s dispatches %4d  VALU %12.0f  SALU %12.0f  LDS %11.0f  VMEM_RD %10.0f" % (n, k, best[:62], dsp, c["SQ_INSTS_VALU"] / dsp, c["SQ_INSTS_SALU"] / dsp, c["SQ_INSTS_LDS"] / dsp, c["SQ_INSTS_VMEM_RD"] / dsp))
P
        rm -rf gpurun_out/${tag}_ph_${n}_$k gpurun_out/${tag}_ph_${n}_$k.log
        tail -1 $out
    done
done
