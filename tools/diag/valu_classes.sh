#!/bin/bash
# Diagnostic (not a test): which SQ_INSTS_VALU_* class counter each instruction of tests/cpp/valu_issue_probe.hip lands in (the probe's kernels
# under two PMC passes), then the same passes around the bench's dominant kernel at A, B, C and A x 64: the DYNAMIC class mix that
# tools/diag/valu_classes_summary.py prices with the probe's constants.  usage: tools/diag/valu_classes.sh TAG
export TMPDIR=/tmp
tag=${1:-r06}
P1="SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT64"
P2="SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_WAVES"
P3="SQ_ACTIVE_INST_VALU2 SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_ANY SQ_BUSY_CU_CYCLES SQ_WAIT_INST_ANY"
mkdir -p gpurun_out
i=0
for P in "$P1" "$P2" "$P3"; do
    i=$((i + 1))
    timeout -k 10 200 rocprofv3 --pmc $P --kernel-trace --output-format csv -d gpurun_out/${tag}_vc_probe_$i -- tests/cpp/valu_issue_probe /tmp/vi.json > gpurun_out/${tag}_vc_probe_$i.log 2>&1 || { echo "probe pass $i failed"; tail -3 gpurun_out/${tag}_vc_probe_$i.log; exit 1; }
    for cfg in "A --steps 5 --warmup 1" "B --config B --steps 3 --warmup 1" "C --config C --steps 2 --warmup 1" "Ax64 --batch 64 --steps 2 --warmup 1"; do
        set -- $cfg; n=$1; shift
        timeout -k 10 240 rocprofv3 --pmc $P --kernel-trace --output-format csv -d gpurun_out/${tag}_vc_${n}_$i -- python3 bench.py --no-cpu-baseline --no-other-configs "$@" > gpurun_out/${tag}_vc_${n}_$i.log 2>&1 || { echo "bench pass $n $i failed"; tail -3 gpurun_out/${tag}_vc_${n}_$i.log; exit 1; }
    done
    echo "pass $i done"
done
python3 tools/diag/valu_classes_summary.py $tag
rm -rf gpurun_out/${tag}_vc_*/
