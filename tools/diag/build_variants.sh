export TMPDIR=/tmp
for l in "$@"; do
export ICP_AMD_LIB=$l
d=gpurun_out/bk; rm -rf $d
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $d -- python3 tools/diag/build_ab.py > $d.log 2>&1 || { tail -3 $d.log; exit 1; }
echo "== $l"; grep buildRBC $d.log
python3 - <<P
import csv,glob,collections
rows=list(csv.DictReader(open(glob.glob("gpurun_out/bk/*/*kernel_trace.csv")[0])))
d=collections.defaultdict(list)
for r in rows:
    if r["Kernel_Name"].startswith("k_reps"): d[(r["Grid_Size_X"],r["Grid_Size_Y"])].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
for k,v in d.items(): print("   k_reps_and_boxes grid",k, len(v), sorted(v)[len(v)//2])
P
done
