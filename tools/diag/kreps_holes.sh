export TMPDIR=/tmp
for cfgcase in "C blobs30" "C scattered10" "B blobs30"; do set -- $cfgcase; export CFG=$1 CASE=$2
d=gpurun_out/kr; rm -rf $d
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $d -- python3 tools/diag/holes.py > $d.log 2>&1 || { tail -3 $d.log; exit 1; }
python3 - <<P
import csv,glob
rows=list(csv.DictReader(open(glob.glob("gpurun_out/kr/*/*kernel_trace.csv")[0])))
v=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in rows if r["Kernel_Name"].startswith("k_reps_and_boxes")]
print("$CFG $CASE k_reps_and_boxes us:", sorted(v))
P
rm -rf $d $d.log
done
