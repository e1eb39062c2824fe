OUT=$GRAFT_REPO_ROOT/gpurun_out/r3p; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for st in 0 1 2 4; do
echo "== field_stagger=$st"
CED_OPTIONS=field_stagger=$st timeout -k 10 300 rocprofv3 --kernel-trace -d $OUT/tr -o t --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/bench_field_sizes.py > $OUT/sizes.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("$OUT/tr/**/*kernel_trace.csv", recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if "field_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
sizes=[32, 98304, 196608, 393216, 500000, 491520, 589824, 1000000, 2000000, 4000000, 8000000]
out=[]
for k,n in enumerate(sizes):
    grp=rows[k*23+3:(k+1)*23]
    d=sorted((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in grp)
    out.append("%d: %.1f"%(n, d[len(d)//2]))
print("  ".join(out))
PY
rm -rf $OUT/tr
done
