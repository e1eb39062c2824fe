mkdir -p gpurun_out/r3m; OUT=$GRAFT_REPO_ROOT/gpurun_out/r3m
cd /tmp && export TMPDIR=/tmp
for sc in dynerf hypernerf; do
PRECISION=f32 timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/kt_$sc -o kt --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/bench_render_image.py $sc > $OUT/ri_$sc.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("$OUT/kt_$sc/*kernel_stats.csv")[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("== $sc total kernel ms", tot/1e6)
for r in rows[:16]:
    print("%6.2f %%  %9.1f us total  calls %5d  avg %8.1f us  %s"%(float(r["Percentage"]), float(r["TotalDurationNs"])/1e3, int(r["Calls"]), float(r["AverageNs"])/1e3, r["Name"][:80]))
PY
rm -rf $OUT/kt_$sc
done
