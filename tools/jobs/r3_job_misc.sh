mkdir -p gpurun_out/r3o; OUT=$GRAFT_REPO_ROOT/gpurun_out/r3o
cd /tmp && export TMPDIR=/tmp
prof() { # name cmd...
  local name=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/kt_$name -o kt --output-format csv -- "$@" > $OUT/$name.log 2>&1
  python3 - <<PY
import csv,glob
f=glob.glob("$OUT/kt_$name/*kernel_stats.csv")[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("== $name total kernel ms %.1f"%(tot/1e6)); print(open("$OUT/$name.log").read().strip().splitlines()[-2:])
for r in rows[:12]:
    print("%6.2f %%  %9.1f us total  calls %5d  avg %8.1f us  %s"%(float(r["Percentage"]), float(r["TotalDurationNs"])/1e3, int(r["Calls"]), float(r["AverageNs"])/1e3, r["Name"][:90]))
PY
  rm -rf $OUT/kt_$name
}
prof occgrid python3 $GRAFT_REPO_ROOT/tools/bench_occgrid.py dnerf dynerf
PRECISION=f32+h16x2 prof ri_dnerf python3 $GRAFT_REPO_ROOT/tools/bench_render_image.py dnerf
