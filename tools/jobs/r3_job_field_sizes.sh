# kernel durations of the field kernel against the sample count, from the profiler's trace (the host's launch rate
# bounds what events around back-to-back python calls can show)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3p; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for tag in ${TAGS:-base stageloop}; do
lib=$GRAFT_REPO_ROOT/ced_nerf_amd/libcednerf_hip.so; [ "$tag" != "base" ] && lib=$GRAFT_REPO_ROOT/build/variants/libcednerf_hip.$tag.so
echo "== $tag"
CED_NERF_LIB=$lib timeout -k 10 300 rocprofv3 --kernel-trace -d $OUT/tr -o t --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/bench_field_sizes.py > $OUT/sizes.log 2>&1
python3 - <<PY
import csv,glob,collections
f=glob.glob("$OUT/tr/**/*kernel_trace.csv", recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if "field_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# 23 launches per size (3 warm-up + 20 timed), sizes in the script's order
sizes=[32, 98304, 196608, 393216, 500000, 491520, 589824, 1000000, 2000000, 4000000, 8000000]
for k,n in enumerate(sizes):
    grp=rows[k*23+3:(k+1)*23]
    d=sorted((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in grp)
    print("n = %8d: kernel duration median %8.1f us (min %.1f)  %.3f Gsamples/s"%(n, d[len(d)//2], d[0], n/d[len(d)//2]/1e3))
PY
rm -rf $OUT/tr
done
