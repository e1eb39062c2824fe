# training path: the fused element-wise pieces -- parity tests, step time, kernel statistics of the 262k-ray step
mkdir -p gpurun_out/r3t && cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3t
timeout -k 10 600 python -m pytest tests -m gpu -x -q --capture=sys -k "mlp_backward or table_gradient or trainable or train_step or student or loss or rendering_train or mlp_chain or weight_grad" > $OUT/tests.log 2>&1; echo "tests rc=$?" | tee -a $OUT/tests.log
tail -4 $OUT/tests.log
for ov in 1; do echo "overlap_table_grad=$ov"; OVERLAP=$ov timeout -k 10 300 python tools/bench_train.py 2>&1 | grep train_step; done | tee $OUT/bench_train.txt
cd /tmp && export TMPDIR=/tmp
N_RAYS=262144 timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/bench_train.py > $OUT/kt.log 2>&1
cp $(ls $OUT/kt/*kernel_stats.csv | head -1) $OUT/train_262k_kernel_stats.csv; rm -rf $OUT/kt
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$OUT/train_262k_kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("kernel time per step %.2f ms (13 steps)"%(tot/13/1e6))
for r in rows[:24]:
    print("%6.2f %%  %8.1f us/step  calls/step %5.1f  %s"%(float(r["Percentage"]), float(r["TotalDurationNs"])/13/1e3, int(r["Calls"])/13, r["Name"][:90]))
PY
